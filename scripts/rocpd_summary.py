#!/usr/bin/env python3
"""Summaries of a rocprofv3 run stored as a rocpd SQLite database (the default output format
of the ROCm 7 profiler): per-kernel statistics (the `--stats` table) and, for a `--pmc` pass,
the per-kernel mean of every collected counter.

  python scripts/rocpd_summary.py stats  <results.db> > profiles/rNN_kernel_stats.csv
  python scripts/rocpd_summary.py pmc    <results.db> > profiles/rNN_pmc_<counter>.csv
"""
import csv
import sqlite3
import sys


def short(name: str) -> str:
    # torch's generator kernels carry kilobyte-long template names
    return name if len(name) < 300 else name[:297] + "..."


def stats(db):
    c = sqlite3.connect(db)
    rows = c.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
        "from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    w = csv.writer(sys.stdout)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for name, calls, tot, avg, mn, mx in rows:
        w.writerow([short(name), calls, tot, f"{avg:.1f}", f"{100.0 * tot / total:.3f}", mn, mx])


def pmc(db):
    c = sqlite3.connect(db)
    rows = c.execute(
        "select kernel_name, counter_name, count(*), avg(value), min(value), max(value), avg(duration) "
        "from counters_collection group by kernel_name, counter_name order by avg(value) desc").fetchall()
    w = csv.writer(sys.stdout)
    w.writerow(["Kernel", "Counter", "Dispatches", "MeanValue", "MinValue", "MaxValue", "MeanDurationNs"])
    for k, cn, n, avg, mn, mx, dur in rows:
        w.writerow([short(k), cn, n, avg, mn, mx, f"{dur:.1f}"])


if __name__ == "__main__":
    {"stats": stats, "pmc": pmc}[sys.argv[1]](sys.argv[2])
