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


def gaps(db):
    """per kernel name, over the dispatches in time order: duration percentiles and the idle time in front of it
    (its start minus the previous kernel's end) -- where a chain of dependent launches loses its time"""
    c = sqlite3.connect(db)
    rows = c.execute("select name, start, end from kernels order by start").fetchall()
    per = {}
    prev_end = None
    for name, st, en in rows:
        d = per.setdefault(name, {"dur": [], "gap": []})
        d["dur"].append(en - st)
        if prev_end is not None:
            d["gap"].append(max(0, st - prev_end))
        prev_end = max(prev_end or en, en)
    w = csv.writer(sys.stdout)
    w.writerow(["Name", "Calls", "DurP10Ns", "DurP50Ns", "DurP90Ns", "DurMaxNs", "DurMeanNs", "GapBeforeP50Ns", "GapBeforeMeanNs"])
    pc = lambda v, q: sorted(v)[int(q * (len(v) - 1))] if v else 0
    for name, d in sorted(per.items(), key=lambda kv: -sum(kv[1]["dur"])):
        g = [x for x in d["gap"] if x < 1_000_000]  # (a gap of a millisecond is the host between two selections, not the chain)
        w.writerow([short(name), len(d["dur"]), pc(d["dur"], .1), pc(d["dur"], .5), pc(d["dur"], .9), max(d["dur"]),
                    f"{sum(d['dur']) / len(d['dur']):.1f}", pc(g, .5), f"{(sum(g) / len(g)) if g else 0:.1f}"])


if __name__ == "__main__":
    {"stats": stats, "pmc": pmc, "gaps": gaps}[sys.argv[1]](sys.argv[2])
