#!/usr/bin/env python3
"""Host + device timeline of one bench step from a rocprofv3 --hip-trace --kernel-trace rocpd
database: every HIP API call and kernel between two consecutive kmer_hist_kernel dispatches."""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
ks = c.execute("select name, start, end from kernels order by start").fetchall()
hist = [k for k in ks if "kmer_hist_kernel" in k[0]]
t0, t1 = hist[-4][1], hist[-2][1]  # (two histogram launches per step: head rows, rest)
cols = [d[1] for d in c.execute("pragma table_info(regions)")]
rows = c.execute("select name, start, end from regions where start >= ? and start < ? order by start",
                 (t0 - 200000, t1)).fetchall()
ev = [("K", n.split("(")[0].split("::")[-1][:40], s, e) for n, s, e in ks if t0 <= s < t1]
ev += [("A", n[:40], s, e) for n, s, e in rows]
ev.sort(key=lambda r: r[2])
for kind, n, s, e in ev:
    if kind == "K" or e - s > 3000:
        print(f"{(s - t0) / 1e3:9.1f} us  {kind} {(e - s) / 1e3:8.1f} us  {n}")
print("step", (t1 - t0) / 1e3, "us; api calls in step:", len(rows))
