#!/usr/bin/env python3
"""GPU-side timeline of the last bench step from a rocprofv3 --kernel-trace rocpd database: every kernel
dispatch of the step with its start and end relative to the step's first kernel (the kernels of a step
overlap: two histogram launches and the selection's head phase run on different streams)."""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
ks = c.execute("select name, start, end from kernels order by start").fetchall()
hist = [i for i, k in enumerate(ks) if "kmer_hist_kernel" in k[0]]
per_step = int(sys.argv[2]) if len(sys.argv) > 2 else 2  # histogram launches per step
a, b = hist[-2 * per_step], hist[-per_step]
t0 = ks[a][1]
for name, s, e in ks[a:b]:
    short = name.split("(")[0].split("::")[-1][:40]
    print(f"{(s - t0) / 1e3:8.1f} .. {(e - t0) / 1e3:8.1f} us  ({(e - s) / 1e3:7.1f})  {short}")
print(f"step {(ks[b][1] - t0) / 1e3:.1f} us")
