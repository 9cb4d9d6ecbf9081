#!/usr/bin/env python3
"""GPU-side timeline of one bench step from a rocprofv3 --kernel-trace rocpd database: every
kernel between two consecutive kmer_hist_kernel dispatches with the idle gap in front of it."""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
ks = c.execute("select name, start, end from kernels order by start").fetchall()
hist = [i for i, k in enumerate(ks) if "kmer_hist_kernel" in k[0]]
a, b = hist[-2], hist[-1]
prev_end = ks[a - 1][2] if a else ks[a][1]
busy = 0
for name, s, e in ks[a:b]:
    short = name.split("(")[0].split("::")[-1][:36]
    print(f"gap {(s - prev_end) / 1e3:7.1f} us | {(e - s) / 1e3:8.1f} us  {short}")
    busy += e - s
    prev_end = e
step = ks[b][1] - ks[a][1]
print(f"step {step / 1e3:.1f} us, kernels {busy / 1e3:.1f} us, idle {(step - busy) / 1e3:.1f} us")
