#!/usr/bin/env python3
"""top kernels (calls, total, average) of a rocprofv3 --kernel-trace rocpd database"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, count(*), sum(duration), avg(duration) from kernels group by name "
                 "order by sum(duration) desc limit 14").fetchall()
for n, k, t, a in rows:
    short = n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:48]
    print(f"{short:50s} calls {k:7d}  total {t / 1e6:9.3f} ms  avg {a / 1e3:9.2f} us")
