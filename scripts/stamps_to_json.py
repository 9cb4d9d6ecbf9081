#!/usr/bin/env python3
"""gpurun_out/<tag>_stamps.txt (what the -DDVS_PERSIST_STAMPS library prints per launch under
DVS_PERSIST_DEBUG=1, scripts/stamps.sh) -> JSON: per launch kind (head-phase / full-grid) the LAST step's
phase times in microseconds for block 0 and the mirror block, and the window statistics.

  python scripts/stamps_to_json.py gpurun_out/r03_b_stamps.txt "nmost n=10, 100000 x 5000 bp, k=6" > profiles/r03_b_stamps.json
"""
import json
import re
import sys

txt = open(sys.argv[1]).read().splitlines()
out = {"workload": sys.argv[2] if len(sys.argv) > 2 else "", "unit": "us per selection (sum over the launch), s_memrealtime at 100 MHz",
       "source": "scripts/stamps.sh: library built with -DDVS_PERSIST_STAMPS, bench.py --steps 4 --no-side-runs, DVS_PERSIST_DEBUG=1",
       "phases": {"scan": "this workgroup's share of the window's rows", "bar1": "arrival record stored -> release seen "
                  "(the accept's frequencies and this workgroup's leave-one-out job are worked out in here)",
                  "resolve": "release -> accept decided", "loo": "publish the job, shift the member arrays, mirror (mirror block)",
                  "bar2": "wait for the leave-one-out totals", "partials": "totals -> delta_jsd", "combine": "argmin, hand-over through LDS",
                  "lowest-row fetch": "the new lowest member's row (LDS for small sets)", "rebuild": "sl = S' - lowest, f32 copy"},
       "launches": {}}
kind = None
for line in txt:
    m = re.match(r"\[dvs persist\] (head-phase|full-grid) launch", line)
    if m:
        kind = m.group(1)
        out["launches"][kind] = {}
        continue
    m = re.match(r"\[dvs persist (block 0|mirror block)\] us: (.*)", line)
    if m and kind:
        vals = dict(re.findall(r"([a-z0-9\- ]+?) ([0-9.]+)(?: \||$| )", m.group(2).replace("|", "")))
        out["launches"][kind][m.group(1)] = {k.strip(): float(v) for k, v in vals.items()}
        continue
    m = re.match(r"\[dvs persist block 0\] us inside the phases: window top ([0-9.]+) own rows scanned ([0-9.]+) hint look \+ record ([0-9.]+) .*behind the rebuild ([0-9.]+)", line)
    if m and kind:
        out["launches"][kind]["block 0, inside the phases"] = {
            "window top (barrier, window words reset)": float(m.group(1)), "own rows scanned": float(m.group(2)),
            "workgroup barrier + arrival record stored": float(m.group(3)), "behind the rebuild": float(m.group(4)),
            "note": "these four are taken out of `scan` (which keeps only what lies between the last row and the arrival) and of `rebuild`"}
        continue
    m = re.match(r"\[dvs persist\] scan \+ rendezvous: row-per-workgroup windows (\d+) \(([0-9.]+) us, (\d+) rows\), row-per-wave windows (\d+) \(([0-9.]+) us, (\d+) rows\)", line)
    if m and kind:
        out["launches"][kind]["windows"] = {"row_per_workgroup": {"count": int(m.group(1)), "us": float(m.group(2)), "rows": int(m.group(3))},
                                            "row_per_wave": {"count": int(m.group(4)), "us": float(m.group(5)), "rows": int(m.group(6))}}
print(json.dumps(out, indent=1))
