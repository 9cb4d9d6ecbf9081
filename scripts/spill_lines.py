#!/usr/bin/env python3
"""Source lines of a kernel's register spills and reloads, from an annotated assembly listing
(hipcc ... --offload-device-only -S -gline-tables-only):
    scripts/spill_lines.py persist_dbg.s <kernel-name-substring>"""
import collections
import re
import sys

path, want = sys.argv[1], sys.argv[2]
cur = None
line = 0
inside = False
sp = collections.Counter()
rl = collections.Counter()
for l in open(path):
    m = re.match(r"^(_Z\S+):", l)
    if m:
        inside = want in m.group(1)
        continue
    if not inside:
        continue
    m = re.match(r"\s+\.loc\s+\d+\s+(\d+)", l)
    if m:
        line = int(m.group(1))
        continue
    if l.startswith("\t.end_amdhsa_kernel") or l.startswith(".Lfunc_end"):
        inside = False
    if "Folded Spill" in l:
        sp[line] += 1
    elif "Folded Reload" in l:
        rl[line] += 1
print("line spills reloads")
for k in sorted(set(sp) | set(rl)):
    print(f"{k:5d} {sp[k]:4d} {rl[k]:4d}")
