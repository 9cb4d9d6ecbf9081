#!/usr/bin/env python3
"""Register pressure along a kernel: highest VGPR index touched per block of instructions, with the
scratch ops of the block.  scripts/vgpr_profile.py <objdump -d file> <kernel-name-substring> [block]"""
import re
import sys

path, want = sys.argv[1], sys.argv[2]
blk = int(sys.argv[3]) if len(sys.argv) > 3 else 250
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^[0-9a-f]+ <", l) and want in l)
end = next((i for i in range(start + 1, len(lines)) if re.match(r"^[0-9a-f]+ <", lines[i])), len(lines))
ins = [m.group(1) for l in lines[start + 1:end] if (m := re.match(r"^\s+(\S.*?)\s+// [0-9A-F]+:", l))]
for b in range(0, len(ins), blk):
    hi = 0
    sc = 0
    gl = 0
    for t in ins[b:b + blk]:
        for m in re.finditer(r"\bv(\d+)\b|v\[(\d+):(\d+)\]", t):
            hi = max(hi, int(m.group(1) or m.group(3)))
        sc += t.startswith("scratch_")
        gl += t.startswith("global_load")
    print(f"{b:6d} maxv {hi:3d} scratch {sc:3d} gload {gl:3d} " + "#" * (hi // 4))
