#!/usr/bin/env python3
"""Where a kernel's scratch traffic sits: scripts/spill_map.py <objdump -d file> <kernel-name-substring>
Lists every loop (backward branch) that contains scratch loads/stores, innermost first, with instruction
counts -- a spill inside a scan loop is paid per row, one in straight-line accept code once per event."""
import re
import sys

path, want = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^[0-9a-f]+ <", l) and want in l)
end = next((i for i in range(start + 1, len(lines)) if re.match(r"^[0-9a-f]+ <", lines[i])), len(lines))
ins = []  # (addr, text)
for l in lines[start + 1:end]:
    m = re.match(r"^\s+(\S.*?)\s+// ([0-9A-F]+):", l)
    if m:
        ins.append((int(m.group(2), 16), m.group(1)))
addr_ix = {a: i for i, (a, _) in enumerate(ins)}
loops = []
for i, (a, t) in enumerate(ins):
    m = re.match(r"s_cbranch_\w+\s+(\d+)|s_branch\s+(\d+)", t)
    if m:
        off = int(m.group(1) or m.group(2))
        if off >= 32768:
            off -= 65536
        tgt = a + 4 + off * 4
        if tgt <= a and tgt in addr_ix:
            loops.append((addr_ix[tgt], i))
sc = [i for i, (_, t) in enumerate(ins) if t.startswith("scratch_")]
print(f"{want}: {len(ins)} instructions, {len(sc)} scratch ops "
      f"({sum(1 for i in sc if 'load' in ins[i][1])} loads, {sum(1 for i in sc if 'store' in ins[i][1])} stores), {len(loops)} loops")
loops.sort(key=lambda r: r[1] - r[0])
seen = set()
for lo, hi in loops:
    inside = [i for i in sc if lo <= i <= hi]
    own = [i for i in inside if i not in seen]
    if own:
        ld = sum(1 for i in own if "load" in ins[i][1])
        vm = sum(1 for i in range(lo, hi + 1) if ins[i][1].startswith(("global_load", "buffer_load")))
        ds = sum(1 for i in range(lo, hi + 1) if ins[i][1].startswith("ds_"))
        print(f"  loop [{lo:6d}..{hi:6d}] len {hi - lo + 1:6d}: {len(own):4d} scratch ops not in an inner loop ({ld} loads), "
              f"{vm} global loads, {ds} LDS ops in the body")
        seen.update(own)
rest = [i for i in sc if i not in seen]
print(f"  outside every loop: {len(rest)}")
