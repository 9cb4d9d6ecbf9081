#!/bin/bash
# Per-kernel resource usage of the shipped library, from the code object's notes:
#   scripts/resource_usage.sh [libdvs_hip.so] > profiles/rNN_resource_usage.txt
# columns: vgprs agprs sgprs spilled-vgprs spilled-sgprs scratch-bytes-per-lane static-lds kernel
set -eu
lib=${1:-diverseseq_amd/libdvs_hip.so}
tmp=$(mktemp -d)
trap 'rm -rf $tmp' EXIT
objcopy -O binary --only-section=.hip_fatbin $lib $tmp/fat.bin
# (one offload bundle per translation unit, laid end to end in the section)
python3 - $tmp <<'PY'
import sys
d = sys.argv[1]
blob = open(d + "/fat.bin", "rb").read()
magic = b"__CLANG_OFFLOAD_BUNDLE__"
at = [i for i in range(len(blob)) if blob.startswith(magic, i)] + [len(blob)]
for n in range(len(at) - 1):
    open(f"{d}/bundle{n}.bin", "wb").write(blob[at[n]:at[n + 1]])
PY
: > $tmp/notes.txt
for b in $tmp/bundle*.bin; do
  /opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$b --output=$b.co
  /opt/rocm/lib/llvm/bin/llvm-readelf --notes $b.co >> $tmp/notes.txt
done
python3 - $tmp/notes.txt <<'PY'
import re, subprocess, sys
txt = open(sys.argv[1]).read()
rows = []
for blk in re.split(r"\n\s+- \.agpr_count:", txt)[1:]:
    blk = ".agpr_count:" + blk
    g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
    name = g("name")
    try:
        name = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip()
    except Exception:
        pass
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"\((?!anonymous).*", "", name)
    rows.append((g("vgpr_count"), g("agpr_count"), g("sgpr_count"), g("vgpr_spill_count"), g("sgpr_spill_count"),
                 g("private_segment_fixed_size"), g("group_segment_fixed_size"), name))
print("vgpr agpr sgpr vgpr_spill sgpr_spill scratch_B lds_static kernel")
for r in sorted(rows, key=lambda r: r[-1]):
    print(" ".join(f"{x:>5}" for x in r[:-1]), r[-1])
PY
