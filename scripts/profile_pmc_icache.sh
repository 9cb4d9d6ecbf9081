#!/bin/bash
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r04_icache
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
db() { find $out/$1 -name '*.db' 2>/dev/null | head -1; }
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  name=$(echo $set | cut -c1-12 | tr ' ' '_')
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set -d $out/$name -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-side-runs > $out/$name.out 2> $out/$name.err || [ -n "$(db $name)" ] || { echo "$name failed"; tail -3 $out/$name.err; continue; }
  python3 $GRAFT_REPO_ROOT/scripts/rocpd_summary.py pmc $(db $name) | grep -E "^Kernel|persist_nmost_kernel" > $out/pmc_$name.csv
  rm -rf $out/$name
  cut -c1-50,190- $out/pmc_$name.csv
done
