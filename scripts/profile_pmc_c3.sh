#!/bin/bash
# SQ / GRBM counters for persist_nmost_kernel on C5 (1000 sketches of 3000), two rocprofv3 --pmc passes:
#   gpurun -- scripts/profile_pmc_pairs.sh r03_x
set -u
tag=${1:-rXX}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
db() { find $out/$1 -name '*.db' 2>/dev/null | head -1; }
A="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
B="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
run() {
  local name=$1 ctr=$2; shift; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $ctr -d $out/$name -- "$@" > $out/$name.out 2> $out/$name.err || [ -n "$(db $name)" ] || { echo "$name failed"; tail -3 $out/$name.err; return 1; }
  python3 $GRAFT_REPO_ROOT/scripts/rocpd_summary.py pmc $(db $name) | grep -E "^Kernel|persist_nmost_kernel" > $out/pmc_$name.csv
  rm -rf $out/$name
}
run c3_a "$A" python3 $GRAFT_REPO_ROOT/scripts/bench_configs.py C3 || exit 1
run c3_b "$B" python3 $GRAFT_REPO_ROOT/scripts/bench_configs.py C3 || exit 1
cd $GRAFT_REPO_ROOT
for f in c3_a c3_b; do echo "== $f"; cut -c1-30,180- $out/pmc_$f.csv | head -30; done
