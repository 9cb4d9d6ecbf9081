#!/bin/bash
# Kernel times + SQ / GRBM counters for hash_filter_dna_kernel on C5 (1000 x 3 Mb, k = 12, s = 3000): one
# --kernel-trace --stats run and two rocprofv3 --pmc passes (never combined with other trace domains):
#   gpurun -- scripts/profile_pmc_hash.sh r04_x
set -u
tag=${1:-rXX}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
db() { find $out/$1 -name '*.db' 2>/dev/null | head -1; }
A="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
B="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/kt -- python3 $GRAFT_REPO_ROOT/scripts/bench_configs.py C5 > $out/c5.jsonl 2> $out/kt.err || [ -n "$(db kt)" ] || { echo kt failed; tail -3 $out/kt.err; exit 1; }
python3 $GRAFT_REPO_ROOT/scripts/rocpd_summary.py stats $(db kt) > $out/c5_kernel_stats.csv
rm -rf $out/kt
run() {
  local name=$1 ctr=$2; shift; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $ctr -d $out/$name -- "$@" > $out/$name.out 2> $out/$name.err || [ -n "$(db $name)" ] || { echo "$name failed"; tail -3 $out/$name.err; return 1; }
  python3 $GRAFT_REPO_ROOT/scripts/rocpd_summary.py pmc $(db $name) | grep -E "^Kernel|hash_filter" > $out/pmc_$name.csv
  rm -rf $out/$name
}
run hash_a "$A" python3 $GRAFT_REPO_ROOT/scripts/bench_configs.py C5 || exit 1
run hash_b "$B" python3 $GRAFT_REPO_ROOT/scripts/bench_configs.py C5 || exit 1
cd $GRAFT_REPO_ROOT
head -6 $out/c5_kernel_stats.csv | cut -c1-40,200-
for f in hash_a hash_b; do echo "== $f"; cut -c1-30,150- $out/pmc_$f.csv | head -24; done
