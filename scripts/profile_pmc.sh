#!/bin/bash
# SQ / GRBM counters for the two streaming kernels of the headline path, in their own rocprofv3 --pmc
# passes (never combined with other trace domains):   gpurun -- scripts/profile_pmc.sh r03_x
#   * kmer_hist_kernel: bench.py --steps 2 --no-side-runs (every launch a headline build)
#   * persist_nmost_kernel as a pure stream: scripts/micro/stream_pass.py (threshold no row reaches)
# Two passes each (8 SQ slots per pass).  Summaries: gpurun_out/<tag>/pmc_{hist,stream}_{a,b}.csv
set -u
tag=${1:-rXX}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
db() { find $out/$1 -name '*.db' 2>/dev/null | head -1; }
A="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
B="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
run() {  # name, counters, program...
  local name=$1 ctr=$2; shift; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $ctr -d $out/$name -- "$@" > $out/$name.out 2> $out/$name.err || [ -n "$(db $name)" ] || { echo "$name failed"; tail -3 $out/$name.err; return 1; }
  python3 $GRAFT_REPO_ROOT/scripts/rocpd_summary.py pmc $(db $name) | grep -E "^Kernel|kmer_hist_kernel|persist_nmost_kernel" > $out/pmc_$name.csv
  rm -rf $out/$name
}
run hist_a "$A" python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-side-runs || exit 1
run hist_b "$B" python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-side-runs || exit 1
run stream_a "$A" python3 $GRAFT_REPO_ROOT/scripts/micro/stream_pass.py || exit 1
run stream_b "$B" python3 $GRAFT_REPO_ROOT/scripts/micro/stream_pass.py || exit 1
cd $GRAFT_REPO_ROOT
for f in hist_a hist_b stream_a stream_b; do echo "== $f"; cut -c1-40,200- $out/pmc_$f.csv | head -30; done
