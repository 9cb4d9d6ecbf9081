#!/bin/bash
# rocprofv3 kernel statistics of the exact (row-sharded, stepwise) mode at world 1.
#   gpurun -- scripts/profile_exact.sh r02_x
set -u
tag=${1:-rXX}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
db() { find $out/$1 -name '*.db' 2>/dev/null | head -1; }
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $out/kt_exact -- python3 $GRAFT_REPO_ROOT/bench.py --mode exact --no-cpu-baseline --no-side-runs --steps 5 --warmup 1 > $out/exact.json 2> $out/exact.err || [ -n "$(db kt_exact)" ] || { echo exact failed; tail -3 $out/exact.err; exit 1; }
cd $GRAFT_REPO_ROOT
python3 scripts/rocpd_summary.py stats $(db kt_exact) > $out/exact_kernel_stats.csv
rm -rf $out/kt_exact
tail -1 $out/exact.json | cut -c1-200
head -12 $out/exact_kernel_stats.csv | cut -c1-160
