#!/bin/bash
# Kernel statistics of the exact (stepwise, row-sharded) mode at world 1, old step kernels and the fast step:
#   gpurun -- scripts/profile_exact.sh <tag>     -> gpurun_out/<tag>/{old,fast}_kernel_stats.csv
set -u
tag=${1:-exact}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for v in old fast; do
  if [ $v = old ]; then export DVS_NO_FAST_STEP=1; else unset DVS_NO_FAST_STEP; fi
  timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $out/prof_$v -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --mode exact --no-side-runs --no-cpu-baseline > $out/bench_$v.json 2> $out/bench_$v.err || [ -n "$(find $out/prof_$v -name '*.db' | head -1)" ] || { echo "profile failed ($v)"; tail -5 $out/bench_$v.err; exit 1; }
  python3 $GRAFT_REPO_ROOT/scripts/rocpd_summary.py stats $(find $out/prof_$v -name '*.db' | head -1) > $out/${v}_kernel_stats.csv
  python3 $GRAFT_REPO_ROOT/scripts/rocpd_summary.py gaps $(find $out/prof_$v -name '*.db' | head -1) > $out/${v}_kernel_gaps.csv
  rm -rf $out/prof_$v
  echo "== $v"; cut -c1-160 $out/${v}_kernel_stats.csv | head -12
done
