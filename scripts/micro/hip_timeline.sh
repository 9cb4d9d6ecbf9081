#!/bin/bash
# host + device timeline of one bench step (every HIP call longer than 3 us and every kernel):
#   gpurun -- scripts/micro/hip_timeline.sh
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/htl
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --hip-trace --kernel-trace -d $out/kt -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-side-runs --steps 5 --warmup 1 > $out/b.json 2> $out/b.err
db=$(find $out/kt -name '*.db' | head -1)
cd $GRAFT_REPO_ROOT
python3 scripts/micro/hip_timeline.py $db | tee $out/timeline.txt
rm -rf $out/kt
