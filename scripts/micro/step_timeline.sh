set -u
out=$GRAFT_REPO_ROOT/gpurun_out/tl
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d $out/kt -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-side-runs --steps 5 --warmup 1 > $out/b.json 2> $out/b.err
db=$(find $out/kt -name '*.db' | head -1)
cd $GRAFT_REPO_ROOT
python3 scripts/micro/step_timeline.py $db 2 | tee $out/timeline.txt
rm -rf $out/kt
