set -e
run() { timeout -k 10 200 python bench.py "$@" --no-cpu-baseline --no-side-runs > gpurun_out/ab.log 2>&1; tail -1 gpurun_out/ab.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
echo exact one-launch; run --mode exact
echo exact three-launch; DVS_EVENT_THREE_LAUNCH=1 run --mode exact
echo multi one-launch; DVS_NO_PERSIST=1 run
echo multi three-launch; DVS_NO_PERSIST=1 DVS_EVENT_THREE_LAUNCH=1 run
