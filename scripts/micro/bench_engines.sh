# the selection engines side by side on the headline workload (n = 10 and 100): persistent, multi-launch
# (DVS_NO_PERSIST=1) and the stepwise exact mode at world 1.   gpurun -- bash scripts/micro/bench_engines.sh
run() { timeout -k 10 200 python bench.py "$@" --no-cpu-baseline --no-side-runs > gpurun_out/eng.log 2>&1; tail -1 gpurun_out/eng.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  %.2f M seq/s  %.3f ms' % (d['value']/1e6, d['ms_per_step']))"; }
for n in 10 100; do
echo n=$n persistent; run -n $n
echo n=$n multi-launch; DVS_NO_PERSIST=1 run -n $n
echo n=$n exact mode, world 1; run --mode exact -n $n
done
