#!/bin/bash
# A/B of one environment switch on one box:   gpurun -- scripts/ab_env.sh <tag> DVS_PERSIST_NO_MBOX [reps]
# bench.py (headline, --steps 20 --warmup 3, no side runs) and scripts/bench_configs.py (C2, north-star n=100, C4)
# with the switch off / on, alternating, `reps` times.  Output: gpurun_out/<tag>/ab.txt
set -u
tag=${1:-ab}; var=${2:-DVS_PERSIST_NO_MBOX}; reps=${3:-2}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
: > $out/ab.txt
for r in $(seq 1 $reps); do
  for v in 0 1; do
    if [ $v = 1 ]; then export $var=1; else unset $var; fi
    echo "== $var=$v rep $r" >> $out/ab.txt
    timeout -k 10 200 python3 bench.py --steps 20 --warmup 3 --no-side-runs --no-cpu-baseline > $out/bench_${v}_$r.json 2> $out/bench_${v}_$r.err || { echo "bench failed ($var=$v)"; tail -5 $out/bench_${v}_$r.err; exit 1; }
    python3 scripts/benchline.py $out/bench_${v}_$r.json >> $out/ab.txt
    timeout -k 10 300 python3 scripts/bench_configs.py C2 NS C4 2> $out/cfg_${v}_$r.err | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    print('  %-28s %8.3f ms  engine_ms %8.3f accepts %4d windows %4d engine %d arb %d' % (d['config'][:28], d['ms'], d['engine_ms'], d['accepts'], d['windows'], d['engine'], d['n_arbitrated']))
" >> $out/ab.txt || { echo "configs failed ($var=$v)"; tail -5 $out/cfg_${v}_$r.err; exit 1; }
  done
done
cat $out/ab.txt
