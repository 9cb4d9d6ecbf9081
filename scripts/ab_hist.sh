#!/bin/bash
# the histogram on its own: bench.py's side run (100k x 5 kb, k=6, bytes and packed input) and C3 scaled / C3 at k=7
python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-large-point 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  headline hist alone %.4f ms (frac %.3f)' % (d['roofline']['histogram_alone']['ms'], d['roofline']['histogram_alone']['frac']))"
python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-large-point --input packed 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  packed   hist alone %.4f ms (frac %.3f)' % (d['roofline']['histogram_alone']['ms'], d['roofline']['histogram_alone']['frac']))"
python3 scripts/bench_configs.py C3 C3K7 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('  %-40s hist %.3f ms total %.3f ms' % (d['config'][:40], d['hist_ms'], d['ms']))"
