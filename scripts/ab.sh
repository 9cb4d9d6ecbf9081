#!/bin/bash
# A/B bench runs in ONE gpurun call: scripts/ab.sh "ENV1=.. ENV2=.." "ENV=..." ...  (use "-" for no env)
# every run is bounded (a hung kernel must not take the box's whole time limit) and a failed run stops the series
for cfg in "$@"; do
  [ "$cfg" = "-" ] && cfg=""
  env $cfg timeout -k 5 90 python bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/ab.json 2>gpurun_out/ab.err
  rc=$?
  if [ $rc -ne 0 ]; then echo "$cfg FAILED rc=$rc"; tail -3 gpurun_out/ab.err; exit 1; fi
  python -c "
import json; d=json.load(open('gpurun_out/ab.json')); print('$cfg', round(d['value']/1e6,2), 'ms', round(d['ms_per_step'],3), 'hist', round(d['config']['hist_host_ms_per_step'],3), 'sel', round(d['config']['scan_ms_per_step'],3), 'rows', d['config']['rows_scored_per_step'], 'stream', round(d['roofline']['scan_streaming']['ms'],3))"
done
