#!/bin/bash
# bench.py with the library built under different EXTRA flags, in ONE gpurun call:
#   scripts/ab_build.sh "-DX=0" "-DX=1" ...   ("-" = no extra flag); the plain library is rebuilt at the end
for cfg in "$@"; do
  [ "$cfg" = "-" ] && cfg=""
  make -C diverseseq_amd/csrc clean > /dev/null; make -C diverseseq_amd/csrc -j8 EXTRA="$cfg" > gpurun_out/ab_build.log 2>&1 || { tail -3 gpurun_out/ab_build.log; exit 1; }
  for rep in 1 2; do
  timeout -k 5 90 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-side-runs > gpurun_out/ab.json 2>gpurun_out/ab.err || { tail -3 gpurun_out/ab.err; exit 1; }
  python -c "
import json; d=json.load(open('gpurun_out/ab.json')); print('[$cfg]', round(d['value']/1e6,2), 'ms', round(d['ms_per_step'],3), 'sel', round(d['config']['scan_ms_per_step'],3))"
  done
done
make -C diverseseq_amd/csrc clean > /dev/null; make -C diverseseq_amd/csrc -j8 > /dev/null 2>&1
