#!/bin/bash
# scripts/bench_configs.py groups with the library built under different EXTRA flags, in ONE gpurun call:
#   scripts/ab_build_cfg.sh "NS C4" "-DX=0" "-DX=1" ...   ("-" = no extra flag); plain rebuild at the end
grp=$1; shift
for cfg in "$@"; do
  [ "$cfg" = "-" ] && cfg=""
  make -C diverseseq_amd/csrc clean > /dev/null; make -C diverseseq_amd/csrc -j8 EXTRA="$cfg" > gpurun_out/ab_build.log 2>&1 || { tail -3 gpurun_out/ab_build.log; exit 1; }
  timeout -k 5 200 python scripts/bench_configs.py $grp 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    r=json.loads(l); print('[$cfg]', r['config'][:24], r['ms'], r['engine_ms'])"
done
make -C diverseseq_amd/csrc clean > /dev/null; make -C diverseseq_amd/csrc -j8 > /dev/null 2>&1
