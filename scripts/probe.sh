#!/bin/bash
# One interval of the persistent engine's event cycle per build, two clock reads per pass and nothing else switched on:
#   gpurun -- scripts/probe.sh <tag> "1 2 3 4 5 6" [bench_configs group, default "": bench.py's headline] ["ENV=1,ENV2=1 ..."]
# the optional fourth argument lists environment settings (comma-separated variables per setting, "-" for none) that
# every probed build is run with, one after the other
# -> gpurun_out/<tag>_probe.txt
set -u
tag=${1:-rXX}; ids=${2:-"1 2 3 4 5 6"}; grp=${3:-}; envs=${4:--}
: > gpurun_out/${tag}_probe.txt
for k in $ids; do
 for e in $envs; do
  ev=$(echo $e | tr ',' ' '); [ "$e" = "-" ] && ev=""
  [ "$envs" != "-" ] && echo "-- env: $e" >> gpurun_out/${tag}_probe.txt
  if [ "$e" = "$(echo $envs | cut -d' ' -f1)" ]; then
   rm -f diverseseq_amd/csrc/build/persist.hip.o diverseseq_amd/csrc/build/select.hip.o   # (the two units that see the switch)
   make -C diverseseq_amd/csrc -j8 EXTRA="-DDVS_PERSIST_STAMPS -DDVS_PROBE=$k" > gpurun_out/${tag}_probe_build.log 2>&1 || { tail -5 gpurun_out/${tag}_probe_build.log; exit 1; }
  fi
  if [ -z "$grp" ]; then
    env $ev DVS_PERSIST_DEBUG=1 timeout -k 5 120 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-side-runs 2> gpurun_out/${tag}_probe_$k.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('probe $k: scan_ms_per_step', round(d['config']['scan_ms_per_step'],4))" >> gpurun_out/${tag}_probe.txt
  else
    env $ev DVS_PERSIST_DEBUG=1 timeout -k 5 200 python scripts/bench_configs.py $grp > /dev/null 2> gpurun_out/${tag}_probe_$k.err
  fi
  grep "dvs persist probe\|\] .* launch" gpurun_out/${tag}_probe_$k.err | tail -4 >> gpurun_out/${tag}_probe.txt
 done
done
rm -f diverseseq_amd/csrc/build/persist.hip.o diverseseq_amd/csrc/build/select.hip.o
make -C diverseseq_amd/csrc -j8 > /dev/null 2>&1
cat gpurun_out/${tag}_probe.txt
