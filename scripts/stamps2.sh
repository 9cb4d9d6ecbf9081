#!/bin/bash
# In-kernel phase stamps and window traces of the persistent engine for several environment settings in ONE gpurun call:
#   gpurun -- scripts/stamps2.sh <tag> "- DVS_PERSIST_NO_MBOX=1" [bench args]
# (a -DDVS_PERSIST_STAMPS build of persist.hip / select.hip for the measurement, the plain library afterwards)
set -u
tag=${1:-rXX}; envs=${2:--}; shift; shift
rm -f diverseseq_amd/csrc/build/persist.hip.o diverseseq_amd/csrc/build/select.hip.o
make -C diverseseq_amd/csrc -j8 EXTRA=-DDVS_PERSIST_STAMPS > gpurun_out/${tag}_stamps_build.log 2>&1 || { tail -5 gpurun_out/${tag}_stamps_build.log; exit 1; }
i=0
for e in $envs; do
  ev=$(echo $e | tr ',' ' '); [ "$e" = "-" ] && ev=""
  env $ev DVS_PERSIST_DEBUG=1 timeout -k 5 120 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-side-runs "$@" > gpurun_out/${tag}_stamps_bench_$i.json 2> gpurun_out/${tag}_stamps_$i.txt
  echo "== env: $e" >> gpurun_out/${tag}_stamps.txt
  tail -40 gpurun_out/${tag}_stamps_$i.txt | grep "trace\|block 0\]" >> gpurun_out/${tag}_stamps.txt
  i=$((i+1))
done
rm -f diverseseq_amd/csrc/build/persist.hip.o diverseseq_amd/csrc/build/select.hip.o
make -C diverseseq_amd/csrc -j8 > /dev/null 2>&1
cat gpurun_out/${tag}_stamps.txt
