#!/bin/bash
# In-kernel phase stamps of the persistent engine in ONE gpurun call:
#   gpurun -- scripts/stamps.sh <tag> ["ENV=.. ENV=.."] [bench args]
# builds the library with -DDVS_PERSIST_STAMPS (s_memrealtime at every phase boundary, costs registers:
# the build is for this measurement only), runs bench.py for a few steps with DVS_PERSIST_DEBUG=1 and
# keeps what the library prints per launch in gpurun_out/<tag>_stamps.txt, then rebuilds the plain library.
set -u
tag=${1:-rXX}; cfg=${2:--}; shift; shift
[ "$cfg" = "-" ] && cfg=""
make -C diverseseq_amd/csrc clean > /dev/null
make -C diverseseq_amd/csrc -j8 EXTRA=-DDVS_PERSIST_STAMPS > gpurun_out/${tag}_stamps_build.log 2>&1 || { tail -5 gpurun_out/${tag}_stamps_build.log; exit 1; }
env $cfg DVS_PERSIST_DEBUG=1 timeout -k 5 120 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-side-runs "$@" > gpurun_out/${tag}_stamps_bench.json 2> gpurun_out/${tag}_stamps.txt
rc=$?
grep -c "dvs persist" gpurun_out/${tag}_stamps.txt
tail -12 gpurun_out/${tag}_stamps.txt
make -C diverseseq_amd/csrc clean > /dev/null
make -C diverseseq_amd/csrc -j8 > /dev/null 2>&1
exit $rc
