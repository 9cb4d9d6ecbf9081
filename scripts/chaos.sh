#!/bin/bash
# The persistent engine with pseudo-randomly delayed workgroups (a -DDVS_PERSIST_CHAOS build: persist.hip P_CHAOS), in ONE
# gpurun call:   gpurun -- scripts/chaos.sh <tag> [repetitions per case]
# runs the selection tests of the parity suite and scripts/repeat_selections.py against it, then rebuilds the plain library.
set -u
tag=${1:-rXX}; reps=${2:-300}
make -C diverseseq_amd/csrc clean > /dev/null
make -C diverseseq_amd/csrc -j8 EXTRA=-DDVS_PERSIST_CHAOS > gpurun_out/${tag}_chaos_build.log 2>&1 || { tail -5 gpurun_out/${tag}_chaos_build.log; exit 1; }
timeout -k 10 600 python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py -x -q -k "not ingest and not mash and not sketch and not c5" > gpurun_out/${tag}_chaos_tests.log 2>&1
echo "tests rc=$?"; tail -2 gpurun_out/${tag}_chaos_tests.log
timeout -k 10 500 python scripts/repeat_selections.py $reps > gpurun_out/${tag}_chaos_repeat.jsonl 2> gpurun_out/${tag}_chaos_repeat.err
echo "repeat rc=$?"
python3 - gpurun_out/${tag}_chaos_repeat.jsonl <<'PY'
import json,sys
for l in open(sys.argv[1]):
    d=json.loads(l); print(d['case'], d['repetitions'], 'distinct', d['distinct_outcomes'], 'divergent', d['divergent_runs'], 'engine', d['engine'], 's', d['seconds'])
PY
make -C diverseseq_amd/csrc clean > /dev/null
make -C diverseseq_amd/csrc -j8 > /dev/null 2>&1
