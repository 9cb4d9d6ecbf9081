#!/bin/bash
# as stamps.sh, for one of scripts/bench_configs.py's groups:  gpurun -- scripts/stamps_configs.sh <tag> C3
set -u
tag=${1:-rXX}; grp=${2:-C3}
make -C diverseseq_amd/csrc clean > /dev/null
make -C diverseseq_amd/csrc -j8 EXTRA=-DDVS_PERSIST_STAMPS > gpurun_out/${tag}_stamps_build.log 2>&1 || { tail -5 gpurun_out/${tag}_stamps_build.log; exit 1; }
env DVS_PERSIST_DEBUG=1 timeout -k 5 200 python scripts/bench_configs.py $grp > gpurun_out/${tag}_stamps_cfg.jsonl 2> gpurun_out/${tag}_stamps.txt
rc=$?
grep -c "dvs persist" gpurun_out/${tag}_stamps.txt
make -C diverseseq_amd/csrc clean > /dev/null
make -C diverseseq_amd/csrc -j8 > /dev/null 2>&1
exit $rc
