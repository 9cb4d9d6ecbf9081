#!/bin/bash
# as stamps.sh, for groups of scripts/bench_configs.py:  gpurun -- scripts/stamps_configs.sh <tag> "C4 NS" ["ENV=.."]
# (one -DDVS_PERSIST_STAMPS build, one run per group, then the plain library again)
set -u
tag=${1:-rXX}; grps=${2:-C3}; cfg=${3:-}
make -C diverseseq_amd/csrc clean > /dev/null
make -C diverseseq_amd/csrc -j8 EXTRA=-DDVS_PERSIST_STAMPS > gpurun_out/${tag}_stamps_build.log 2>&1 || { tail -5 gpurun_out/${tag}_stamps_build.log; exit 1; }
rc=0
for grp in $grps; do
  env $cfg DVS_PERSIST_DEBUG=1 timeout -k 5 200 python scripts/bench_configs.py $grp > gpurun_out/${tag}_${grp}_stamps_cfg.jsonl 2> gpurun_out/${tag}_${grp}_stamps.txt || rc=$?
  grep -c "dvs persist" gpurun_out/${tag}_${grp}_stamps.txt
done
make -C diverseseq_amd/csrc clean > /dev/null
make -C diverseseq_amd/csrc -j8 > /dev/null 2>&1
exit $rc
