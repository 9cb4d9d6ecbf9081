#!/bin/bash
# mash_pairs_kernel built with different trip lengths, C5 timed for each (one gpurun call)
set -u
cd $GRAFT_REPO_ROOT
for W in "$@"; do
  touch diverseseq_amd/csrc/mash.hip
  make -C diverseseq_amd/csrc EXTRA=-DDVS_PAIR_W=$W > gpurun_out/ab_pairs_build.log 2>&1 || { tail -5 gpurun_out/ab_pairs_build.log; exit 1; }
  echo "== W=$W"
  python scripts/bench_configs.py C5 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print(r['config'], 'pairs_ms', r['pairs_ms'], 'kept', r['pairs_ms_into_a_kept_matrix'], 'host', r['pairs_ms_from_host_sketches'])"
done
