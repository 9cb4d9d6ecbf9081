#!/bin/bash
# A/B of environment switches over scripts/bench_configs.py cases in ONE gpurun call (same box, same build):
#   gpurun -- scripts/ab_configs.sh <tag> "<cases>" "ENV=.. ENV=.." ["ENV=.."] ...
# one jsonl per setting under gpurun_out/<tag>/ ("-" = no switch)
set -u
tag=$1; cases=$2; shift; shift
out=gpurun_out/$tag
mkdir -p $out
i=0
for cfg in "$@"; do
  c=$cfg; [ "$c" = "-" ] && c=""
  echo "== $cfg" >> $out/summary.txt
  env $c timeout -k 10 240 python scripts/bench_configs.py $cases > $out/cfg_$i.jsonl 2> $out/cfg_$i.err || { echo "failed: $cfg"; tail -3 $out/cfg_$i.err; }
  python3 - $out/cfg_$i.jsonl >> $out/summary.txt <<'PY'
import json,sys
for l in open(sys.argv[1]):
    d=json.loads(l); print({k:d[k] for k in ("config","ms","engine_ms","accepts","windows","hist_ms","sketch_ms","pairs_ms") if k in d})
PY
  i=$((i+1))
done
cat $out/summary.txt
