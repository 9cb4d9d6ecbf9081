#!/bin/bash
# Round 5's evidence in one gpurun call:  gpurun -- scripts/profile_r05.sh r05_x
#   bench line + rocprofv3 kernel stats + FETCH_SIZE / WRITE_SIZE passes (scripts/profile_round.sh), the other
#   configurations (scripts/bench_configs.py), C5 / ingest kernel stats (scripts/profile_other.sh) and the sketch
#   call's breakdown
set -u
tag=${1:-r05_x}
cd $GRAFT_REPO_ROOT
bash scripts/profile_round.sh $tag > gpurun_out/${tag}_round.log 2>&1 || { tail -5 gpurun_out/${tag}_round.log; exit 1; }
timeout -k 10 200 python3 bench.py --steps 20 --warmup 3 > gpurun_out/$tag/bench_20.json 2> gpurun_out/$tag/bench_20.err || { tail -3 gpurun_out/$tag/bench_20.err; exit 1; }
timeout -k 10 400 python3 scripts/bench_configs.py C2 NS C4 C3 C3K7 ING C5 > gpurun_out/$tag/configs.jsonl 2> gpurun_out/$tag/configs.err || { tail -3 gpurun_out/$tag/configs.err; exit 1; }
bash scripts/profile_other.sh $tag > gpurun_out/${tag}_other.log 2>&1 || { tail -5 gpurun_out/${tag}_other.log; exit 1; }
timeout -k 10 200 python3 scripts/micro/sketch_call_breakdown.py > gpurun_out/$tag/sketch_call.json 2> gpurun_out/$tag/sketch_call.err || { tail -3 gpurun_out/$tag/sketch_call.err; exit 1; }
python3 scripts/benchline.py gpurun_out/$tag/bench_20.json
cat gpurun_out/$tag/sketch_call.json
cut -c1-220 gpurun_out/$tag/configs.jsonl
