# repeats the configuration groups in front of C4 the way the profile round runs them: do the counters of the
# C4 share (accepts, arbitrations, launches) ever change from run to run?
for i in 1 2 3 4 5; do DVS_PERSIST_DEBUG=1 timeout -k 10 200 python scripts/bench_configs.py C2 NS C4 2> gpurun_out/c4dbg_$i.err | python -c "
import sys,json
for l in sys.stdin:
    r=json.loads(l); print(r['config'][:22], r['ms'], r['accepts'], r['arbitrations'], r['launches'], r['events'])"; grep "ended early" gpurun_out/c4dbg_$i.err | sort | uniq -c | sort -rn | head -3; done
