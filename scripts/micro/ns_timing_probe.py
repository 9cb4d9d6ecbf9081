"""north-star n=10 through scripts/bench_configs.py's path, with and without the engine's launch timing"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench_configs as bc
seqs, offs = bc.synth(100_000, 5000, 5000, 20260421 + len("north-star n=10"))
ctx = bc.ctx
for timing in (True, False, True, False):
    ctx.set_timing(timing)
    ts = []
    for rep in range(6):
        t0 = time.perf_counter()
        m = ctx.build_matrix_device(seqs.data_ptr(), offs, 6, 4)
        sel = m.nmost(10)
        s = sel.summary()
        ts.append(time.perf_counter() - t0)
        sel.close(); m.close()
    print("timing", timing, [round(t * 1e3, 3) for t in ts], "launches", s.scan_launches, flush=True)
