"""step time of the north-star selection under engine knobs given as KEY=VAL,... sets on the command line"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
from diverseseq_amd import engine
ctx = engine.Context(0)
n = int(os.environ.get("SWEEP_N", "10")); k = int(os.environ.get("SWEEP_K", "6"))
g = torch.Generator(device="cuda:0"); g.manual_seed(20260421)
seqs = torch.randint(0, 4, (100_000 * 5000,), dtype=torch.uint8, device="cuda:0", generator=g)
offs = np.arange(100_001, dtype=np.uint64) * np.uint64(5000)
torch.cuda.synchronize()
ctx.set_timing(True)
for spec in sys.argv[1:]:
    env = dict(kv.split("=") for kv in spec.split(",") if kv and kv != "-")
    os.environ.update(env)
    ctx.refresh_knobs()
    best = None
    for it in range(6):
        t0 = time.perf_counter()
        m = ctx.build_matrix_device(seqs.data_ptr(), offs, k, 4)
        sel = m.nmost(n)
        dt = time.perf_counter() - t0
        s = sel.summary()
        if it and (best is None or dt < best[0]):
            best = (dt, s.scan_ms, s.rows_scored, s.n_windows, s.n_accepts)
        sel.close(); m.close()
    print(f"{spec:50s} step {best[0]*1e3:.3f} ms persist {best[1]:.3f} rows {best[2]} windows {best[3]} accepts {best[4]}", flush=True)
    for k_ in env: os.environ.pop(k_, None)
    ctx.refresh_knobs()
