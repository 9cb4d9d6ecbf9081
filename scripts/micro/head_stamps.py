"""per-phase times of the head kernel (build with EXTRA=-DDVS_HEAD_STAMPS, run with DVS_PERSIST_DEBUG=1)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
from diverseseq_amd import engine
ctx = engine.Context(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
g = torch.Generator(device="cuda:0"); g.manual_seed(20260421)
seqs = torch.randint(0, 4, (100_000 * 5000,), dtype=torch.uint8, device="cuda:0", generator=g)
offs = np.arange(100_001, dtype=np.uint64) * np.uint64(5000)
torch.cuda.synchronize()
ctx.set_timing(True)
for it in range(3):
    t0 = time.perf_counter()
    m = ctx.build_matrix_device(seqs.data_ptr(), offs, 6, 4)
    sel = m.nmost(n)
    dt = time.perf_counter() - t0
    s = sel.summary()
    print(f"step {dt*1e3:.3f} ms persist {s.scan_ms:.3f} head {s.head_ms:.3f} head_rows {s.head_rows} head_accepts {s.head_accepts} accepts {s.n_accepts}", file=sys.stderr)
    sel.close(); m.close()
