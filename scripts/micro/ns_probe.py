import sys, os, time, numpy as np, torch
sys.path.insert(0, os.getcwd())
from diverseseq_amd import engine
dev = torch.device("cuda:0"); ctx = engine.Context(0); ctx.set_timing(True)
nseq, L = 100_000, 5000
g = torch.Generator(device=dev); g.manual_seed(1)
seqs = torch.randint(0, 4, (nseq * L + 16,), dtype=torch.uint8, device=dev, generator=g)
offsets = np.arange(nseq + 1, dtype=np.uint64) * np.uint64(L)
torch.cuda.synchronize()
def run():
    m = ctx.build_matrix_device(seqs.data_ptr(), offsets, 6, 4)
    sel = m.nmost(10); s = sel.summary(); sel.close(); m.close(); return s
for mode in ("plain", "after_waited_build"):
    ts = []
    for i in range(6):
        if mode == "after_waited_build":
            m = ctx.build_matrix_device(seqs.data_ptr(), offsets, 6, 4); ctx.sync(); m.close()
        t0 = time.perf_counter(); s = run(); ctx.sync(); ts.append((time.perf_counter() - t0) * 1e3)
    print(mode, [round(t, 3) for t in ts], "engine_ms", round(s.scan_ms, 3), "launches", s.scan_launches)
