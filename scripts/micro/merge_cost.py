#!/usr/bin/env python3
"""Cost of the chunk merge (final_nmost over the winners of G chunks) on one GPU: the winners of G
independent selections are merged as merge_nmost does after the all_gather (device rows)."""
import sys, time, numpy as np, torch
sys.path.insert(0, ".")
from diverseseq_amd import engine
G, N, L, K, n = 8, 20000, 5000, 6, 10
dev = torch.device("cuda:0")
ctx = engine.Context(0)
rows = []
for r in range(G):
    g = torch.Generator(device=dev); g.manual_seed(100 + r)
    seqs = torch.randint(0, 4, (N * L,), dtype=torch.uint8, device=dev, generator=g)
    m = ctx.build_matrix_device(seqs.data_ptr(), np.arange(N + 1, dtype=np.uint64) * L, K, 4)
    sel = m.nmost(n)
    rows.append(sel.members(with_freqs=True).kfreqs)
    sel.close(); m.close()
allr = torch.from_numpy(np.vstack(rows)).to(dev)
meta = torch.ones((G * n, 2), dtype=torch.float64, device=dev)
torch.cuda.synchronize()
for world in (1, 2, 4, 8):
    ts = []
    for it in range(6):
        t0 = time.perf_counter()
        m = ctx.matrix_from_device_freqs(allr.data_ptr(), world * n, 4 ** K, meta.data_ptr())
        sel = m.nmost(n); s = sel.summary()
        t1 = time.perf_counter()
        sel.close(); m.close()
        if it >= 2: ts.append(t1 - t0)
    print(f"merge of {world} x {n} rows: {min(ts)*1e6:.0f} us, accepts {s.n_accepts}, engine {s.engine}")
