#!/usr/bin/env python3
"""Host-timed histogram build of the north-star input (100k x 5 kb, k=6); DVS_HIST_THREADS sweeps."""
import sys, time, numpy as np, torch
sys.path.insert(0, ".")
from diverseseq_amd import engine
N, L, K = 100000, 5000, 6
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(1)
seqs = torch.randint(0, 4, (N * L,), dtype=torch.uint8, device=dev, generator=g)
offsets = np.arange(N + 1, dtype=np.uint64) * L
torch.cuda.synchronize()
ctx = engine.Context(0)
ts = []
for it in range(8):
    t0 = time.perf_counter(); m = ctx.build_matrix_device(seqs.data_ptr(), offsets, K, 4); t1 = time.perf_counter(); m.close()
    if it >= 2: ts.append(t1 - t0)
print("build us", round(min(ts) * 1e6, 1))
