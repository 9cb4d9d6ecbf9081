#!/usr/bin/env python3
"""Host-side breakdown of one bench step (wall-clock per call, averaged)."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from diverseseq_amd import engine  # noqa: E402

N, L, K, n = 100000, 5000, 6, 10
dev = torch.device("cuda:0")
g = torch.Generator(device=dev)
g.manual_seed(1)
seqs = torch.randint(0, 4, (N * L,), dtype=torch.uint8, device=dev, generator=g)
offsets = np.arange(N + 1, dtype=np.uint64) * L
torch.cuda.synchronize()
ctx = engine.Context(0)
acc = {}
for it in range(12):
    t = [time.perf_counter()]
    m = ctx.build_matrix_device(seqs.data_ptr(), offsets, K, 4); t.append(time.perf_counter())
    sel = m.nmost(n); t.append(time.perf_counter())
    s = sel.summary(); t.append(time.perf_counter())
    sel.close(); t.append(time.perf_counter())
    m.close(); t.append(time.perf_counter())
    if it >= 2:
        for name, a, b in zip(["build", "nmost", "summary", "sel.close", "m.close"], t, t[1:]):
            acc[name] = acc.get(name, 0.0) + (b - a) * 1e6 / 10
print({k: round(v, 1) for k, v in acc.items()}, "total", round(sum(acc.values()), 1), "us")
