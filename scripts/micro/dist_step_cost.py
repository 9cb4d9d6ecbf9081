"""Host-side cost of the pieces of a chunk-mode step with the exchange forced on ONE GPU (RCCL, world 1):
how long each call holds the host, and what the whole step takes.   gpurun -- python scripts/micro/dist_step_cost.py"""
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from diverseseq_amd import engine, parallel  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
stream = torch.cuda.Stream()
torch.cuda.set_stream(stream)
ctx = engine.Context(0, stream=stream.cuda_stream)
nseq, length, k, n = 100_000, 5_000, 6, 10
g = torch.Generator(device=dev)
g.manual_seed(20260421)
seqs = torch.randint(0, 4, (nseq * length,), dtype=torch.uint8, device=dev, generator=g)
offsets = np.arange(nseq + 1, dtype=np.uint64) * np.uint64(length)
os.environ["DVS_NO_OFFSETS_CACHE"] = "1"
buffers = {}
names = ["build", "nmost", "gather kernel", "all_gather x2", "matrix_from_device_freqs", "merge nmost", "close"]
acc = np.zeros(len(names))
tot = []
for it in range(8):
    torch.cuda.synchronize()
    t = [time.perf_counter()]
    m = ctx.build_matrix_device(seqs.data_ptr(), offsets, k, 4); t.append(time.perf_counter())
    sel = m.nmost(n); t.append(time.perf_counter())
    B = m.nbins
    if "t" not in buffers:
        buffers["t"] = (torch.empty((n, B), dtype=torch.float64, device=dev), torch.empty((n, 2), dtype=torch.float64, device=dev),
                        torch.empty((n, B), dtype=torch.float64, device=dev), torch.empty((n, 2), dtype=torch.float64, device=dev))
    t_rows, t_meta, all_rows, all_meta = buffers["t"]
    sel.gather_members(t_rows.data_ptr(), t_meta.data_ptr(), n); t.append(time.perf_counter())
    dist.all_gather_into_tensor(all_rows, t_rows)
    dist.all_gather_into_tensor(all_meta, t_meta); t.append(time.perf_counter())
    mm = ctx.matrix_from_device_freqs(all_rows.data_ptr(), n, B, all_meta.data_ptr()); t.append(time.perf_counter())
    merged = mm.nmost(n); t.append(time.perf_counter())
    merged.close(); mm.close(); sel.close(); m.close()
    torch.cuda.synchronize(); t.append(time.perf_counter())
    if it >= 3:
        acc += np.diff(t)
        tot.append(t[-1] - t[0])
cnt = len(tot)
for nm, v in zip(names, acc / cnt):
    print(f"{nm:28s} {v * 1e6:8.1f} us")
print(f"{'step':28s} {np.mean(tot) * 1e6:8.1f} us")
dist.destroy_process_group()
