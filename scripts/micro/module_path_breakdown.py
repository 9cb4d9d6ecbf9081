#!/usr/bin/env python3
"""Where the time of `_dvs.nmost_divergent(store, n, k, seqids)` goes for 100 000 x 5 kb sequences held on the
host (bench.py's `through_dvs_module` side line): the host's packing rate by thread count, the PCIe copy of the
packed planes in one piece and in the upload's chunk sizes, the packed upload as the library runs it, the
build, and a cProfile of one whole call.   python scripts/micro/module_path_breakdown.py > gpurun_out/module_path.txt"""
import cProfile
import ctypes as C
import os
import pstats
import subprocess
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from diverseseq_amd import _dvs, _lib, engine  # noqa: E402

print(subprocess.run("lscpu | grep -E 'Model name|^CPU\\(s\\)|Thread|Socket|NUMA node\\(s\\)'; nproc", shell=True, capture_output=True,
                     text=True).stdout, flush=True)
L = _lib.load()
L.dvs_pack_bases.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
L.dvs_pack_bases.restype = None
nseq, length = 100_000, 5000
n = nseq * length
src = np.random.default_rng(1).integers(0, 4, n, dtype=np.uint8)
codes = np.zeros(n // 16 + 4, np.uint32)
mask = np.zeros(n // 16 + 8, np.uint16)


def pack_only(nthr, chunk=4 << 20):
    nch = (n + chunk - 1) // chunk
    nxt = [0]
    lock = threading.Lock()

    def work():
        while True:
            with lock:
                c = nxt[0]
                nxt[0] += 1
            if c >= nch:
                return
            a = c * chunk
            L.dvs_pack_bases(src.ctypes.data + a, min(chunk, n - a), codes.ctypes.data + a // 4, mask.ctypes.data + a // 8)

    ts = [threading.Thread(target=work) for _ in range(nthr)]
    t0 = time.perf_counter()
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    return time.perf_counter() - t0


for nthr in (1, 2, 4, 8, 12, 16):
    best = min(pack_only(nthr) for _ in range(3))
    print(f"pack only, {nthr:2d} threads: {best * 1e3:7.2f} ms  {n / best / 1e9:6.1f} GB/s of bases", flush=True)

dev = torch.device("cuda:0")
pin = torch.empty(n // 16 * 6, dtype=torch.uint8).pin_memory()
dst = torch.empty_like(pin, device=dev)
for piece in (pin.numel(), 6 << 20, 3 << 20, 3 << 19, 3 << 18):
    best = None
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for a in range(0, pin.numel(), piece):
            dst[a:a + piece].copy_(pin[a:a + piece], non_blocking=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None or dt < best else best
    print(f"H2D of the packed planes ({pin.numel() / 1e6:.0f} MB) from pinned memory in pieces of {piece / 1e6:7.2f} MB: "
          f"{best * 1e3:6.2f} ms  {pin.numel() / best / 1e9:5.1f} GB/s", flush=True)

ctx = engine.default_context()
offsets = (np.arange(nseq + 1, dtype=np.uint64) * np.uint64(length))
for name, fn in (("pack_sequences (host -> planes in HBM)", lambda: ctx.pack_host(src) if hasattr(ctx, "pack_host") else None),
                 ("build_matrix_concat (upload + histogram)", lambda: ctx.build_matrix_concat(src, offsets, 6, 4))):
    best = None
    for _ in range(4):
        t0 = time.perf_counter()
        r = fn()
        ctx.sync()
        dt = time.perf_counter() - t0
        if r is not None:
            r.close()
        best = dt if best is None or dt < best else best
    print(f"{name}: {best * 1e3:6.2f} ms", flush=True)

store = _dvs.make_zarr_store()
for i in range(nseq):
    store.write(f"s{i:06d}", src[i * length:(i + 1) * length].tobytes())
ids = [f"s{i:06d}" for i in range(nseq)]
for _ in range(3):
    t0 = time.perf_counter()
    _dvs.nmost_divergent(store, 10, 6, seqids=ids)
    print(f"nmost_divergent: {(time.perf_counter() - t0) * 1e3:6.2f} ms", flush=True)
t0 = time.perf_counter()
g = _dvs._gather(store, ids)
print(f"_gather: {(time.perf_counter() - t0) * 1e3:6.2f} ms", flush=True)
del g
pr = cProfile.Profile()
pr.enable()
_dvs.nmost_divergent(store, 10, 6, seqids=ids)
pr.disable()
pstats.Stats(pr, stream=sys.stdout).sort_stats("cumulative").print_stats(18)
