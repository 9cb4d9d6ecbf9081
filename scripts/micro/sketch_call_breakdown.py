#!/usr/bin/env python3
"""Where a C5 sketch call's time goes (1000 x ~3 Mb, k=12, s=3000; sequences resident in HBM): the build that leaves
the sketches in HBM (set-up + kernels + the round's status read-back), the copy of the 12 MB of sketches to pageable host
memory, and the one-call form dvs_mash_sketch that does both.  Kernel time itself: scripts/profile_other.sh (rocprofv3)."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from diverseseq_amd import _lib, distance, engine  # noqa: E402

ctx = engine.Context(0)
nseq, k, s = 1000, 12, 3000
rng = np.random.default_rng(777)
lens = rng.integers(2_900_000, 3_100_001, size=nseq, dtype=np.int64)
offsets = np.zeros(nseq + 1, dtype=np.uint64)
offsets[1:] = np.cumsum(lens)
g = torch.Generator(device="cuda:0")
g.manual_seed(777)
seqs = torch.randint(0, 4, (int(offsets[-1]) + 16,), dtype=torch.uint8, device="cuda:0", generator=g)
torch.cuda.synchronize()


def best(fn, reps=4):
    fn()
    ctx.sync()
    t = []
    for _ in range(reps):
        t0 = time.perf_counter()
        r = fn()
        ctx.sync()
        t.append(time.perf_counter() - t0)
        if hasattr(r, "close"):
            r.close()
    return min(t) * 1e3


def build():
    return distance.Sketches(None, k, s, 4, False, ctx=ctx, dev_ptr=seqs.data_ptr(), offsets=offsets)


sk_host = np.zeros((nseq, s), dtype=np.uint32)
ln_host = np.zeros(nseq, dtype=np.uint32)


def one_call():
    ctx.check(ctx._L.dvs_mash_sketch(ctx._h, C.c_void_p(seqs.data_ptr()), 1, _lib.ptr(offsets, C.c_uint64), nseq, k, s, 4, 0,
                                     _lib.ptr(sk_host, C.c_uint32), _lib.ptr(ln_host, C.c_uint32)))


kept = build()
ctx.sync()
out = {"config": "C5 sketch call breakdown", "nseq": nseq, "k": k, "sketch_size": s,
       "build_left_in_hbm_ms": round(best(build), 3),
       "copy_12MB_to_pageable_host_ms": round(best(lambda: kept.to_host()), 3),
       "dvs_mash_sketch_one_call_ms": round(best(one_call), 3)}
kept.close()
print(json.dumps(out))
