#!/usr/bin/env python3
"""ONE selection repeated many times over the same input, for every instantiation of the persistent kernel that
a caller can reach: do the members, their order, total_jsd (bit for bit), the accepts, the events and the
windows ever change?  (The leave-one-out partials lost in round 3 showed as 22 of 3000 repetitions; the
hand-overs have been self-describing words since round 4 -- DESIGN.md 4.3c -- and this is the evidence.)

    python scripts/repeat_selections.py [reps] [case ...] > profiles/rNN_repeat.json

Cases (T = element type of the matrix rows, CACHED = 4^k <= 4096, the kernel's template arguments):
  u16-small   <u16, CACHED, nmost, SMALL>   n = 10,  k = 6   (member rows in every workgroup's LDS)
  u16-n64     <u16, CACHED, nmost>          n = 64,  k = 6   (jobs by slot; totals polled by one wave)
  u16-n100    <u16, CACHED, nmost>          n = 100, k = 6
  u16-n200    <u16, CACHED, nmost>          n = 200, k = 6   (>= 128 members: every thread waits for its members' totals)
  u32-n100    <u32, CACHED, nmost>          n = 100, k = 6   (DVS_COUNTS_U32)
  u32-k7      <u32, !CACHED, nmost>         n = 100, k = 7   (the C4 share's shape)
  u16-max     <u16, CACHED, max>            k = 6, batches of rows
  u32-max     <u32, CACHED, max>            k = 6, sequences of several tiles
  f64-merge   <f64, CACHED, nmost>          the chunk merge: final_nmost over frequency rows, n = 100, k = 6
  f64-maxmerge <f64, CACHED, max>           final_max over frequency rows, k = 6
  f64-k7      <f64, !CACHED, nmost>         final_nmost over frequency rows at 4^7 bins
(<u16, !CACHED> is compiled but not reachable: rows beyond 4096 bins are built as 32-bit counts.)"""
import collections
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from diverseseq_amd import engine  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_configs  # noqa: E402  (its synth(): sequences generated on the device)

args = sys.argv[1:]
reps = int(args[0]) if args and args[0].isdigit() else 5000
wanted = [a for a in args if not a.isdigit()]
ctx = bench_configs.ctx


def key_of(sel):
    s = sel.summary()
    mem = sel.members(False)
    return (s.engine, s.size, s.n_accepts, s.n_arbitrated, s.n_events, s.n_windows, repr(s.total_jsd),
            hash(mem.positions.tobytes()), hash(np.asarray(mem.delta_jsd).tobytes()))


def run_case(name, make_matrix, select, what):
    seen = collections.Counter()
    t0 = time.perf_counter()
    last = t0
    for i in range(reps):
        m = make_matrix()
        sel = select(m)
        seen[key_of(sel)] += 1
        sel.close()
        m.close()
        if time.perf_counter() - last > 30:
            print(f"[{name}] {i + 1} of {reps}", file=sys.stderr, flush=True)
            last = time.perf_counter()
    top = seen.most_common(1)[0]
    rec = dict(case=name, what=what, repetitions=reps, distinct_outcomes=len(seen), divergent_runs=reps - top[1],
               engine=top[0][0], set_size=top[0][1], accepts=top[0][2], arbitrations=top[0][3], events=top[0][4],
               windows=top[0][5], total_jsd=top[0][6], seconds=round(time.perf_counter() - t0, 1))
    print(json.dumps(rec), flush=True)
    return rec


def counts_case(name, nseq, lo, hi, k, what, select, env=None, composition=False):
    if wanted and name not in wanted:
        return
    for k_, v in (env or {}).items():
        os.environ[k_] = v
    ctx.refresh_knobs()
    seqs, offs = bench_configs.synth(nseq, lo, hi, 20261004 + len(name), composition)
    run_case(name, lambda: ctx.build_matrix_device(seqs.data_ptr(), offs, k, 4), select, what)
    for k_ in (env or {}):
        del os.environ[k_]
    ctx.refresh_knobs()
    del seqs
    torch.cuda.empty_cache()


def freqs_case(name, nrows, k, what, select):
    if wanted and name not in wanted:
        return
    rng = np.random.default_rng(20261004 + len(name))
    # frequency rows as a chunk merge sees them: rows of sequences with a base composition each
    seqs, offs = bench_configs.synth(nrows, 4000, 6000, 77 + k, composition=True)
    m0 = ctx.build_matrix_device(seqs.data_ptr(), offs, k, 4)
    c = m0.counts().astype(np.float64)
    m0.close()
    f = torch.from_numpy(c / c.sum(axis=1, keepdims=True)).to(bench_configs.dev)  # (resident in HBM: wrapped, not uploaded, every time)
    torch.cuda.synchronize()
    del rng
    run_case(name, lambda: ctx.matrix_from_device_freqs(f.data_ptr(), f.shape[0], f.shape[1]), select, what)


counts_case("u16-small", 20_000, 2000, 2000, 6, "<u16, CACHED, nmost, SMALL> n=10, 20000 x 2 kb, k=6", lambda m: m.nmost(10))
counts_case("u16-n64", 12_500, 5000, 5000, 6, "<u16, CACHED, nmost> n=64, 12500 x 5 kb, k=6", lambda m: m.nmost(64))
counts_case("u16-n100", 12_500, 5000, 5000, 6, "<u16, CACHED, nmost> n=100, 12500 x 5 kb, k=6", lambda m: m.nmost(100))
counts_case("u16-n200", 12_500, 5000, 5000, 6, "<u16, CACHED, nmost> n=200, 12500 x 5 kb, k=6", lambda m: m.nmost(200))
counts_case("u32-n100", 12_500, 5000, 5000, 6, "<u32, CACHED, nmost> n=100, 12500 x 5 kb, k=6 (DVS_COUNTS_U32)",
            lambda m: m.nmost(100), env={"DVS_COUNTS_U32": "1"})
counts_case("u32-k7", 12_500, 5000, 5000, 7, "<u32, !CACHED, nmost> n=100, 12500 x 5 kb, k=7 (the C4 share's shape)",
            lambda m: m.nmost(100))
counts_case("u16-max", 2000, 8000, 12_000, 6, "<u16, CACHED, max> min_size=40, 2000 x 8-12 kb with a base composition each, k=6",
            lambda m: m.max_divergent(40, 2000, "stdev"), composition=True)
counts_case("u32-max", 900, 40_000, 60_000, 6, "<u32, CACHED, max> min_size=40, 900 x 40-60 kb with a base composition each, k=6",
            lambda m: m.max_divergent(40, 900, "stdev"), composition=True)
freqs_case("f64-merge", 1600, 6, "<f64, CACHED, nmost> final_nmost n=100 over 1600 frequency rows, k=6", lambda m: m.nmost(100))
freqs_case("f64-maxmerge", 800, 6, "<f64, CACHED, max> final_max min_size=40 over 800 frequency rows, k=6",
           lambda m: m.max_divergent(40, 800, "stdev"))
freqs_case("f64-k7", 800, 7, "<f64, !CACHED, nmost> final_nmost n=100 over 800 frequency rows, k=7", lambda m: m.nmost(100))
