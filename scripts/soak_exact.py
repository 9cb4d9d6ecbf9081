"""Soak of the exact (stepwise) mode's fast step against the oracle: many small selections over fresh random streams, ids and
accept counts compared (a race in the step kernel's arrival / pack / event-word logic would show as a wrong member or a
hang).  python scripts/soak_exact.py [rounds]   (on a GPU box; ~1 s a round)"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import oracle  # noqa: E402
from diverseseq_amd import engine, parallel  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dev = torch.device("cuda", 0)
ctx = engine.Context(0, stream=torch.cuda.current_stream().cuda_stream)
bad = 0
t0 = time.time()
for it in range(rounds):
    rng = np.random.default_rng(10_000 + it)
    k = (6, 6, 5, 4)[it % 4]
    n = int(rng.integers(2, 14))
    nseq = int(rng.integers(1500, 5000))
    length = int(rng.integers(800, 3000))
    seqs = [rng.integers(0, 4, size=length, dtype=np.uint8) for _ in range(nseq)]
    m = ctx.build_matrix(seqs, k, 4)
    world = (1, 2, 4)[it % 3]  # (a rank's share of a sharded stream: the other ranks' positions are skipped)
    if world == 1:
        _, order = parallel.shard_order(nseq, n, 0, 1, block=32)
        sel = parallel.nmost_exact(ctx, m, order, n, dev, 1, poll_every=(4, 16)[it % 2])
        exp, acc = oracle.nmost_concat(*engine.concat(seqs), n, k, 4)
        got = sel.members(with_freqs=False).positions.tolist()
        ok = got == exp.members()[0].tolist() and sel.summary().n_accepts == acc
    else:
        # one rank of `world` alone: its own rows only -- the oracle over the same sub-stream (seeds + owned rows)
        owned, order = parallel.shard_order(nseq, n, 0, world, block=32)
        local = seqs[:n] + [seqs[int(p)] for p in owned]
        m.close()
        m = ctx.build_matrix(local, k, 4)
        sel = parallel.nmost_exact(ctx, m, order, n, dev, 1, poll_every=(4, 16)[it % 2])
        exp, acc = oracle.nmost_concat(*engine.concat(local), n, k, 4)
        pos = np.concatenate([np.arange(n), owned])
        got = sel.members(with_freqs=False).positions.tolist()
        ok = got == pos[exp.members()[0]].tolist() and sel.summary().n_accepts == acc
    if not ok:
        bad += 1
        print(f"round {it}: k={k} n={n} nseq={nseq} length={length} world={world}: MISMATCH {got}")
    sel.close()
    m.close()
print(f"{rounds} rounds, {bad} mismatches, {time.time() - t0:.1f} s")
sys.exit(1 if bad else 0)
