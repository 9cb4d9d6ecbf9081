"""One rank's share of a W-way sharded stream in the exact mode, alone on one GPU (the other ranks' positions are REMOTE and
never answer): how long its steps take when only one position in W is its own.
python scripts/experiments/exact_shard_emul.py [W ...]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from diverseseq_amd import engine, parallel  # noqa: E402

worlds = [int(x) for x in sys.argv[1:]] or [1, 2, 8]
dev = torch.device("cuda", 0)
ctx = engine.Context(0, stream=torch.cuda.current_stream().cuda_stream)
rng = np.random.default_rng(7)
n, k, nloc, L = 10, 6, 100_000, 5000
data = rng.integers(0, 4, size=(nloc + n) * L, dtype=np.uint8)
seqs = [data[i * L:(i + 1) * L] for i in range(nloc + n)]
m = ctx.build_matrix(seqs, k, 4)
for W in worlds:
    npos = n + (nloc - 256) * W
    owned, order = parallel.shard_order(npos, n, 0, W, block=256)
    assert owned.size <= nloc  # (the last block may be short: a few rows of the matrix go unused)
    ts = []
    for rep in range(6):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sel = parallel.nmost_exact(ctx, m, order, n, dev, 1)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
        s = sel.summary()
        sel.close()
    print(f"W={W}: {npos} positions, {nloc} local rows: {1e3 * np.median(ts[1:]):.3f} ms a selection (accepts {s.n_accepts}, events {s.n_events}, rows scored {s.rows_scored})")
m.close()
