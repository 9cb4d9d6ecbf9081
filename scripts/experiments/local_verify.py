"""(needs r05_local_argmin_small_sets.patch applied.)  The persistent engine's local decisions against the exchange's, on the north-star shape and a few others:
DVS_PERSIST_VERIFY_LOCAL=1 python scripts/experiments/local_verify.py"""
import sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from diverseseq_amd import engine
ctx = engine.default_context()
rng = np.random.default_rng(3)
for nseq, L, n in ((100000, 5000, 10), (30000, 1200, 13), (50000, 3000, 4), (20000, 800, 2)):
    data = rng.integers(0, 4, size=nseq * L, dtype=np.uint8)
    seqs = [data[i * L:(i + 1) * L] for i in range(nseq)]
    m = ctx.build_matrix(seqs, 6, 4)
    sel = m.nmost(n)
    s = sel.summary()
    print(f"{nseq} x {L}, n={n}: engine {s.engine} accepts {s.n_accepts} local {s.n_local_decisions} exchanged {s.n_exchanged_decisions} "
          f"mismatch {s.n_local_mismatch} total_jsd {s.total_jsd:.15f} lowest {s.lowest_index} mean_delta {s.mean_delta_jsd:.12e}")
    print("   members", sel.members(with_freqs=False).positions.tolist()[:13], np.array2string(sel.members(with_freqs=False).delta_jsd[:4], precision=12))
    sel.close(); m.close()
