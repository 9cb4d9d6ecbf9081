"""Repeated selections (and a merge over their members) in one process: every repetition must give the
same bits.   gpurun -- python scripts/micro/determinism.py"""
import os, sys
import os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import synth_seqs
from diverseseq_amd import engine
import oracle
ctx = engine.default_context()
for (nseq, L, k, n) in [(300, 500, 4, 8), (24, 500, 4, 8), (16, 400, 4, 8), (3000, 400, 6, 10)]:
    seqs = synth_seqs(nseq, L, 31 + nseq, invalid_frac=0.002, ragged=True)
    outs = []
    for it in range(6):
        m = ctx.build_matrix(seqs, k, 4)
        sel = m.nmost(n)
        mem = sel.members(with_freqs=True)
        s = sel.summary()
        outs.append((mem.positions.tolist(), mem.delta_jsd.tolist(), s.total_jsd, s.n_accepts, s.engine))
        # merge-like: f64 matrix from the members' rows
        m2 = ctx.matrix_from_freqs(np.vstack([mem.kfreqs, mem.kfreqs[::-1] * 1.0]))
        sel2 = m2.nmost(min(n, 2 * len(mem.positions) - 1))
        mem2 = sel2.members(with_freqs=False)
        outs[-1] += (mem2.delta_jsd.tolist(), sel2.summary().total_jsd)
        sel2.close(); m2.close(); sel.close(); m.close()
    same = all(o == outs[0] for o in outs)
    print(nseq, L, k, n, "deterministic" if same else "VARIES", outs[0][3], outs[0][4])
    if not same:
        for o in outs: print("   ", o[1][:3], o[2], o[5][:3], o[6])
