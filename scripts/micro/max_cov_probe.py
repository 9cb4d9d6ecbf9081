"""`max` with stat=cov over a stream of events: which engine handles them (DVS_PERSIST_DEBUG=1 prints why launches end)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
from diverseseq_amd import engine
from test_gpu_parity import _own_composition_seqs

ctx = engine.Context(0)
ctx.set_timing(True)
for stat in ("stdev", "cov"):
    rng = np.random.default_rng(900 + 7 * 6 + 30)
    seqs = _own_composition_seqs(rng, 700, 3000, 5000, {33, 34, 200, 201, 202, 450, 699})
    m = ctx.build_matrix(seqs, 6, 4)
    import time
    sel = m.max_divergent(30, 700, stat); sel.close()
    t0 = time.perf_counter()
    sel = m.max_divergent(30, 700, stat)
    s = sel.summary()
    wall = time.perf_counter() - t0
    print(stat, "size", s.size, "events", s.n_events, "windows", s.n_windows, "launches", s.scan_launches, "arb", s.n_arbitrated,
          "engine_ms", round(s.scan_ms, 3), "wall_ms", round(wall * 1e3, 2), flush=True)
    sel.close(); m.close()
