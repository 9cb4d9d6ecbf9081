"""the C4 share's selection (12 500 x 5 kb, k=7, nmost n=100; or: reps k nseq n) many times over the same
sequences: are the counters and the members the same every time?  (This is how the lost leave-one-out
partials of persist.hip's grid_barrier comment were found.)"""
import os, sys, collections
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from diverseseq_amd import engine

n_rep = int(sys.argv[1]) if len(sys.argv) > 1 else 200
k = int(sys.argv[2]) if len(sys.argv) > 2 else 7
nseq = int(sys.argv[3]) if len(sys.argv) > 3 else 12_500
nsel = int(sys.argv[4]) if len(sys.argv) > 4 else 100
dev = torch.device("cuda:0")
ctx = engine.Context(0)
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench_configs  # (its synth(): the very sequences of scripts/bench_configs.py's "C4 per-GPU share (1/8)")
seqs, offs = bench_configs.synth(nseq, 5000, 5000, 20260421 + len("C4 per-GPU share (1/8)"))
seen = collections.Counter()
for i in range(n_rep):
    m = ctx.build_matrix_device(seqs.data_ptr(), offs, k, 4)
    sel = m.nmost(nsel)
    s = sel.summary()
    mem = sel.members(False)
    key = (s.n_accepts, s.n_arbitrated, s.n_events, s.n_windows, s.rows_rechecked, repr(s.total_jsd), hash(mem.positions.tobytes()))
    seen[key] += 1
    print(f"[iter {i}] {key[:5]}", file=sys.stderr, flush=True)
    sel.close(); m.close()
for key, c in seen.most_common():
    print(c, key)
