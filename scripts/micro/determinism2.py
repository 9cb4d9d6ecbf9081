"""chunk + merge of tests/test_gpu_parity.py::test_chunk_merge_mode_across_ranks in ONE process (the three
ranks one after the other), twice: every number of the second pass must equal the first bit for bit"""
import sys
import os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from conftest import synth_seqs
from diverseseq_amd import engine, parallel
dev = torch.device("cuda:0")
stream = torch.cuda.Stream()
world, n, k = 3, 8, 4
seqs = synth_seqs(907, 500, 31, invalid_frac=0.002, ragged=True)
bounds = parallel.chunk_bounds(len(seqs), world)
with torch.cuda.stream(stream):
    ctx = engine.Context(0, stream=stream.cuda_stream)
    outs = []
    for it in range(3):
        rows, metas, local = [], [], []
        for lo, hi in bounds:
            m = ctx.build_matrix(seqs[lo:hi], k, 4)
            sel = m.nmost(n)
            mem = sel.members(with_freqs=False)
            local.append((mem.positions.tolist(), mem.delta_jsd.tolist(), sel.summary().total_jsd, sel.summary().engine))
            t_rows = torch.empty((n, m.nbins), dtype=torch.float64, device=dev)
            t_meta = torch.empty((n, 2), dtype=torch.float64, device=dev)
            sel.gather_members(t_rows.data_ptr(), t_meta.data_ptr(), n)
            ctx.sync()
            rows.append(t_rows); metas.append(t_meta)
            sel.close(); m.close()
        all_rows, all_meta = torch.cat(rows), torch.cat(metas)
        torch.cuda.synchronize()
        mm = ctx.matrix_from_device_freqs(all_rows.data_ptr(), world * n, 4 ** k, all_meta.data_ptr())
        ms = mm.nmost(n)
        mem = ms.members(with_freqs=False)
        s = ms.summary()
        outs.append((local, mem.positions.tolist(), mem.delta_jsd.tolist(), s.total_jsd, s.engine, s.n_accepts, s.n_arbitrated))
        ms.close(); mm.close()
for i, o in enumerate(outs):
    print(i, "local same" if o[0] == outs[0][0] else "LOCAL VARIES", "merged same" if o[1:] == outs[0][1:] else "MERGED VARIES",
          o[2][:2], o[3], "engine", o[4], "accepts", o[5], "arb", o[6], [l[3] for l in o[0]])
