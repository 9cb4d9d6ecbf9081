"""The split build + head phase + seeded start without torch in the process (torch's lazy nvrtc load does
not survive an LD_PRELOADed ASan runtime): 20 000 FASTA records encoded on the device, the matrix built
from the bases in HBM, nmost against the oracle.  Run it with the host-ASan library:
    LD_PRELOAD=$(hipcc -print-file-name=libclang_rt.asan-x86_64.so) DVS_HIP_LIB=.../libdvs_hip_asan.so \\
    ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0 python scripts/micro/asan_head_phase.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402
from conftest import synth_seqs  # noqa: E402

from diverseseq_amd import engine  # noqa: E402

ctx = engine.default_context()
ctx.set_timing(True)
letters = np.frombuffer(b"TCAGN", dtype=np.uint8)
for k, n in ((6, 10), (5, 20), (6, 40)):
    seqs = synth_seqs(20_000, 240, seed=5 + k + n, ragged=True, invalid_frac=0.001)
    raw = b"".join(b">s%d\n" % i + letters[s].tobytes() + b"\n" for i, s in enumerate(seqs))
    batch = ctx.encode_fasta(raw)
    m = batch.build_matrix(k, 4)
    sel = m.nmost(n)
    s = sel.summary()
    exp = oracle.nmost(seqs, n, k, 4)
    got = sel.members(with_freqs=False).positions.tolist()
    assert got == exp.members()[0].tolist(), (k, n)
    assert abs(s.total_jsd - exp.total_jsd) <= 1e-9 * abs(exp.total_jsd)
    print("k", k, "n", n, "engine", s.engine, "persistent launches", s.scan_launches, "ok", flush=True)
    sel.close()
    m.close()
    batch.close()
print("done")
