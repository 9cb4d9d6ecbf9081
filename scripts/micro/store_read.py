"""Host-side cost of reading an on-disk .dvseqsz store into the stream a selection uploads:
per-id read + join (the path until round 2) against DvseqszDir.read_many (thread pool, zstd in place).
    python scripts/micro/store_read.py [nseq] [length]"""
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from diverseseq_amd import _dvs, engine  # noqa: E402

nseq = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
length = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
rng = np.random.default_rng(1)
with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
    st = _dvs.make_zarr_store(os.path.join(tmp, "s.dvseqsz"), mode="w")
    t0 = time.perf_counter()
    for i in range(nseq):
        st.write(f"s{i:05d}", rng.integers(0, 4, length, dtype=np.uint8).tobytes())
    st.close()
    print(f"wrote {nseq} x {length}: {time.perf_counter() - t0:.2f} s")
    st = _dvs.make_zarr_store(os.path.join(tmp, "s.dvseqsz"), mode="r")
    ids = st.get_seqids()
    t0 = time.perf_counter()
    d0, o0 = engine.concat([st.read(s) for s in ids])
    t_old = time.perf_counter() - t0
    for w in (1, 4, None):
        t0 = time.perf_counter()
        d1, o1 = st._disk.read_many(ids, workers=w)
        t_new = time.perf_counter() - t0
        assert np.array_equal(d0, d1) and np.array_equal(o0, o1)
        print(f"read_many workers={w}: {t_new:.3f} s = {d1.size / t_new / 1e9:.2f} GB/s of decoded bases "
              f"(per-id read + join {t_old:.3f} s)")
