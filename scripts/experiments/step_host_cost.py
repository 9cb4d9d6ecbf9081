"""Host cost of one enqueued step of the exact mode (pack + apply through ctypes), measured on steps behind the end of a
selection (no-ops on the device): python scripts/experiments/step_host_cost.py"""
import sys, time
import numpy as np
import torch

sys.path.insert(0, ".")
from diverseseq_amd import _lib, engine, parallel  # noqa: E402

dev = torch.device("cuda", 0)
ctx = engine.Context(0, stream=torch.cuda.current_stream().cuda_stream)
rng = np.random.default_rng(1)
seqs = [rng.integers(0, 4, 600, dtype=np.uint8) for _ in range(2000)]
m = ctx.build_matrix(seqs, 6, 4)
_, order = parallel.shard_order(len(seqs), 10, 0, 1, block=32)
sel = m.select(_lib.MODE_NMOST, 10, order=order, window=4096, flags=_lib.SELECT_STEPWISE)
st = parallel.HipStepper(ctx, sel, m.nbins, dev)
parallel.drive_exact(st, 1, dev)
torch.cuda.synchronize()
for n in (2000, 2000):
    t0 = time.perf_counter()
    for i in range(n):
        st.apply(st.pack(), 1)
        if i % 64 == 63:
            st.done()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{n} steps: enqueue {1e6 * (t1 - t0) / n:.1f} us a step, drained {1e6 * (t2 - t0) / n:.1f} us a step")
t0 = time.perf_counter()
for i in range(2000):
    st.peek(4)
print(f"peek: {1e6 * (time.perf_counter() - t0) / 2000:.2f} us")
