import time, numpy as np, torch, os, sys
n = 1 << 30
a = np.random.default_rng(0).integers(0, 255, size=n, dtype=np.uint8)
rt = torch.cuda.cudart()
torch.cuda.init()
d = torch.empty(n, dtype=torch.uint8, device="cuda:0")
for chunk in (32 << 20, 256 << 20, n):
    t0 = time.perf_counter()
    for off in range(0, n, chunk):
        r = rt.cudaHostRegister(a.ctypes.data + off, min(chunk, n - off), 0)
    t1 = time.perf_counter()
    t = torch.from_numpy(a)
    d.copy_(t, non_blocking=True); torch.cuda.synchronize()
    t2 = time.perf_counter()
    for off in range(0, n, chunk):
        rt.cudaHostUnregister(a.ctypes.data + off)
    t3 = time.perf_counter()
    print(f"chunk {chunk>>20} MiB: register {1e3*(t1-t0):.1f} ms/GiB, copy {1e3*(t2-t1):.1f} ms, unregister {1e3*(t3-t2):.1f} ms", r, flush=True)
t0 = time.perf_counter(); d.copy_(torch.from_numpy(a)); torch.cuda.synchronize(); print("pageable copy 1 GiB", 1e3*(time.perf_counter()-t0), "ms")
print("cpus", os.cpu_count(), len(os.sched_getaffinity(0)))
