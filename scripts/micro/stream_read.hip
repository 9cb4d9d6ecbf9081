// Practical read roof of the box for the scan's access pattern: every wave reads whole 16 KiB rows
// (16 x dwordx4 per lane, one burst) of a 1.64 GB matrix and only XORs them.
//   hipcc -O3 --offload-arch=gfx950 scripts/micro/stream_read.hip -o /tmp/stream_read && /tmp/stream_read
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int CH>
__global__ __launch_bounds__(512) void rows_kernel(const uint4 *m, uint64_t nrows, uint32_t *out) {
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t nw = uint64_t(gridDim.x) * (blockDim.x / 64);
    uint32_t acc = 0;
    for (uint64_t r = uint64_t(blockIdx.x) * (blockDim.x / 64) + wave; r < nrows; r += nw) {
        const uint4 *rp = m + r * 1024;  // 4096 u32 = 1024 uint4
        for (int b = 0; b < 16 / CH; b++) {
            uint4 v[CH];
#pragma unroll
            for (int j = 0; j < CH; j++) v[j] = rp[(b * CH + j) * 64 + lane];
#pragma unroll
            for (int j = 0; j < CH; j++) acc ^= v[j].x ^ v[j].y ^ v[j].z ^ v[j].w;
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

__global__ __launch_bounds__(256) void flat_kernel(const uint4 *m, uint64_t n4, uint32_t *out) {
    uint32_t acc = 0;
    const uint64_t stride = uint64_t(gridDim.x) * blockDim.x;
    uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n4; i += 4 * stride) {
        const uint4 a = m[i], b = m[i + stride], c = m[i + 2 * stride], d = m[i + 3 * stride];
        acc ^= a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w ^ c.x ^ c.y ^ c.z ^ c.w ^ d.x ^ d.y ^ d.z ^ d.w;
    }
    for (; i < n4; i += stride) acc ^= m[i].x;
    if (acc == 0x12345678u) out[0] = acc;
}

int main() {
    const uint64_t nrows = 100000, bytes = nrows * 16384;
    uint4 *m;
    uint32_t *out;
    CK(hipMalloc(&m, bytes));
    CK(hipMalloc(&out, 4));
    CK(hipMemset(m, 1, bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto time = [&](const char *name, auto launch) {
        launch();
        CK(hipDeviceSynchronize());
        float best = 1e9f;
        for (int it = 0; it < 5; it++) {
            CK(hipEventRecord(e0));
            launch();
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        printf("%-40s %.3f ms  %.0f GB/s\n", name, best, bytes / (best * 1e-3) / 1e9);
    };
    for (int g : {256, 512, 1024, 2048}) {
        char nm[64];
        snprintf(nm, 64, "rows, burst 16, grid %d x 512", g);
        time(nm, [&] { hipLaunchKernelGGL(rows_kernel<16>, dim3(g), dim3(512), 0, 0, m, nrows, out); });
        snprintf(nm, 64, "rows, burst 8, grid %d x 512", g);
        time(nm, [&] { hipLaunchKernelGGL(rows_kernel<8>, dim3(g), dim3(512), 0, 0, m, nrows, out); });
        snprintf(nm, 64, "rows, burst 4, grid %d x 512", g);
        time(nm, [&] { hipLaunchKernelGGL(rows_kernel<4>, dim3(g), dim3(512), 0, 0, m, nrows, out); });
    }
    for (int g : {1024, 4096, 16384})  {
        char nm[64];
        snprintf(nm, 64, "flat grid-stride, grid %d x 256", g);
        time(nm, [&] { hipLaunchKernelGGL(flat_kernel, dim3(g), dim3(256), 0, 0, m, bytes / 16, out); });
    }
    return 0;
}
