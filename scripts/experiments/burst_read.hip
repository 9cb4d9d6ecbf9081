// How fast can ONE short launch pull 4096 cold rows of 8 KiB (what a greedy step of the exact mode reads)?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/burst_read scripts/experiments/burst_read.hip && /tmp/burst_read
// Each variant is launched 24 times over successive 32 MiB windows of a 3 GiB buffer (cold in every cache).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int ROWB = 8192;  // bytes a row

// A: a wave a row, 16 loads of 8 B a lane (512 B a wave instruction), all in flight
__global__ __launch_bounds__(512) void k_wave_row_8B(const char *base, uint64_t nrows, unsigned *out) {
    const uint32_t lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint64_t r = uint64_t(blockIdx.x) * 8 + wave;
    if (r >= nrows) return;
    const uint2 *p = reinterpret_cast<const uint2 *>(base + r * ROWB) + lane;
    uint2 v[16];
#pragma unroll
    for (int j = 0; j < 16; j++) v[j] = p[j * 64];
    unsigned a = 0;
#pragma unroll
    for (int j = 0; j < 16; j++) a += v[j].x ^ v[j].y;
    if (a == 0x12345678u) out[r] = a;
}
// B: a wave a row, 8 loads of 16 B a lane (1 KiB a wave instruction)
__global__ __launch_bounds__(512) void k_wave_row_16B(const char *base, uint64_t nrows, unsigned *out) {
    const uint32_t lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint64_t r = uint64_t(blockIdx.x) * 8 + wave;
    if (r >= nrows) return;
    const uint4 *p = reinterpret_cast<const uint4 *>(base + r * ROWB) + lane;
    uint4 v[8];
#pragma unroll
    for (int j = 0; j < 8; j++) v[j] = p[j * 64];
    unsigned a = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) a += v[j].x ^ v[j].y ^ v[j].z ^ v[j].w;
    if (a == 0x12345678u) out[r] = a;
}
// C: a workgroup a row at a time (8 waves x 1 KiB = the row), 8 rows a workgroup, all 8 loads in flight
__global__ __launch_bounds__(512) void k_wg_row(const char *base, uint64_t nrows, unsigned *out) {
    const uint64_t r0 = uint64_t(blockIdx.x) * 8;
    if (r0 >= nrows) return;
    const uint4 *p = reinterpret_cast<const uint4 *>(base + r0 * ROWB) + threadIdx.x;
    uint4 v[8];
#pragma unroll
    for (int j = 0; j < 8; j++) v[j] = p[j * 512];
    unsigned a = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) a += v[j].x ^ v[j].y ^ v[j].z ^ v[j].w;
    if (a == 0x12345678u) out[r0] = a;
}
// D: the whole window as one flat array, thread t reads 16 B at t, t + T, ... (T = all threads): the most regular pattern
__global__ __launch_bounds__(512) void k_flat(const char *base, uint64_t nrows, unsigned *out) {
    const uint64_t T = uint64_t(gridDim.x) * 512, t = uint64_t(blockIdx.x) * 512 + threadIdx.x;
    const uint4 *p = reinterpret_cast<const uint4 *>(base);
    const uint64_t n16 = nrows * ROWB / 16;
    uint4 v[8];
#pragma unroll
    for (int j = 0; j < 8; j++) v[j] = (t + j * T < n16) ? p[t + j * T] : make_uint4(0, 0, 0, 0);
    unsigned a = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) a += v[j].x ^ v[j].y ^ v[j].z ^ v[j].w;
    if (a == 0x12345678u) out[t & 1023] = a;
}
// E: as A with one load in flight at a time (the dependent-chain floor)
__global__ __launch_bounds__(512) void k_wave_row_4x(const char *base, uint64_t nrows, unsigned *out) {
    const uint32_t lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint64_t r = uint64_t(blockIdx.x) * 8 + wave;
    if (r >= nrows) return;
    const uint2 *p = reinterpret_cast<const uint2 *>(base + r * ROWB) + lane;
    unsigned a = 0;
    for (int i = 0; i < 4; i++) {
        uint2 v[4];
#pragma unroll
        for (int j = 0; j < 4; j++) v[j] = p[(i * 4 + j) * 64];
#pragma unroll
        for (int j = 0; j < 4; j++) a += v[j].x ^ v[j].y;
        asm volatile("" : "+v"(a));
    }
    if (a == 0x12345678u) out[r] = a;
}

int main() {
    const uint64_t total = 3ull << 30;
    char *buf;
    unsigned *out;
    CK(hipMalloc(&buf, total));
    CK(hipMalloc(&out, 1 << 20));
    CK(hipMemset(buf, 1, total));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    struct V { const char *name; void (*k)(const char *, uint64_t, unsigned *); };
    const V vs[] = {{"wave a row, 16 x 512 B in flight", k_wave_row_8B}, {"wave a row, 8 x 1 KiB in flight", k_wave_row_16B},
                    {"workgroup a row, 8 x 8 KiB in flight", k_wg_row}, {"flat, grid-strided 16 B a thread", k_flat},
                    {"wave a row, 4 x 512 B in flight, 4 round trips", k_wave_row_4x}};
    for (uint64_t nrows : {1024ull, 2048ull, 4096ull, 8192ull, 32768ull}) {
        printf("rows a launch: %llu (%.1f MiB)\n", (unsigned long long)nrows, nrows * ROWB / 1048576.0);
        for (const V &v : vs) {
            std::vector<float> ms;
            uint64_t off = 0;
            for (int it = 0; it < 24; it++) {
                if (off + nrows * ROWB > total) off = 0;
                CK(hipEventRecord(e0, 0));
                hipLaunchKernelGGL(v.k, dim3((nrows + 7) / 8), dim3(512), 0, 0, buf + off, nrows, out);
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                float t;
                CK(hipEventElapsedTime(&t, e0, e1));
                ms.push_back(t);
                off += std::max<uint64_t>(nrows * ROWB, 64ull << 20);  // (at least 64 MiB on: never a line the last launch pulled in)
            }
            std::sort(ms.begin(), ms.end());
            const double med = ms[ms.size() / 2];
            printf("  %-50s median %7.1f us  min %7.1f us   %6.2f TB/s\n", v.name, med * 1e3, ms[0] * 1e3, nrows * ROWB / (med * 1e-3) / 1e12);
        }
    }
    // the same windows a second time round (3 GiB later: cold in the caches, warm in the TLBs' reach?) vs a re-read at once
    {
        const uint64_t nrows = 4096;
        float t;
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_wave_row_16B, dim3(nrows / 8), dim3(512), 0, 0, buf, nrows, out);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&t, e0, e1));
        printf("4096 rows, first touch in a while: %.1f us;", t * 1e3);
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_wave_row_16B, dim3(nrows / 8), dim3(512), 0, 0, buf, nrows, out);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&t, e0, e1));
        printf(" the same rows again at once: %.1f us\n", t * 1e3);
    }
    return 0;
}
