// The streaming scan over 16-BIT count rows (8 KiB per row at 4^6 bins): what rate the row-per-wave
// pattern can reach, with the scan's extras (coarse arithmetic from LDS state, row scalars, a poll per
// row), one row in flight per wave or two (the next row requested before the current one is used), and
// 8 or 16 waves per CU.
//   hipcc -O3 --offload-arch=gfx950 scripts/micro/stream_scan16.hip -o /tmp/ss16 && /tmp/ss16
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float coarse8(const uint4 c, const float *b, const f2 r2) {
    float s = 0.f;
    const uint32_t w[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const f2 cc = {float(w[q] & 0xFFFFu), float(w[q] >> 16)};
        const f2 y = __builtin_elementwise_fma(cc, r2, (f2){b[2 * q], b[2 * q + 1]});
        s += y.x * __builtin_amdgcn_logf(y.x) + y.y * __builtin_amdgcn_logf(y.y);
    }
    return s;
}

__device__ __forceinline__ double score(const uint4 (&v)[8], const float *slf, uint32_t lane, float rtn) {
    const f2 r2 = {rtn, rtn};
    float c0 = 0.f, c1 = 0.f;
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
        float b0[8], b1[8];
        *reinterpret_cast<float4 *>(b0) = *reinterpret_cast<const float4 *>(slf + j * 512 + lane * 8);
        *reinterpret_cast<float4 *>(b0 + 4) = *reinterpret_cast<const float4 *>(slf + j * 512 + lane * 8 + 4);
        *reinterpret_cast<float4 *>(b1) = *reinterpret_cast<const float4 *>(slf + j * 512 + 512 + lane * 8);
        *reinterpret_cast<float4 *>(b1 + 4) = *reinterpret_cast<const float4 *>(slf + j * 512 + 512 + lane * 8 + 4);
        c0 += coarse8(v[j], b0, r2);
        c1 += coarse8(v[j + 1], b1, r2);
    }
    double s = double(c0) + double(c1);
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    return s;
}

// DEPTH 1: load row, use it.  DEPTH 2: the next row's loads are issued before the current row is used.
template <int DEPTH, int WPC>
__global__ __launch_bounds__(512, WPC / 8) void k(const uint4 *m, const uint32_t *totals, const double *rowH,
                                                  unsigned long long *ev, uint64_t nrows, double *out) {
    __shared__ float slf[4096];
    for (int i = threadIdx.x; i < 4096; i += 512) slf[i] = 1.0f / 4096.0f;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t nw = uint64_t(gridDim.x) * 8;
    double acc = 0.0;
    uint64_t r = uint64_t(blockIdx.x) * 8 + wave;
    uint4 v[8], w[8];
    if (DEPTH == 2 && r < nrows) {
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = m[r * 512 + j * 64 + lane];
    }
    for (; r < nrows; r += nw) {
        const unsigned long long e = __hip_atomic_load(ev + (blockIdx.x & 7) * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (e < r) break;
        const uint32_t t = totals[r];
        const double hrow = rowH[r];
        const float rtn = float(0.1 / double(t ? t : 1));
        if (DEPTH == 1) {
#pragma unroll
            for (int j = 0; j < 8; j++) v[j] = m[r * 512 + j * 64 + lane];
            acc += score(v, slf, lane, rtn) - hrow;
        } else {
            const uint64_t rn = r + nw < nrows ? r + nw : r;
#pragma unroll
            for (int j = 0; j < 8; j++) w[j] = m[rn * 512 + j * 64 + lane];
#pragma unroll
            for (int j = 0; j < 8; j++) asm volatile("" : "+v"(w[j].x), "+v"(w[j].y), "+v"(w[j].z), "+v"(w[j].w));
            acc += score(v, slf, lane, rtn) - hrow;
#pragma unroll
            for (int j = 0; j < 8; j++) v[j] = w[j];
        }
    }
    if (acc == 123.456) out[0] = acc;
}

int main() {
    const uint64_t nrows = 100000, bytes = nrows * 8192;
    uint4 *m; uint32_t *tot; double *rh, *out; unsigned long long *ev;
    CK(hipMalloc(&m, bytes)); CK(hipMalloc(&tot, nrows * 4)); CK(hipMalloc(&rh, nrows * 8)); CK(hipMalloc(&out, 8));
    CK(hipMalloc(&ev, 8 * 32 * 8));
    CK(hipMemset(m, 1, bytes)); CK(hipMemset(tot, 1, nrows * 4)); CK(hipMemset(rh, 0, nrows * 8)); CK(hipMemset(ev, 0xFF, 8 * 32 * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time = [&](const char *name, auto launch) {
        launch(); CK(hipDeviceSynchronize());
        float best = 1e9f;
        for (int it = 0; it < 5; it++) {
            CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        printf("%-58s %.3f ms  %.0f GB/s\n", name, best, bytes / (best * 1e-3) / 1e9);
    };
    time("1 row in flight per wave,  8 waves/CU (255 WGs)", [&] { hipLaunchKernelGGL((k<1, 8>), dim3(255), dim3(512), 0, 0, m, tot, rh, ev, nrows, out); });
    time("2 rows in flight per wave, 8 waves/CU (255 WGs)", [&] { hipLaunchKernelGGL((k<2, 8>), dim3(255), dim3(512), 0, 0, m, tot, rh, ev, nrows, out); });
    time("1 row in flight per wave, 16 waves/CU (510 WGs)", [&] { hipLaunchKernelGGL((k<1, 16>), dim3(510), dim3(512), 0, 0, m, tot, rh, ev, nrows, out); });
    time("2 rows in flight per wave, 16 waves/CU (510 WGs)", [&] { hipLaunchKernelGGL((k<2, 16>), dim3(510), dim3(512), 0, 0, m, tot, rh, ev, nrows, out); });
    return 0;
}
