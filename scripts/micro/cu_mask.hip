// Does a CU-masked stream confine a kernel to the CUs of its mask on this box, and how do mask bits
// map to XCDs?  Launches many short workgroups on (a) an unmasked stream, (b) a stream masked to bits
// 0..63, (c) bits 64..255, and counts the distinct (XCC, SE, CU) places they ran on; then runs a
// long-running kernel on (b) and a wide one on (c) at the same time and checks they never share a CU.
//   hipcc -O3 --offload-arch=gfx950 scripts/micro/cu_mask.hip -o /tmp/cum && /tmp/cum
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <set>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ uint32_t xcc_id() {
    uint32_t v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 15u;
}
__device__ __forceinline__ uint32_t hw_id() {
    uint32_t v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(v));
    return v;
}

__global__ void where_kernel(uint32_t *out, int spin) {
    if (threadIdx.x == 0) {
        const uint32_t h = hw_id();
        // HW_ID: [3:0] wave, [5:4] simd, [7:6] pipe, [11:8] cu, [12] sh, [15:13] se
        out[blockIdx.x] = (xcc_id() << 16) | (((h >> 13) & 7u) << 8) | (((h >> 12) & 1u) << 4) | ((h >> 8) & 15u);
    }
    for (int i = 0; i < spin; i++) __builtin_amdgcn_s_sleep(64);
}

static int places(hipStream_t st, uint32_t *d, int nwg, int spin, std::set<uint32_t> &seen, int per_xcc[16]) {
    std::vector<uint32_t> h(nwg);
    hipLaunchKernelGGL(where_kernel, dim3(nwg), dim3(512), 0, st, d, spin);
    if (hipStreamSynchronize(st) != hipSuccess) return -1;
    if (hipMemcpy(h.data(), d, nwg * 4, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    seen.clear();
    for (int i = 0; i < nwg; i++) seen.insert(h[i]);
    for (int i = 0; i < 16; i++) per_xcc[i] = 0;
    for (uint32_t p : seen) per_xcc[p >> 16]++;
    return int(seen.size());
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    printf("%s: %d CUs\n", prop.gcnArchName, prop.multiProcessorCount);
    const int ncu = prop.multiProcessorCount, words = (ncu + 31) / 32;
    std::vector<uint32_t> lo(words, 0), hi(words, 0);
    for (int i = 0; i < ncu; i++) (i < 64 ? lo : hi)[i / 32] |= 1u << (i % 32);
    hipStream_t s0, s_lo, s_hi;
    CHECK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
    CHECK(hipExtStreamCreateWithCUMask(&s_lo, words, lo.data()));
    CHECK(hipExtStreamCreateWithCUMask(&s_hi, words, hi.data()));
    uint32_t *d, *d2;
    CHECK(hipMalloc(&d, 1 << 20));
    CHECK(hipMalloc(&d2, 1 << 20));
    std::set<uint32_t> a, b, c;
    int px[16];
    const char *names[] = {"unmasked", "bits 0..63", "bits 64.."};
    hipStream_t ss[] = {s0, s_lo, s_hi};
    std::set<uint32_t> *sets[] = {&a, &b, &c};
    for (int k = 0; k < 3; k++) {
        int n = places(ss[k], d, 8192, 20, *sets[k], px);
        printf("%-12s distinct places %3d   per XCC:", names[k], n);
        for (int i = 0; i < 8; i++) printf(" %d", px[i]);
        printf("\n");
    }
    int shared = 0;
    for (uint32_t p : b) shared += c.count(p);
    printf("places used by both masked streams: %d\n", shared);
    // concurrently: 64 long workgroups on the low mask, a wide stream of short ones on the high mask
    hipLaunchKernelGGL(where_kernel, dim3(64), dim3(512), 0, s_lo, d, 20000);
    hipLaunchKernelGGL(where_kernel, dim3(65536), dim3(256), 0, s_hi, d2, 1);
    CHECK(hipDeviceSynchronize());
    std::vector<uint32_t> h1(64), h2(65536);
    CHECK(hipMemcpy(h1.data(), d, 64 * 4, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(h2.data(), d2, 65536 * 4, hipMemcpyDeviceToHost));
    std::set<uint32_t> p1(h1.begin(), h1.end()), p2(h2.begin(), h2.end());
    shared = 0;
    for (uint32_t p : p1) shared += p2.count(p);
    printf("concurrent: 64 long workgroups on %zu places, wide kernel on %zu places, shared %d\n", p1.size(), p2.size(), shared);
    // cooperative launch on the masked stream
    {
        int spin = 10;
        uint32_t *dd = d;
        void *args[] = {&dd, &spin};
        hipError_t e = hipLaunchCooperativeKernel(reinterpret_cast<const void *>(where_kernel), dim3(64), dim3(512), args, 0, s_lo);
        printf("cooperative launch of 64 workgroups on the masked stream: %s\n", hipGetErrorString(e));
        CHECK(hipDeviceSynchronize());
    }
    return 0;
}
