#include <hip/hip_runtime.h>
#include <cstdio>
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
    const long long b = __double_as_longlong(v);
    int lo = int(b), hi = int(b >> 32);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
    return __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo));
}
__device__ __forceinline__ double wave_sum_dpp(double v) {
    v += dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]
    v += dpp_mov<0x141>(v);  // row_half_mirror
    v += dpp_mov<0x140>(v);  // row_mirror
    const long long b = __double_as_longlong(v);
    const int lo = int(b), hi = int(b >> 32);
    auto rl = [&](int l) { return __longlong_as_double((long long)(((unsigned long long)(unsigned)__builtin_amdgcn_readlane(hi, l) << 32) | (unsigned)__builtin_amdgcn_readlane(lo, l))); };
    return (rl(0) + rl(16)) + (rl(32) + rl(48));
}
__global__ void k(const double* in, double* out) {
    double v = in[threadIdx.x];
    out[threadIdx.x] = wave_sum_dpp(v);
}
int main() {
    double h[64], *d, *o, r[64]; double ref = 0;
    for (int i = 0; i < 64; i++) { h[i] = 1.0 / (i + 3) * (i % 3 ? 1 : -1); }
    hipMalloc(&d, 512); hipMalloc(&o, 512); hipMemcpy(d, h, 512, hipMemcpyHostToDevice);
    k<<<1, 64>>>(d, o); hipMemcpy(r, o, 512, hipMemcpyDeviceToHost);
    for (int i = 0; i < 64; i++) ref += h[i];
    int same = 1; for (int i = 1; i < 64; i++) same &= (r[i] == r[0]);
    printf("ref %.17g got %.17g all-lanes-same %d\n", ref, r[0], same);
    return 0;
}
