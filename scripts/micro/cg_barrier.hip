// Microbenchmark: cooperative_groups grid.sync() under hipLaunchCooperativeKernel (what the runtime
// offers for a grid-wide barrier) on 256 resident 512-thread workgroups, beside nothing else.
//   hipcc -O3 --offload-arch=gfx950 scripts/micro/cg_barrier.hip -o /tmp/cgb && /tmp/cgb
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdio>
namespace cg = cooperative_groups;

__global__ __launch_bounds__(512) void kern(int iters, unsigned long long *dt) {
    cg::grid_group g = cg::this_grid();
    g.sync();
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; i++) g.sync();
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (blockIdx.x == 0 && threadIdx.x == 0) *dt = t1 - t0;
}

int main() {
    unsigned long long *dt;
    hipMalloc(&dt, 8);
    for (int G : {64, 128, 256}) {
        int iters = 2000;
        void *args[] = {&iters, &dt};
        hipError_t e = hipLaunchCooperativeKernel((const void *)kern, dim3(G), dim3(512), args, 0, 0);
        hipDeviceSynchronize();
        unsigned long long t = 0;
        hipMemcpy(&t, dt, 8, hipMemcpyDeviceToHost);
        printf("cg grid.sync G=%d: %.3f us/barrier (launch %s)\n", G, double(t) * 0.01 / iters, hipGetErrorString(e));
    }
    return 0;
}
