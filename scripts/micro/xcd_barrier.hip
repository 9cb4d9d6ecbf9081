// Microbenchmark: a rendezvous among the workgroups of ONE XCD through that XCD's L2 (workgroup-scope
// atomic read-modify-writes on ordinary device memory execute in the L2 the 32 CUs of an XCD share),
// against the agent-scope barrier across all 8 XCDs the persistent engine uses.  Also checks that a
// value published with such an atomic is seen by the other workgroups of the XCD after the barrier.
//   hipcc -O3 --offload-arch=gfx950 scripts/micro/xcd_barrier.hip -o /tmp/xb && /tmp/xb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define AG __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
#define WG __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP
constexpr uint32_t SPIN_LIMIT = 1u << 20;

struct Line { uint32_t v; uint32_t pad[63]; };
struct Sync {
    Line timeout, count;
    Line gcount[8], ggen[8];  // agent-scope barrier (groups = blockIdx % 8)
    Line members[8];          // workgroups that found themselves on XCD x
    Line xcount[8], xgen[8];  // XCD-local barrier
    Line xdata[8];            // word published by one workgroup of the XCD per round
    Line xflags[8];           // one arrival word per workgroup of the XCD (64 words = this line)
    Line bad;
};

__device__ __forceinline__ uint32_t xcc_id() {
    uint32_t v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 15u;
}

__device__ bool agent_barrier(Sync *s, uint32_t G, uint32_t target) {
    __syncthreads();
    __shared__ int ok_s;
    if (threadIdx.x == 0) {
        bool ok = true;
        const uint32_t x = blockIdx.x & 7u, ng = G < 8 ? G : 8, gsz = (G - x + 7) >> 3;
        bool released = false;
        if (__hip_atomic_fetch_add(&s->gcount[x].v, 1u, AG) == gsz * target - 1 &&
            __hip_atomic_fetch_add(&s->count.v, 1u, AG) == ng * target - 1) {
            for (uint32_t g = 0; g < ng; g++) __hip_atomic_store(&s->ggen[g].v, target, AG);
            released = true;
        }
        uint32_t spins = 0;
        while (!released && __hip_atomic_load(&s->ggen[x].v, AG) < target) {
            if ((++spins & 255u) == 0 && (spins > SPIN_LIMIT || __hip_atomic_load(&s->timeout.v, AG))) {
                __hip_atomic_store(&s->timeout.v, 1u, AG);
                ok = false;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        ok_s = ok;
    }
    __syncthreads();
    return ok_s;
}

// POLL 0: read-modify-write (or 0) at workgroup scope; 1: workgroup-scope atomic load (may be served by
// the CU's own L1 -- bounded, so a stale line shows as TIMEOUT instead of a hang)
template <int POLL>
__device__ bool xcd_barrier(Sync *s, uint32_t x, uint32_t nmemb, uint32_t target) {
    __syncthreads();
    __shared__ int ok_s;
    if (threadIdx.x == 0) {
        bool ok = true;
        if (__hip_atomic_fetch_add(&s->xcount[x].v, 1u, WG) == nmemb * target - 1) {
            __hip_atomic_fetch_max(&s->xgen[x].v, target, WG);
        } else {
            uint32_t spins = 0;
            for (;;) {
                const uint32_t g = POLL == 0 ? __hip_atomic_fetch_or(&s->xgen[x].v, 0u, WG)
                                             : __hip_atomic_load(&s->xgen[x].v, WG);
                if (g >= target) break;
                if ((++spins & 255u) == 0 && (spins > SPIN_LIMIT || __hip_atomic_load(&s->timeout.v, AG))) {
                    __hip_atomic_store(&s->timeout.v, 1u, AG);
                    ok = false;
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        ok_s = ok;
    }
    __syncthreads();
    return ok_s;
}

// No atomics (those execute memory-side, past the L2): every workgroup stores its own arrival word,
// wave 0 of every workgroup polls the XCD's words with workgroup-scope loads (sc0: past the L1, into
// the L2 the XCD shares).
template <int INV>
__device__ bool xcd_flag_barrier(Sync *s, uint32_t x, uint32_t rank, uint32_t nmemb, uint32_t target) {
    __syncthreads();
    __shared__ int ok_s;
    if (threadIdx.x < 64) {
        uint32_t *fl = &s->xflags[x].v;
        if (threadIdx.x == 0) __hip_atomic_store(fl + rank, target, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        bool ok = true;
        uint32_t spins = 0;
        for (;;) {
            if (INV == 1) asm volatile("buffer_inv sc0" ::: "memory");  // drop the L1's lines: the load goes to the L2
            if (INV == 2) asm volatile("buffer_inv sc1" ::: "memory");
            const bool mine = threadIdx.x >= nmemb ||
                              __hip_atomic_load(fl + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= target;
            if (__all(mine)) break;
            if ((++spins & 255u) == 0 && (spins > SPIN_LIMIT || __hip_atomic_load(&s->timeout.v, AG))) {
                __hip_atomic_store(&s->timeout.v, 1u, AG);
                ok = false;
                break;
            }
        }
        if (threadIdx.x == 0) ok_s = ok;
    }
    __syncthreads();
    return ok_s;
}

template <int V>
__global__ __launch_bounds__(512) void kern(Sync *s, uint32_t iters, unsigned long long *ticks) {
    const uint32_t G = gridDim.x;
    __shared__ uint32_t s_x, s_rank, s_n;
    if (threadIdx.x == 0) {
        s_x = xcc_id() & 7u;
        s_rank = __hip_atomic_fetch_add(&s->members[s_x].v, 1u, AG);
    }
    if (!agent_barrier(s, G, 1)) return;
    if (threadIdx.x == 0) s_n = __hip_atomic_load(&s->members[s_x].v, AG);
    __syncthreads();
    const uint32_t x = s_x, rank = s_rank, nm = s_n;
    if (threadIdx.x == 0) ticks[8 + blockIdx.x] = (uint64_t(x) << 32) | (uint64_t(rank) << 16) | nm;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    uint32_t bad = 0;
    for (uint32_t it = 1; it <= iters; it++) {
        bool ok;
        if (V == 0) {
            ok = agent_barrier(s, G, it + 1);
        } else if (V >= 3) {
            if (threadIdx.x == 0 && rank == it % nm)
                __hip_atomic_store(&s->xdata[x].v, it * 8u + x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            ok = xcd_flag_barrier<V - 3>(s, x, rank, nm, 2 * it - 1);
            if (V == 4) asm volatile("buffer_inv sc0" ::: "memory");
            if (V == 5) asm volatile("buffer_inv sc1" ::: "memory");
            if (ok && threadIdx.x == 0 &&
                __hip_atomic_load(&s->xdata[x].v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != it * 8u + x)
                bad++;
            if (ok) ok = xcd_flag_barrier<V - 3>(s, x, rank, nm, 2 * it);
        } else {
            // the round's publisher stores before it arrives; everybody reads after the barrier
            if (threadIdx.x == 0 && rank == it % nm) __hip_atomic_exchange(&s->xdata[x].v, it * 8u + x, WG);
            ok = xcd_barrier<V - 1>(s, x, nm, 2 * it - 1);
            if (ok && threadIdx.x == 0 && __hip_atomic_fetch_or(&s->xdata[x].v, 0u, WG) != it * 8u + x) bad++;
            // (a second barrier so that the next round's publisher does not overwrite before everybody read)
            if (ok) ok = xcd_barrier<V - 1>(s, x, nm, 2 * it);
        }
        if (!ok) break;
    }
    if (threadIdx.x == 0) {
        if (bad) __hip_atomic_fetch_add(&s->bad.v, bad, AG);
        if (blockIdx.x == 0) ticks[0] = __builtin_amdgcn_s_memrealtime() - t0;
    }
}

template <int V>
void run(const char *name, Sync *d, unsigned long long *dt, int G, int per_iter) {
    const uint32_t iters = 2000;
    for (int rep = 0; rep < 2; rep++) {
        (void)hipMemset(d, 0, sizeof(Sync));
        hipLaunchKernelGGL((kern<V>), dim3(G), dim3(512), 0, 0, d, iters, dt);
        if (hipDeviceSynchronize() != hipSuccess) { printf("%s: kernel failed\n", name); exit(1); }
    }
    unsigned long long t[8 + 256];
    Sync h;
    (void)hipMemcpy(t, dt, sizeof t, hipMemcpyDeviceToHost);
    (void)hipMemcpy(&h, d, sizeof h, hipMemcpyDeviceToHost);
    printf("%-40s G=%d  %.3f us/barrier%s  wrong reads %u  members/XCD", name, G,
           double(t[0]) * 0.01 / iters / per_iter, h.timeout.v ? "  TIMEOUT" : "", h.bad.v);
    for (int x = 0; x < 8; x++) printf(" %u", h.members[x].v);
    int agree = 0;
    for (int b = 0; b < G; b++) agree += int((t[8 + b] >> 32) == uint64_t(b & 7));
    printf("  blockIdx%%8==XCC_ID for %d/%d\n", agree, G);
}

int main() {
    Sync *d;
    unsigned long long *dt;
    (void)hipMalloc(&d, sizeof(Sync));
    (void)hipMalloc(&dt, (8 + 256) * 8);
    for (int G : {256, 128}) {
        run<0>("agent scope, all XCDs", d, dt, G, 1);
        run<1>("XCD-local, workgroup-scope RMW poll", d, dt, G, 2);
        run<2>("XCD-local, workgroup-scope load poll", d, dt, G, 2);
        run<3>("XCD-local, arrival words, no atomics", d, dt, G, 2);
        run<4>("XCD-local, arrival words + buffer_inv sc0", d, dt, G, 2);
        run<5>("XCD-local, arrival words + buffer_inv sc1", d, dt, G, 2);
    }
    return 0;
}
