// Microbenchmark: cost of one grid barrier across 256 resident workgroups on gfx950, for the
// variants considered for the persistent selection engine.  Build + run on the GPU box:
//   hipcc -O3 --offload-arch=gfx950 scripts/micro/barrier_bench.hip -o /tmp/bb && /tmp/bb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define RLX __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
constexpr uint32_t SPIN_LIMIT = 1u << 22;

struct Line { uint32_t v; uint32_t pad[63]; };
struct Sync {
    Line count, gen, timeout;
    Line gcount[32], ggen[32];
    Line flags[4];  // 256 words
    Line gdone[8];  // replica r: word g = generation group g has completed
};

template <bool FENCED>
__device__ void pre(void) {
    if (FENCED) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (FENCED && threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
}
template <bool FENCED>
__device__ void post(void) {
    if (FENCED && threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
}
__device__ bool spin_until(uint32_t *w, uint32_t target, Sync *s) {
    uint32_t spins = 0;
    while (__hip_atomic_load(w, RLX) < target) {
        if ((++spins & 255u) == 0 && (spins > SPIN_LIMIT || __hip_atomic_load(&s->timeout.v, RLX))) {
            __hip_atomic_store(&s->timeout.v, 1u, RLX);
            return false;
        }
        __builtin_amdgcn_s_sleep(1);
    }
    return true;
}

// V0: one counter, one generation word
template <bool FENCED>
__device__ bool bar_flat(Sync *s, uint32_t G, uint32_t target) {
    pre<FENCED>();
    bool ok = true;
    if (threadIdx.x == 0) {
        const uint32_t old = __hip_atomic_fetch_add(&s->count.v, 1u, RLX);
        if (old == G * target - 1) __hip_atomic_store(&s->gen.v, target, RLX);
        else ok = spin_until(&s->gen.v, target, s);
    }
    post<FENCED>();
    return ok;
}
// V1: 8 group counters (blockIdx % 8 ~ XCD), a top counter, one generation word per group
template <bool FENCED, uint32_t NG = 8>
__device__ bool bar_hier(Sync *s, uint32_t G, uint32_t target) {
    pre<FENCED>();
    bool ok = true;
    if (threadIdx.x == 0) {
        const uint32_t x = blockIdx.x & (NG - 1), ng = G < NG ? G : NG;
        const uint32_t gsz = (G - x + NG - 1) / NG;
        bool released = false;
        if (__hip_atomic_fetch_add(&s->gcount[x].v, 1u, RLX) == gsz * target - 1) {
            if (__hip_atomic_fetch_add(&s->count.v, 1u, RLX) == ng * target - 1) {
                for (uint32_t g = 0; g < ng; g++) __hip_atomic_store(&s->ggen[g].v, target, RLX);
                released = true;
            }
        }
        if (!released) ok = spin_until(&s->ggen[x].v, target, s);
    }
    post<FENCED>();
    return ok;
}
// V7: group counters only; the last arrival of a group publishes the group's generation in every
// replica line, wave 0's lanes 0..7 poll the eight words of their group's replica (3 hops, not 4)
template <bool FENCED>
__device__ bool bar_groups(Sync *s, uint32_t G, uint32_t target) {
    pre<FENCED>();
    __shared__ int s_ok2;
    if (threadIdx.x < 64) {
        const uint32_t x = blockIdx.x & 7u, ng = G < 8 ? G : 8;
        const uint32_t gsz = (G - x + 7) >> 3;
        uint32_t last = 0;
        if (threadIdx.x == 0) last = __hip_atomic_fetch_add(&s->gcount[x].v, 1u, RLX) == gsz * target - 1;
        last = __shfl(last, 0, 64);
        if (last && threadIdx.x < ng) __hip_atomic_store(&s->gdone[threadIdx.x].v + x, target, RLX);
        bool ok = true;
        uint32_t spins = 0;
        const uint32_t *mine = &s->gdone[x].v;
        for (;;) {
            const bool done = threadIdx.x >= ng || __hip_atomic_load(mine + threadIdx.x, RLX) >= target;
            if (__all(done)) break;
            if ((++spins & 255u) == 0 && (spins > SPIN_LIMIT || __hip_atomic_load(&s->timeout.v, RLX))) {
                __hip_atomic_store(&s->timeout.v, 1u, RLX);
                ok = false;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        if (threadIdx.x == 0) s_ok2 = ok;
    }
    post<FENCED>();
    return s_ok2;
}
// V2: one flag word per workgroup (plain stores, no atomics); wave 0 of every workgroup polls all flags
template <bool FENCED>
__device__ bool bar_flags(Sync *s, uint32_t G, uint32_t target) {
    pre<FENCED>();
    __shared__ int s_ok;
    if (threadIdx.x < 64) {
        uint32_t *fl = &s->flags[0].v;
        if (threadIdx.x == 0) __hip_atomic_store(fl + blockIdx.x, target, RLX);
        uint32_t spins = 0;
        bool ok = true;
        for (;;) {
            bool mine = true;
            for (uint32_t i = threadIdx.x; i < G; i += 64) mine &= __hip_atomic_load(fl + i, RLX) >= target;
            if (__all(mine)) break;
            if ((++spins & 255u) == 0 && (spins > SPIN_LIMIT || __hip_atomic_load(&s->timeout.v, RLX))) {
                __hip_atomic_store(&s->timeout.v, 1u, RLX);
                ok = false;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        if (threadIdx.x == 0) s_ok = ok;
    }
    post<FENCED>();
    return s_ok;
}
// V3: flags for the arrival, workgroup 0 collects and publishes one generation word per group
template <bool FENCED>
__device__ bool bar_flags_master(Sync *s, uint32_t G, uint32_t target) {
    pre<FENCED>();
    __shared__ int s_ok;
    if (threadIdx.x < 64) {
        uint32_t *fl = &s->flags[0].v;
        bool ok = true;
        if (blockIdx.x == 0) {
            uint32_t spins = 0;
            for (;;) {
                bool mine = true;
                for (uint32_t i = threadIdx.x; i < G; i += 64)
                    mine &= (i == 0) || __hip_atomic_load(fl + i, RLX) >= target;
                if (__all(mine)) break;
                if ((++spins & 255u) == 0 && (spins > SPIN_LIMIT || __hip_atomic_load(&s->timeout.v, RLX))) {
                    __hip_atomic_store(&s->timeout.v, 1u, RLX);
                    ok = false;
                    break;
                }
            }
            if (threadIdx.x < 8) __hip_atomic_store(&s->ggen[threadIdx.x].v, target, RLX);
        } else if (threadIdx.x == 0) {
            __hip_atomic_store(fl + blockIdx.x, target, RLX);
            ok = spin_until(&s->ggen[blockIdx.x & 7u].v, target, s);
        }
        if (threadIdx.x == 0) s_ok = ok;
    }
    post<FENCED>();
    return s_ok;
}

template <int V, bool FENCED>
__global__ __launch_bounds__(512) void kern(Sync *s, uint32_t iters, unsigned long long *ticks, double *sink) {
    const uint32_t G = gridDim.x;
    unsigned long long t0 = __builtin_readcyclecounter();
    t0 = __builtin_amdgcn_s_memrealtime();
    double acc = 0;
    for (uint32_t it = 1; it <= iters; it++) {
        bool ok;
        if (V == 0) ok = bar_flat<FENCED>(s, G, it);
        else if (V == 1) ok = bar_hier<FENCED>(s, G, it);
        else if (V == 7) ok = bar_groups<FENCED>(s, G, it);
        else if (V == 4) ok = bar_hier<FENCED, 4>(s, G, it);
        else if (V == 5) ok = bar_hier<FENCED, 16>(s, G, it);
        else if (V == 6) ok = bar_hier<FENCED, 32>(s, G, it);
        else if (V == 2) ok = bar_flags<FENCED>(s, G, it);
        else ok = bar_flags_master<FENCED>(s, G, it);
        if (!ok) break;
        if (FENCED) sink[blockIdx.x * 512 + threadIdx.x] = acc += 1.0;  // a dirty line to write back
    }
    if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = __builtin_amdgcn_s_memrealtime() - t0;
}

template <int V, bool F>
void run(const char *name, Sync *d, unsigned long long *dt, double *sink, int G) {
    const uint32_t iters = 2000;
    for (int rep = 0; rep < 2; rep++) {
        hipMemset(d, 0, sizeof(Sync));
        hipLaunchKernelGGL((kern<V, F>), dim3(G), dim3(512), 0, 0, d, iters, dt, sink);
        hipDeviceSynchronize();
    }
    unsigned long long t;
    uint32_t to;
    hipMemcpy(&t, dt, 8, hipMemcpyDeviceToHost);
    hipMemcpy(&to, &d->timeout.v, 4, hipMemcpyDeviceToHost);
    printf("%-28s G=%d  %.3f us/barrier%s\n", name, G, double(t) * 0.01 / iters, to ? "  TIMEOUT" : "");
}

int main() {
    Sync *d;
    unsigned long long *dt;
    double *sink;
    hipMalloc(&d, sizeof(Sync));
    hipMalloc(&dt, 64);
    hipMalloc(&sink, 256 * 512 * 8);
    for (int G : {256, 64}) {
        run<0, false>("flat unfenced", d, dt, sink, G);
        run<1, false>("hier unfenced", d, dt, sink, G);
        run<7, false>("group words, 3 hops, unfenced", d, dt, sink, G);
        run<4, false>("hier unfenced, 4 groups", d, dt, sink, G);
        run<5, false>("hier unfenced, 16 groups", d, dt, sink, G);
        run<6, false>("hier unfenced, 32 groups", d, dt, sink, G);
        run<2, false>("flags all-poll unfenced", d, dt, sink, G);
        run<3, false>("flags master unfenced", d, dt, sink, G);
        run<0, true>("flat fenced", d, dt, sink, G);
        run<1, true>("hier fenced", d, dt, sink, G);
        run<2, true>("flags all-poll fenced", d, dt, sink, G);
        run<3, true>("flags master fenced", d, dt, sink, G);
    }
    return 0;
}
