// Persistent form of the greedy selection (MODE_NMOST and MODE_MAX, identity order, unique ids).
//
// The multi-launch engine in select.hip pays, for every greedy event, a scan launch
// plus resolve / leave-one-out / finalize launches and their serial latencies.  Here
// ONE launch of one workgroup per CU walks the whole candidate stream:
//
//   * every workgroup keeps a private replica of the scalar set state (size, lowest,
//     sumH, total_jsd, threshold, cursor, window, member order) in registers / LDS,
//     and the vector sl_i = S_i - lowest_i in LDS (32 KB at k=6) -- what a
//     candidate's frequencies are added to (src/records.rs:78-81);
//   * SCAN: waves score the rows of the current window exactly as scan_kernel does
//     (x = (sl + c/T)/n: all-f32 tier, f32-log tier, f64 recheck); a workgroup's first
//     event goes into its arrival record, and as a hint into a word the others poll;
//   * the window's rendezvous (the gathering block takes the minimum of the records and
//     releases the window); then EVERY workgroup resolves the first event redundantly and
//     deterministically (same code, same reduction tree -> same bits, so no
//     broadcast is needed): exact delta_jsd of the candidate, decision,
//     replace_lowest, new total_jsd;
//   * leave-one-out as (n + 1) * K jobs over the workgroups (get_lowest_record_index,
//     src/records.rs:220-252), their sums added to accumulators that every workgroup polls
//     for completeness; every workgroup takes the argmin, rebuilds sl in its LDS and scans on
//     from the event + 1.
//   * the last two workgroups scan nothing: one GATHERS the arrival records, the other
//     MIRRORS the state into the global SelDev / SelCtl arrays (what the host reads back,
//     and what the multi-launch kernels resume from).
//
// Two grid-wide exchanges per accepted event and none of the launch latencies.  Any decision
// inside the rounding band stops the kernel with SEL_ARBITER; the host arbitrates and
// runs that one event through the multi-launch kernels, then relaunches this one.
//
// Inter-workgroup protocol (MI355X: 8 XCDs, private L2s; written down in DESIGN.md 4.3c).  Nothing a
// workgroup stores with plain stores is read by another workgroup during the launch: the matrix, totals
// and row entropies are read-only (a member's frequency row is re-derived from its matrix row, whose
// position every workgroup keeps in LDS) and the set state is replicated.  What crosses workgroups are
// single 64-bit words, read and written with relaxed agent-scope atomics, and every one of them carries
// in itself what a reader needs to know that it is the word it is waiting for -- no word's meaning depends
// on the order in which stores to two different addresses become visible:
//   * WINDOW words carry the window's number in their top bits (p_word).  A workgroup that has scanned its
//     share stores ONE arrival record (its first event, how many near-threshold candidates it listed); the
//     mirror block, which scans nothing, polls the records, and when all carry the window's number stores
//     the release word (the window's first event) to eight copies that the others poll.  Events found
//     early also go, by atomicMin, to a hint word beside the release word so that waves can stop scanning
//     rows behind them; a newer window's tag is SMALLER, so a straggler of an older window never wins, and
//     nothing is ever cleared.  The outcome never depends on a hint: a wave only skips rows behind a hint
//     of the current window, and whoever posted that hint has the same event in its record.
//   * LEAVE-ONE-OUT sums are added to monotonic accumulators (fixed point + a contribution count in the low
//     bits, p_acc_word) that are never cleared: every workgroup remembers the totals it read at the previous
//     use (LDS) and takes the difference; the count's difference says when the K contributions are in.
// The counter barrier below (grid_barrier) is a pure rendezvous for the few places that still want one (the
// seeded start, `max` batches): two-level monotonic arrival counters (per group of blockIdx % 8, then one
// top counter) and a generation word per group, all on their own cache lines, zeroed by the host before
// every launch; every spin is bounded (a grid that is not fully resident ends with SEL_ERROR instead of
// hanging).
#include "select_dev.h"

#include <algorithm>
#include <type_traits>
#include <cstddef>
#include <cstdlib>
#include <cstring>

namespace {

constexpr int P_THREADS = 512;
constexpr int P_J = 8;                      // bins per thread kept in registers (B <= 4096)
constexpr int P_CH = 16;                    // row chunks requested per burst (4096 bins)
constexpr uint32_t P_MAXN = 512;            // members the LDS replica holds when 4^k <= 4096 ...
constexpr uint32_t P_MAXN_BIG = 256;        // ... and beyond (k = 7: 128 KB of LDS go to the set state)
constexpr uint32_t P_MAXN_MAX = 896;        // ... `max` with 4^k <= 4096 (sets grow; what 160 KB of LDS hold beside S, sl and the batches)
constexpr uint32_t p_maxn(bool cached, bool maxm = false) { return cached ? (maxm ? P_MAXN_MAX : P_MAXN) : P_MAXN_BIG; }
// SMALL sets (nmost over 16-bit count rows of 4096 bins): the members' count rows live in every
// workgroup's LDS (see the kernel)
constexpr uint32_t P_SMALLN = 16;      // members the replica arrays of a SMALL launch hold ...
constexpr uint32_t P_SMALL_ROWS = 13;  // ... and rows (8 KB each) that fit beside the state in 160 KB of LDS
// leave-one-out jobs per event: (n + 1) * K <= max(G - 1, n + 1) <= maxn + 1 (G <= maxn + 2 is checked)
constexpr uint32_t p_maxjobs(bool cached, bool maxm = false) { return p_maxn(cached, maxm) + 1; }
// leave-one-out accumulators: 8 group replicas x (maxn + 1) members x 2 words, zeroed by the host before the
// launch and never cleared again.  A word = (2^-50 fixed-point sum << 6) + number of contributions, added to
// by every job of every use; a reader keeps the totals it read at the previous use (LDS, p_prev) and works
// with the DIFFERENCE: its low six bits are the contributions of this use -- complete at K -- and the rest
// is their sum (two's complement: wrap-around cancels in the difference; |sum| < 64 per use and K <= 32
// keep one use inside the word).  Integer addition is associative, so the totals do not depend on the order
// the jobs arrive in; nothing is ever reset, so there is no clear that a late add could race with, and an
// add can never be mistaken for one of an earlier use: the earlier use was complete, in every replica,
// when it was read (each job adds the same word to all eight replicas).
constexpr size_t p_acc_bytes(uint32_t maxn) { return size_t(8) * (maxn + 1) * 2 * sizeof(unsigned long long); }
__device__ __forceinline__ unsigned long long p_acc_word(double v) {
    return ((unsigned long long)__double2ll_rn(v * 0x1p50) << 6) + 1ull;
}
__device__ __forceinline__ double p_acc_value(unsigned long long w) { return double((long long)w >> 6) * 0x1p-50; }
// A job's two words added to an accumulator pair (non-returning adds: the readers wait for the counts)
__device__ __forceinline__ void p_acc_add(unsigned long long *dst, double th, double ts) {
    atomicAdd(dst, p_acc_word(th));
    atomicAdd(dst + 1, p_acc_word(ts));
}
__device__ __forceinline__ uint32_t p_acc_count(unsigned long long w) { return uint32_t(w & 63ull); }
// this use's share of an accumulator word once all K contributions are in: (total now) - (total at the previous
// use, *prev, which is brought up to date).  Bounded spin: ok = false on a time-out (an error and the
// multi-launch kernels, never a wrong total).
__device__ __forceinline__ unsigned long long p_acc_complete(const unsigned long long *w, unsigned long long *prev,
                                                             uint32_t K, bool &ok) {
    const unsigned long long before = *prev;
    unsigned long long v = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t spins = 0;
    while (uint32_t((v - before) & 63ull) != K) {
        if (++spins > (1u << 20)) {
            ok = false;
            break;
        }
        __builtin_amdgcn_s_sleep(1);
        v = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    *prev = v;
    return v - before;
}
constexpr uint32_t P_SPIN_LIMIT = 1u << 22;  // ~0.5 s of polling
constexpr uint32_t P_MAXG = 256;             // workgroups a launch may have (one arrival record each)
constexpr uint32_t P_FLAT_GRID = 64;         // counter barrier: grids of up to this many workgroups use one counter
constexpr uint32_t P_WINWORDS = 64;          // LDS words of a window's bookkeeping (s_win)
constexpr uint32_t P_LIST = 2;               // near-threshold candidates a workgroup lists per window (more become plain events)
// MODE_MAX, growth phase: consecutive rows evaluated against the same set in one go (see the kernel)
constexpr uint32_t P_BATCH = 32;             // rows per batch at most
constexpr uint32_t P_BATCH_MEMBERS = 256;    // set size + 2 up to which batches are formed (4 members per lane)
constexpr size_t p_batch_bytes() { return size_t(P_BATCH) * (P_BATCH_MEMBERS + 2) * 4 * sizeof(double); }
constexpr size_t p_batch_lds() { return size_t(P_BATCH) * (2 + 12 + 24) * sizeof(double); }

struct PLine {  // a polled word on a cache line of its own (256 B apart)
    uint32_t v;
    uint32_t pad[63];
};
struct PRel {  // what a waiting workgroup polls with ONE 16-byte load, on a cache line of its own
    unsigned long long rel;   // the release word of the current window: its first event (window word, below)
    unsigned long long hint;  // events posted so far (atomicMin of window words; starts as all ones)
    unsigned long long pad[30];
};

struct PSync {
    uint32_t count;  // counter barrier: top-level arrivals (one per group and barrier), monotonic over the launch
    uint32_t pad0[63];
    PLine gcount[8];  // ... arrivals of group g = blockIdx % 8 (workgroups are dealt round-robin to the 8 XCDs)
    PLine ggen[8];    // ... completed barriers, one copy per group so 32 pollers share a line, not 256
    uint32_t timeout;
    uint32_t wg_thresh;  // windows of up to wg_thresh rows per workgroup are scanned a row per WORKGROUP (0: never)
    float wg_scale;      // ... and are cursor * wg_scale / size rows long (rounded up to whole rounds)
    uint32_t no_coarse;  // measurement aid: skip the COARSE tier
    uint32_t stop_at;    // head phase: leave at this stream position with status RUN (0: walk the whole stream)
    uint32_t seeded;     // the set is still only its seed positions: the launch works the initial state out itself
    const unsigned long long *seed_list;  // ... those positions (ctl->size of them)
    uint32_t small_rows;  // SMALL instantiation: member rows the LDS replica has room for
    uint32_t lds_bytes;   // the launch's dynamic LDS (a -DDVS_PERSIST_STAMPS build keeps its ticks in the last 128 bytes)
    uint32_t wmax;        // longest window of this engine (the control block's cap is the multi-launch scan's)
    uint32_t pad2[53];
    PRel rel[8];                            // one copy per group g = blockIdx % 8
    // Two sets of arrival records and listed candidates, taking turns by the window's parity: a workgroup that is
    // still walking window e's listed candidates reads the others' records and entries of window e, while a quick
    // one may already have scanned window e + 1 and stored its record for it.  Window e + 2's words (the same set as
    // e's) cannot be written before every workgroup has stored its record for e + 1, i.e. has left window e.
    unsigned long long wrec[2][P_MAXG];          // arrival record of every workgroup (a window word)
    unsigned long long soft[2][P_MAXG][P_LIST];  // the candidates a workgroup listed in the window (window words)
#ifdef DVS_PERSIST_STAMPS
    // four traced windows (epochs 12, 24, 36, 48): 100 MHz ticks per workgroup at the window's top, at its arrival record,
    // when it saw the release; [3] = the gathering block's own: gather begun, last record seen, release stored
    // [4..6]: the accept behind the window: this workgroup's job published (or nothing to publish), its totals read, its
    // rebuild of sl done
    unsigned long long trace[4][8][P_MAXG];  // ([7]: the speculative job's sums handed to the accumulators, or that point passed)
#endif
    unsigned long long dbg2[16];   // block 0 (owns a job): phase ticks
    unsigned long long dbg[16];    // mirror block: 100 MHz ticks per phase (scan, bar1, resolve, loo, bar2, finalize)
};

struct PState {  // replicated scalars (identical in every workgroup)
    uint64_t cursor, npos;
    uint32_t window, wmin, wmax;
    uint32_t n, li;
    double sumH, total_jsd, thr, band, wscale;
    uint32_t n_windows, n_events, n_accepts;
    // MODE_MAX: the set grows while it is below max_size (tentative pushes, records.rs:427-451)
    uint32_t max_size, stat;
    double mean_d, std_d, cov_d;  // statistics of the current set's delta_jsd
};

#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

// ---- window words.  [63:40] tag = P_TAG0 - window number (a NEWER window has the SMALLER tag: an atomicMin
// prefers it and a straggler of an older window can never replace it; all ones -- the hint words' initial
// value -- and zero -- everything else's -- are no window's tag), [39:3] stream position (all ones: none),
// [2:1] candidates listed (arrival record: how many this workgroup listed; release word: non-zero if anybody
// did), [0] set when the event is not a sure one (its exact score decides).
constexpr uint32_t P_TAG0 = 0xFFFFFEu;
constexpr uint32_t P_EPOCH_MAX = 0xFFFFF0u;  // windows a launch may walk (it leaves with status RUN there)
constexpr unsigned long long P_POS_NONE = (1ull << 37) - 1;
__device__ __forceinline__ unsigned long long p_word(uint32_t epoch, uint64_t pos, bool sure) {
    return ((unsigned long long)(P_TAG0 - epoch) << 40) | ((pos == SEL_NONE ? P_POS_NONE : pos) << 3) | (sure ? 0ull : 1ull);
}
__device__ __forceinline__ bool p_word_is(unsigned long long w, uint32_t epoch) { return uint32_t(w >> 40) == P_TAG0 - epoch; }
// the position a word of window `epoch` names (SEL_NONE: none, or a word of another window)
__device__ __forceinline__ uint64_t p_word_pos(unsigned long long w, uint32_t epoch) {
    const unsigned long long pos = (w >> 3) & P_POS_NONE;
    return (!p_word_is(w, epoch) || pos == P_POS_NONE) ? SEL_NONE : uint64_t(pos);
}
__device__ __forceinline__ bool p_word_sure(unsigned long long w) { return (w & 1ull) == 0; }
__device__ __forceinline__ uint32_t p_word_listed(unsigned long long w) { return uint32_t(w >> 1) & 3u; }

// returns false on timeout (every thread of the block gets the same answer).
// Two levels: a workgroup arrives on its group's counter, the last of a group arrives on the top
// counter, the last group publishes the generation to all eight group words.  Measured on MI355X
// (scripts/micro/barrier_bench.hip, 256 workgroups): 1.9 us against 3.7 us for one counter + one
// word -- the 256 same-address atomics serialise at ~11 ns each.
// The barrier is a pure rendezvous: no fences.  What is handed across it are atomic words whose readers
// can tell that they are complete (the accumulators' contribution counts) or whose writer consumed the
// result of the exchange that put them there (the `max` batches' results); the matrix is read-only for the
// whole launch, and what the mirror block stores to global memory is read by nobody before the kernel
// ends.  (An agent-scope release + acquire pair around the barrier -- L2 write-back and invalidate on
// every XCD -- cost another 4 us per barrier.)
__device__ bool grid_barrier(PSync *sync, uint32_t G, uint32_t &gen, int *s_ok) {
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t target = gen + 1;
        const uint32_t x = blockIdx.x & 7u, ng = G < 8u ? G : 8u;
        const uint32_t gsz = (G - x + 7u) >> 3;  // workgroups with blockIdx % 8 == x
        int ok = 1;
        bool released = false;
        const bool last = G <= P_FLAT_GRID
                              ? __hip_atomic_fetch_add(&sync->count, 1u, RLX_AGENT) == G * target - 1
                              : (__hip_atomic_fetch_add(&sync->gcount[x].v, 1u, RLX_AGENT) == gsz * target - 1 &&
                                 __hip_atomic_fetch_add(&sync->count, 1u, RLX_AGENT) == ng * target - 1);
        if (last) {
            for (uint32_t g = 0; g < ng; g++) __hip_atomic_store(&sync->ggen[g].v, target, RLX_AGENT);
            released = true;
        }
        if (!released) {
            uint32_t spins = 0;
            while (__hip_atomic_load(&sync->ggen[x].v, RLX_AGENT) < target) {
                if ((++spins & 255u) == 0 &&
                    (spins > P_SPIN_LIMIT || __hip_atomic_load(&sync->timeout, RLX_AGENT))) {
                    __hip_atomic_store(&sync->timeout, 1u, RLX_AGENT);
                    ok = 0;
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        *s_ok = ok;
    }
    __syncthreads();
    gen++;
    return *s_ok != 0;
}

// 16 bytes in one agent-scope load (one request for a PRel's two words; each word validates itself)
__device__ __forceinline__ uint4 p_load16_agent(const void *ptr) {
    uint4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(ptr) : "memory");
    return v;
}

// What a scanning wave needs of the window it is in (uniform values).
struct PWin {
    uint32_t epoch;               // the window's number
    const unsigned long long *hintp;  // this group's hint word
    PRel *rel;                    // the eight group copies (hints are posted to all of them)
    unsigned long long *wgev;     // LDS: this workgroup's first event so far (becomes its arrival record)
    uint32_t *nlist;              // LDS: candidates this workgroup has listed in this window
    unsigned long long *mysoft;   // this workgroup's list entries (sync->soft[epoch & 1][blockIdx.x])
};
// the first event posted so far in this window, as far as this group's hint word knows
__device__ __forceinline__ uint64_t p_hint_pos(const PWin &w) {
    return p_word_pos(__hip_atomic_load(w.hintp, RLX_AGENT), w.epoch);
}
// An event found by a wave (called by its lanes 0..7 at least): into the workgroup's record (LDS) and, as a
// hint for the waves still scanning, into every group's hint word -- fire and forget, nothing waits for it.
__device__ __forceinline__ void p_post_event_wave(const PWin &w, uint64_t p, bool sure, uint32_t lane) {
    const unsigned long long word = p_word(w.epoch, p, sure);
    if (lane == 0) atomicMin(w.wgev, word);
    if (lane < 8) atomicMin(&w.rel[lane].hint, word);
}
__device__ __forceinline__ void p_post_event_thread(const PWin &w, uint64_t p, bool sure) {
    const unsigned long long word = p_word(w.epoch, p, sure);
    atomicMin(w.wgev, word);
    for (uint32_t g = 0; g < 8; g++) atomicMin(&w.rel[g].hint, word);
}
// A candidate whose fast score is within FAST_BAND of the threshold: listed, not an event -- the scan goes on
// and the workgroups settle it in f64 after the rendezvous.  A workgroup's list holds P_LIST entries per
// window (window words: a reader knows which window an entry belongs to); the count travels in its arrival
// record.  A full list makes the candidate a plain (unsure) event.  One thread.
__device__ __forceinline__ void p_list_candidate(const PWin &w, uint64_t p) {
    const uint32_t idx = atomicAdd(w.nlist, 1u);
    if (idx < P_LIST) __hip_atomic_store(w.mysoft + idx, p_word(w.epoch, p, false), RLX_AGENT);
    else p_post_event_thread(w, p, false);
}
// wave-wide minimum of 64-bit words by DPP (register cross-lane moves, as dvs_wave_sum_dpp): every lane gets it
template <int CTRL>
__device__ __forceinline__ unsigned long long p_dpp_mov_u64(unsigned long long v) {
    int lo = int(uint32_t(v)), hi = int(uint32_t(v >> 32));
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
    return ((unsigned long long)uint32_t(hi) << 32) | uint32_t(lo);
}
__device__ __forceinline__ unsigned long long p_wave_min_u64(unsigned long long v) {
    auto mn = [](unsigned long long a, unsigned long long b) { return b < a ? b : a; };
    v = mn(v, p_dpp_mov_u64<0xB1>(v));   // quad_perm [1,0,3,2]
    v = mn(v, p_dpp_mov_u64<0x4E>(v));   // quad_perm [2,3,0,1]
    v = mn(v, p_dpp_mov_u64<0x141>(v));  // row_half_mirror
    v = mn(v, p_dpp_mov_u64<0x140>(v));  // row_mirror: every lane holds the minimum of its row of 16
    const int lo = int(uint32_t(v)), hi = int(uint32_t(v >> 32));
    auto row = [&](int l) {
        return ((unsigned long long)uint32_t(__builtin_amdgcn_readlane(hi, l)) << 32) | uint32_t(__builtin_amdgcn_readlane(lo, l));
    };
    return mn(mn(row(0), row(16)), mn(row(32), row(48)));
}

// The gathering block's part of a window's rendezvous, by ONE wave: lane l waits for the arrival records of
// workgroups l, l + 64, l + 128, l + 192 (four loads in flight per look; records of other windows are not looked
// at twice), the minimum is taken by DPP, lanes 0..7 store the release word -- no LDS, no barrier between the last
// record's arrival and the release.  `own`: the gathering block's own record.  Returns the release word; ok = false
// on a time-out (the launch is then given up: sync->timeout).  seen_all / stored: 100 MHz ticks (stamps builds).
__device__ __forceinline__ unsigned long long p_gather(PSync *sync, uint32_t G, uint32_t epoch, unsigned long long own,
                                                       uint32_t lane, bool &ok, unsigned long long *seen_all = nullptr,
                                                       unsigned long long *stored = nullptr) {
    constexpr uint32_t Q = P_MAXG / 64;
    unsigned long long w[Q];
    bool have[Q];
#pragma unroll
    for (uint32_t q = 0; q < Q; q++) {
        w[q] = ~0ull;
        have[q] = lane + 64 * q >= G - 1;  // (this workgroup's own record is `own`)
    }
    uint32_t spins = 0;
    ok = true;
    for (;;) {
        unsigned long long got[Q];
#pragma unroll
        for (uint32_t q = 0; q < Q; q++)
            got[q] = have[q] ? 0ull : __hip_atomic_load(&sync->wrec[epoch & 1u][lane + 64 * q], RLX_AGENT);
        bool all = true;
#pragma unroll
        for (uint32_t q = 0; q < Q; q++) {
            if (!have[q] && p_word_is(got[q], epoch)) {
                w[q] = got[q];
                have[q] = true;
            }
            all = all && have[q];
        }
        if (__ballot(!all) == 0ull) break;
        if ((++spins & 255u) == 0 && (spins > P_SPIN_LIMIT || __hip_atomic_load(&sync->timeout, RLX_AGENT))) {
            ok = false;
            break;
        }
        __builtin_amdgcn_s_sleep(1);
    }
    if (seen_all) *seen_all = __builtin_amdgcn_s_memrealtime();
    unsigned long long m = own & ~6ull, fl = own & 6ull;
#pragma unroll
    for (uint32_t q = 0; q < Q; q++)
        if (lane + 64 * q < G - 1 && have[q]) {
            m = (w[q] & ~6ull) < m ? (w[q] & ~6ull) : m;
            fl |= w[q] & 6ull;
        }
    m = p_wave_min_u64(m);
    const bool anyl = __ballot(fl != 0ull) != 0ull;
    const unsigned long long relw = m | (anyl ? 2ull : 0ull);
    if (ok && lane < 8) __hip_atomic_store(&sync->rel[lane].rel, relw, RLX_AGENT);
    if (!ok && lane == 0) __hip_atomic_store(&sync->timeout, 1u, RLX_AGENT);
    if (stored) *stored = __builtin_amdgcn_s_memrealtime();
    return relw;
}

// One wave's share of the window, as scan_rows_hot but against the unscaled vector sl:
// x = (sl_i + c_i / T) / n  computed as  fma(c, 1/T, sl) * (1/n).
template <typename T, bool COARSE>
__device__ __forceinline__ void p_scan_rows(const T *__restrict__ mat,
                                            const uint32_t *__restrict__ totals,
                                            const double *__restrict__ rowH, const double *sl,
                                            const float *slf, uint64_t B, const PState &st, double he_base,
                                            const PWin &win,
                                            uint64_t first, uint64_t stride, uint64_t nrows,
                                            uint32_t lane, uint32_t &nread,
                                            uint32_t &nprecise, uint32_t &nmid, bool coarse_on,
                                            bool burst_drop = true) {
    const double dn = double(st.n), rn = 1.0 / dn;
    const double thr_lo = st.thr - st.band;
    const double thr_fast = thr_lo - FAST_BAND, thr_sure = st.thr + st.band + FAST_BAND;
    (void)thr_lo;
    const bool vec = (B & 255) == 0;
    const double cband = coarse_band(B);
    const double thr_c_lo = st.thr - st.band - cband, thr_c_hi = st.thr + st.band + cband;
    for (uint64_t r = first; r < nrows; r += stride) {
        const uint64_t p = st.cursor + r;
        const T *rp = mat + p * B;
        const uint64_t ev = p_hint_pos(win);
        const uint32_t tot = totals[p];
        const double hrow = rowH[p];
        if (ev < p) break;
        if (tot == 0) continue;
        const double rt = 1.0 / double(tot);
        const double mean_entropy = (he_base + hrow) / dn;
        nread++;
        if constexpr (COARSE) {
            if (vec && coarse_on) {  // COARSE tier: decides every row farther than cband from the threshold
                const float rtn = float(rt * rn);
                const dvs_f2 r2 = {rtn, rtn};
                double c0 = 0.0, c1 = 0.0;
                bool done16 = false;
                if constexpr (sizeof(T) == 2) {
                    // 16-bit rows of 4096 bins: eight 16-byte loads per lane (lane l owns bins
                    // 512 j + 8 l .. + 7), the whole 8 KiB row requested at once.  (Requesting it beside the
                    // look at the event word instead of behind it -- all of it, half, a quarter -- measured
                    // the same step and event-free pass, and half again as much traffic for rows dropped.)
                    if (B == 4096) {
                        uint4 raw8[8];
#pragma unroll
                        for (int j = 0; j < 8; j++) raw8[j] = *reinterpret_cast<const uint4 *>(rp + j * 512 + lane * 8);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int j = 0; j < 8; j++) {
                            asm volatile("" : "+v"(raw8[j].x), "+v"(raw8[j].y), "+v"(raw8[j].z), "+v"(raw8[j].w));
                            coarse8(raw8[j], slf + j * 512 + lane * 8, r2, c0, c1);
                        }
                        done16 = true;
                    }
                }
                if (!done16) {
                // Bursts of C_CH chunks: 4 KiB of a row requested at a time keep the memory pipe as full
                // as 16 KiB do (scripts/micro/stream_read.hip), and a row that an earlier event has made
                // pointless is dropped at the next burst with its other bytes never requested.
                constexpr int C_CH = sizeof(T) == 2 ? 16 : 8;  // (16-bit rows: the same bytes per burst)
                const uint64_t full = B - B % (256 * C_CH);
                uint64_t i0 = 0;
                bool dropped = false;
                for (; i0 < full; i0 += 256 * C_CH) {
                    if (i0 && burst_drop && p_hint_pos(win) < p) {
                        dropped = true;
                        break;
                    }
                    Raw4<T> raw[C_CH];
#pragma unroll
                    for (int j = 0; j < C_CH; j++) raw[j].load(rp + i0 + uint64_t(j) * 256 + lane * 4);
                    // (the scheduler would otherwise sink each load to its use: one 1 KiB request in
                    // flight per wave instead of four)
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < C_CH; j += 2) {
                        const uint64_t i = i0 + uint64_t(j) * 256 + lane * 4;
                        raw[j].pin();  // nothing of chunk j is consumed (converted) above this point
                        raw[j + 1].pin();
                        c0 += double(coarse4(raw[j].c, *reinterpret_cast<const float4 *>(slf + i), r2));
                        c1 += double(coarse4(raw[j + 1].c, *reinterpret_cast<const float4 *>(slf + i + 256), r2));
                    }
                }
                if (dropped) {  // (an earlier event exists: this wave's later rows are pointless too)
                    nread--;    // not a row scored: it does not count towards the algorithmic bytes
                    break;
                }
                for (; i0 < B; i0 += 256) {
                    const uint64_t i = i0 + lane * 4;
                    Raw4<T> raw;
                    raw.load(rp + i);
                    c0 += double(coarse4(raw.c, *reinterpret_cast<const float4 *>(slf + i), r2));
                }
                }  // !done16
                const double jf0 = -dvs_wave_sum_dpp(c0 + c1) - mean_entropy;
                if (!(jf0 > thr_c_lo)) continue;  // (NaN: a negative bin, rejected as the reference does)
                if (jf0 > thr_c_hi) {
                    p_post_event_wave(win, p, true, lane);
                    continue;
                }
                nmid++;
            }
        }
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0, xmin = 0.0;
        if (vec) {
            // (behind the COARSE tier this pass sees a few dozen rows per selection: half the burst, half the registers)
            constexpr int F_CH = COARSE ? P_CH / 2 : P_CH;
            const uint64_t full = B - B % (256 * F_CH);
            uint64_t i0 = 0;
            for (; i0 < full; i0 += 256 * F_CH) {  // F_CH chunks of 1 KiB per wave instruction per burst
                Raw4<T> raw[F_CH];
#pragma unroll
                for (int j = 0; j < F_CH; j++) raw[j].load(rp + i0 + uint64_t(j) * 256 + lane * 4);
#pragma unroll
                for (int j = 0; j < F_CH; j++) {
                    const uint64_t i = i0 + uint64_t(j) * 256 + lane * 4;
                    const double2 b01 = *reinterpret_cast<const double2 *>(sl + i);
                    const double2 b23 = *reinterpret_cast<const double2 *>(sl + i + 2);
                    double v0, v1, v2, v3;
                    raw[j].get(v0, v1, v2, v3);
                    const double x0 = fma(v0, rt, b01.x) * rn, x1 = fma(v1, rt, b01.y) * rn;
                    const double x2 = fma(v2, rt, b23.x) * rn, x3 = fma(v3, rt, b23.y) * rn;
                    a0 += fast_neg_xlog2x(x0);
                    a1 += fast_neg_xlog2x(x1);
                    a2 += fast_neg_xlog2x(x2);
                    a3 += fast_neg_xlog2x(x3);
                    xmin = fmin(fmin(xmin, fmin(x0, x1)), fmin(x2, x3));
                }
            }
            for (; i0 < B; i0 += 256) {
                const uint64_t i = i0 + lane * 4;
                double v0, v1, v2, v3;
                load4(rp, i, v0, v1, v2, v3);
                const double2 b01 = *reinterpret_cast<const double2 *>(sl + i);
                const double2 b23 = *reinterpret_cast<const double2 *>(sl + i + 2);
                const double x0 = fma(v0, rt, b01.x) * rn, x1 = fma(v1, rt, b01.y) * rn;
                const double x2 = fma(v2, rt, b23.x) * rn, x3 = fma(v3, rt, b23.y) * rn;
                a0 += fast_neg_xlog2x(x0);
                a1 += fast_neg_xlog2x(x1);
                a2 += fast_neg_xlog2x(x2);
                a3 += fast_neg_xlog2x(x3);
                xmin = fmin(fmin(xmin, fmin(x0, x1)), fmin(x2, x3));
            }
        } else {
            for (uint64_t i = lane; i < B; i += 64) {
                const double x = fma(row_value(rp, i), rt, sl[i]) * rn;
                a0 += fast_neg_xlog2x(x);
                xmin = fmin(xmin, x);
            }
        }
        const double hf = dvs_wave_sum((a0 + a1) + (a2 + a3));
        const double mn = dvs_wave_min(xmin);
        const double jf = hf - mean_entropy;
        if (!(mn < 0.0) && jf > thr_sure) {
            // above the threshold by more than every error bound: no f64 re-evaluation needed
            p_post_event_wave(win, p, true, lane);
        } else if (!(mn < 0.0) && jf > thr_fast && lane == 0) {
            {
                // within FAST_BAND of the threshold: listed, not an event -- the scan goes on and
                // the workgroups settle it in f64 after the rendezvous (a wave doing that alone
                // would hold the whole grid at the barrier for 4^k f64 logarithms)
                nprecise++;
                p_list_candidate(win, p);
            }
        }
    }
}

// The same scores with one row per WORKGROUP (8 waves x 512 bins at k=6): a row takes an eighth of
// the time a lone wave needs for it, so a short window -- early in the stream an accept comes every
// few hundred rows and the scan is pure latency -- ends that much sooner, and the rows in flight
// when an event is found are 255, not 2040.  One barrier per row: the waves' partial sums alternate
// between two sets of LDS slots.  Lane l of wave w owns bins 4 (512 c + 64 w + l) .. + 3 of every
// 2048-bin chunk c.  red: 2 x 24 doubles.
template <typename T, bool COARSE>
__device__ __forceinline__ void p_scan_rows_wg(const T *__restrict__ mat, const uint32_t *__restrict__ totals,
                                               const double *__restrict__ rowH, const double *sl,
                                               const float *slf, uint64_t B,
                                               const PState &st, double he_base, const PWin &win,
                                               uint64_t first, uint64_t stride,
                                               uint64_t nrows, double *red, uint32_t &nread,
                                               uint32_t &nprecise, uint32_t &nmid, bool coarse_on,
                                            bool burst_drop = true) {
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const double dn = double(st.n), rn = 1.0 / dn;
    const double thr_fast = st.thr - st.band - FAST_BAND, thr_sure = st.thr + st.band + FAST_BAND;
    const bool vec = (B & 2047) == 0;
    const double cband = coarse_band(B);
    const double thr_c_lo = st.thr - st.band - cband, thr_c_hi = st.thr + st.band + cband;
    uint32_t par = 0;
    for (uint64_t r = first; r < nrows; r += stride, par ^= 1) {
        const uint64_t p = st.cursor + r;
        const T *rp = mat + p * B;
        const uint64_t ev = wave == 0 ? p_hint_pos(win) : 0ull;
        const uint32_t tot = totals[p];
        const double hrow = rowH[p];
        const double rt = tot ? 1.0 / double(tot) : 0.0;
        if constexpr (COARSE) {
            if (vec && B <= 4 * 2048 && coarse_on) {  // COARSE tier first (select_dev.h); its sums use red[48..]
                const float rtn = float(rt * rn);
                const dvs_f2 r2 = {rtn, rtn};
                double c0 = 0.0;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const uint64_t i = uint64_t(j) * 2048 + tid * 4;
                    if (i < B) {
                        Raw4<T> raw;
                        raw.load(rp + i);
                        c0 += double(coarse4(raw.c, *reinterpret_cast<const float4 *>(slf + i), r2));
                    }
                }
                c0 = dvs_wave_sum_dpp(c0);
                double *cslot = red + 48 + par * 16;
                if (lane == 0) {
                    cslot[wave] = c0;
                    if (wave == 0) cslot[8] = __longlong_as_double((long long)ev);
                }
                __syncthreads();
                if ((unsigned long long)__double_as_longlong(cslot[8]) < p) break;
                if (tot == 0) continue;
                double hc = 0.0;
#pragma unroll
                for (int w = 0; w < P_THREADS / 64; w++) hc += cslot[w];
                const double jf0 = -hc - (he_base + hrow) / dn;
                if (tid == 0) nread++;
                if (!(jf0 > thr_c_lo)) continue;
                if (jf0 > thr_c_hi) {
                    if (tid < 8) p_post_event_wave(win, p, true, tid);  // (eight lanes, one instruction: every thread knows the score)
                    continue;
                }
                if (tid == 0) {
                    nmid++;
                    nread--;  // counted again below
                }
            }
        }
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0, xmin = 0.0;
        if (vec) {
            for (uint64_t i0 = 0; i0 < B; i0 += 4 * 2048) {  // up to four chunks requested at once
                Raw4<T> raw[4];
#pragma unroll
                for (int j = 0; j < 4; j++)
                    if (i0 + uint64_t(j) * 2048 < B) raw[j].load(rp + i0 + uint64_t(j) * 2048 + tid * 4);
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const uint64_t i = i0 + uint64_t(j) * 2048 + tid * 4;
                    if (i < B) {
                        const double2 b01 = *reinterpret_cast<const double2 *>(sl + i);
                        const double2 b23 = *reinterpret_cast<const double2 *>(sl + i + 2);
                        double v0, v1, v2, v3;
                        raw[j].get(v0, v1, v2, v3);
                        const double x0 = fma(v0, rt, b01.x) * rn, x1 = fma(v1, rt, b01.y) * rn;
                        const double x2 = fma(v2, rt, b23.x) * rn, x3 = fma(v3, rt, b23.y) * rn;
                        a0 += fast_neg_xlog2x(x0);
                        a1 += fast_neg_xlog2x(x1);
                        a2 += fast_neg_xlog2x(x2);
                        a3 += fast_neg_xlog2x(x3);
                        xmin = fmin(fmin(xmin, fmin(x0, x1)), fmin(x2, x3));
                    }
                }
            }
        } else {
            for (uint64_t i = tid; i < B; i += P_THREADS) {
                const double x = fma(row_value(rp, i), rt, sl[i]) * rn;
                a0 += fast_neg_xlog2x(x);
                xmin = fmin(xmin, x);
            }
        }
        const double hw = dvs_wave_sum((a0 + a1) + (a2 + a3));
        const bool negw = __ballot(xmin < 0.0) != 0ull;
        double *slot = red + par * 24;
        if (lane == 0) {
            slot[wave] = hw;
            slot[8 + wave] = negw ? 1.0 : 0.0;
            if (wave == 0) slot[16] = __longlong_as_double((long long)ev);
        }
        __syncthreads();
        // every thread takes the same decisions from the same LDS words
        if ((unsigned long long)__double_as_longlong(slot[16]) < p) break;
        if (tot == 0) continue;
        double hf = 0.0, neg = 0.0;
#pragma unroll
        for (int w = 0; w < P_THREADS / 64; w++) {
            hf += slot[w];
            neg += slot[8 + w];
        }
        const double jf = hf - (he_base + hrow) / dn;
        // (a sure event goes out from eight lanes in one instruction: every thread knows the score)
        if (tid < 8 && neg == 0.0 && jf > thr_fast && jf > thr_sure) p_post_event_wave(win, p, true, tid);
        if (tid == 0) {
            nread++;
            if (neg == 0.0 && jf > thr_fast) {
                if (jf > thr_sure) {
                } else {
                    nprecise++;
                    p_list_candidate(win, p);
                }
            }
        }
    }
}

// The row-per-workgroup scan as a stream (4^k = 4096 bins, COARSE tier): every thread keeps the
// two 16-byte requests of each of the next D rows of its workgroup in flight, so the grid streams
// rows at memory speed while an event still leaves only ~one row per workgroup to drain.  A row the
// COARSE tier cannot decide is scored again by p_scan_rows_wg (FAST tier) as a one-row window.
// red: the 128-double scratch area minus its first 32 (slots [48..80) are this function's).
template <typename T>
__device__ __forceinline__ void p_scan_rows_wg_stream(const T *__restrict__ mat,
                                                      const uint32_t *__restrict__ totals,
                                                      const double *__restrict__ rowH, const double *sl,
                                                      const float *slf, const PState &st, double he_base,
                                                      const PWin &win, uint64_t first,
                                                      uint64_t stride, uint64_t nrows, double *red,
                                                      uint32_t &nread, uint32_t &nprecise, uint32_t &nmid,
                                                      bool no_first_poll = true) {
    constexpr uint64_t B = 4096;
    constexpr int D = 4;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const double dn = double(st.n), rn = 1.0 / dn;
    const double cband = coarse_band(B);
    const double thr_c_lo = st.thr - st.band - cband, thr_c_hi = st.thr + st.band + cband;
    constexpr bool W16 = sizeof(T) == 2;  // 16-bit rows: ONE 16-byte load per thread (bins 8 tid .. + 7)
    struct Row {
        Raw4<T> r0, r1;
        uint4 q;
        uint32_t tot;
        double hrow;
        unsigned long long ev;
    };
    // poll: also look at the event word (a device-scope round trip, longer than an L2 hit on the row).  The
    // window's first rows go without it -- the window has only just begun, and a row scored for nothing
    // changes no answer -- so the first score waits for its row alone.
    auto issue = [&](Row &w, uint64_t r, bool poll) {
        const uint64_t p = st.cursor + r;
        const T *rp = mat + p * B;
        if constexpr (W16) {
            w.q = *reinterpret_cast<const uint4 *>(rp + tid * 8);
        } else {
            w.r0.load(rp + tid * 4);
            w.r1.load(rp + 2048 + tid * 4);
        }
        w.tot = totals[p];
        if (wave == 0) {  // the row's entropy and the event word travel through LDS with the partial sums
            w.hrow = rowH[p];
            w.ev = poll ? p_hint_pos(win) : SEL_NONE;
        }
    };
    uint32_t par = 0;
    auto process = [&](Row &w, uint64_t r) -> bool {  // false: an earlier event exists, the workgroup is done
        const uint64_t p = st.cursor + r;
        const float rtn = w.tot ? float(rn / double(w.tot)) : 0.0f;
        const dvs_f2 r2 = {rtn, rtn};
        double c;
        if constexpr (W16) {
            asm volatile("" : "+v"(w.q.x), "+v"(w.q.y), "+v"(w.q.z), "+v"(w.q.w));
            double ca = 0.0, cb = 0.0;
            coarse8(w.q, slf + tid * 8, r2, ca, cb);
            c = ca + cb;
        } else {
            w.r0.pin();
            w.r1.pin();
            c = double(coarse4(w.r0.c, *reinterpret_cast<const float4 *>(slf + tid * 4), r2)) +
                double(coarse4(w.r1.c, *reinterpret_cast<const float4 *>(slf + 2048 + tid * 4), r2));
        }
        const double cs = dvs_wave_sum_dpp(c);
        double *slot = red + 48 + par * 16;
        par ^= 1;
        if (lane == 0) {
            slot[wave] = cs;
            if (wave == 0) {
                slot[8] = __longlong_as_double((long long)w.ev);
                slot[9] = w.hrow;
            }
        }
        __syncthreads();
        if ((unsigned long long)__double_as_longlong(slot[8]) < p) return false;
        if (w.tot == 0) return true;
        double hc = 0.0;
#pragma unroll
        for (int q = 0; q < P_THREADS / 64; q++) hc += slot[q];
        const double jf0 = -hc - (he_base + slot[9]) / dn;
        if (tid == 0) nread++;
        if (!(jf0 > thr_c_lo)) return true;  // (NaN: a negative bin, rejected as the reference does)
        if (jf0 > thr_c_hi) {
            if (tid < 8) p_post_event_wave(win, p, true, tid);  // (eight lanes, one instruction: every thread knows the score)
            return true;
        }
        if (tid == 0) {
            nmid++;
            nread--;  // counted again by the FAST pass
        }
        uint32_t nm = 0;
        p_scan_rows_wg<T, false>(mat, totals, rowH, sl, slf, B, st, he_base, win, r, nrows, nrows, red,
                                 nread, nprecise, nm, false);
        return true;
    };
    Row w[D];
#pragma unroll
    for (int d = 0; d < D; d++)
        if (first + uint64_t(d) * stride < nrows) issue(w[d], first + uint64_t(d) * stride, no_first_poll ? d >= 2 : true);
    for (uint64_t base = first; base < nrows; base += uint64_t(D) * stride) {
#pragma unroll
        for (int d = 0; d < D; d++) {
            const uint64_t r = base + uint64_t(d) * stride;
            if (r >= nrows) return;
            if (!process(w[d], r)) return;
            if (r + uint64_t(D) * stride < nrows) issue(w[d], r + uint64_t(D) * stride, true);
        }
    }
}

// Window policy of the persistent engine.  Short windows run a row per workgroup (p_scan_rows_wg) and
// are whole rounds of the scanning workgroups long; the scale aims at ~3 in 4 windows ending in an
// accept (the accept probability at stream position i is ~ size / i).
__device__ __forceinline__ uint32_t p_next_window(const PState &st, uint32_t nwg, uint32_t wg_thresh,
                                                  double wg_scale, bool &wgmode) {
    if (wg_thresh) {
        uint64_t w = uint64_t(double(st.cursor) * wg_scale / double(st.n ? st.n : 1u));
        w = (w + nwg - 1) / nwg * nwg;
        if (w < nwg) w = nwg;
        if (w <= uint64_t(wg_thresh) * nwg) {
            wgmode = true;
            return uint32_t(w);
        }
    }
    wgmode = false;
    return sel_next_window(st.cursor, st.n, st.wmin, st.wmax, st.wscale);
}

// argmin (strict '<' from 1e6, first index), runner-up, mean and standard deviation of the
// members' delta_jsd by ONE wave: lane l owns members l, l + 64, ... (Q per lane; re-read from LDS
// in every pass when Q > 1 -- registers are the scan loop's).
// scratch[100..105] = min, index, second, mean, sd, any-risky.
template <uint32_t Q>
__device__ __forceinline__ void p_argmin(const double *s_dl, const double *s_ds, uint32_t n, uint64_t B,
                                         uint32_t lane, double *scratch) {
    const double v0 = lane < n ? s_dl[lane] : 1e6;
    auto val = [&](uint32_t q) { return (Q == 1 || q == 0) ? v0 : s_dl[lane + 64 * q]; };
    double best = 1e6, acc = 0.0;
    bool rk = false;
#pragma unroll
    for (uint32_t q = 0; q < Q; q++) {
        const uint32_t r = lane + 64 * q;
        if (r < n) {
            const double v = val(q);
            rk |= sum_risky(s_ds[r], B);
            acc += v;
            if (v < best) best = v;
        }
    }
    const double dn = double(n);
    const double mnv = dvs_wave_min(best);
    double fi = 4294967295.0;
#pragma unroll
    for (uint32_t q = 0; q < Q; q++)
        if (mnv < 1e6 && lane + 64 * q < n && val(q) == mnv) fi = fmin(fi, double(lane + 64 * q));
    const double first = dvs_wave_min(fi);
    const uint32_t lw = (first < 4294967295.0) ? uint32_t(first) : 0u;
    const double mu = dvs_wave_sum(acc) / dn;
    double sec = 1e6, tv = 0.0;
#pragma unroll
    for (uint32_t q = 0; q < Q; q++) {
        const uint32_t r = lane + 64 * q;
        if (r < n) {
            const double v = val(q);
            if (r != lw && v < sec) sec = v;
            const double t = v - mu;
            tv += t * t;
        }
    }
    sec = dvs_wave_min(sec);
    const double var = dvs_wave_sum(tv);
    const unsigned long long anyr = __ballot(rk);
    if (lane == 0) {
        scratch[100] = mnv;
        scratch[101] = double(lw);
        scratch[102] = sec;
        scratch[103] = mu;
        scratch[104] = sqrt(var / (dn - 1.0));
        scratch[105] = anyr ? 1.0 : 0.0;
    }
}

// LDS: [sl B f64][scratch 128 f64][s_mH, s_tot, s_rt, s_dl, s_ds maxn f64][s_pos maxn u64]
//      [s_slot maxn u32][s_win P_WINWORDS u64][flags][log2 table][s_prev (maxn + 1) x 2 u64]
// maxn = p_maxn(CACHED): compile-time offsets (runtime ones cost registers the scan loop needs),
// smaller beyond 4096 bins so that 4^7 bins (128 KB of sl) still fit.
// CACHED: B <= P_J * 512, so a thread's share of the candidate row stays in registers
// MAXM: select_max_divergent (records.rs:390-454).  While the set is below max_size an accepted
// candidate is a TENTATIVE push: clone + push (the running sums ARE the clone's re-sums while the
// set has only grown, see resolve_kernel), leave-one-out over the n + 1 members, and the bigger set
// is kept iff the standard deviation (or the coefficient of variation) of the members' delta_jsd
// rose; a rollback leaves every replica untouched.  At max_size the stream continues as above
// (replace_lowest).  The summed vector S lives in LDS beside sl.  Anything too close to call ends
// the launch with the event unconsumed; the multi-launch kernels (and the arbiter) take it.
// SMALL: nmost over 16-bit count rows of exactly 4096 bins with a set of <= P_SMALL_ROWS members.  The
// members' count rows (8 KB each) sit in every workgroup's LDS, thread-major (thread t's bins
// j * 512 + t, j < 8, are the 16 bytes at t * 16: no thread ever reads another's part).  An accept then
// takes its leave-one-out job's member counts and the new lowest member's row from LDS instead of two
// memory round trips, and the candidate's own row is requested before the rendezvous that ends the
// window (the event word already names it), not behind it.
template <typename T, bool CACHED, bool MAXM = false, bool SMALL = false>
__global__ __launch_bounds__(P_THREADS, 2) void persist_nmost_kernel(SelDev d, const T *__restrict__ mat,
                                                                    PSync *sync, unsigned long long *part, uint32_t G) {
    static_assert(!SMALL || (CACHED && !MAXM && sizeof(T) == 2), "SMALL: nmost, 16-bit rows, state in the register cache");
    constexpr uint32_t maxn = SMALL ? P_SMALLN : p_maxn(CACHED, MAXM);
    // SPEC: nmost over a count matrix whose rows fit the register cache -- the accept's first steps (the
    // candidate's row, its frequencies, this workgroup's leave-one-out job) are taken while the window's
    // rendezvous is still completing (see the event loop)
    constexpr bool SPEC = CACHED && !MAXM && sizeof(T) <= 4;
    // SPEC_BIG: the same idea for count rows that do not fit the register cache (4^7 bins and beyond): this
    // workgroup's leave-one-out job for the candidate the event record names is worked out while the
    // rendezvous completes -- at 4^7 bins a job is sixteen bins a thread in f64 plus 2 x 32 KB of counts from
    // L2, 6 us that used to follow the release
    constexpr bool SPEC_BIG = !CACHED && !MAXM && sizeof(T) <= 4;
    // OWN (nmost over count rows in the register cache, sets too large for SMALL): the leave-one-out jobs are
    // dealt by member SLOT, not by member order -- a slot keeps its member until that member is replaced, so
    // a workgroup serves the same member accept after accept and keeps that member's count row in its LDS
    // (s_own, thread-major like the candidate's registers; refreshed from the candidate's registers when the
    // slot changes hands): a job never waits for a row from memory.  The accumulators are indexed by slot too.
    // (Tried on top of it, round 4: an LDS cache of the rows of the members closest to being the lowest, filled by
    // LDS-DMA, for the rebuild of sl -- 97 % of the accepts found their row there and the step did not move: the
    // wait at that point is not for the row.  DESIGN.md 4.3.)
    constexpr bool OWN = SPEC && !SMALL;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint64_t B0 = d.B;
    const uint64_t B = B0;  // (the event loop below takes its own, laundered copy)
    double *sl = reinterpret_cast<double *>(smem);
    // f32 copy of sl / n for the COARSE tier (count matrices whose state fits the register cache)
    constexpr bool COARSE = CACHED && sizeof(T) <= 4;  // (count matrices, 32- or 16-bit)
    float *slf = reinterpret_cast<float *>(sl + ((B + 1) & ~1ull));
    static_assert(!MAXM || CACHED, "the growth phase keeps the candidate's frequencies in registers");
    double *Sl = sl + ((B + 1) & ~1ull) + (COARSE ? ((B + 3) & ~3ull) / 2 : 0);  // S (MAXM only)
    double *scratch = Sl + (MAXM ? ((B + 1) & ~1ull) : 0);
    double *s_mH = scratch + 128;
    double *s_tot = s_mH + maxn;  // member row totals and their correctly rounded reciprocals
    double *s_rt = s_tot + maxn;
    double *s_dl = s_rt + maxn;   // delta_jsd per member (sets of 64 members and more)
    double *s_ds = s_dl + maxn;   // sum of each member's mean vector
    uint64_t *s_pos = reinterpret_cast<uint64_t *>(s_ds + maxn);  // matrix row of each member
    uint32_t *s_slot = reinterpret_cast<uint32_t *>(s_pos + maxn);
    // the window's words: [0] this workgroup's first event (its arrival record), [1] candidates it listed,
    // [2] scratch minimum of the listed candidates' walk, [4..6] what the polling thread saw (release word, hint
    // word, released?)
    unsigned long long *s_win = reinterpret_cast<unsigned long long *>(s_slot + maxn + (maxn & 1));
    int *s_flag = reinterpret_cast<int *>(s_win + P_WINWORDS);
    static_assert(maxn % 4 == 0 && P_WINWORDS % 2 == 0, "s_ltab below must sit on a 16-byte boundary");
    double2 *s_ltab = reinterpret_cast<double2 *>(s_win + P_WINWORDS + 2);  // log2_tab's 128 entries
    // the leave-one-out accumulators' totals as of their previous use, this group's replica (p_acc_complete)
    unsigned long long *s_prev = reinterpret_cast<unsigned long long *>(s_ltab + 128);
    unsigned char *s_tail = reinterpret_cast<unsigned char *>(s_prev + uint64_t(maxn + 1) * 2);
    [[maybe_unused]] uint16_t *s_rows = reinterpret_cast<uint16_t *>(s_tail);  // SMALL: member count rows by slot
    // OWN: slot -> member order, this workgroup's own member's counts
    [[maybe_unused]] uint32_t *s_inv = reinterpret_cast<uint32_t *>(s_tail);
    [[maybe_unused]] T *s_own = reinterpret_cast<T *>(s_inv + maxn);
    // MAXM: a batch's row totals, row entropies and decisions (12 doubles per row)
    [[maybe_unused]] double *s_bt = reinterpret_cast<double *>(s_tail);
    [[maybe_unused]] double *s_bH = s_bt + P_BATCH;
    [[maybe_unused]] double *s_bev = s_bH + P_BATCH;
    [[maybe_unused]] double *s_bpart = s_bev + 12 * P_BATCH;  // a batch job's sums by wave: [row][3][8 waves]
    if (threadIdx.x < 128) log2_tab_fill(s_ltab, threadIdx.x);
    for (uint32_t i = threadIdx.x; i < (maxn + 1) * 2; i += P_THREADS) s_prev[i] = 0ull;  // (the host zeroed the accumulators)
    SelCtl *ctl = d.ctl;
    const int tid0 = threadIdx.x;
    const int tid = tid0;
    [[maybe_unused]] const uint32_t lane = tid & 63, wave = tid >> 6;
    // (one block mirrors the state into global memory: it owns no member in the leave-one-out pass and scans
    // nothing, so its stores overlap the other blocks' work)
    // Two workgroups scan nothing (grids of three and more): the last one GATHERS the arrival records of every
    // window and stores its release word -- it stores nothing else, so no store of its own ever sits in front of
    // its polling loads --, the one before it MIRRORS the state into global memory (`lead`).  The mirror
    // block's stores (64 KB per accept at 4^6 bins) used to sit in front of the same block's arrival: every
    // window waited ~3 us for them to drain (profiles/r04_d stamps).  It announces itself for the NEXT window as
    // soon as it has read everything the others wrote in this one, before it stores anything (see early_rec).
    const bool two_roles = G >= 3;
    const bool gath = blockIdx.x == G - 1;
    const bool lead = blockIdx.x == (two_roles ? G - 2 : G - 1);
    const uint32_t n_work = G > 1 ? G - (two_roles ? 2u : 1u) : 1u;  // workgroups that scan and take single jobs

    // ---- replica of the state (global memory is quiescent: written by earlier launches)
    PState st;
    st.cursor = ctl->cursor;
    st.npos = ctl->npos;
    const bool head_phase = sync->stop_at != 0u && uint64_t(sync->stop_at) < st.npos;
    if (head_phase) st.npos = sync->stop_at;  // (the next launch carries on from the mirrored state)
    st.window = ctl->window;
    st.wmin = ctl->window_min;
    // (a window costs this engine nothing for being long: every wave looks at the hint word before every row and
    // stops behind the first event -- the cap only decides how often an event-free stretch of the late stream is
    // interrupted by a rendezvous and its drain)
    st.wmax = sync->wmax > ctl->window_max ? sync->wmax : ctl->window_max;
    st.n = ctl->size;
    st.li = ctl->lowest;
    st.sumH = ctl->sum_entropy;
    st.total_jsd = ctl->total_jsd;
    st.thr = ctl->thr;
    st.band = ctl->band;
    st.wscale = ctl->wscale;
    st.n_windows = st.n_events = st.n_accepts = 0;
    st.max_size = MAXM ? ctl->max_size : st.n;
    st.stat = ctl->stat;
    st.mean_d = ctl->mean_delta;
    st.std_d = ctl->std_delta;
    st.cov_d = ctl->cov_delta;
    // (rows scored by the launches in front of this one: global memory is quiescent here, and this launch's own
    // additions come at its exit)
    if (lead && tid == 0) ctl->rows_before_launch = ctl->rows_scored;
    if (ctl->status != SEL_RUN || ctl->ev_kind != 0 || st.n > maxn || st.n < 2) {
        if (lead && tid == 0 && ctl->status == SEL_RUN) ctl->why[7]++;  // (a pending event or a set the replica cannot hold)
        return;
    }
    if (MAXM && st.n < st.max_size && (ctl->s_is_resum == 0 || st.n + 2 > maxn)) {  // (multi-launch kernels)
        if (lead && tid == 0) ctl->why[6]++;
        return;
    }
    // SEEDED start (nmost, the state in the register cache): nothing but the control block and the
    // seed positions exists yet -- no seed / rebuild / loo / finalize launches ran; the initial set
    // (SummedRecords::new, records.rs:27-68) is worked out further down by the grid itself.
    const bool seeded = !MAXM && CACHED && sync->seeded != 0u;
    for (uint32_t r = tid; r < st.n; r += P_THREADS) {
        const uint32_t slot = seeded ? r : d.ord[r];
        const uint64_t mp = seeded ? sync->seed_list[r] : d.mPos[slot];
        const double t = double(d.totals[mp]);
        s_slot[r] = slot;
        s_mH[r] = seeded ? d.rowH[mp] : d.mH[slot];
        s_pos[r] = mp;
        s_tot[r] = t;
        s_rt[r] = 1.0 / t;
    }
    __syncthreads();
    if (!seeded) {
        const double *low = d.M + uint64_t(s_slot[st.li]) * B;
        const double rn0 = 1.0 / double(st.n);
        for (uint64_t i = tid; i < B; i += P_THREADS) {
            const double v = d.S[i] - low[i];
            sl[i] = v;
            if (COARSE) slf[i] = coarse_sl(v, rn0);
            if (MAXM) Sl[i] = d.S[i];
        }
    }
    __syncthreads();

    uint32_t gen = 0, epoch = 0;
    uint32_t nread = 0, nprecise = 0, nmid = 0;
    uint32_t exit_status = SEL_RUN;  // what the lead block writes to ctl->status on exit
    uint32_t arb_stage = 0;
    uint64_t arb_pos = 0;
    uint32_t pend_kind = 0;  // ev_kind left pending for the multi-launch kernels (finalize tie)
    bool bail = false;       // leave with the state as it stands and status RUN (MAXM: an undecidable push)
    [[maybe_unused]] uint32_t mx_batch = 1;  // MAXM: rows the next tentative push takes along (identical in every workgroup)
    [[maybe_unused]] uint32_t n_batches = 0;  // MAXM: batches of this launch so far (their results alternate between two areas)
    // the mirror block (two_roles): the window whose arrival record it has already stored -- at the end of the
    // window before, the moment it had read the last word another workgroup wrote for that window (the release,
    // the listed candidates, the leave-one-out totals), ahead of its own stores
    uint32_t early_rec = 0xFFFFFFFFu;
    auto announce_early = [&](uint32_t next_epoch) {
        if (lead && two_roles && !MAXM) {
            if (threadIdx.x == 0)
                __hip_atomic_store(&sync->wrec[next_epoch & 1u][blockIdx.x], p_word(next_epoch, SEL_NONE, false), RLX_AGENT);
            early_rec = next_epoch;
        }
    };
    const uint64_t wpb = P_THREADS / 64;
    const uint32_t nwg = n_work;                      // scanning workgroups
    const uint64_t nwaves = uint64_t(nwg) * wpb;      // scanning waves
    const uint32_t wg_thresh = sync->wg_thresh;       // (written by the host before the launch)
    const double wg_scale = double(sync->wg_scale);
    const bool coarse_on = (sync->no_coarse & 1u) == 0;
    bool wgmode = false;
    if (wg_thresh) st.window = p_next_window(st, nwg, wg_thresh, wg_scale, wgmode);
    // Leave-one-out jobs (the set size is constant in this mode): job (r, part) covers the
    // 512-bin chunks c = part, part + K, ... of member r's leave-one-out vector (r < n) or of the
    // whole new set (r == n).  K is a power of two so that, with the candidate's frequencies in
    // registers (fr[c], chunk c = bins c * 512 + tid), a thread's job bins are its own.
    const uint32_t nchunk = uint32_t((B + P_THREADS - 1) / P_THREADS);
    uint32_t K = 1, jobs = 0;
    bool one_job = false, has_job = false;
    auto set_geometry = [&](uint32_t members) {  // (again after every kept push: MAXM sets grow)
        K = 1;
        const uint32_t kmax = (members + 1 <= n_work) ? n_work / (members + 1) : 1u;
        while (K * 2 <= kmax && K * 2 <= nchunk && K * 2 <= 32u) K *= 2;
        jobs = (members + 1) * K;
        one_job = jobs <= n_work && G > 1;  // at most one job per workgroup, none for the mirror and the gathering block
        has_job = one_job ? (blockIdx.x < jobs) : true;
    };
    set_geometry(st.n);

    if constexpr (!MAXM && CACHED) {
    if (seeded) {
        // ---- the initial set, by the grid: S = sum of the members' rows in member order and the sum
        // of their entropies (records.rs:36-47), the leave-one-out pass as (n + 1) K jobs like after an
        // accept (records.rs:220-252), argmin, and sl = S - lowest.  The mirror block writes what the
        // set-up kernels would have left in global memory.  A decision too close to call -- or a sum
        // check that is not sure -- is not taken here: the launch ends having changed nothing and the
        // host runs the set-up kernels (SEL_NEED_SETUP).
        const uint32_t n = st.n;
        const double dn0 = double(n), rn0 = 1.0 / dn0, rdiv0 = 1.0 / (dn0 - 1.0);
        double Sreg[P_J];
#pragma unroll
        for (int j = 0; j < P_J; j++) Sreg[j] = 0.0;
        for (uint32_t r0 = 0; r0 < n; r0 += 4) {  // four members' counts requested at a time
            T cv[4][P_J];
#pragma unroll
            for (uint32_t q = 0; q < 4; q++) {
                const T *mrow = mat + s_pos[r0 + q < n ? r0 + q : r0] * B;
#pragma unroll
                for (int j = 0; j < P_J; j++) {
                    const uint64_t i = uint64_t(j) * P_THREADS + tid;
                    if (i < B) cv[q][j] = mrow[i];
                }
            }
#pragma unroll
            for (uint32_t q = 0; q < 4; q++) {
                if (r0 + q < n) {
                    const double mt = s_tot[r0 + q], mr = s_rt[r0 + q];
#pragma unroll
                    for (int j = 0; j < P_J; j++) {
                        const uint64_t i = uint64_t(j) * P_THREADS + tid;
                        if (i < B) Sreg[j] += count_freq_x(cv[q][j], mt, mr);
                    }
                }
            }
        }
        double sh = 0.0;
        for (uint32_t r = 0; r < n; r++) sh += s_mH[r];  // (the same LDS words in the same order in every thread)
#pragma unroll
        for (int j = 0; j < P_J; j++) {
            const uint64_t i = uint64_t(j) * P_THREADS + tid;
            if (i < B) {
                sl[i] = Sreg[j];  // (S for the jobs below; becomes S - lowest afterwards)
                if (lead) d.S[i] = Sreg[j];
            }
        }
        __syncthreads();
        unsigned long long *acc_all = part;  // the accumulators (zeroed by the host before the launch)
        const unsigned long long *acc = acc_all + uint64_t(blockIdx.x & 7u) * (maxn + 1) * 2;
        bool first_job = true;
        for (uint32_t job = blockIdx.x; job < jobs && has_job; job += G) {
            if ((lead || gath) && one_job) break;
            const uint32_t r = job / K, part_i = job % K;
            const T *mrow = mat + (r < n ? s_pos[r] : 0) * B;
            const double mtot = r < n ? s_tot[r] : 1.0, mrt = r < n ? s_rt[r] : 1.0;
            double h = 0.0, sv = 0.0;
#pragma unroll
            for (int j = 0; j < P_J; j++) {
                const uint64_t i = uint64_t(j) * P_THREADS + tid;
                if ((uint32_t(j) & (K - 1)) == part_i && i < B) {
                    double u;
                    if (r == n) {
                        u = Sreg[j] * rn0;
                    } else {
                        u = (Sreg[j] - count_freq_x(mrow[i], mtot, mrt)) * rdiv0;  // updated_mean_freqs, records.rs:276-286
                        if (u <= DVS_EPS) u = 0.0;
                    }
                    if (u > 0.0) h -= u * log2_tab(u, s_ltab);
                    sv += u;
                }
            }
            h = dvs_wave_sum_dpp(h);
            sv = dvs_wave_sum_dpp(sv);
            if (!first_job) __syncthreads();
            first_job = false;
            if (lane == 0) {
                scratch[64 + wave] = h;
                scratch[80 + wave] = sv;
            }
            __syncthreads();
            if (tid < 8) {
                double th = 0.0, ts = 0.0;
                for (uint32_t w = 0; w < P_THREADS / 64; w++) {
                    th += scratch[64 + w];
                    ts += scratch[80 + w];
                }
                unsigned long long *dst = acc_all + (uint64_t(tid) * (maxn + 1) + r) * 2;
                p_acc_add(dst, th, ts);
            }
        }
        if (!grid_barrier(sync, G, gen, s_flag)) {
            if (lead && tid == 0) ctl->status = SEL_ERROR;
            return;
        }
        bool acc_ok = true;
        for (uint32_t r = tid; r <= n; r += P_THREADS) {
            const double h = p_acc_value(p_acc_complete(acc + uint64_t(r) * 2, s_prev + uint64_t(r) * 2, K, acc_ok));
            const double sv = p_acc_value(p_acc_complete(acc + uint64_t(r) * 2 + 1, s_prev + uint64_t(r) * 2 + 1, K, acc_ok));
            if (r == n) {
                scratch[110] = h;
                scratch[111] = sv;
            } else {
                s_dl[r] = h - (sh - s_mH[r]) * rdiv0;  // JSD of the set without member r
                s_ds[r] = sv;
            }
        }
        if (__syncthreads_or(acc_ok ? 0 : 1)) {  // (a contribution that never arrived: every workgroup sees the same)
            if (lead && tid == 0) ctl->status = SEL_ERROR;
            return;
        }
        const double hm = scratch[110];
        const double tj = hm - sh / dn0;
        const bool evr = sum_risky(scratch[111], B) || !(hm == hm);
        for (uint32_t r = tid; r < n; r += P_THREADS) s_dl[r] = tj - s_dl[r];  // delta_jsd
        __syncthreads();
        if (wave == 0) p_argmin<(maxn + 63) / 64>(s_dl, s_ds, n, B, lane, scratch);
        __syncthreads();
        const double dmin0 = scratch[100], dsec0 = scratch[102], mean0 = scratch[103], sd0 = scratch[104];
        const uint32_t low0 = uint32_t(scratch[101]);
        const bool anyr0 = scratch[105] != 0.0;
        const double band0 = sel_band(tj + sh / dn0, B);
        if (anyr0 || evr || (n > 1 && dsec0 - dmin0 <= band0 && dsec0 < 1e6)) {
            if (lead && tid == 0) ctl->status = SEL_NEED_SETUP;  // (every workgroup reads the same words: all leave)
            return;
        }
        st.sumH = sh;
        st.total_jsd = tj;
        st.li = low0;
        st.band = band0;
        st.thr = tj + DVS_EPS;
        st.mean_d = mean0;
        st.std_d = sd0;
        st.cov_d = sd0 / mean0;
        {   // sl <- S - lowest (no clamp: what the set-up kernels' base vector holds)
            const T *lrow = mat + s_pos[low0] * B;
            const double ltot = s_tot[low0], lrt = s_rt[low0];
            T lv[P_J];
#pragma unroll
            for (int j = 0; j < P_J; j++) {
                const uint64_t i = uint64_t(j) * P_THREADS + tid;
                if (i < B) lv[j] = lrow[i];
            }
#pragma unroll
            for (int j = 0; j < P_J; j++) {
                const uint64_t i = uint64_t(j) * P_THREADS + tid;
                if (i < B) {
                    const double nv = Sreg[j] - count_freq_x(lv[j], ltot, lrt);
                    sl[i] = nv;
                    if (COARSE) slf[i] = coarse_sl(nv, rn0);
                    (void)dn0;
                }
            }
        }
        // (the mirror is the lead workgroup's alone: the accepts rewrite these words later, and plain
        // stores of two workgroups to one address sit in two XCDs' L2s until the kernel ends -- whichever
        // is written back last would win; a row per workgroup here corrupted exactly that way)
        if (lead) {  // what seed_kernel / rebuild_kernel / loo_kernel / finalize_kernel leave behind
            for (uint32_t r = 0; r < n; r++) {
                const T *mrow = mat + s_pos[r] * B;
                const double mt = s_tot[r], mr = s_rt[r];
                double *dst = d.M + uint64_t(r) * B;
                for (uint64_t i = tid; i < B; i += P_THREADS) dst[i] = count_freq_x(mrow[i], mt, mr);
            }
            for (uint32_t r = tid; r < n; r += P_THREADS) {
                const uint64_t mp = s_pos[r];
                d.ord[r] = r;
                d.mH[r] = s_mH[r];
                d.mLabel[r] = uint32_t(mp);  // (label-free: the label of a position is the position)
                d.mPos[r] = mp;
                if (uint32_t(mp) < d.nlabels) d.inset[uint32_t(mp)] = 1;
                d.dtmp[r] = s_dl[r];
                d.dsum[r] = s_ds[r];
                d.mDelta[r] = s_dl[r];
            }
            if (tid == 0) {
                ctl->sum_entropy = sh;
                ctl->s_is_resum = 1;
                ctl->total_jsd = tj;
                ctl->lowest = low0;
                ctl->mean_delta = mean0;
                ctl->std_delta = sd0;
                ctl->cov_delta = sd0 / mean0;
                ctl->band = band0;
                ctl->he_base = sh - s_mH[low0];
                ctl->thr = tj + DVS_EPS;
                ctl->ev_kind = 0;
                ctl->ev_risky = 0;
                ctl->ev_n = n;
            }
        }
        __syncthreads();
    }
    }

    [[maybe_unused]] const uint32_t own_slot = one_job ? blockIdx.x / K : 0u;  // OWN: the slot this workgroup's job serves (n: the whole set)
    [[maybe_unused]] const bool use_own = OWN && one_job;
    if constexpr (OWN) {
        bool slots_ok = true;
        for (uint32_t r = tid; r < st.n; r += P_THREADS) {
            const uint32_t a = s_slot[r];
            if (a < st.n) s_inv[a] = r;
            else slots_ok = false;
        }
        if (__syncthreads_or(slots_ok ? 0 : 1)) {  // (never: an nmost set occupies slots 0 .. n - 1)
            if (lead && tid == 0) ctl->why[7]++;
            return;
        }
        if (use_own && has_job && own_slot < st.n) {
            const T *mrow = mat + s_pos[s_inv[own_slot]] * B;
#pragma unroll
            for (int j = 0; j < P_J; j++) {
                const uint64_t i = uint64_t(j) * P_THREADS + tid;
                s_own[uint32_t(tid) * P_J + j] = i < B ? mrow[i] : T(0);
            }
        }
    }
    // SMALL: the members' count rows into LDS, thread-major (see above).  A slot the replica has no room
    // for (never: an nmost set occupies slots 0 .. n - 1) leaves the launch with the state untouched;
    // the host then carries on with the multi-launch kernels.
    if constexpr (SMALL) {
        const uint32_t rows_cap = sync->small_rows;
        bool fits = B == 4096 && st.n <= rows_cap;
        for (uint32_t r = 0; r < st.n; r++) fits = fits && s_slot[r] < rows_cap;
        if (!fits) return;
        for (uint32_t r = 0; r < st.n; r++) {
            const T *mrow = mat + s_pos[r] * B;
            T cv[P_J];
#pragma unroll
            for (int j = 0; j < P_J; j++) cv[j] = mrow[uint64_t(j) * P_THREADS + tid];
            uint4 q;
            q.x = uint32_t(cv[0]) | (uint32_t(cv[1]) << 16);
            q.y = uint32_t(cv[2]) | (uint32_t(cv[3]) << 16);
            q.z = uint32_t(cv[4]) | (uint32_t(cv[5]) << 16);
            q.w = uint32_t(cv[6]) | (uint32_t(cv[7]) << 16);
            *reinterpret_cast<uint4 *>(s_rows + uint64_t(s_slot[r]) * 4096 + uint32_t(tid) * 8) = q;
        }
    }
    if (sync->no_coarse & 2u) st.thr = 1e300;  // measurement aid: no row is ever an event (pure streaming)
#ifdef DVS_PERSIST_STAMPS  // per-phase in-kernel timing (DVS_PERSIST_DEBUG prints it): costs registers
    // (accumulated in LDS by thread 0 of block 0 and of the mirror block, written out when the launch
    // ends: a stamp costs a clock read and an LDS add, not a memory round trip)
    unsigned long long *s_dbg = reinterpret_cast<unsigned long long *>(smem + sync->lds_bytes - 128);  // (the last 128 bytes)
    if (tid < 16) s_dbg[tid] = 0ull;
    __syncthreads();
    unsigned long long t_prev = __builtin_amdgcn_s_memrealtime();
#ifdef DVS_PROBE
// ONE interval per build (-DDVS_PERSIST_STAMPS -DDVS_PROBE=k, scripts/probe.sh): thread 0 of block 0 (it owns a leave-one-out
// job) and of the last scanning block (it owns none) take two clock reads per pass -- everything else is switched off, so
// what is measured is not the stamps.  [0] the ticks, [1] the passes.
#define P_PROBE_BEGIN(id)                                                                          \
    do {                                                                                           \
        if ((id) == DVS_PROBE && (blockIdx.x == 0 || blockIdx.x == n_work - 1) && tid == 0)        \
            t_prev = __builtin_amdgcn_s_memrealtime();                                             \
    } while (0)
#define P_PROBE_END(id)                                                                            \
    do {                                                                                           \
        if ((id) == DVS_PROBE && (blockIdx.x == 0 || blockIdx.x == n_work - 1) && tid == 0) {      \
            s_dbg[0] += __builtin_amdgcn_s_memrealtime() - t_prev;                                 \
            s_dbg[1] += 1;                                                                         \
        }                                                                                          \
    } while (0)
#define P_STAMP(k) do { } while (0)
#define P_STAMP_B0(k) do { } while (0)
#define P_TRACE(which) do { } while (0)
#define P_TRACE_G(slot) do { } while (0)
#else
#define P_PROBE_BEGIN(id) do { } while (0)
#define P_PROBE_END(id) do { } while (0)
#define P_STAMP(k)                                                         \
    do {                                                                   \
        if ((lead || blockIdx.x == 0) && tid == 0) {                       \
            const unsigned long long t_now = __builtin_amdgcn_s_memrealtime(); \
            s_dbg[k] += t_now - t_prev;                                    \
            t_prev = t_now;                                                \
        }                                                                  \
    } while (0)
// (block 0 only: finer stamps inside a phase -- slots 9..14 are the mirror block's window statistics)
#define P_STAMP_B0(k)                                                      \
    do {                                                                   \
        if (blockIdx.x == 0 && !lead && tid == 0) {                        \
            const unsigned long long t_now = __builtin_amdgcn_s_memrealtime(); \
            s_dbg[k] += t_now - t_prev;                                    \
            t_prev = t_now;                                                \
        }                                                                  \
    } while (0)
// one window's timeline across the grid (DVS_PERSIST_DEBUG prints it): slot `which` of the traced window
#define P_TRACE(which)                                                                             \
    do {                                                                                           \
        if (tid == 0 && (epoch == 12 || epoch == 24 || epoch == 36 || epoch == 48))               \
            sync->trace[epoch / 12 - 1][which][blockIdx.x] = __builtin_amdgcn_s_memrealtime();     \
    } while (0)
#define P_TRACE_G(slot)                                                                            \
    do {                                                                                           \
        if (tid == 0 && (epoch == 12 || epoch == 24 || epoch == 36 || epoch == 48))               \
            sync->trace[epoch / 12 - 1][3][slot] = __builtin_amdgcn_s_memrealtime();               \
    } while (0)
#endif  // DVS_PROBE
#else
#define P_STAMP(k) do { } while (0)
#define P_STAMP_B0(k) do { } while (0)
#define P_TRACE(which) do { } while (0)
#define P_TRACE_G(slot) do { } while (0)
#define P_PROBE_BEGIN(id) do { } while (0)
#define P_PROBE_END(id) do { } while (0)
#endif
// A -DDVS_PERSIST_CHAOS build holds pseudo-randomly chosen workgroups back (one in sixteen up to ~10 us, one in four
// thousand for ~60 us) at the points where they are about to write or read a word another workgroup reads or
// writes: the answers must not change
// (scripts/chaos.sh runs the parity suite and the repeat script against such a build).
#ifdef DVS_PERSIST_CHAOS
#define P_CHAOS(k)                                                                                         \
    do {                                                                                                   \
        uint32_t h_ = (epoch * 0x9E3779B1u) ^ ((blockIdx.x + 1u) * 0x85EBCA77u) ^ (uint32_t(k) * 0xC2B2AE3Du); \
        h_ ^= h_ >> 15;                                                                                    \
        h_ *= 0x27D4EB2Fu;                                                                                 \
        h_ ^= h_ >> 13;                                                                                    \
        if ((h_ & 15u) == 0u)                                                                              \
            for (uint32_t z_ = (h_ >> 8) & 31u; z_ > 0; z_--) __builtin_amdgcn_s_sleep(12);               \
        if ((h_ & 0xFFF0u) == 0x1230u) /* now and then for the length of several windows (~60 us) */      \
            for (uint32_t z_ = 0; z_ < 160; z_++) __builtin_amdgcn_s_sleep(14);                            \
    } while (0)
#else
#define P_CHAOS(k) do { } while (0)
#endif
    for (;;) {
        // The loop body's view of the bin count and the thread index goes through an empty asm:
        // otherwise every loop-invariant mask and address derived from them (dozens) is hoisted in
        // front of the loop and kept alive across the scan, which the 256-VGPR budget pays for in
        // scratch spills on every window.
        uint64_t B = B0;
        int tid = tid0;
        asm volatile("" : "+s"(B));
        asm volatile("" : "+v"(tid));
        const uint32_t lane = uint32_t(tid) & 63u, wave = uint32_t(tid) >> 6;
        if (epoch >= P_EPOCH_MAX) {  // (window words have 24 bits for the window's number: the host launches again)
            bail = true;
            break;
        }
        PRel *myrel = &sync->rel[blockIdx.x & 7u];  // this group's copy of the release word and the hints
        PWin win;
        win.epoch = epoch;
        win.hintp = &myrel->hint;
        win.rel = sync->rel;
        win.wgev = s_win;
        win.nlist = reinterpret_cast<uint32_t *>(s_win + 1);
        win.mysoft = &sync->soft[epoch & 1u][blockIdx.x][0];
        if (tid == 0) {  // (the previous window's words were last read before the barrier that ended its walk)
            s_win[0] = p_word(epoch, SEL_NONE, false);
            *win.nlist = 0u;
        }
        __syncthreads();
        P_PROBE_END(6);
        P_PROBE_BEGIN(1);  // 1: the window's top -> this workgroup's arrival record stored
        P_CHAOS(1);  // (late into the window)
        P_STAMP_B0(9);  // (the top of the window: its barrier)
        P_TRACE(0);
#ifdef DVS_PERSIST_STAMPS
        const unsigned long long t_window = __builtin_amdgcn_s_memrealtime();
#endif
        const uint64_t end = umin64(st.cursor + uint64_t(st.window), st.npos);
        const uint64_t nrows = end - st.cursor;
        // ================= scan
        // (the mirror block scans nothing when there are other blocks: its global stores of the
        // previous event overlap the others' scan instead of delaying the rendezvous)
        if ((!lead && !gath) || G == 1) {
            if (wgmode && COARSE && B == 4096 && coarse_on) {
                if constexpr (COARSE)
                    p_scan_rows_wg_stream<T>(mat, d.totals, d.rowH, sl, slf, st, st.sumH - s_mH[st.li], win,
                                             blockIdx.x, nwg, nrows, scratch + 32, nread, nprecise, nmid);
            } else if (wgmode)
                p_scan_rows_wg<T, COARSE>(mat, d.totals, d.rowH, sl, slf, B, st, st.sumH - s_mH[st.li], win,
                                          blockIdx.x, nwg, nrows, scratch + 32, nread, nprecise, nmid, coarse_on);
            else
                p_scan_rows<T, COARSE>(mat, d.totals, d.rowH, sl, slf, B, st, st.sumH - s_mH[st.li], win,
                                       uint64_t(blockIdx.x) * wpb + wave, nwaves, nrows, lane, nread, nprecise, nmid, coarse_on,
                                       true);
        }
        // SPEC: the candidate this window will most likely end with -- the event word as it stands when
        // this workgroup leaves the scan -- is taken up BEFORE the rendezvous has completed: its counts,
        // total and entropy are requested ahead of the arrival (the row's memory round trip runs beside the
        // barrier, not behind it), and between arrival and release the workgroup already converts them to
        // frequencies and works out its own leave-one-out job for that candidate (the arithmetic an
        // accept would start with after the release; nothing leaves the workgroup).  An earlier event
        // posted later, an unsure one or a listed candidate ahead of it only means that the work is done
        // again below.
        double fr[P_J];  // the candidate's frequencies of this thread's bins (B <= P_J * 512)
        [[maybe_unused]] T craw[P_J];
        [[maybe_unused]] uint64_t craw_pos = SEL_NONE;  // the position whose counts craw holds ...
        [[maybe_unused]] uint32_t craw_tot = 1;         // ... its total and its entropy
        [[maybe_unused]] double craw_H = 0.0;
        [[maybe_unused]] uint64_t fr_pos = SEL_NONE;    // the position fr[] holds the frequencies of
        [[maybe_unused]] uint64_t spec_job_pos = SEL_NONE;  // ... and the candidate spec_th / spec_ts were worked out for
        [[maybe_unused]] double spec_th = 0.0, spec_ts = 0.0;
        [[maybe_unused]] auto fetch_raw = [&](uint64_t q) {
            if (craw_pos != q) {
                const T *gp = mat + q * B;
#pragma unroll
                for (int j = 0; j < P_J; j++) {
                    const uint64_t i = uint64_t(j) * P_THREADS + tid;
                    craw[j] = i < B ? gp[i] : T(0);
                }
                craw_tot = d.totals[q];
                craw_H = d.rowH[q];
                craw_pos = q;
            }
        };
        // one leave-one-out job of an accept, SPEC form: member r of the NEW order over this thread's
        // bins j with j % K == part -- the whole new set for r == n, the candidate itself for r == n - 1,
        // else the member whose replica arrays sit at index `at` (before the order has been shifted that
        // is r or r + 1, afterwards r), its counts from LDS (SMALL) or from its matrix row; -> this
        // workgroup's sums on threads 0..7 (updated_mean_freqs, records.rs:276-286)
        [[maybe_unused]] auto small_job = [&](uint32_t r, uint32_t part_i, uint32_t at, uint32_t n_, double &th, double &ts) {
            const double dn_ = double(n_), rn_ = 1.0 / dn_, rdiv_ = 1.0 / (dn_ - 1.0);
            const bool is_new = r == n_ - 1;
            uint32_t mcv[P_J];
#pragma unroll
            for (int j = 0; j < P_J; j++) mcv[j] = 0u;
            double mtot = 1.0, mrt = 1.0;
            if (r < n_ && !is_new) {
                if constexpr (SMALL) {
                    const uint4 q = *reinterpret_cast<const uint4 *>(s_rows + uint64_t(s_slot[at]) * 4096 + uint32_t(tid) * 8);
                    const uint32_t qw[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                    for (int j = 0; j < P_J; j++) mcv[j] = (qw[j >> 1] >> ((j & 1) * 16)) & 0xFFFFu;
                } else if (use_own) {  // (the member this workgroup's slot holds: its counts are here)
#pragma unroll
                    for (int j = 0; j < P_J; j++) mcv[j] = uint32_t(s_own[uint32_t(tid) * P_J + j]);
                } else {
                    const T *mrow = mat + s_pos[at] * B;
#pragma unroll
                    for (int j = 0; j < P_J; j++) {
                        const uint64_t i = uint64_t(j) * P_THREADS + tid;
                        if ((uint32_t(j) & (K - 1)) == part_i && i < B) mcv[j] = uint32_t(mrow[i]);
                    }
                }
                mtot = s_tot[at];
                mrt = s_rt[at];
            }
            double h = 0.0, sv = 0.0;
#pragma unroll
            for (int j = 0; j < P_J; j++) {
                if ((uint32_t(j) & (K - 1)) == part_i && uint64_t(j) * P_THREADS + tid < B) {
                    double v = sl[uint64_t(j) * P_THREADS + tid];
                    if (v <= DVS_EPS) v = 0.0;
                    const double sn = v + fr[j];
                    double u;
                    if (r == n_) {
                        u = sn * rn_;
                    } else {
                        u = (sn - (is_new ? fr[j] : exact_div_u32(double(mcv[j]), mtot, mrt))) * rdiv_;
                        if (u <= DVS_EPS) u = 0.0;
                    }
                    if (u > 0.0) h -= u * log2_tab(u, s_ltab);
                    sv += u;
                }
            }
            h = dvs_wave_sum_dpp(h);
            sv = dvs_wave_sum_dpp(sv);
            __syncthreads();  // (scratch[64..] of an earlier use has been read)
            if (lane == 0) {
                scratch[64 + wave] = h;
                scratch[80 + wave] = sv;
            }
            __syncthreads();
            th = ts = 0.0;
            if (tid < 8) {
                for (uint32_t w = 0; w < P_THREADS / 64; w++) {
                    th += scratch[64 + w];
                    ts += scratch[80 + w];
                }
            }
        };
        // the same job for rows beyond the register cache: candidate q (its counts at rq, total tq) as member
        // n_ - 1 of the new order; member r's replica arrays at index `at`.  Eight chunks at a time: sixteen loads
        // in flight per thread, then the bins WITHOUT a test inside (the job's kind is the workgroup's; as tests
        // and branches per bin, and with four chunks a time, a job of sixteen chunks at 4^7 bins took 8.6 us) and
        // their logarithms in three passes (all table reads together).  A term that is zero is multiplied out
        // (log2_tab of a tiny number is finite).
        [[maybe_unused]] auto big_job = [&](uint32_t r, uint32_t part_i, uint32_t at, uint32_t n_, const T *rq, double tq,
                                            double rtq, double &th, double &ts) {
            const double dn_ = double(n_), rn_ = 1.0 / dn_, rdiv_ = 1.0 / (dn_ - 1.0);
            const bool is_new = r == n_ - 1;
            const bool need_m = r < n_ && !is_new;
            const T *mrow = mat + (need_m ? s_pos[at] : 0) * B;
            const double mtot = need_m ? s_tot[at] : 1.0, mrt = need_m ? s_rt[at] : 1.0;
            double h = 0.0, sv = 0.0;
            constexpr int NQ = sizeof(T) <= 4 ? 8 : 4;
            for (uint32_t c0 = part_i; c0 < nchunk; c0 += NQ * K) {
                T cv[NQ], mv[NQ];
#pragma unroll
                for (int q = 0; q < NQ; q++) {
                    const uint64_t i = uint64_t(c0 + q * K) * P_THREADS + tid;
                    if (c0 + q * K < nchunk && i < B) {
                        cv[q] = rq[i];
                        if (need_m) mv[q] = mrow[i];
                    }
                }
                double u[NQ];
#pragma unroll
                for (int q = 0; q < NQ; q++) {
                    const uint64_t i = uint64_t(c0 + q * K) * P_THREADS + tid;
                    u[q] = 0.0;
                    if (c0 + q * K < nchunk && i < B) {
                        double v = sl[i];
                        if (v <= DVS_EPS) v = 0.0;
                        const double f = count_freq_x(cv[q], tq, rtq);
                        u[q] = v + f;  // S' of the bin
                    }
                }
                if (r == n_) {  // the whole set
#pragma unroll
                    for (int q = 0; q < NQ; q++) u[q] = fmax(u[q] * rn_, 0.0);
                } else if (is_new) {  // without the new member itself
#pragma unroll
                    for (int q = 0; q < NQ; q++) {
                        const uint64_t i = uint64_t(c0 + q * K) * P_THREADS + tid;
                        if (c0 + q * K < nchunk && i < B) {
                            double x = (u[q] - count_freq_x(cv[q], tq, rtq)) * rdiv_;
                            if (x <= DVS_EPS) x = 0.0;
                            u[q] = x;
                        }
                    }
                } else {  // without member r
#pragma unroll
                    for (int q = 0; q < NQ; q++) {
                        const uint64_t i = uint64_t(c0 + q * K) * P_THREADS + tid;
                        if (c0 + q * K < nchunk && i < B) {
                            double x = (u[q] - count_freq_x(mv[q], mtot, mrt)) * rdiv_;
                            if (x <= DVS_EPS) x = 0.0;
                            u[q] = x;
                        }
                    }
                }
                double mnt[NQ];
                int ex[NQ];
                double2 tb[NQ];
#pragma unroll
                for (int q = 0; q < NQ; q++) {
                    const double x = fmax(u[q], 1e-300);
                    mnt[q] = __builtin_amdgcn_frexp_mant(x);
                    ex[q] = __builtin_amdgcn_frexp_exp(x);
                    tb[q] = s_ltab[(uint32_t(__double2hiint(mnt[q])) >> 13) & 127u];
                }
#pragma unroll
                for (int q = 0; q < NQ; q++) {
                    const double r_ = fma(mnt[q], tb[q].y, -1.0);
                    double pl = 1.0 / 7.0;
                    pl = fma(pl, r_, -1.0 / 6.0);
                    pl = fma(pl, r_, 1.0 / 5.0);
                    pl = fma(pl, r_, -1.0 / 4.0);
                    pl = fma(pl, r_, 1.0 / 3.0);
                    pl = fma(pl, r_, -1.0 / 2.0);
                    pl = fma(pl, r_, 1.0);
                    const double lg = fma(pl * r_, 1.4426950408889634, double(ex[q]) + tb[q].x);
                    h -= u[q] * lg;
                    sv += u[q];
                }
            }
            h = dvs_wave_sum_dpp(h);
            sv = dvs_wave_sum_dpp(sv);
            __syncthreads();  // (scratch[64..] of an earlier use has been read)
            if (lane == 0) {
                scratch[64 + wave] = h;
                scratch[80 + wave] = sv;
            }
            __syncthreads();
            th = ts = 0.0;
            if (tid < 8) {
                for (uint32_t w = 0; w < P_THREADS / 64; w++) {
                    th += scratch[64 + w];
                    ts += scratch[80 + w];
                }
            }
        };
        P_STAMP_B0(10);  // (this workgroup's rows scanned)
        P_STAMP(0);
        // ---- the window's rendezvous.  Every workgroup stores ONE arrival record -- its first event and how
        // many candidates it listed, tagged with the window's number --; the mirror block, which has scanned
        // nothing, waits for all of them (thread t polls workgroup t's) and stores the release word: the
        // window's first event.  The others poll their group's copy of it together with the hint word beside
        // it (one 16-byte load; each word says itself which window it belongs to).
        P_CHAOS(2);  // (late with its arrival record)
        __syncthreads();  // (every wave's events and listings are in s_win)
        unsigned long long rel_w = 0ull;
        bool bar_ok = true;
        {
            const uint32_t nl = *win.nlist;
            const unsigned long long own = s_win[0] | ((unsigned long long)(nl < P_LIST ? nl : P_LIST) << 1);
            if (tid == 0 && early_rec != epoch) __hip_atomic_store(&sync->wrec[epoch & 1u][blockIdx.x], own, RLX_AGENT);
            P_PROBE_END(1);
            P_PROBE_BEGIN(2);  // 2: record stored -> release seen
            P_STAMP_B0(11);  // (the workgroup's barrier, the record stored)
            P_TRACE(1);
            if (gath) {
                P_TRACE_G(0);
                if (wave == 0) {  // (p_gather: one wave, no LDS, no barrier between the last record and the release)
                    bool ok;
#ifdef DVS_PERSIST_STAMPS
                    unsigned long long t_seen = 0, t_stored = 0;
                    const unsigned long long relw = p_gather(sync, G, epoch, own, lane, ok, &t_seen, &t_stored);
                    if (tid == 0 && (epoch == 12 || epoch == 24 || epoch == 36 || epoch == 48)) {
                        sync->trace[epoch / 12 - 1][3][1] = t_seen;
                        sync->trace[epoch / 12 - 1][3][2] = t_stored;
                    }
#else
                    const unsigned long long relw = p_gather(sync, G, epoch, own, lane, ok);
#endif
                    if (lane == 0) {
                        s_win[4] = relw;
                        s_flag[0] = ok ? 1 : 0;
                    }
                }
                __syncthreads();
                rel_w = s_win[4];
                bar_ok = s_flag[0] != 0;
                __syncthreads();  // (the words are rewritten in the next window)
            } else {
                // Thread 0 polls and hands every look to the workgroup; SPEC: as soon as the hint names a candidate
                // -- usually well before the last workgroup has arrived -- the workgroup takes it up.
                uint32_t spins = 0;
                for (;;) {
                    if (tid == 0) {
                        const uint4 v = p_load16_agent(myrel);
                        const unsigned long long r_ = ((unsigned long long)v.y << 32) | v.x;
                        const bool rel_ = p_word_is(r_, epoch);
                        int ok = 1;
                        if (!rel_ && (++spins & 255u) == 0 &&
                            (spins > P_SPIN_LIMIT || __hip_atomic_load(&sync->timeout, RLX_AGENT))) {
                            __hip_atomic_store(&sync->timeout, 1u, RLX_AGENT);
                            ok = 0;
                        }
                        s_win[4] = r_;
                        s_win[5] = ((unsigned long long)v.w << 32) | v.z;
                        s_win[6] = rel_ ? 1ull : 0ull;
                        s_flag[0] = ok;
                    }
                    __syncthreads();
                    rel_w = s_win[4];
                    const uint64_t seen = p_word_pos(s_win[5], epoch);  // the first event posted so far
                    const bool released = s_win[6] != 0ull;
                    bar_ok = s_flag[0] != 0;
                    __syncthreads();  // (thread 0 rewrites the words in the next round)
                    if (released || !bar_ok) break;
                    bool worked = false;
                    if constexpr (SPEC) {
                        if (seen != SEL_NONE && seen != fr_pos) {
                            fetch_raw(seen);
                            const double t_ = double(craw_tot), rt_ = 1.0 / t_;
#pragma unroll
                            for (int j = 0; j < P_J; j++) fr[j] = count_freq_x(craw[j], t_, rt_);
                            fr_pos = seen;
                            if (one_job && has_job && st.n < 128) {
                                if constexpr (OWN) {  // this workgroup's slot: the whole set, the member to be replaced, or a member that stays
                                    const uint32_t at = own_slot < st.n ? s_inv[own_slot] : st.n;
                                    const uint32_t r = own_slot == st.n ? st.n : at == st.li ? st.n - 1 : at < st.li ? at : at - 1;
                                    small_job(r, blockIdx.x % K, at, st.n, spec_th, spec_ts);
                                } else {
                                    const uint32_t r = blockIdx.x / K;
                                    small_job(r, blockIdx.x % K, r < st.li ? r : r + 1, st.n, spec_th, spec_ts);
                                }
                                spec_job_pos = seen;
                            }
                            worked = true;
                        }
                    }
                    if constexpr (SPEC_BIG) {
                        if (seen != SEL_NONE && seen != spec_job_pos && one_job && has_job && st.n < 128) {
                            const uint32_t r = blockIdx.x / K;
                            const double tq = double(d.totals[seen]);
                            big_job(r, blockIdx.x % K, r < st.li ? r : r + 1, st.n, mat + seen * B, tq, 1.0 / tq, spec_th, spec_ts);
                            spec_job_pos = seen;
                            worked = true;
                        }
                    }
                    // (one poller, one look in flight: a second poller half a round trip behind the first -- and a second
                    // gathering wave -- made every round trip slower, 1.53 -> 1.71 ms per step; waiting 0.1-0.4 us
                    // longer between looks changed nothing, 0.85 us cost 4 %: profiles/r04_polling_ab.txt)
                    if (!worked) __builtin_amdgcn_s_sleep(1);
                }
            }
        }
        if (!bar_ok) { exit_status = SEL_ERROR; break; }
        // The release names a SURE event and nobody listed a candidate in front of it: the window ends in the accept
        // of exactly that row (a sure event is accepted outright, below) -- so a job worked out for it while the
        // workgroup waited goes to the accumulators HERE, before anything else is looked at: the other workgroups'
        // wait for the totals ends when the last job's additions land.
        [[maybe_unused]] bool early_published = false;
        if constexpr (SPEC || SPEC_BIG) {
            const uint64_t h0 = p_word_pos(rel_w, epoch);
            if (h0 != SEL_NONE && p_word_sure(rel_w) && p_word_listed(rel_w) == 0u && one_job && has_job && !lead &&
                spec_job_pos == h0) {
                if (tid < 8) p_acc_add(part + (uint64_t(tid) * (maxn + 1) + blockIdx.x / K) * 2, spec_th, spec_ts);
                early_published = true;
            }
        }
        P_PROBE_END(2);
        P_PROBE_BEGIN(3);  // 3: release seen -> this workgroup's job handed to the accumulators (an accept)
        P_CHAOS(3);  // (late to act on the release: the lists are walked, the job published, later than the others')
        P_TRACE(2);
        P_STAMP(1);
#ifdef DVS_PERSIST_STAMPS
        if (lead && tid == 0) {  // scan + rendezvous time and window count by scan mode, rows per mode
            const unsigned long long t_w = __builtin_amdgcn_s_memrealtime();
            s_dbg[wgmode ? 9 : 11] += t_w - t_window;
            s_dbg[wgmode ? 10 : 12] += 1;
            s_dbg[wgmode ? 13 : 14] += nrows;
        }
#endif
        // the window's outcome came with the release (grid_wait): no further round trip, except for the
        // list of near-threshold candidates when there is one
        const uint64_t hard = p_word_pos(rel_w, epoch);
        const bool hard_is_sure = hard != SEL_NONE && p_word_sure(rel_w);
        const bool any_listed = p_word_listed(rel_w) != 0u;
        st.n_windows++;
        // this workgroup's leave-one-out job, should the window end in an accept: the member's
        // counts are requested now (its row does not depend on the event), in the same memory
        // round trip as the candidate's row below.  Member r of the NEW order is member r of the
        // old one before the lowest, r + 1 after it; the candidate becomes member n - 1.
        const uint32_t job_r = one_job ? blockIdx.x / K : 0u, job_part = one_job ? blockIdx.x % K : 0u;
        T mc[P_J];
        double job_tot = 1.0, job_rt = 1.0;
        if (CACHED && !SPEC && one_job && has_job && job_r + 1 < st.n) {
            const uint32_t old = job_r < st.li ? job_r : job_r + 1;
            const T *mrow = mat + s_pos[old] * B;
            job_tot = s_tot[old];
            job_rt = s_rt[old];
#pragma unroll
            for (int j = 0; j < P_J; j++) {
                const uint64_t i = uint64_t(j) * P_THREADS + tid;
                if ((uint32_t(j) & (K - 1)) == job_part && i < B) mc[j] = mrow[i];
            }
        }
        // ================= resolve (every workgroup, identical arithmetic)
        // Scores only need to land inside the decision band (4 B eps H), so the f64
        // evaluations multiply by reciprocals; everything that feeds S / sl keeps the
        // reference's exact add / subtract / clamp order.
        const double dn = double(st.n), rn = 1.0 / dn;
        uint64_t p = SEL_NONE;
        double tot = 1.0, rtot = 1.0, cand_H = 0.0;
        const T *rp = mat;
        double jsd = 0.0, sm = 1.0;
        auto evaluate = [&](uint64_t q) {  // exact score of candidate q, by the whole workgroup
            if constexpr (SPEC) {
                fetch_raw(q);
                tot = double(craw_tot);
                cand_H = craw_H;
                fr_pos = q;
            } else {
                tot = double(d.totals[q]);
                cand_H = d.rowH[q];
            }
            rtot = 1.0 / tot;
            rp = mat + q * B;
            Ent e;
            for (uint64_t b0 = 0; b0 < B; b0 += uint64_t(P_J) * P_THREADS) {
#pragma unroll
                for (int j = 0; j < P_J; j++) {
                    const uint64_t i = b0 + uint64_t(j) * P_THREADS + tid;
                    if (i < B) {
                        double f;
                        if constexpr (SPEC) f = count_freq_x(craw[j], tot, rtot);
                        else f = cand_freq_x(rp, i, tot, rtot);
                        if (CACHED) fr[j] = f;
                        e.add((sl[i] + f) * rn, s_ltab);
                    }
                }
            }
            double h = e.h, mn = e.mn;
            sm = e.sum;
            block_red3(h, mn, sm, scratch);
            const double mean_entropy = (st.sumH - s_mH[st.li] + cand_H) / dn;
            jsd = (mn < 0.0) ? NAN : h - mean_entropy;
        };
        bool accepted = false, to_arbiter = false;
        // listed candidates ahead of the first event, in stream order: the first one whose exact
        // score beats the threshold is the event; a rejected one changes nothing, so the window's
        // other scores stay valid and nothing is scanned again
        bool list_ok = true;
        if (any_listed) {
            // every scanning workgroup's list: thread t takes workgroup t's arrival record (the count) and its
            // entries.  (The gathering block saw the record before it released the window; an entry is waited
            // for until it carries the window's number.)
            unsigned long long le[P_LIST];
#pragma unroll
            for (uint32_t i = 0; i < P_LIST; i++) le[i] = SEL_NONE;
            if (uint32_t(tid) < n_work) {  // (only the scanning workgroups list; the mirror block's record may be a window ahead)
                auto tagged = [&](const unsigned long long *src) {
                    unsigned long long w = __hip_atomic_load(src, RLX_AGENT);
                    uint32_t spins = 0;
                    while (!p_word_is(w, epoch)) {
                        if (++spins > (1u << 20)) {
                            list_ok = false;
                            break;
                        }
                        __builtin_amdgcn_s_sleep(1);
                        w = __hip_atomic_load(src, RLX_AGENT);
                    }
                    return w;
                };
                const unsigned long long rec = tagged(&sync->wrec[epoch & 1u][tid]);
                const uint32_t c = list_ok ? p_word_listed(rec) : 0u;
#pragma unroll
                for (uint32_t i = 0; i < P_LIST; i++)
                    if (i < c) le[i] = p_word_pos(tagged(&sync->soft[epoch & 1u][tid][i]), epoch);
            }
            list_ok = __syncthreads_or(list_ok ? 0 : 1) == 0;
            uint64_t floor = st.cursor;
            while (list_ok) {
                if (tid == 0) s_win[2] = SEL_NONE;
                __syncthreads();
                unsigned long long mine = SEL_NONE;
#pragma unroll
                for (uint32_t i = 0; i < P_LIST; i++)
                    if (le[i] != SEL_NONE && le[i] >= floor && le[i] < hard && le[i] < mine) mine = le[i];
                if (mine != SEL_NONE) atomicMin(s_win + 2, mine);
                __syncthreads();
                const uint64_t cand = s_win[2];
                __syncthreads();  // (thread 0 resets the word in the next round)
                if (cand == SEL_NONE) break;
                evaluate(cand);
                p = cand;
                if (sum_risky(sm, B) || fabs(jsd - st.thr) <= st.band) {
                    to_arbiter = true;
                    break;
                }
                st.n_events++;
                if (jsd > st.thr) {
                    accepted = true;
                    break;
                }
                floor = cand + 1;
            }
        }
        if (!list_ok) { exit_status = SEL_ERROR; break; }
        if (!accepted && !to_arbiter) {
            if (hard == SEL_NONE) {
                st.cursor = end;
                if (end >= st.npos) { exit_status = SEL_DONE; break; }
                st.window = p_next_window(st, nwg, wg_thresh, wg_scale, wgmode);
                announce_early(epoch + 1);
                epoch++;
                continue;
            }
            p = hard;
            // A sure event (fast score above thr + band + FAST_BAND, so the exact score is above
            // thr + band) is accepted outright.  The sum-to-one guard needs no second look either:
            // the candidate's mean vector sums to the previous whole-set sum up to ~B u, and that
            // sum passed its guard at the last finalize.
            const bool sure = hard_is_sure;
            if (sure) {
                if constexpr (SPEC) {
                    fetch_raw(p);
                    tot = double(craw_tot);
                    cand_H = craw_H;
                } else {
                    tot = double(d.totals[p]);
                    cand_H = d.rowH[p];
                }
                rp = mat + p * B;
                rtot = 1.0 / tot;
                if constexpr (SPEC) {
                    if (fr_pos != p) {  // (else: worked out while the rendezvous was completing)
#pragma unroll
                        for (int j = 0; j < P_J; j++) fr[j] = count_freq_x(craw[j], tot, rtot);
                        fr_pos = p;
                    }
                } else if (CACHED) {
#pragma unroll
                    for (int j = 0; j < P_J; j++) {
                        const uint64_t i = uint64_t(j) * P_THREADS + tid;
                        if (i < B) fr[j] = cand_freq_x(rp, i, tot, rtot);
                    }
                }
                jsd = INFINITY;
                sm = 1.0;
            } else {
                evaluate(p);
            }
            to_arbiter = sum_risky(sm, B) || fabs(jsd - st.thr) <= st.band;
            if (!to_arbiter) {
                st.n_events++;
                if (!(jsd > st.thr)) {  // rejected (NaN included, records.rs:91)
                    st.cursor = p + 1;
                    if (st.cursor >= st.npos) { exit_status = SEL_DONE; break; }
                    announce_early(epoch + 1);
                    epoch++;
                    continue;
                }
            }
        }
        if (to_arbiter) {
            st.n_windows--;  // the multi-launch resolve will count this window
            if (lead && tid == 0) ctl->why[4]++;
            exit_status = SEL_ARBITER;
            arb_stage = ARB_RESOLVE;
            arb_pos = p;
            break;
        }
        P_STAMP(2);
        if constexpr (MAXM) {
            if (st.n < st.max_size) {
                // ================= tentative push (records.rs:427-451): clone + push, keep iff the stat rose
                const uint32_t n = st.n, n1 = n + 1;
                if (n1 + 1 > maxn) {  // the LDS replica holds no more members: the multi-launch kernels go on
                    if (lead && tid == 0) ctl->why[0]++;
                    bail = true;
                    st.n_windows--;
                    break;
                }
                const double rn1 = 1.0 / double(n1), rdiv1 = 1.0 / double(n);
                // what the decision needs, whichever way it was reached
                double sumH_t = st.sumH + cand_H, tj = 0.0, band1 = 0.0, mean1 = 0.0, sd1 = 0.0, cov1 = 0.0;
                uint32_t low1 = 0;
                bool unclear = false, grow = false;
                uint32_t why_i = 0;
                auto decide = [&](double hm, double svm, double dmin1, double dsec1, double mean_, double sd_, bool anyr) {
                    tj = hm - sumH_t * rn1;
                    mean1 = mean_;
                    sd1 = sd_;
                    const bool evr = sum_risky(svm, B) || !(hm == hm);
                    band1 = sel_band(tj + sumH_t * rn1, B);
                    cov1 = sd1 / mean1;
                    const double a = st.stat == DVS_STAT_STDEV ? sd1 : cov1;
                    const double b = st.stat == DVS_STAT_STDEV ? st.std_d : st.cov_d;
                    // every delta_jsd carries an error <= band, so std moves by <= ~band and cov = std / mean
                    // by ~ band (1 + |cov|) / |mean| (finalize_kernel); NaN compares false
                    const double mm = fmin(fabs(mean1), fabs(st.mean_d));
                    const double sband = st.stat == DVS_STAT_STDEV
                                             ? 4.0 * band1
                                             : 4.0 * band1 * (1.0 + fmax(fabs(a), fabs(b))) / fmax(mm, 1e-300);
                    const bool tie = dsec1 - dmin1 <= band1 && dsec1 < 1e6;
                    unclear = anyr || evr || tie || !(fabs(a - b) > sband);
                    why_i = (anyr || evr) ? 1 : tie ? 2 : 3;
                    grow = a > b;
                };
                // ---- BATCH: the rows behind the event, while the set does not change.  A tentative push
                // that is rolled back leaves the set as it was (records.rs:439-450: `summed` stays), and so
                // does a row that is no event -- so the rows p, p + 1, ... up to the first push that is KEPT
                // all face the same set, and their scores and leave-one-out passes are independent jobs.
                // When events come back to back (genome collections under `max`: nearly every row is one,
                // one in fifty is kept) the next E rows are worked out in one go: E (n + 3) jobs over the
                // grid, ONE rendezvous, eight decisions at a time by the eight waves of every workgroup,
                // then the rows are walked in stream order up to the first kept push or the first row too
                // close to call (which stays unconsumed: the next window meets it as its first event).
                const bool dense = p == st.cursor;  // the window's very first row was the event
                uint32_t E = mx_batch;
                if (uint64_t(E) > st.npos - p) E = uint32_t(st.npos - p);
                if (n1 + 1 > P_BATCH_MEMBERS || !CACHED) E = 1;
                bool known0 = true;  // row p is known to be an event (it came out of the resolve phase)
                bool left = false;   // the batches are over: on with the main loop
                if (E >= 2) {
                for (;;) {  // batch after batch while the rows keep being events
                    constexpr uint32_t JW = P_BATCH_MEMBERS + 2;  // result rows per batch entry
                    // (the results of a batch are read behind its rendezvous -- and a workgroup that is slow to read them must
                    // not meet the NEXT batch's: two result areas taking turns; the one two batches on is written behind the
                    // next batch's rendezvous, which every workgroup reaches only after it has read this batch's)
                    double *bres = reinterpret_cast<double *>(part + p_acc_bytes(maxn) / 8) + uint64_t(n_batches & 1u) * (p_batch_bytes() / 8);
                    n_batches++;
                    const uint32_t jpe = n + 3;  // leave-one-out of n + 1 members, the whole bigger set, the score
                    __syncthreads();  // every thread has read the member arrays of the resolve phase
                    if (tid < E) {
                        const double t = double(d.totals[p + tid]);
                        s_bt[tid] = t;
                        s_bH[tid] = d.rowH[p + tid];
                    }
                    __syncthreads();
                    // Jobs by member: a workgroup takes ONE member r (or the whole set, or the score) and every
                    // ng-th row of the batch -- the member's counts are read once, the next row's are requested
                    // while this row's are worked on.
                    P_STAMP(8);  // (everything since the accept was decided, or since the previous batch's walk)
                    const uint32_t ng = G >= jpe ? G / jpe : 1u;
                    const uint32_t r_step = G >= jpe ? jpe : G;
                    const uint32_t e_first = G >= jpe ? blockIdx.x / jpe : 0u;
                    for (uint32_t r = G >= jpe ? blockIdx.x % jpe : blockIdx.x; r < jpe && e_first < ng; r += r_step) {
                        T cv[P_J];
                        double fmv[P_J];  // the member's frequencies
                        if (r < n) {
                            const T *mrow = mat + s_pos[r] * B;
                            const double mtot = s_tot[r], mrt = s_rt[r];
#pragma unroll
                            for (int j = 0; j < P_J; j++) {
                                const uint64_t i = uint64_t(j) * P_THREADS + tid;
                                fmv[j] = i < B ? count_freq_x(mrow[i], mtot, mrt) : 0.0;
                            }
                        }
                        auto request = [&](uint32_t e) {
                            const T *crow = mat + (p + e) * B;
#pragma unroll
                            for (int j = 0; j < P_J; j++) {
                                const uint64_t i = uint64_t(j) * P_THREADS + tid;
                                if (i < B) cv[j] = crow[i];
                            }
                        };
                        if (e_first < E) request(e_first);
                        for (uint32_t e = e_first; e < E; e += ng) {
                            const double tot_e = s_bt[e], rt_e = 1.0 / tot_e;
                            double fv[P_J];
#pragma unroll
                            for (int j = 0; j < P_J; j++) fv[j] = count_freq_x(cv[j], tot_e, rt_e);
                            if (e + ng < E) request(e + ng);
                            if (tot_e == 0.0 || (known0 && e == 0 && r == n + 2)) continue;  // (no k-mers: never an event; nobody reads the result)
                            // The job's kind is the workgroup's, not the bin's: one loop per kind, no test inside
                            // (as eight tests and branches per bin the job was 560 vector + scalar instructions, twice
                            // its arithmetic).  A term that is zero is multiplied out instead of skipped: log2_tab of a
                            // tiny positive number is finite, so u = 0 adds -0 * finite -- bit for bit the sum of the
                            // single-row path.
                            double h = 0.0, sv = 0.0, mn = 0.0;
                            // sum of -u log2 u over the thread's bins, u >= 0: log2_tab (select_dev.h) in three passes --
                            // all mantissas and table reads first, then the polynomials -- so that the eight LDS
                            // reads travel together instead of one wait per bin (two waves per SIMD hide nothing)
                            auto entropy_of = [&](const double (&u)[P_J]) {
                                double mnt[P_J];
                                int ex[P_J];
                                double2 tb[P_J];
#pragma unroll
                                for (int j = 0; j < P_J; j++) {
                                    const double x = fmax(u[j], 1e-300);
                                    mnt[j] = __builtin_amdgcn_frexp_mant(x);
                                    ex[j] = __builtin_amdgcn_frexp_exp(x);
                                    tb[j] = s_ltab[(uint32_t(__double2hiint(mnt[j])) >> 13) & 127u];
                                }
                                double acc = 0.0;
#pragma unroll
                                for (int j = 0; j < P_J; j++) {
                                    const double r_ = fma(mnt[j], tb[j].y, -1.0);
                                    double q = 1.0 / 7.0;
                                    q = fma(q, r_, -1.0 / 6.0);
                                    q = fma(q, r_, 1.0 / 5.0);
                                    q = fma(q, r_, -1.0 / 4.0);
                                    q = fma(q, r_, 1.0 / 3.0);
                                    q = fma(q, r_, -1.0 / 2.0);
                                    q = fma(q, r_, 1.0);
                                    const double lg = fma(q * r_, 1.4426950408889634, double(ex[j]) + tb[j].x);
                                    acc -= u[j] * lg;  // (in bin order, as the single-row path adds them)
                                }
                                return acc;
                            };
                            auto job = [&](auto full_c) {
                                constexpr bool FULL = decltype(full_c)::value;  // 4^k = 4096: every thread owns P_J bins
                                double u[P_J];
                                if (r == n + 2) {  // increases_jsd of row p + e (records.rs:70-92), as `evaluate` above
#pragma unroll
                                    for (int j = 0; j < P_J; j++) {
                                        const uint64_t i = uint64_t(j) * P_THREADS + tid;
                                        u[j] = 0.0;
                                        if (FULL || i < B) {
                                            const double x = (sl[i] + fv[j]) * rn;
                                            u[j] = fmax(x, 0.0);
                                            sv += x;
                                            mn = fmin(mn, x);
                                        }
                                    }
                                } else if (r == n1) {  // the bigger set as a whole
#pragma unroll
                                    for (int j = 0; j < P_J; j++) {
                                        const uint64_t i = uint64_t(j) * P_THREADS + tid;
                                        u[j] = 0.0;
                                        if (FULL || i < B) {
                                            const double x = (Sl[i] + fv[j]) * rn1;
                                            u[j] = fmax(x, 0.0);
                                            sv += x;
                                        }
                                    }
                                } else {  // without member r (r == n: without the candidate itself)
                                    if (r == n) {
#pragma unroll
                                        for (int j = 0; j < P_J; j++) fmv[j] = fv[j];
                                    }
#pragma unroll
                                    for (int j = 0; j < P_J; j++) {
                                        const uint64_t i = uint64_t(j) * P_THREADS + tid;
                                        u[j] = 0.0;
                                        if (FULL || i < B) {
                                            double x = ((Sl[i] + fv[j]) - fmv[j]) * rdiv1;  // updated_mean_freqs, records.rs:276-286
                                            if (x <= DVS_EPS) x = 0.0;
                                            u[j] = x;
                                            sv += x;
                                        }
                                    }
                                }
                                h = entropy_of(u);
                            };
                            if (B == uint64_t(P_J) * P_THREADS) job(std::true_type{});
                            else job(std::false_type{});
                            h = dvs_wave_sum_dpp(h);
                            sv = dvs_wave_sum_dpp(sv);
                            mn = dvs_wave_min(mn);
                            // (no barrier per job: every wave leaves its share in a slot of its own and goes on)
                            if (lane == 0) {
                                s_bpart[(e * 3 + 0) * 8 + wave] = h;
                                s_bpart[(e * 3 + 1) * 8 + wave] = sv;
                                s_bpart[(e * 3 + 2) * 8 + wave] = mn;
                            }
                        }
                        // this member's results for the workgroup's rows, by wave 0 (whose thread 0 then arrives at the
                        // rendezvous: a wave's memory operations are acknowledged in order); waves summed in a fixed order
                        __syncthreads();
                        if (wave == 0) {
                            unsigned long long seen = 0;  // (exchanges, results consumed: performed before the rendezvous)
                            for (uint32_t x = lane; x < 3 * E; x += 64) {
                                const uint32_t e = x / 3, k = x % 3;
                                if (e >= e_first && (e - e_first) % ng == 0) {
                                    const double *pw = s_bpart + x * 8;
                                    double t = pw[0];
                                    for (uint32_t w = 1; w < P_THREADS / 64; w++) t = k == 2 ? fmin(t, pw[w]) : t + pw[w];
                                    seen |= __hip_atomic_exchange(reinterpret_cast<unsigned long long *>(bres) + (uint64_t(e) * 3 + k) * JW + r,
                                                                  (unsigned long long)__double_as_longlong(t), RLX_AGENT);
                                }
                            }
                            asm volatile("" ::"v"(seen) : "memory");
                        }
                        __syncthreads();  // (s_bpart is rewritten by the next member's jobs)
                    }
                    P_STAMP(3);  // the batch's jobs
                    if (!grid_barrier(sync, G, gen, s_flag)) { exit_status = SEL_ERROR; break; }
                    P_STAMP(4);  // its rendezvous
                    // ---- the decisions: wave w takes rows w, w + 8, ... of the batch; lane l the members l, l + 64, ...
                    auto rd = [&](uint32_t e, uint32_t r, uint32_t k) {  // result k of member r for row e
                        return __longlong_as_double((long long)__hip_atomic_load(
                            reinterpret_cast<unsigned long long *>(bres) + (uint64_t(e) * 3 + k) * JW + r, RLX_AGENT));
                    };
                    constexpr uint32_t QB = (P_BATCH_MEMBERS + 63) / 64;
                    for (uint32_t e = wave; e < E; e += P_THREADS / 64) {
                        double *out = s_bev + e * 12;
                        const double tot_e = s_bt[e], H_e = s_bH[e];
                        // every word this row's decision may need is requested before the first one is looked at
                        const double hs = rd(e, n + 2, 0), ss = rd(e, n + 2, 1), ms = rd(e, n + 2, 2);
                        const double hm = rd(e, n1, 0), svm = rd(e, n1, 1);
                        double hr[QB], sr[QB];
#pragma unroll
                        for (uint32_t q = 0; q < QB; q++) {
                            const uint32_t r = lane + 64 * q;
                            hr[q] = r < n1 ? rd(e, r, 0) : 0.0;
                            sr[q] = r < n1 ? rd(e, r, 1) : 1.0;
                        }
                        double kind = 0.0;  // 0 no event, 1 event, 2 too close to call (score)
                        if (tot_e != 0.0) {
                            kind = 1.0;
                            if (e > 0 || !known0) {
                                const double js = (ms < 0.0) ? NAN : hs - (st.sumH - s_mH[st.li] + H_e) / dn;
                                if (sum_risky(ss, B) || fabs(js - st.thr) <= st.band) kind = 2.0;
                                else if (!(js > st.thr)) kind = 0.0;
                                if (lane == 0) out[9] = js;
                            }
                        }
                        if (kind == 1.0) {
                            const double sumH_e = st.sumH + H_e;
                            const double tj_e = hm - sumH_e * rn1;
                            double v[QB], best = 1e6, acc = 0.0;
                            bool rk = false;
#pragma unroll
                            for (uint32_t q = 0; q < QB; q++) {
                                const uint32_t r = lane + 64 * q;
                                v[q] = 1e6;
                                if (r < n1) {
                                    const double mh = r < n ? s_mH[r] : H_e;
                                    v[q] = tj_e - (hr[q] - (sumH_e - mh) * rdiv1);  // delta_jsd of member r
                                    rk |= sum_risky(sr[q], B);
                                    acc += v[q];
                                    if (v[q] < best) best = v[q];
                                }
                            }
                            const double mnv = dvs_wave_min(best);
                            double fi = 4294967295.0;
#pragma unroll
                            for (uint32_t q = 0; q < QB; q++)
                                if (mnv < 1e6 && lane + 64 * q < n1 && v[q] == mnv) fi = fmin(fi, double(lane + 64 * q));
                            const double first = dvs_wave_min(fi);
                            const uint32_t lw = (first < 4294967295.0) ? uint32_t(first) : 0u;
                            const double mu = dvs_wave_sum(acc) / double(n1);
                            double sec = 1e6, tv = 0.0;
#pragma unroll
                            for (uint32_t q = 0; q < QB; q++) {
                                const uint32_t r = lane + 64 * q;
                                if (r < n1) {
                                    if (r != lw && v[q] < sec) sec = v[q];
                                    const double t = v[q] - mu;
                                    tv += t * t;
                                }
                            }
                            sec = dvs_wave_min(sec);
                            const double var = dvs_wave_sum(tv);
                            const unsigned long long anyr = __ballot(rk);
                            const double sd_e = sqrt(var / (double(n1) - 1.0));
                            sumH_t = sumH_e;
                            decide(hm, svm, mnv, sec, mu, sd_e, anyr != 0ull);  // (here, by eight waves at a time, not in the walk)
                            if (lane == 0) {
                                out[1] = hm;
                                out[2] = svm;
                                out[3] = mnv;
                                out[4] = sec;
                                out[5] = mu;
                                out[6] = sd_e;
                                out[7] = anyr ? 1.0 : 0.0;
                                out[8] = double(lw);
                                out[10] = unclear ? 1.0 : 0.0;
                                out[11] = grow ? 1.0 : 0.0;
                            }
                        }
                        if (lane == 0) out[0] = kind;
                    }
                    __syncthreads();
                    P_STAMP(6);  // the decisions
                    // ---- the walk, in stream order (every workgroup alike)
                    uint32_t e_commit = E, e_stop = E, n_ev = 0;
                    for (uint32_t e = 0; e < E; e++) {
                        const double *ev = s_bev + e * 12;
                        const bool counted = known0 && e == 0;  // (the resolve phase has counted row p)
                        if (ev[0] == 0.0) {
                            if (!counted && s_bt[e] != 0.0) st.n_events++;  // (an exact score below the threshold: rejected)
                            continue;
                        }
                        if (ev[0] == 2.0) { e_stop = e; break; }
                        if (ev[10] != 0.0) { e_stop = e; break; }
                        if (!counted) st.n_events++;
                        n_ev++;
                        if (ev[11] != 0.0) { e_commit = e; break; }
                    }
                    if (e_stop == 0 && known0) {  // the event itself is too close to call: it stays unconsumed
                        if (s_bev[0] == 1.0) {
                            sumH_t = st.sumH + s_bH[0];
                            decide(s_bev[1], s_bev[2], s_bev[3], s_bev[4], s_bev[5], s_bev[6], s_bev[7] != 0.0);  // (which check it was)
                        }
                        if (lead && tid == 0) ctl->why[why_i]++;
                        bail = true;
                        st.n_windows--;
                        st.n_events--;
                        __syncthreads();
                        break;
                    }
                    if (e_commit == E) {  // nothing kept: the stream moves on behind the rows walked
                        st.cursor = p + e_stop;
                        const bool dense_b = e_stop == E && n_ev * 2 >= E;
                        if (dense_b) mx_batch = E * 2 <= P_BATCH ? E * 2 : P_BATCH;
                        else if (n_ev * 2 < e_stop) mx_batch = E / 2;
                        if (lead && tid == 0) {
                            ctl->cursor = st.cursor;
                            ctl->event_pos = SEL_NONE;
                        }
                        __syncthreads();  // (s_bev, scratch)
                        if (st.cursor >= st.npos) { exit_status = SEL_DONE; left = true; break; }
                        if (dense_b) {  // the next batch right away: its first row is scored like the others
                            p = st.cursor;
                            known0 = false;
                            E = mx_batch;
                            if (uint64_t(E) > st.npos - p) E = uint32_t(st.npos - p);
                            if (E >= 2) continue;
                        }
                        st.window = p_next_window(st, nwg, wg_thresh, wg_scale, wgmode);
                        epoch++;
                        left = true;
                        break;
                    }
                    // ---- row p + e_commit is kept: it becomes the candidate of the commit below
                    {
                        const uint32_t e = e_commit;
                        const double *ev = s_bev + e * 12;
                        if (2 * (e + 1) < mx_batch) mx_batch = 2 * (e + 1);  // (kept pushes this close together: shorter batches)
                        sumH_t = st.sumH + s_bH[e];
                        decide(ev[1], ev[2], ev[3], ev[4], ev[5], ev[6], ev[7] != 0.0);  // (tj, band, mean, sd of the kept set)
                        low1 = uint32_t(ev[8]);
                        if (e > 0) jsd = ev[9];
                        p += e;
                        tot = s_bt[e];
                        rtot = 1.0 / tot;
                        cand_H = s_bH[e];
                        rp = mat + p * B;
#pragma unroll
                        for (int j = 0; j < P_J; j++) {
                            const uint64_t i = uint64_t(j) * P_THREADS + tid;
                            if (i < B) fr[j] = cand_freq_x(rp, i, tot, rtot);
                        }
                        __syncthreads();  // (s_bev has been read by everybody)
                        for (uint32_t r = tid; r < n1; r += P_THREADS) {
                            const double mh = r < n ? s_mH[r] : cand_H;
                            s_dl[r] = tj - (rd(e, r, 0) - (sumH_t - mh) * rdiv1);
                            s_ds[r] = rd(e, r, 1);
                        }
                        if (tid == 0) {
                            s_slot[n] = n;
                            s_mH[n] = cand_H;
                            s_pos[n] = p;
                            s_tot[n] = tot;
                            s_rt[n] = rtot;
                        }
                        __syncthreads();
                    }
                    break;  // (on to the commit)
                }
                if (exit_status == SEL_ERROR || bail || (left && exit_status == SEL_DONE)) break;
                if (left) continue;
                } else {
                uint32_t K1 = 1;
                {
                    const uint32_t kmax = (n1 + 1 <= n_work) ? n_work / (n1 + 1) : 1u;
                    while (K1 * 2 <= kmax && K1 * 2 <= nchunk && K1 * 2 <= 32u) K1 *= 2;
                }
                const uint32_t jobs1 = (n1 + 1) * K1;
                const bool one1 = jobs1 <= n_work && G > 1;
                __syncthreads();  // every thread has read the member arrays of the resolve phase
                if (tid == 0) {   // the candidate as member n (harmless beyond the set if rolled back)
                    s_slot[n] = n;
                    s_mH[n] = cand_H;
                    s_pos[n] = p;
                    s_tot[n] = tot;
                    s_rt[n] = rtot;
                }
                __syncthreads();
                unsigned long long *accw = part;
                const unsigned long long *accr = accw + uint64_t(blockIdx.x & 7u) * (maxn + 1) * 2;
                bool first1 = true;
                for (uint32_t job = blockIdx.x; job < jobs1; job += G) {
                    if ((lead || gath) && one1) break;
                    const uint32_t r = job / K1, part_i = job % K1;
                    const T *mrow = mat + (r < n ? s_pos[r] : 0) * B;
                    const double mtot = r < n ? s_tot[r] : 1.0, mrt = r < n ? s_rt[r] : 1.0;
                    double h = 0.0, sv = 0.0;
#pragma unroll
                    for (int j = 0; j < P_J; j++) {
                        const uint64_t i = uint64_t(j) * P_THREADS + tid;
                        if ((uint32_t(j) & (K1 - 1)) == part_i && i < B) {
                            const double f = fr[j];
                            const double stv = Sl[i] + f;  // S of the bigger set
                            double u;
                            if (r == n1) {
                                u = stv * rn1;
                            } else {
                                const double fm = r == n ? f : count_freq_x(mrow[i], mtot, mrt);
                                u = (stv - fm) * rdiv1;  // updated_mean_freqs, records.rs:276-286
                                if (u <= DVS_EPS) u = 0.0;
                            }
                            if (u > 0.0) h -= u * log2_tab(u, s_ltab);
                            sv += u;
                        }
                    }
                    h = dvs_wave_sum_dpp(h);
                    sv = dvs_wave_sum_dpp(sv);
                    if (!first1) __syncthreads();
                    first1 = false;
                    if (lane == 0) {
                        scratch[64 + wave] = h;
                        scratch[80 + wave] = sv;
                    }
                    __syncthreads();
                    if (tid < 8) {
                        double th = 0.0, ts = 0.0;
                        for (uint32_t w = 0; w < P_THREADS / 64; w++) {
                            th += scratch[64 + w];
                            ts += scratch[80 + w];
                        }
                        unsigned long long *dst = accw + (uint64_t(tid) * (maxn + 1) + r) * 2;
                        p_acc_add(dst, th, ts);
                    }
                }
                if (!grid_barrier(sync, G, gen, s_flag)) { exit_status = SEL_ERROR; break; }
                bool acc_ok = true;
                for (uint32_t r = tid; r <= n1; r += P_THREADS) {
                    const double h = p_acc_value(p_acc_complete(accr + uint64_t(r) * 2, s_prev + uint64_t(r) * 2, K1, acc_ok));
                    const double sv = p_acc_value(p_acc_complete(accr + uint64_t(r) * 2 + 1, s_prev + uint64_t(r) * 2 + 1, K1, acc_ok));
                    if (r == n1) {
                        scratch[110] = h;
                        scratch[111] = sv;
                    } else {
                        s_dl[r] = h - (sumH_t - s_mH[r]) * rdiv1;  // JSD of the bigger set without member r
                        s_ds[r] = sv;
                    }
                }
                if (__syncthreads_or(acc_ok ? 0 : 1)) { exit_status = SEL_ERROR; break; }
                {
                    const double hm0 = scratch[110];
                    const double tj0 = hm0 - sumH_t * rn1;
                    for (uint32_t r = tid; r < n1; r += P_THREADS) s_dl[r] = tj0 - s_dl[r];  // delta_jsd
                }
                __syncthreads();
                if (wave == 0) p_argmin<(maxn + 63) / 64>(s_dl, s_ds, n1, B, lane, scratch);
                __syncthreads();
                low1 = uint32_t(scratch[101]);
                decide(scratch[110], scratch[111], scratch[100], scratch[102], scratch[103], scratch[104], scratch[105] != 0.0);
                if (unclear) {
                    if (lead && tid == 0) ctl->why[why_i]++;
                    bail = true;  // too close to call (or NaN): the event stays unconsumed
                    st.n_windows--;
                    st.n_events--;
                    __syncthreads();
                    break;
                }
                // (a rolled-back push right behind the previous event: the next rows go in batches; a kept one
                // changes the set, and whatever was worked out for the rows behind it would be thrown away)
                mx_batch = dense && !grow ? 2u : 1u;
                }
                st.cursor = p + 1;
                if (grow) {  // ---- commit: the bigger set is the set
                    st.n_accepts++;
                    st.n = n1;
                    st.sumH = sumH_t;
                    st.total_jsd = tj;
                    st.li = low1;
                    st.band = band1;
                    st.thr = tj + DVS_EPS;
                    st.mean_d = mean1;
                    st.std_d = sd1;
                    st.cov_d = cov1;
                    set_geometry(n1);
                    const bool low_is_new = low1 == n;
                    const T *lrow = mat + s_pos[low1] * B;
                    const double ltot = s_tot[low1], lrt = s_rt[low1];
#pragma unroll
                    for (int j = 0; j < P_J; j++) {
                        const uint64_t i = uint64_t(j) * P_THREADS + tid;
                        if (i < B) {
                            const double f = fr[j];
                            const double stv = Sl[i] + f;
                            const double nv = stv - (low_is_new ? f : count_freq_x(lrow[i], ltot, lrt));
                            Sl[i] = stv;
                            sl[i] = nv;
                            if (COARSE) slf[i] = coarse_sl(nv, rn1);
                            if (lead) {
                                d.S[i] = stv;
                                d.M[uint64_t(n) * B + i] = f;
                            }
                        }
                    }
                    if (lead) {
                        for (uint32_t r = tid; r < n1; r += P_THREADS) {
                            d.dtmp[r] = s_dl[r];
                            d.dsum[r] = s_ds[r];
                            d.mDelta[r] = s_dl[r];
                        }
                        if (tid == 0) {
                            d.ord[n] = n;
                            d.mH[n] = cand_H;
                            d.mLabel[n] = uint32_t(p);
                            d.mPos[n] = p;
                            if (uint32_t(p) < d.nlabels) d.inset[uint32_t(p)] = 1;
                            d.evlog_pos[ctl->n_logged] = p;
                            d.evlog_kind[ctl->n_logged] = 2;
                            ctl->n_logged++;
                            ctl->size = n1;
                            ctl->sum_entropy = sumH_t;
                            ctl->total_jsd = tj;
                            ctl->lowest = low1;
                            ctl->mean_delta = mean1;
                            ctl->std_delta = sd1;
                            ctl->cov_delta = cov1;
                            ctl->band = band1;
                            ctl->he_base = sumH_t - s_mH[low1];
                            ctl->thr = st.thr;
                            ctl->cursor = st.cursor;
                            ctl->event_pos = SEL_NONE;
                            ctl->last_jsd = jsd;
                            ctl->ev_n = n1;
                            ctl->ev_risky = 0;
                        }
                    }
                } else if (lead && tid == 0) {  // ---- rollback: nothing changed but the cursor
                    ctl->cursor = st.cursor;
                    ctl->event_pos = SEL_NONE;
                }
                __syncthreads();  // sl / scratch are rewritten by the next window
                if (st.cursor >= st.npos) { exit_status = SEL_DONE; break; }
                st.window = p_next_window(st, nwg, wg_thresh, wg_scale, wgmode);
                epoch++;
                continue;
            }
        }
        // ================= replace_lowest (records.rs:94-147) + leave-one-out
        st.n_accepts++;
        // SPEC: a leave-one-out job worked out for this very candidate while the rendezvous was completing
        // goes out FIRST -- its memory-side additions travel while the member arrays are shifted below
        [[maybe_unused]] bool job_published = false;
        if constexpr (SPEC || SPEC_BIG) {
            if (one_job && has_job && !lead && spec_job_pos == p) {
                if (tid < 8 && !early_published) {  // lane g adds the job's words to group g's replica
                    unsigned long long *dst = part + (uint64_t(tid) * (maxn + 1) + blockIdx.x / K) * 2;
                    p_acc_add(dst, spec_th, spec_ts);
                }
                job_published = true;
            }
        }
        P_PROBE_END(3);
        P_PROBE_BEGIN(12);  // 12: job handed over -> member arrays shifted (in front of the job loop)
        P_PROBE_BEGIN(4);  // 4: job handed over -> the totals read and the argmin taken
        P_TRACE(7);
#ifdef DVS_PERSIST_STAMPS
        if constexpr (SPEC || SPEC_BIG) {  // (block 0: how often its job was ready when the release came)
            if (blockIdx.x == 0 && tid == 0) {
                s_dbg[15] += job_published ? 1u : 0u;
                s_dbg[14] += (fr_pos == p) ? 1u : 0u;
            }
        }
#endif
        const uint32_t n = st.n, li = st.li;
        const uint32_t slot_low = s_slot[li];
        const uint32_t old_lab = lead ? d.mLabel[slot_low] : 0;
        const double sh = (st.sumH - s_mH[li]) + cand_H;
        __syncthreads();  // every thread has read the old member arrays
        if (n <= 64) {
            // Vec::remove(li) + push by one wave: a wave's LDS reads are all performed before its
            // (data-dependent) writes, so the members move one place down without a barrier
            if (wave == 0) {
                const uint32_t i = li + lane;
                const bool mv = i + 1 < n;
                uint32_t a = 0;
                uint64_t c = 0;
                double b = 0.0, t = 1.0, rt = 1.0;
                if (mv) {
                    a = s_slot[i + 1];
                    b = s_mH[i + 1];
                    c = s_pos[i + 1];
                    t = s_tot[i + 1];
                    rt = s_rt[i + 1];
                }
                if (mv) {
                    s_slot[i] = a;
                    s_mH[i] = b;
                    s_pos[i] = c;
                    s_tot[i] = t;
                    s_rt[i] = rt;
                    if constexpr (OWN) s_inv[a] = i;
                }
                if (lane == 0) {
                    s_slot[n - 1] = slot_low;
                    s_mH[n - 1] = cand_H;
                    s_pos[n - 1] = p;
                    s_tot[n - 1] = tot;
                    s_rt[n - 1] = rtot;
                    if constexpr (OWN) s_inv[slot_low] = n - 1;
                }
            }
        } else {  // every thread moves its members one place down
            constexpr uint32_t Q = (maxn + P_THREADS - 1) / P_THREADS;
            uint32_t mv_slot[Q];
            double mv_H[Q], mv_t[Q], mv_rt[Q];
            uint64_t mv_pos[Q];
#pragma unroll
            for (uint32_t q = 0; q < Q; q++) {
                const uint32_t i = li + q * P_THREADS + tid;
                if (i + 1 < n) {
                    mv_slot[q] = s_slot[i + 1];
                    mv_H[q] = s_mH[i + 1];
                    mv_pos[q] = s_pos[i + 1];
                    mv_t[q] = s_tot[i + 1];
                    mv_rt[q] = s_rt[i + 1];
                }
            }
            __syncthreads();
#pragma unroll
            for (uint32_t q = 0; q < Q; q++) {
                const uint32_t i = li + q * P_THREADS + tid;
                if (i + 1 < n) {
                    s_slot[i] = mv_slot[q];
                    s_mH[i] = mv_H[q];
                    s_pos[i] = mv_pos[q];
                    s_tot[i] = mv_t[q];
                    s_rt[i] = mv_rt[q];
                    if constexpr (OWN) s_inv[mv_slot[q]] = i;
                }
            }
            if (tid == 0) {
                s_slot[n - 1] = slot_low;
                s_mH[n - 1] = cand_H;
                s_pos[n - 1] = p;
                s_tot[n - 1] = tot;
                s_rt[n - 1] = rtot;
                if constexpr (OWN) s_inv[slot_low] = n - 1;
            }
        }
        if constexpr (OWN) {
            // the slot has changed hands: its workgroups' own row becomes the candidate's counts (registers)
            if (use_own && has_job && own_slot == slot_low) {
#pragma unroll
                for (int j = 0; j < P_J; j++) s_own[uint32_t(tid) * P_J + j] = craw[j];
            }
        }
        __syncthreads();
        st.sumH = sh;
        if constexpr (SMALL) {  // the candidate's counts take the LDS slot of the member it replaces
            uint4 q;
            q.x = uint32_t(craw[0]) | (uint32_t(craw[1]) << 16);
            q.y = uint32_t(craw[2]) | (uint32_t(craw[3]) << 16);
            q.z = uint32_t(craw[4]) | (uint32_t(craw[5]) << 16);
            q.w = uint32_t(craw[6]) | (uint32_t(craw[7]) << 16);
            *reinterpret_cast<uint4 *>(s_rows + uint64_t(slot_low) * 4096 + uint32_t(tid) * 8) = q;
        }
        // ================= leave-one-out (get_lowest_record_index, records.rs:220-252, with
        // updated_mean_freqs :276-286) as (n + 1) * K jobs over the workgroups.  A job's two sums
        // (entropy terms, mean-vector total) are added to its member's accumulators as 2^-56
        // fixed-point integers: integer addition is associative, so the totals do not depend on the
        // order the K workgroups of a member arrive in, and partial sums of this magnitude
        // (>= 2^-3) convert without rounding.  (Every term is >= 0: the clamps of the reference
        // leave no negative bin here, so there is no NaN to carry.)
        P_PROBE_END(12);
        P_PROBE_BEGIN(13);  // 13: the job loop and wave 0's wait for the totals
        const double rdiv = 1.0 / (dn - 1.0);
        unsigned long long *acc_all = part;                                                    // the eight replicas
        const unsigned long long *acc = acc_all + uint64_t(blockIdx.x & 7u) * (maxn + 1) * 2;  // this group's
        bool first_job = true;
        for (uint32_t job = blockIdx.x; job < jobs && has_job; job += G) {
            if ((lead || gath) && one_job) break;
            const uint32_t r = job / K, part_i = job % K;
            const bool is_new = r == n - 1;
            const bool pre = CACHED && !SPEC && one_job;  // member counts already requested above
            const uint64_t mp = r < n ? s_pos[r] : 0;
            const T *mrow = mat + mp * B;
            const double mtot = pre ? job_tot : (r < n ? s_tot[r] : 1.0);
            const double mrt = pre ? job_rt : (r < n ? s_rt[r] : 1.0);
            double h = 0.0, sv = 0.0;
            auto bin = [&](uint64_t i, double f, double fm) {
                double v = sl[i];
                if (v <= DVS_EPS) v = 0.0;
                const double sn = v + f;
                double u;
                if (r == n) {
                    u = sn * rn;
                } else {
                    u = (sn - (is_new ? f : fm)) * rdiv;
                    if (u <= DVS_EPS) u = 0.0;
                }
                if (u > 0.0) h -= u * log2_tab(u, s_ltab);
                sv += u;
            };
            if constexpr (SPEC) {
                // (the workgroup's one job may already be there, worked out for this very candidate while
                // the rendezvous was completing)
                if (job_published) continue;
                double th, ts;
                // OWN: r is a SLOT (n: the whole set) -- the member it holds sits at s_inv[r] of the new order
                const uint32_t ro = (OWN && r < n) ? s_inv[r] : r;
                small_job(ro, part_i, ro, n, th, ts);
                if (tid < 8) {  // lane g adds the job's words to group g's replica
                    unsigned long long *dst = acc_all + (uint64_t(tid) * (maxn + 1) + r) * 2;
                    p_acc_add(dst, th, ts);
                }
                first_job = false;
                continue;
            } else if (CACHED) {
#pragma unroll
                for (int j = 0; j < P_J; j++) {
                    const uint64_t i = uint64_t(j) * P_THREADS + tid;
                    if ((uint32_t(j) & (K - 1)) == part_i && i < B) {
                        const double fm = (r >= n || is_new) ? 0.0
                                          : count_freq_x(pre ? mc[j] : mrow[i], mtot, mrt);
                        bin(i, fr[j], fm);
                    }
                }
            } else if constexpr (SPEC_BIG) {
                // (the workgroup's one job may already be there, worked out for this very candidate while
                // the rendezvous was completing)
                if (job_published) continue;
                double th, ts;
                big_job(r, part_i, r, n, rp, tot, rtot, th, ts);
                if (tid < 8) {  // lane g adds the job's words to group g's replica
                    unsigned long long *dst = acc_all + (uint64_t(tid) * (maxn + 1) + r) * 2;
                    p_acc_add(dst, th, ts);
                }
                first_job = false;
                continue;
            } else {
                double th, ts;  // (frequency rows beyond the register cache: chunk-merge matrices of 4^7 bins)
                big_job(r, part_i, r, n, rp, tot, rtot, th, ts);
                if (tid < 8) {
                    unsigned long long *dst = acc_all + (uint64_t(tid) * (maxn + 1) + r) * 2;
                    p_acc_add(dst, th, ts);
                }
                first_job = false;
                continue;
            }
            h = dvs_wave_sum_dpp(h);
            sv = dvs_wave_sum_dpp(sv);
            if (!first_job) __syncthreads();  // scratch[64..] still being read by thread 0
            first_job = false;
            if (lane == 0) {
                scratch[64 + wave] = h;
                scratch[80 + wave] = sv;
            }
            __syncthreads();
            if (tid < 8) {  // lane g adds the job's words to group g's replica
                double th = 0.0, ts = 0.0;
                for (uint32_t w = 0; w < P_THREADS / 64; w++) {
                    th += scratch[64 + w];
                    ts += scratch[80 + w];
                }
                unsigned long long *dst = acc_all + (uint64_t(tid) * (maxn + 1) + r) * 2;
                p_acc_add(dst, th, ts);
            }
        }
        st.cursor = p + 1;
        P_TRACE(4);
        P_STAMP(3);
        P_CHAOS(4);  // (late to read the totals)
        // ================= finalize (every workgroup): totals -> delta_jsd -> argmin (strict '<'
        // from 1e6, first index), all from the accumulators
        uint32_t lowest;
        double dmin, dsecond;
        bool any_risky, ev_risky;
        if (n < 128) {
            // No barrier: one wave polls this group's replica (lane l holds members l and l + 64, entry
            // n is the whole set) until every word carries K contributions, takes the decisions and
            // hands them to the other waves through LDS.  Bounded like every other spin of the kernel.
            if (wave == 0) {
                const uint32_t r0 = lane, r1 = lane + 64;
                const bool valid0 = r0 <= n, valid1 = r1 <= n;
                // (OWN: the accumulators are indexed by slot; entry n is the whole set either way)
                const uint32_t a0 = (OWN && r0 < n) ? s_slot[r0] : r0, a1 = (OWN && r1 < n) ? s_slot[r1] : r1;
                // (what the words held when the previous use was read; the differences are this use's)
                const unsigned long long ph0 = valid0 ? s_prev[a0 * 2] : 0ull, ps0 = valid0 ? s_prev[a0 * 2 + 1] : 0ull;
                const unsigned long long ph1 = valid1 ? s_prev[a1 * 2] : 0ull, ps1 = valid1 ? s_prev[a1 * 2 + 1] : 0ull;
                unsigned long long wh0 = 0, ws0 = 0, wh1 = 0, ws1 = 0;
                uint32_t spins = 0;
                int ok = 1;
                for (;;) {
                    if (valid0) {
                        wh0 = __hip_atomic_load(acc + uint64_t(a0) * 2, RLX_AGENT) - ph0;
                        ws0 = __hip_atomic_load(acc + uint64_t(a0) * 2 + 1, RLX_AGENT) - ps0;
                    }
                    if (valid1) {
                        wh1 = __hip_atomic_load(acc + uint64_t(a1) * 2, RLX_AGENT) - ph1;
                        ws1 = __hip_atomic_load(acc + uint64_t(a1) * 2 + 1, RLX_AGENT) - ps1;
                    }
                    if (__ballot((valid0 && (p_acc_count(wh0) != K || p_acc_count(ws0) != K)) ||
                                 (valid1 && (p_acc_count(wh1) != K || p_acc_count(ws1) != K))) == 0ull)
                        break;
                    if ((++spins & 255u) == 0 &&
                        (spins > P_SPIN_LIMIT || __hip_atomic_load(&sync->timeout, RLX_AGENT))) {
                        __hip_atomic_store(&sync->timeout, 1u, RLX_AGENT);
                        ok = 0;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
                if (valid0) {
                    s_prev[a0 * 2] = ph0 + wh0;
                    s_prev[a0 * 2 + 1] = ps0 + ws0;
                }
                if (valid1) {
                    s_prev[a1 * 2] = ph1 + wh1;
                    s_prev[a1 * 2 + 1] = ps1 + ws1;
                }
                P_PROBE_END(13);
                P_PROBE_BEGIN(14);  // 14: totals complete -> argmin handed to the other waves
                P_STAMP(4);
                const double h0 = p_acc_value(wh0), sv0 = p_acc_value(ws0);
                const double h1 = p_acc_value(wh1), sv1 = p_acc_value(ws1);
                const bool whole_hi = n >= 64;  // which of a lane's two entries the whole set is
                const double hm = __shfl(whole_hi ? h1 : h0, int(n & 63u), 64);
                const double svn = __shfl(whole_hi ? sv1 : sv0, int(n & 63u), 64);
                const double tj = hm - st.sumH / dn;
                const bool evr = sum_risky(svn, B) || !(hm == hm);
                const bool mem0 = r0 < n, mem1 = r1 < n;
                const double mH0 = mem0 ? s_mH[r0] : 0.0, mH1 = mem1 ? s_mH[r1] : 0.0;
                const double dl0 = mem0 ? tj - (h0 - (st.sumH - mH0) * rdiv) : 1e6;  // delta_jsd
                const double dl1 = mem1 ? tj - (h1 - (st.sumH - mH1) * rdiv) : 1e6;
                P_STAMP(6);
                const double mn = dvs_wave_min_dpp(fmin(dl0, dl1));
                const unsigned long long at0 = __ballot(mem0 && dl0 == mn && mn < 1e6);
                const unsigned long long at1 = __ballot(mem1 && dl1 == mn && mn < 1e6);
                const uint32_t lw = at0 ? uint32_t(__builtin_ctzll(at0)) : at1 ? 64u + uint32_t(__builtin_ctzll(at1)) : 0u;
                const double sec = dvs_wave_min_dpp(fmin((mem0 && r0 != lw) ? dl0 : 1e6, (mem1 && r1 != lw) ? dl1 : 1e6));
                const bool anyr = __ballot((mem0 && sum_risky(sv0, B)) || (mem1 && sum_risky(sv1, B))) != 0ull;
                if (lane == 0) {
                    scratch[100] = mn;
                    scratch[101] = double(lw);
                    scratch[102] = sec;
                    scratch[103] = tj;
                    scratch[104] = anyr ? 1.0 : 0.0;
                    scratch[105] = evr ? 1.0 : 0.0;
                    scratch[106] = double(ok);
                }
                P_STAMP(7);
                if (lead) {
                    const double mu = dvs_wave_sum((mem0 ? dl0 : 0.0) + (mem1 ? dl1 : 0.0)) / dn;
                    const double t0 = mem0 ? dl0 - mu : 0.0, t1 = mem1 ? dl1 - mu : 0.0;
                    const double sd = sqrt(dvs_wave_sum(t0 * t0 + t1 * t1) / (dn - 1.0));
                    if (mem0) {
                        d.dtmp[r0] = dl0;
                        d.dsum[r0] = sv0;
                        d.mDelta[r0] = dl0;
                    }
                    if (mem1) {
                        d.dtmp[r1] = dl1;
                        d.dsum[r1] = sv1;
                        d.mDelta[r1] = dl1;
                    }
                    if (lane == 0) {
                        ctl->total_jsd = tj;
                        ctl->mean_delta = mu;
                        ctl->std_delta = sd;
                        ctl->cov_delta = sd / mu;
                    }
                }
            }
            __syncthreads();
            dmin = scratch[100];
            lowest = uint32_t(scratch[101]);
            dsecond = scratch[102];
            st.total_jsd = scratch[103];
            any_risky = scratch[104] != 0.0;
            ev_risky = scratch[105] != 0.0;
            if (scratch[106] == 0.0) { exit_status = SEL_ERROR; break; }
        } else {
            // larger sets: thread t waits for the words of members t, t + 512, ... (their contribution counts)
            P_STAMP(4);
            bool acc_ok = true;
            for (uint32_t r = tid; r <= n; r += P_THREADS) {
                const uint64_t a = (OWN && r < n) ? s_slot[r] : r;  // (OWN: the accumulators are indexed by slot)
                const double h = p_acc_value(p_acc_complete(acc + a * 2, s_prev + a * 2, K, acc_ok));
                const double sv = p_acc_value(p_acc_complete(acc + a * 2 + 1, s_prev + a * 2 + 1, K, acc_ok));
                if (r == n) {
                    scratch[110] = h;
                    scratch[111] = sv;
                } else {
                    s_dl[r] = h - (st.sumH - s_mH[r]) * rdiv;  // JSD of the set without member r
                    s_ds[r] = sv;
                }
            }
            if (__syncthreads_or(acc_ok ? 0 : 1)) { exit_status = SEL_ERROR; break; }
            P_STAMP(6);
            const double hm = scratch[110];
            st.total_jsd = hm - st.sumH / dn;
            ev_risky = sum_risky(scratch[111], B) || !(hm == hm);
            for (uint32_t r = tid; r < n; r += P_THREADS) s_dl[r] = st.total_jsd - s_dl[r];  // delta_jsd
            __syncthreads();
            if (wave == 0) p_argmin<(maxn + 63) / 64>(s_dl, s_ds, n, B, lane, scratch);
            __syncthreads();
            dmin = scratch[100];
            lowest = uint32_t(scratch[101]);
            dsecond = scratch[102];
            any_risky = scratch[105] != 0.0;
            P_STAMP(7);
            if (lead) {
                for (uint32_t r = tid; r < n; r += P_THREADS) {
                    d.dtmp[r] = s_dl[r];
                    d.dsum[r] = s_ds[r];
                    d.mDelta[r] = s_dl[r];
                }
                if (tid == 0) {
                    ctl->total_jsd = st.total_jsd;
                    ctl->mean_delta = scratch[103];
                    ctl->std_delta = scratch[104];
                    ctl->cov_delta = scratch[104] / scratch[103];
                }
            }
            __syncthreads();  // scratch[100..] is rewritten by the next accept
        }
        // The mirror block has read the last word another workgroup wrote for this window: it announces itself
        // for the next one now, ahead of its stores.  Then S_new_i = clamp(S_i - low_i) + f_i and the new
        // member's row go to global memory (sl is still the old vector: the rebuild below rewrites it) -- what
        // resolve_kernel would have left behind, also for the kernels that take over an argmin too close to call.
        P_PROBE_END(4);
        P_PROBE_END(14);
        P_PROBE_BEGIN(7);  // 7: argmin handed over -> the rebuild's first load (the mirror's stores, the band checks)
        P_PROBE_BEGIN(5);  // 5: argmin known -> sl rebuilt
        P_CHAOS(5);  // (the mirror block: late with its early announcement; everybody: late into the rebuild)
        P_TRACE(5);
        announce_early(epoch + 1);
        if (lead)
            for (uint32_t i = tid; i < n; i += P_THREADS) d.ord[i] = s_slot[i];
        if (lead && tid == 0) {  // post-resolve mirror (what resolve_kernel leaves behind)
            ctl->sum_entropy = st.sumH;
            ctl->s_is_resum = 0;
            if (old_lab < d.nlabels) d.inset[old_lab] = 0;
            if (uint32_t(p) < d.nlabels) d.inset[uint32_t(p)] = 1;
            d.mH[slot_low] = cand_H;
            d.mLabel[slot_low] = uint32_t(p);
            d.mPos[slot_low] = p;
            d.evlog_pos[ctl->n_logged] = p;
            d.evlog_kind[ctl->n_logged] = 1;
            ctl->n_logged++;
            ctl->cursor = st.cursor;
            ctl->event_pos = SEL_NONE;
            ctl->last_jsd = jsd;
            ctl->ev_n = n;
        }
        if (lead) {
            for (uint64_t b0 = 0; b0 < B; b0 += uint64_t(P_J) * P_THREADS) {
                T cv[P_J];
                if (!CACHED) {
#pragma unroll
                    for (int j = 0; j < P_J; j++) {
                        const uint64_t i = b0 + uint64_t(j) * P_THREADS + tid;
                        if (i < B) cv[j] = rp[i];
                    }
                }
#pragma unroll
                for (int j = 0; j < P_J; j++) {
                    const uint64_t i = b0 + uint64_t(j) * P_THREADS + tid;
                    if (i < B) {
                        double v = sl[i];
                        if (v <= DVS_EPS) v = 0.0;
                        const double f = CACHED ? fr[j] : count_freq_x(cv[j], tot, rtot);
                        d.S[i] = v + f;
                        d.M[uint64_t(slot_low) * B + i] = f;
                    }
                }
            }
        }
        const double band = sel_band(st.total_jsd + st.sumH / dn, B);
        if (any_risky || ev_risky || (n > 1 && dsecond - dmin <= band && dsecond < 1e6)) {
            if (lead && tid == 0) ctl->ev_risky = ev_risky ? 1 : 0;
            if (lead && tid == 0) ctl->why[5]++;
            exit_status = SEL_ARBITER;  // argmin too close to call: loo + finalize kernels resume
            arb_stage = ARB_FINALIZE;
            arb_pos = p;
            pend_kind = 1;
            break;
        }
        st.li = lowest;
        st.band = band;
        st.thr = st.total_jsd + DVS_EPS;
        P_PROBE_END(7);
        P_PROBE_BEGIN(8);  // 8: the rebuild's loads and arithmetic
        {   // sl <- S_new - new lowest, in place (each thread owns its bins)
            const bool low_is_new = lowest == n - 1;
            const T *lrow = mat + s_pos[lowest] * B;
            const double ltot = s_tot[lowest], lrt = s_rt[lowest];
            // (4^7 bins of 16- or 32-bit counts: sixteen chunks a block -- two memory round trips instead of four)
            constexpr int RB = (!CACHED && sizeof(T) <= 4) ? 2 * P_J : P_J;
            for (uint64_t b0 = 0; b0 < B; b0 += uint64_t(RB) * P_THREADS) {
                // every count of the block is requested before the first is used: 2 RB loads in
                // flight per thread instead of one memory round trip per bin (k = 7: 32 bins a thread)
                T cv[RB], lv[RB];
                if constexpr (SMALL) {  // the new lowest member's counts: this thread's 16 bytes of its LDS row
                    const uint4 q = *reinterpret_cast<const uint4 *>(s_rows + uint64_t(s_slot[lowest]) * 4096 + uint32_t(tid) * 8);
                    lv[0] = T(q.x & 0xFFFFu); lv[1] = T(q.x >> 16);
                    lv[2] = T(q.y & 0xFFFFu); lv[3] = T(q.y >> 16);
                    lv[4] = T(q.z & 0xFFFFu); lv[5] = T(q.z >> 16);
                    lv[6] = T(q.w & 0xFFFFu); lv[7] = T(q.w >> 16);
                } else {
#pragma unroll
                for (int j = 0; j < RB; j++) {
                    const uint64_t i = b0 + uint64_t(j) * P_THREADS + tid;
                    if (i < B) {
                        if (!CACHED) cv[j] = rp[i];
                        if (!low_is_new) lv[j] = lrow[i];
                    }
                }
                }
#ifdef DVS_PERSIST_STAMPS
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                P_STAMP(8);
#endif
#pragma unroll
                for (int j = 0; j < RB; j++) {
                    const uint64_t i = b0 + uint64_t(j) * P_THREADS + tid;
                    if (i < B) {
                        double v = sl[i];
                        if (v <= DVS_EPS) v = 0.0;
                        const double f = CACHED ? fr[j % P_J] : count_freq_x(cv[j], tot, rtot);
                        const double sn = v + f;
                        const double nv = sn - (low_is_new ? f : count_freq_x(lv[j], ltot, lrt));
                        sl[i] = nv;
                        if (MAXM) Sl[i] = sn;
                        if (COARSE) slf[i] = coarse_sl(nv, rn);
                        (void)dn;
                    }
                }
            }
        }
        P_PROBE_END(8);
        P_PROBE_BEGIN(9);  // 9: the barrier behind the rebuild
        if (lead && tid == 0) {
            ctl->lowest = lowest;
            ctl->band = band;
            ctl->he_base = st.sumH - s_mH[lowest];
            ctl->thr = st.thr;
            ctl->ev_risky = 0;
        }
        __syncthreads();
        P_STAMP(5);
        P_PROBE_END(9);
        P_PROBE_END(5);
        P_PROBE_BEGIN(6);  // 6: sl rebuilt -> the next window's top
        P_TRACE(6);
        P_STAMP_B0(12);  // (behind the rebuild: the cache's decision)
        if (st.cursor >= st.npos) { exit_status = SEL_DONE; break; }
        st.window = p_next_window(st, nwg, wg_thresh, wg_scale, wgmode);
        epoch++;
    }

#ifdef DVS_PERSIST_STAMPS
#ifdef DVS_PROBE
    if ((blockIdx.x == n_work - 1 || blockIdx.x == 0) && tid == 0)
        for (int k_ = 0; k_ < 16; k_++) (blockIdx.x == 0 ? sync->dbg2 : sync->dbg)[k_] += s_dbg[k_];
#else
    if ((lead || blockIdx.x == 0) && tid == 0)
        for (int k_ = 0; k_ < 16; k_++) (lead ? sync->dbg : sync->dbg2)[k_] += s_dbg[k_];
#endif
#endif
    // ---- exit: the scan vector of the multi-launch kernels, base = (S - lowest) / size -- sl / n, bin by bin
    // the value their finalize kernel forms -- once per launch instead of once per accept.  (Not behind an
    // argmin left to the arbiter: the kernels that take over rewrite it before anything scans.)
    if (lead && exit_status != SEL_ERROR && !(exit_status == SEL_ARBITER && arb_stage == ARB_FINALIZE)) {
        const double dn_x = double(st.n);
        for (uint64_t i = tid; i < B; i += P_THREADS) d.base[i] = sl[i] / dn_x;
    }
    // ---- exit: counters, and the lead block's scalar mirror
    if (lane == 0 && nread) {
        atomicAdd(&ctl->rows_scored, (unsigned long long)nread);
        if (nprecise) atomicAdd(&ctl->rows_rechecked, (unsigned long long)nprecise);
        if (nmid) atomicAdd(&ctl->rows_coarse_passed, (unsigned long long)nmid);
    }
    if (lead && tid == 0) {
        ctl->cursor = st.cursor;
        ctl->window = st.window < ctl->window_max ? st.window : ctl->window_max;  // (the multi-launch scan's cap)
        ctl->n_windows += st.n_windows;
        ctl->n_events += st.n_events;
        ctl->n_accepts += st.n_accepts;
        ctl->ev_kind = pend_kind;
        if (exit_status == SEL_ARBITER) {
            ctl->arb_stage = arb_stage;
            ctl->arb_pos = arb_pos;
            if (arb_stage == ARB_RESOLVE) ctl->event_pos = arb_pos;  // resolve_kernel re-evaluates it
        }
        ctl->status = (bail || (head_phase && exit_status == SEL_DONE)) ? SEL_RUN
                      : (exit_status == SEL_RUN ? SEL_ERROR : exit_status);
    }
}

// ---- Self-test of the hand-over words on their own (dvs_selftest_handover): `rounds` synthetic
// windows through the very functions the engine uses -- arrival records, hints, listed candidates, p_gather, the
// release word, and a use of the never-cleared accumulators per round -- with every workgroup's contribution a hash
// of (round, workgroup) that every workgroup can recompute, and a pseudo-random pause in front of every step.  A
// workgroup counts a failure whenever what it reads differs from what the hashes say it must read.
__device__ __forceinline__ uint32_t ho_hash(uint32_t round, uint32_t b) {
    uint32_t h = round * 0x9E3779B1u ^ (b + 1u) * 0x85EBCA77u;
    h ^= h >> 15;
    h *= 0xC2B2AE3Du;
    h ^= h >> 13;
    return h;
}
// the event workgroup b finds in round `round` (SEL_NONE: none), whether it is a sure one, and a candidate it lists
__device__ __forceinline__ uint64_t ho_event(uint32_t round, uint32_t b, bool &sure, bool &lists) {
    const uint32_t h = ho_hash(round, b);
    sure = (h >> 2) & 1u;
    lists = ((h >> 3) & 15u) == 0u;
    return (h & 3u) == 0u ? uint64_t(round) * 4096u + ((h >> 8) % 4000u) : SEL_NONE;
}
constexpr uint32_t HO_MEMBERS = 24, HO_K = 8;  // accumulator words in use: 24 "members" x 8 contributions each

__global__ __launch_bounds__(P_THREADS) void handover_selftest_kernel(PSync *sync, unsigned long long *acc_all, uint32_t G,
                                                                      uint32_t rounds, unsigned long long *bad) {
    __shared__ unsigned long long s_win[8];
    __shared__ unsigned long long s_prev[(HO_MEMBERS + 1) * 2];
    __shared__ int s_flag[2];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6, b = blockIdx.x;
    const bool gath = b == G - 1;
    // (the workgroup in front of the gathering block plays the MIRROR block: it finds nothing, announces itself for the
    // next round as soon as it has read this round's last word, and then stays away for a while -- its stores)
    const bool mirror = b == G - 2;
    const uint32_t n_work = G - 2;
    uint32_t early_rec = 0xFFFFFFFFu;
    for (uint32_t i = tid; i < (HO_MEMBERS + 1) * 2; i += P_THREADS) s_prev[i] = 0ull;
    unsigned long long fails = 0;
    PRel *myrel = &sync->rel[b & 7u];
    const unsigned long long *acc = acc_all + uint64_t(b & 7u) * (HO_MEMBERS + 1) * 2;
    for (uint32_t epoch = 0; epoch < rounds; epoch++) {
        PWin win;
        win.epoch = epoch;
        win.hintp = &myrel->hint;
        win.rel = sync->rel;
        win.wgev = s_win;
        win.nlist = reinterpret_cast<uint32_t *>(s_win + 1);
        win.mysoft = &sync->soft[epoch & 1u][b][0];
        if (tid == 0) {
            s_win[0] = p_word(epoch, SEL_NONE, false);
            *win.nlist = 0u;
        }
        __syncthreads();
        const uint32_t h = ho_hash(epoch, b);
        for (uint32_t z = (h >> 24) & 15u; z > 0; z--) __builtin_amdgcn_s_sleep(3);  // (workgroups arrive out of step)
        bool sure, lists;
        const uint64_t mine = (gath || mirror) ? SEL_NONE : ho_event(epoch, b, sure, lists);
        if (mirror) lists = false;
        if (!gath && !mirror && tid == 0) {
            if (mine != SEL_NONE) p_post_event_thread(win, mine, sure);
            if (lists) p_list_candidate(win, uint64_t(epoch) * 4096u + 4001u + b);
        }
        __syncthreads();
        const uint32_t nl = *win.nlist;
        const unsigned long long own = s_win[0] | ((unsigned long long)(nl < P_LIST ? nl : P_LIST) << 1);
        if (tid == 0 && early_rec != epoch) __hip_atomic_store(&sync->wrec[epoch & 1u][b], own, RLX_AGENT);
        unsigned long long rel_w = 0ull;
        bool ok = true;
        if (gath) {
            if (wave == 0) {
                const unsigned long long relw = p_gather(sync, G, epoch, own, lane, ok);
                if (lane == 0) {
                    s_win[4] = relw;
                    s_flag[0] = ok ? 1 : 0;
                }
            }
        } else if (tid == 0) {
            uint32_t spins = 0;
            unsigned long long r_ = 0ull;
            int okk = 1;
            for (;;) {
                const uint4 v = p_load16_agent(myrel);
                r_ = ((unsigned long long)v.y << 32) | v.x;
                if (p_word_is(r_, epoch)) break;
                // a hint of this window never names a position in front of the window's first event
                if ((++spins & 255u) == 0 && (spins > P_SPIN_LIMIT || __hip_atomic_load(&sync->timeout, RLX_AGENT))) {
                    __hip_atomic_store(&sync->timeout, 1u, RLX_AGENT);
                    okk = 0;
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            s_win[4] = r_;
            s_flag[0] = okk;
        }
        __syncthreads();
        rel_w = s_win[4];
        ok = s_flag[0] != 0;
        if (!ok) break;  // (every workgroup sees the time-out flag sooner or later)
        // (now and then a workgroup is late to look -- as one that was inside its speculative job when the release
        // came: by then quick workgroups have stored their records of the NEXT window)
        if (((h >> 12) & 31u) == 0u)
            for (uint32_t z = 0; z < 24; z++) __builtin_amdgcn_s_sleep(8);
        // what the release must say: the minimum of the workgroups' events, and whether anybody listed
        if (wave == 0) {
            unsigned long long best = p_word(epoch, SEL_NONE, false);
            bool anyl = false;
            for (uint32_t q = lane; q < n_work; q += 64) {
                bool s2, l2;
                const uint64_t e2 = ho_event(epoch, q, s2, l2);
                if (e2 != SEL_NONE) {
                    const unsigned long long w2 = p_word(epoch, e2, s2);
                    best = w2 < best ? w2 : best;
                }
                anyl = anyl || l2;
            }
            best = p_wave_min_u64(best);
            const bool any = __ballot(anyl) != 0ull;
            const unsigned long long want = best | (any ? 2ull : 0ull);
            const unsigned long long hint = __hip_atomic_load(&myrel->hint, RLX_AGENT);
            bool wrong = rel_w != want;
            // a hint that carries this window's tag names an event some workgroup did find: never one in front of the first
            if (p_word_is(hint, epoch) && p_word_pos(hint, epoch) != SEL_NONE && p_word_pos(hint, epoch) < p_word_pos(want, epoch)) wrong = true;
            // the listed candidates of every workgroup that says it listed one
            if (any) {
                for (uint32_t q = lane; q < n_work; q += 64) {
                    bool s2, l2;
                    (void)ho_event(epoch, q, s2, l2);
                    unsigned long long rec = __hip_atomic_load(&sync->wrec[epoch & 1u][q], RLX_AGENT);
                    for (uint32_t spins = 0; !p_word_is(rec, epoch) && spins < (1u << 20); spins++)
                        rec = __hip_atomic_load(&sync->wrec[epoch & 1u][q], RLX_AGENT);
                    if (p_word_listed(rec) != (l2 ? 1u : 0u)) wrong = true;
                    if (l2) {
                        unsigned long long ent = __hip_atomic_load(&sync->soft[epoch & 1u][q][0], RLX_AGENT);
                        for (uint32_t spins = 0; !p_word_is(ent, epoch) && spins < (1u << 20); spins++)
                            ent = __hip_atomic_load(&sync->soft[epoch & 1u][q][0], RLX_AGENT);
                        if (p_word_pos(ent, epoch) != uint64_t(epoch) * 4096u + 4001u + q) wrong = true;
                    }
                }
            }
            if (__ballot(wrong) != 0ull && lane == 0) fails++;
        }
        // a use of the accumulators: workgroup j < HO_MEMBERS * HO_K adds its value to member j / HO_K (eight replicas,
        // lane g adds to group g's), everybody reads every member's total as the difference to the previous use
        for (uint32_t z = (h >> 20) & 7u; z > 0; z--) __builtin_amdgcn_s_sleep(2);
        if (b < HO_MEMBERS * HO_K && tid < 8) {
            const double v = double((h >> 4) & 0xFFFFu) * (1.0 / 65536.0), v2 = double(h & 0xFFu) * (1.0 / 4096.0);
            p_acc_add(acc_all + (uint64_t(tid) * (HO_MEMBERS + 1) + b / HO_K) * 2, v, v2);
        }
        bool acc_ok = true, acc_wrong = false;
        if (tid < HO_MEMBERS) {
            const unsigned long long d0 = p_acc_complete(acc + uint64_t(tid) * 2, s_prev + tid * 2, HO_K, acc_ok);
            const unsigned long long d1 = p_acc_complete(acc + uint64_t(tid) * 2 + 1, s_prev + tid * 2 + 1, HO_K, acc_ok);
            unsigned long long e0 = 0ull, e1 = 0ull;
            for (uint32_t q = 0; q < HO_K; q++) {
                const uint32_t h2 = ho_hash(epoch, tid * HO_K + q);
                e0 += p_acc_word(double((h2 >> 4) & 0xFFFFu) * (1.0 / 65536.0));
                e1 += p_acc_word(double(h2 & 0xFFu) * (1.0 / 4096.0));
            }
            acc_wrong = !acc_ok || d0 != e0 || d1 != e1;
        }
        if (__syncthreads_or(acc_wrong ? 1 : 0) && tid == 0) fails++;
        if (mirror && epoch + 1 < rounds) {  // announce_early: the next round's record now, then the "stores"
            if (tid == 0) __hip_atomic_store(&sync->wrec[(epoch + 1) & 1u][b], p_word(epoch + 1, SEL_NONE, false), RLX_AGENT);
            early_rec = epoch + 1;
            for (uint32_t z = (h >> 16) & 15u; z > 0; z--) __builtin_amdgcn_s_sleep(16);
        }
    }
    if (tid == 0 && fails) atomicAdd(bad, fails);
    if (tid == 0 && __hip_atomic_load(&sync->timeout, RLX_AGENT)) atomicAdd(bad, 1ull << 32);  // (a time-out: reported apart)
}

}  // namespace

// the instantiation that serves a selection (dvs_persist_setup decided maxm / cached / small)
template <typename T>
static const void *persist_fn(const dvs_select *s) {
    const bool cached = s->dev.B <= uint64_t(P_J) * P_THREADS;
    if (s->params.mode == DVS_MODE_MAX) return reinterpret_cast<const void *>(persist_nmost_kernel<T, true, true>);
    if constexpr (std::is_same_v<T, uint16_t>) {
        if (s->persist_small) return reinterpret_cast<const void *>(persist_nmost_kernel<T, true, false, true>);
    }
    return cached ? reinterpret_cast<const void *>(persist_nmost_kernel<T, true>)
                  : reinterpret_cast<const void *>(persist_nmost_kernel<T, false>);
}

// One persistent launch.  Returns DVS_OK with *ran = false when the selection does not
// qualify (the caller then uses the multi-launch engine).
// Fresh sync block + cleared accumulators of the next launch, enqueued on `on`.  The head phase has
// blocks (and a host image: the async upload may read it after this returns) of its own, so the
// full-grid launch's can be made ready while the head phase is still running.
static int persist_prepare(dvs_ctx *ctx, dvs_select *s, uint32_t head_stop, hipStream_t on) {
    std::vector<unsigned char> &image = head_stop ? s->h_psync_head : s->h_psync;
    image.assign(sizeof(PSync), 0);
    PSync &init = *reinterpret_cast<PSync *>(image.data());
    init.stop_at = head_stop;
    init.seeded = s->persist_seeded ? 1u : 0u;  // (the first launch of a selection whose set-up kernels were skipped)
    init.seed_list = static_cast<const unsigned long long *>(s->d_seed_list);
    for (int g = 0; g < 8; g++) init.rel[g].hint = ~0ull;  // (atomicMin targets; every other word starts as zero)
    // a row per workgroup while a window is at most this many rounds of the grid (default policy only)
    init.wg_thresh = s->params.window ? 0u : 4u;
    init.wg_scale = 1.5f;
    if (ctx->knobs.persist_wg_rounds >= 0) init.wg_thresh = uint32_t(ctx->knobs.persist_wg_rounds);
    init.no_coarse = (ctx->knobs.persist_no_coarse ? 1u : 0u) | (ctx->knobs.persist_no_events ? 2u : 0u) |
                     0u;
    init.small_rows = s->persist_small ? s->persist_small_rows : 0u;
    init.lds_bytes = uint32_t(s->persist_lds);
    init.wmax = uint32_t(std::min<uint64_t>(s->npos, 0xFFFFFFFFull));
    DVS_HIP(ctx, hipMemcpyAsync(head_stop ? s->psync_head : s->psync, &init, sizeof init, hipMemcpyHostToDevice, on));
    DVS_HIP(ctx, hipMemsetAsync(head_stop ? s->ppart_head : s->ppart, 0, p_acc_bytes(s->persist_maxn), on));
    return DVS_OK;
}

// head_stop != 0: the HEAD PHASE -- `grid` workgroups on stream `on` (the context's CU-masked head
// stream) walk positions below head_stop and leave the state mirrored with status RUN.
template <typename T>
static int persist_launch(dvs_ctx *ctx, dvs_select *s, const T *mat, uint32_t grid, uint32_t head_stop,
                          hipStream_t on) {
    const SelDev &d = s->dev;
    if (head_stop ? !s->head_prepared : !s->persist_prepared) {
        int prc = persist_prepare(ctx, s, head_stop, on);
        if (prc) return prc;
    }
    if (head_stop) s->head_prepared = false;
    else s->persist_prepared = false;
    s->persist_seeded = false;  // (only the first launch of a selection starts from the seeds)
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (s->time_scan) {
        if (s->ev_used + 2 > s->ev_pool.size()) {
            hipEvent_t a = dvs_event_get(ctx), b = dvs_event_get(ctx);
            s->ev_pool.push_back(a);
            s->ev_pool.push_back(b);
        }
        e0 = s->ev_pool[s->ev_used];
        e1 = s->ev_pool[s->ev_used + 1];
        s->ev_used += 2;
        (void)hipEventRecord(e0, on);
    }
    const void *fn = persist_fn<T>(s);
    SelDev d_arg = d;
    const T *mat_arg = mat;
    PSync *sync_arg = static_cast<PSync *>(head_stop ? s->psync_head : s->psync);
    unsigned long long *part_arg = static_cast<unsigned long long *>(head_stop ? s->ppart_head : s->ppart);
    uint32_t g_arg = grid;
    void *args[] = {&d_arg, &mat_arg, &sync_arg, &part_arg, &g_arg};
    // The grid barrier needs every workgroup resident: dvs_persist_setup checked that the device holds
    // the grid (one workgroup with this LDS per CU, grid <= CUs), every spin is bounded, and a launch
    // that still gives up at a barrier -- CUs held by another stream's kernels, a CU mask the runtime
    // does not report -- sends the selection to the multi-launch engine, and after three such
    // time-outs in a row every later selection of the context too (ctx->persist_timeouts).  hipLaunchCooperativeKernel adds a runtime
    // check of the same arithmetic and no reservation, and costs ~0.2 ms per selection on this stack
    // (a step of 2.13 -> 1.91 ms without it: the launch itself starts 40 us later, the memset in front
    // of it and the copy behind it take 35 us longer each, and the kernel runs 3 % slower): it is not used.
    hipError_t le = hipLaunchKernel(fn, dim3(grid), dim3(P_THREADS), args, s->persist_lds, on);
    if (s->time_scan) (void)hipEventRecord(e1, on);
    if (le != hipSuccess) {
        (void)hipGetLastError();
        return DVS_ERR_UNSUPPORTED;  // (the caller falls back; no message: nothing failed for the user)
    }
    DVS_HIP(ctx, hipGetLastError());
    return DVS_OK;
}

int dvs_persist_setup(dvs_ctx *ctx, dvs_select *s) {
    const uint64_t B = s->dev.B;
    s->persist = false;
    if (ctx->knobs.no_persist || ctx->persist_timeouts >= 3) return DVS_OK;
    // a process-wide CU mask hides CUs the device still reports: the grid below could never be resident
    if (ctx->knobs.cu_mask_set) return DVS_OK;
    const bool maxm = s->params.mode == DVS_MODE_MAX;
    if ((s->params.mode != DVS_MODE_NMOST && !maxm) || !s->h_order.empty() || !s->h_labels.empty()) return DVS_OK;
    if (s->npos >= P_POS_NONE) return DVS_OK;  // (window words hold 37-bit positions)
    s->persist_grid = uint32_t(std::min(ctx->n_cu, int(P_MAXG)));  // one 512-thread workgroup per CU: all resident
    const bool cached = B <= uint64_t(P_J) * P_THREADS;
    if (maxm && !cached) return DVS_OK;  // (the growth phase wants the candidate in registers and S in LDS)
    // SMALL sets (see the kernel): nmost, 16-bit rows of 4096 bins, every member's row in LDS
    s->persist_small_rows = s->params.n_seed;
    s->persist_small = !maxm && s->params.mode == DVS_MODE_NMOST && s->mat_kind == 2 && B == 4096 &&
                       s->params.n_seed >= 2 && s->params.n_seed <= P_SMALL_ROWS && !ctx->knobs.persist_no_small;
    auto lds_for = [&](uint32_t maxn_) {
        return ((B + 1) & ~1ull) * 8 + (cached && s->mat_kind != 1 ? ((B + 3) & ~3ull) * 4 : 0) +
               (maxm ? ((B + 1) & ~1ull) * 8 : 0) + 128 * 8 + size_t(maxn_) * 52 + 8 + P_WINWORDS * 8 + 64 +
               128 * 16 + 128 +  // (+ log2_tab's table, + the stamps of a -DDVS_PERSIST_STAMPS build)
               size_t(maxn_ + 1) * 16 +  // (the accumulators' previous totals)
               (maxm ? p_batch_lds() : 0);
    };
    size_t lds = lds_for(p_maxn(cached, maxm));
    if (s->persist_small) {
        const size_t lds_small = lds_for(P_SMALLN) + size_t(s->persist_small_rows) * B * 2 + 16;
        if (lds_small <= ctx->lds_per_block) lds = lds_small;
        else s->persist_small = false;
    }
    // OWN (see the kernel): nmost over count rows in the register cache beyond SMALL -- the slot map and this
    // workgroup's own member's counts
    if (cached && !maxm && s->mat_kind != 1 && !s->persist_small)
        lds += size_t(p_maxn(cached, maxm)) * 4 + size_t(P_J) * P_THREADS * (s->mat_kind == 2 ? 2 : 4);
    s->persist_maxn = s->persist_small ? P_SMALLN : p_maxn(cached, maxm);
    s->persist_maxjobs = s->persist_small ? P_SMALLN + 1 : p_maxjobs(cached, maxm);
    // (MODE_MAX: max_size may be the whole stream; the kernel hands over when its LDS replica is full)
    if (!maxm && s->cap > s->persist_maxn) return DVS_OK;
    if (lds > ctx->lds_per_block) return DVS_OK;
    s->persist_lds = lds;
    const void *fn = dvs_mat_dispatch(s->mat, [&](auto *mp) -> const void * {
        using T = std::remove_cv_t<std::remove_pointer_t<decltype(mp)>>;
        return persist_fn<T>(s);
    });
    int rc = dvs_raise_dyn_lds(ctx, fn, lds);
    if (rc) return rc;
    {   // one 512-thread workgroup with this much LDS must fit a CU, or the grid can never be resident
        // (asked of the runtime once per kernel and LDS size: the answer does not change)
        auto key = std::make_pair(fn, lds);
        auto hit = ctx->persist_fits.find(key);
        if (hit == ctx->persist_fits.end()) {
            int per_cu = 0;
            const bool fits = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, P_THREADS, lds) == hipSuccess && per_cu >= 1;
            if (!fits) (void)hipGetLastError();
            hit = ctx->persist_fits.emplace(key, fits).first;
        }
        if (!hit->second) return DVS_OK;
    }
    rc = dvs_dev_alloc(ctx, &s->psync, sizeof(PSync), "persistent sync block");
    if (!rc) rc = dvs_dev_alloc(ctx, &s->ppart, p_acc_bytes(s->persist_maxn) + (maxm ? 2 * p_batch_bytes() : 0),
                                "leave-one-out accumulators");  // (MODE_MAX: + the results of a batch's jobs)
    if (rc) return rc;
    s->persist = true;
    return DVS_OK;
}

size_t dvs_persist_dbg_offset(void) { return offsetof(PSync, dbg2); }  // dbg2[16] then dbg[16]
#ifdef DVS_PERSIST_STAMPS
size_t dvs_persist_trace_offset(void) { return offsetof(PSync, trace); }
#else
size_t dvs_persist_trace_offset(void) { return 0; }
#endif
#if defined(DVS_PERSIST_STAMPS) && defined(DVS_PROBE)
int dvs_persist_probe_id(void) { return DVS_PROBE; }
#else
int dvs_persist_probe_id(void) { return 0; }
#endif

int dvs_persist_launch(dvs_ctx *ctx, dvs_select *s) {
    return dvs_mat_dispatch(s->mat, [&](auto *mp) { return persist_launch(ctx, s, mp, s->persist_grid, 0u, ctx->stream); });
}

static int persist_head_blocks(dvs_ctx *ctx, dvs_select *s) {
    if (s->psync_head) return DVS_OK;
    int rc = dvs_dev_alloc(ctx, &s->psync_head, sizeof(PSync), "head phase sync block");
    if (!rc) rc = dvs_dev_alloc(ctx, &s->ppart_head, p_acc_bytes(s->persist_maxn), "head phase accumulators");
    return rc;
}

int dvs_persist_launch_head(dvs_ctx *ctx, dvs_select *s, uint32_t grid, uint32_t stop_at, hipStream_t on) {
    int rc = persist_head_blocks(ctx, s);
    if (rc) return rc;
    return dvs_mat_dispatch(s->mat, [&](auto *mp) { return persist_launch(ctx, s, mp, grid, stop_at, on); });
}

// the head phase's blocks, made ready on its stream before the selection waits for the head rows' totals
int dvs_persist_prepare_head(dvs_ctx *ctx, dvs_select *s, uint32_t stop_at, hipStream_t on) {
    int rc = persist_head_blocks(ctx, s);
    if (!rc) rc = persist_prepare(ctx, s, stop_at, on);
    if (!rc) s->head_prepared = true;
    return rc;
}

// the full-grid launch's blocks, made ready on the context's stream ahead of the launch itself
int dvs_persist_prepare_main(dvs_ctx *ctx, dvs_select *s) {
    int rc = persist_prepare(ctx, s, 0u, ctx->stream);
    if (!rc) s->persist_prepared = true;
    return rc;
}

// Self-test of the hand-over words (see handover_selftest_kernel): *failures = workgroup-rounds in which a word read
// was not the word the hashes demand (+ 2^32 if a spin ran into its bound).  One workgroup per CU, all resident.
extern "C" int dvs_selftest_handover(dvs_ctx *ctx, uint32_t rounds, uint64_t *failures) {
    if (!ctx || !failures || !rounds) return dvs_set_error(ctx, DVS_ERR_VALUE, "bad argument");
    if (rounds >= P_EPOCH_MAX) return dvs_set_error(ctx, DVS_ERR_VALUE, "at most %u rounds", P_EPOCH_MAX - 1);
    DVS_HIP(ctx, hipSetDevice(ctx->device));
    const uint32_t G = uint32_t(std::min(ctx->n_cu, int(P_MAXG)));
    if (G < HO_MEMBERS * HO_K + 2 || ctx->knobs.cu_mask_set)
        return dvs_set_error(ctx, DVS_ERR_UNSUPPORTED, "the self-test wants %u co-resident workgroups", HO_MEMBERS * HO_K + 2);
    void *d_sync = nullptr, *d_acc = nullptr, *d_bad = nullptr;
    int rc = dvs_dev_alloc(ctx, &d_sync, sizeof(PSync), "self-test sync block");
    if (!rc) rc = dvs_dev_alloc(ctx, &d_acc, p_acc_bytes(HO_MEMBERS), "self-test accumulators");
    if (!rc) rc = dvs_dev_alloc(ctx, &d_bad, 8, "self-test counter");
    if (!rc) {
        std::vector<unsigned char> image(sizeof(PSync), 0);
        PSync &init = *reinterpret_cast<PSync *>(image.data());
        for (int g = 0; g < 8; g++) init.rel[g].hint = ~0ull;
        hipError_t e = hipMemcpyAsync(d_sync, image.data(), sizeof(PSync), hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipMemsetAsync(d_acc, 0, p_acc_bytes(HO_MEMBERS), ctx->stream);
        if (e == hipSuccess) e = hipMemsetAsync(d_bad, 0, 8, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);  // (the image is a local)
        if (e == hipSuccess) {
            hipLaunchKernelGGL(handover_selftest_kernel, dim3(G), dim3(P_THREADS), 0, ctx->stream, static_cast<PSync *>(d_sync),
                               static_cast<unsigned long long *>(d_acc), G, rounds, static_cast<unsigned long long *>(d_bad));
            e = hipGetLastError();
        }
        unsigned long long bad = 0;
        if (e == hipSuccess) e = hipMemcpyAsync(&bad, d_bad, 8, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) rc = dvs_hip_fail(ctx, e, "hand-over self-test");
        *failures = bad;
    }
    dvs_dev_free(ctx, d_sync);
    dvs_dev_free(ctx, d_acc);
    dvs_dev_free(ctx, d_bad);
    return rc;
}
