// The head of the candidate stream on ONE workgroup (MODE_NMOST, identity order, unique ids,
// small sets of 16-bit or 32-bit count rows with at most 4096 bins).
//
// Early in the stream the greedy accepts a candidate every few rows (the accept probability at
// position i is ~ n / i), and what an accept costs on the grid engine (persist.hip) is a chain of
// grid-wide hand-overs: rendezvous, leave-one-out totals, the new lowest member's row -- ~19 us
// whatever the arithmetic.  One workgroup that keeps the WHOLE set in its LDS pays none of them:
//
//   LDS   S (summed_kfreqs, f64), sl / n as f32 (the COARSE tier's vector), the members' COUNT rows
//         (n + 1 rows of 4^k 16-bit counts: 88 KB at k = 6, n = 10) and the member scalars;
//   scan  thread t owns bins 8 t .. 8 t + 7 of every row (one 16-byte load per row of 16-bit counts),
//         the next H_D rows of the stream are in flight in registers, one s_barrier per row; the
//         same three tiers with the same bands as the grid engine (select_dev.h);
//   event replace_lowest + the new total_jsd in f64 exactly as the grid engine orders them, the
//         leave-one-out pass in the FAST tier (f64 values, f32-log) over the LDS rows, and an exact
//         f64 pass only over the members whose FAST delta_jsd is within its band of the minimum.
//         Everything is computed BEFORE anything is stored, so a decision that is too close to call
//         leaves the event unconsumed and the state untouched for the engines that can arbitrate.
//
// It runs on a second stream beside the histogram of the rest of the matrix (the first rows are
// built by a launch of their own, kmer_hist.hip), hands the state over through the same global
// mirror the other engines use (ev_kind = 1: the exact leave-one-out + finalize kernels of
// select.hip recompute every member's delta_jsd in f64 before anything else reads them), and the
// grid engine takes the stream from there.  Reference: select_nmost_divergent, src/records.rs:311-342;
// replace_lowest :94-147; get_lowest_record_index :220-252.
#include "select_dev.h"

#include <algorithm>
#include <cstdlib>
#include <type_traits>

namespace {

constexpr int H_THREADS = 512;
constexpr int H_WAVES = H_THREADS / 64;
constexpr int H_NB = 8;             // bins per thread
constexpr uint32_t H_MAXN = 16;     // members the register accumulators of the leave-one-out pass hold
constexpr int H_D = 8;              // rows requested ahead (one 16-byte load per thread each)
constexpr uint64_t H_MAXB = uint64_t(H_THREADS) * H_NB;

// eight consecutive counts of a thread, as loaded
template <typename T> struct Cnt8;
template <> struct Cnt8<uint16_t> {
    uint4 q;
    __device__ __forceinline__ void load(const uint16_t *p) { q = *reinterpret_cast<const uint4 *>(p); }
    __device__ __forceinline__ void pin() { asm volatile("" : "+v"(q.x), "+v"(q.y), "+v"(q.z), "+v"(q.w)); }
    __device__ __forceinline__ uint32_t at(int j) const {
        const uint32_t w = j < 2 ? q.x : j < 4 ? q.y : j < 6 ? q.z : q.w;
        return (j & 1) ? (w >> 16) : (w & 0xFFFFu);
    }
    __device__ __forceinline__ void coarse(const float *b, dvs_f2 r2, double &a0, double &a1) const {
        coarse8(q, b, r2, a0, a1);
    }
    // member-wise conditional copy (a struct copy under a condition becomes a pointer select, and the
    // register blocks it points into would have to live in memory)
    __device__ __forceinline__ void take(const Cnt8 &o, bool c) {
        q.x = c ? o.q.x : q.x; q.y = c ? o.q.y : q.y; q.z = c ? o.q.z : q.z; q.w = c ? o.q.w : q.w;
    }
};
template <> struct Cnt8<uint32_t> {
    uint4 a, b;
    __device__ __forceinline__ void load(const uint32_t *p) {
        a = *reinterpret_cast<const uint4 *>(p);
        b = *reinterpret_cast<const uint4 *>(p + 4);
    }
    __device__ __forceinline__ void pin() {
        asm volatile("" : "+v"(a.x), "+v"(a.y), "+v"(a.z), "+v"(a.w), "+v"(b.x), "+v"(b.y), "+v"(b.z), "+v"(b.w));
    }
    __device__ __forceinline__ uint32_t at(int j) const {
        return j == 0 ? a.x : j == 1 ? a.y : j == 2 ? a.z : j == 3 ? a.w : j == 4 ? b.x : j == 5 ? b.y : j == 6 ? b.z : b.w;
    }
    __device__ __forceinline__ void coarse(const float *bb, dvs_f2 r2, double &a0, double &a1) const {
        a0 += double(coarse4(a, *reinterpret_cast<const float4 *>(bb), r2));
        a1 += double(coarse4(b, *reinterpret_cast<const float4 *>(bb + 4), r2));
    }
    __device__ __forceinline__ void take(const Cnt8 &o, bool c) {
        a.x = c ? o.a.x : a.x; a.y = c ? o.a.y : a.y; a.z = c ? o.a.z : a.z; a.w = c ? o.a.w : a.w;
        b.x = c ? o.b.x : b.x; b.y = c ? o.b.y : b.y; b.z = c ? o.b.z : b.z; b.w = c ? o.b.w : b.w;
    }
};

// sums of NV values over the block, every thread gets them (fixed tree: same inputs, same bits).
// scratch: NV * H_WAVES doubles.
template <int NV>
__device__ __forceinline__ void h_block_sums(double (&v)[NV], double *scratch) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NV; k++) v[k] = dvs_wave_sum_dpp(v[k]);
    __syncthreads();  // (scratch may still be read from the previous call)
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < NV; k++) scratch[k * H_WAVES + wave] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NV; k++) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < H_WAVES; w++) t += scratch[k * H_WAVES + w];
        v[k] = t;
    }
}

// The same for many values at once, cheaper per value: four DPP steps leave the sum of each row of 16
// lanes, the 4 x H_WAVES row sums of a value go through LDS and one thread per value adds them up
// in a fixed order.  scratch: NV * 4 * H_WAVES + NV doubles.
template <int NV>
__device__ __forceinline__ void h_block_sums_many(double (&v)[NV], double *scratch) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int PER = 4 * H_WAVES;
#pragma unroll
    for (int k = 0; k < NV; k++) {
        double x = v[k];
        x += dvs_dpp_mov<0xB1>(x);   // quad_perm [1,0,3,2]
        x += dvs_dpp_mov<0x4E>(x);   // quad_perm [2,3,0,1]
        x += dvs_dpp_mov<0x141>(x);  // row_half_mirror
        x += dvs_dpp_mov<0x140>(x);  // row_mirror: every lane holds the sum of its row of 16
        v[k] = x;
    }
    __syncthreads();  // (scratch may still be read from the previous call)
    if ((lane & 15) == 0) {
#pragma unroll
        for (int k = 0; k < NV; k++) scratch[k * PER + wave * 4 + (lane >> 4)] = v[k];
    }
    __syncthreads();
    if (threadIdx.x < NV) {
        double t = 0.0;
#pragma unroll
        for (int q = 0; q < PER; q++) t += scratch[threadIdx.x * PER + q];
        scratch[NV * PER + threadIdx.x] = t;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NV; k++) v[k] = scratch[NV * PER + k];
}
constexpr int H_LB = 8;  // members per leave-one-out batch (2 values each)
constexpr int H_SCRATCH = 2 * H_LB * 4 * H_WAVES + 2 * H_LB;  // doubles

// LDS: [S B f64][slf B f32][Mc (H_MAXN + 1) rows of B counts][scratch 64 + 2 * H_MAXN * H_WAVES f64]
//      [s_mH, s_tot, s_rt, s_res (H_MAXN + 1) f64 each][s_pos (H_MAXN + 1) u64][s_slot, s_row (H_MAXN + 1) u32]
//      [log2 table 128 x double2]
template <typename T>
__host__ __device__ constexpr size_t head_lds_bytes(uint64_t B, uint32_t n) {
    return B * 8 + B * 4 + size_t(n + 1) * B * sizeof(T) + (64 + H_SCRATCH) * 8 + 4 * (H_MAXN + 1) * 8 +
           (H_MAXN + 1) * 8 + 2 * (H_MAXN + 1) * 4 + 32 + 128 * 16;
}

template <typename T>
__global__ __launch_bounds__(H_THREADS, 2) void head_nmost_kernel(SelDev d, const T *__restrict__ mat, uint64_t stop_arg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    SelCtl *ctl = d.ctl;
    const uint64_t B = d.B;
    const uint32_t n = ctl->size;
    // (the set-up kernels must have left a clean running state; anything else is the other engines')
    if (ctl->status != SEL_RUN || ctl->ev_kind != 0 || n < 2 || n > H_MAXN || B > H_MAXB || (B & 7)) return;
    double *S = reinterpret_cast<double *>(smem);
    float *slf = reinterpret_cast<float *>(S + B);
    T *Mc = reinterpret_cast<T *>(slf + B);
    double *scratch = reinterpret_cast<double *>(reinterpret_cast<unsigned char *>(Mc) + size_t(n + 1) * B * sizeof(T));
    double *red = scratch + H_SCRATCH;  // 64 doubles: the per-row partial sums (two parities)
    double *s_mH = red + 64;
    double *s_tot = s_mH + (H_MAXN + 1);
    double *s_rt = s_tot + (H_MAXN + 1);
    double *s_res = s_rt + (H_MAXN + 1);
    uint64_t *s_pos = reinterpret_cast<uint64_t *>(s_res + (H_MAXN + 1));
    uint32_t *s_slot = reinterpret_cast<uint32_t *>(s_pos + (H_MAXN + 1));
    uint32_t *s_row = s_slot + (H_MAXN + 1);
    double2 *s_ltab = reinterpret_cast<double2 *>((reinterpret_cast<uintptr_t>(s_row + (H_MAXN + 1)) + 15) & ~uintptr_t(15));

    const int tid = threadIdx.x;
    const uint32_t lane = tid & 63, wave = tid >> 6;
#ifdef DVS_HEAD_STAMPS  // per-phase timing (DVS_PERSIST_DEBUG prints it)
    unsigned long long t_prev = __builtin_amdgcn_s_memrealtime();
#define H_STAMP(k)                                                             \
    do {                                                                       \
        if (tid == 0) {                                                        \
            const unsigned long long t_now = __builtin_amdgcn_s_memrealtime(); \
            ctl->head_dbg[k] += t_now - t_prev;                                \
            t_prev = t_now;                                                    \
        }                                                                      \
    } while (0)
#else
#define H_STAMP(k) do { } while (0)
#endif
    const uint64_t i0 = uint64_t(tid) * H_NB;
    const bool active = i0 < B;
    if (tid < 128) log2_tab_fill(s_ltab, tid);

    // ---- replica of the state
    uint32_t li = ctl->lowest;
    double sumH = ctl->sum_entropy, total_jsd = ctl->total_jsd, thr = ctl->thr, band = ctl->band;
    uint64_t p = ctl->cursor;
    const uint64_t npos = ctl->npos;
    const uint64_t stop = stop_arg < npos ? stop_arg : npos;
    if (tid < int(n)) {
        const uint32_t slot = d.ord[tid];
        const uint64_t mp = d.mPos[slot];
        const double t = double(d.totals[mp]);
        s_slot[tid] = slot;
        s_row[tid] = tid;
        s_mH[tid] = d.mH[slot];
        s_pos[tid] = mp;
        s_tot[tid] = t;
        s_rt[tid] = 1.0 / t;
    }
    if (tid == 0) s_row[n] = n;  // the spare LDS row
    __syncthreads();
    if (active) {
        for (uint32_t r = 0; r < n; r++) {
            Cnt8<T> c;
            c.load(mat + s_pos[r] * B + i0);
            T *dst = Mc + uint64_t(r) * B + i0;
#pragma unroll
            for (int j = 0; j < H_NB; j++) dst[j] = T(c.at(j));
        }
#pragma unroll
        for (int j = 0; j < H_NB; j++) S[i0 + j] = d.S[i0 + j];
    }
    __syncthreads();
    const double dn = double(n), rn = 1.0 / dn, rdiv = 1.0 / (dn - 1.0);
    // A member's eight counts of this thread's bins: ONE 16-byte LDS read (rows of 16-bit counts; the
    // bytes a thread owns are contiguous, so the read is conflict-free), not eight 2-byte ones.
    auto mcounts = [&](uint32_t r) -> Cnt8<T> {
        Cnt8<T> c;
        c.load(Mc + uint64_t(s_row[r]) * B + i0);
        return c;
    };
    // f64 frequencies of member r's bins (record.rs:139, correctly rounded: exact_div_u32)
    auto mfreqs = [&](uint32_t r, double (&f)[H_NB]) {
        const Cnt8<T> c = mcounts(r);
        const double t = s_tot[r], rt_ = s_rt[r];
#pragma unroll
        for (int j = 0; j < H_NB; j++) f[j] = exact_div_u32(double(c.at(j)), t, rt_);
    };
    auto load_S = [&](double (&v)[H_NB]) {
#pragma unroll
        for (int j = 0; j < H_NB; j += 2) {
            const double2 t = *reinterpret_cast<const double2 *>(S + i0 + j);
            v[j] = t.x;
            v[j + 1] = t.y;
        }
    };
    auto refresh_slf = [&]() {  // slf = (S - lowest) / n as the COARSE tier wants it
        if (active) {
            double sv[H_NB], fl[H_NB];
            load_S(sv);
            mfreqs(li, fl);
            float o[H_NB];
#pragma unroll
            for (int j = 0; j < H_NB; j++) o[j] = coarse_sl(sv[j] - fl[j], rn);
            *reinterpret_cast<float4 *>(slf + i0) = make_float4(o[0], o[1], o[2], o[3]);
            *reinterpret_cast<float4 *>(slf + i0 + 4) = make_float4(o[4], o[5], o[6], o[7]);
        }
    };
    refresh_slf();
    __syncthreads();
    H_STAMP(0);

    uint32_t nread = 0, nprecise = 0, nmid = 0, n_events = 0, n_accepts = 0;
    bool bail = false;

    struct Row {
        Cnt8<T> c;
        uint32_t tot;
        double hrow;
    };
    // Requests are UNCONDITIONAL (a position past the end is clamped to the last row, a thread beyond
    // the bins reads bin 0) and so is the COARSE tier below: with no branch around a load or its use the
    // compiler can count the requests between a row's issue and its use and wait for that row only.
    const uint64_t i0c = active ? i0 : 0;
    const uint64_t last = stop ? stop - 1 : 0;
    auto issue = [&](Row &w, uint64_t q) __attribute__((always_inline)) {
        q = q < last ? q : last;
        w.c.load(mat + q * B + i0c);
        w.tot = d.totals[q];
        w.hrow = d.rowH[q];
    };
    uint32_t par = 0;
    const double cband = coarse_band(B);

    // COARSE tier on one row: 0 = rejected, 1 = the FAST tier must look, 2 = sure event, 3 = no valid k-mers
    // ("No valid k-mers": skipped, records.rs:332-335).  Straight-line, one barrier.
    auto coarse = [&](Row &w) __attribute__((always_inline)) -> int {
        // 1 / (T n), correctly rounded: T n is an integer below 2^24 for every row of 16-bit counts (one
        // f32 division); else by the f64 quotient
        const uint64_t tn = uint64_t(w.tot ? w.tot : 1u) * n;
        const float rtn = tn < (1ull << 24) ? 1.0f / float(uint32_t(tn)) : float(rn / double(w.tot));
        const dvs_f2 r2 = {rtn, rtn};
        double c0 = 0.0, c1 = 0.0;
        w.c.pin();
        w.c.coarse(slf + i0c, r2, c0, c1);
        const double cs = dvs_wave_sum_dpp(active ? c0 + c1 : 0.0);
        double *slot = red + par * 16;
        par ^= 1;
        if (lane == 0) slot[wave] = cs;
        __syncthreads();
        double hc = 0.0;
#pragma unroll
        for (int q = 0; q < H_WAVES; q++) hc += slot[q];
        const double jf0 = -hc - ((sumH - s_mH[li]) + w.hrow) * rn;  // (a reciprocal is good enough for this tier)
        // (NaN: a negative bin -- rejected as the reference does)
        return w.tot == 0 ? 3 : !(jf0 > thr - band - cband) ? 0 : jf0 > thr + band + cband ? 2 : 1;
    };

    // Everything past the COARSE tier for the row in `w` at stream position p.  Returns false when the
    // head phase must end here with the event unconsumed (a decision inside the rounding band).
    auto heavy = [&](Row &w, int tier) __attribute__((always_inline)) -> bool {
        const double tot = double(w.tot), rt = 1.0 / tot;
        const double mean_entropy = ((sumH - s_mH[li]) + w.hrow) / dn;
        bool sure = tier == 2;
        H_STAMP(1);
        double sv[H_NB], fl[H_NB];  // S and the lowest member's frequencies of this thread's bins
        if (active) {
            load_S(sv);
            mfreqs(li, fl);
        }
        if (!sure) {  // ---- FAST
            if (tid == 0) nmid++;
            double v[2] = {0.0, 0.0};  // (sum of -x log2 x, number of negative bins)
            if (active) {
#pragma unroll
                for (int j = 0; j < H_NB; j++) {
                    const double x = fma(double(w.c.at(j)), rt, sv[j] - fl[j]) * rn;
                    v[0] += fast_neg_xlog2x(x);
                    if (x < 0.0) v[1] += 1.0;
                }
            }
            h_block_sums<2>(v, scratch);
            const double jf = v[0] - mean_entropy;
            if (v[1] != 0.0 || !(jf > thr - band - FAST_BAND)) return true;
            sure = jf > thr + band + FAST_BAND;
        }
        if (!sure) {  // ---- exact f64, the reference's per-bin order (records.rs:78-81)
            if (tid == 0) nprecise++;
            Ent e;
            if (active) {
#pragma unroll
                for (int j = 0; j < H_NB; j++) {
                    const double f = exact_div_u32(double(w.c.at(j)), tot, rt);
                    e.add(((sv[j] - fl[j]) + f) * rn, s_ltab);
                }
            }
            double h = e.h, mn = e.mn, sm = e.sum;
            block_red3(h, mn, sm, scratch);
            const double jsd = (mn < 0.0) ? NAN : h - mean_entropy;
            if (sum_risky(sm, B) || fabs(jsd - thr) <= band) return false;  // the arbiter's
            n_events++;
            if (!(jsd > thr)) return true;  // rejected (NaN included, records.rs:91)
            n_events--;
        }
        n_events++;
        H_STAMP(2);
        // ================= replace_lowest (records.rs:94-147), computed before anything is stored
        double sn[H_NB];
        const double sumH_n = (sumH - s_mH[li]) + w.hrow;
        double tv[2] = {0.0, 0.0};  // whole set: entropy terms and sum of S' / n
        if (active) {
#pragma unroll
            for (int j = 0; j < H_NB; j++) {
                const double f = exact_div_u32(double(w.c.at(j)), tot, rt);
                double v = sv[j] - fl[j];  // drop_lowest, with its clamp
                if (v <= DVS_EPS) v = 0.0;
                sn[j] = v + f;              // push
                const double u = sn[j] * rn;
                if (u > 0.0) tv[0] -= u * log2_tab(u, s_ltab);
                tv[1] += u;
            }
        }
        h_block_sums<2>(tv, scratch);
        const double tj = tv[0] - sumH_n / dn;
        if (sum_risky(tv[1], B) || !(tv[0] == tv[0])) return false;
        const double band_n = sel_band(tj + sumH_n / dn, B);
        H_STAMP(3);
        // ---- leave-one-out (get_lowest_record_index, records.rs:220-252; updated_mean_freqs :276-286),
        // H_LB members at a time: member r of the NEW order is member r of the old one before the lowest,
        // r + 1 after it; the candidate is member n - 1.  The mean vector u = clamp((S' - f_r) / (n - 1)) is
        // formed in f64 (its subtraction cancels wherever member r alone fills a bin), its entropy by
        // v_log_f32 of the f32-rounded value, four products folded in f32 and added in f64: the COARSE
        // tier's arithmetic on an exactly rounded input, so the COARSE band bounds its error (select_dev.h:
        // one input rounding instead of three).  The sums of u are exact f64 adds (tolerance check).
        // delta_jsd of member r -> s_res[r].
        bool risky = false;
        for (uint32_t r0 = 0; r0 < n; r0 += H_LB) {
            double acc[2 * H_LB];
#pragma unroll
            for (int q = 0; q < H_LB; q++) {
                const uint32_t r = r0 + q;
                acc[2 * q] = acc[2 * q + 1] = 0.0;
                if (r < n && active) {
                    const uint32_t old = r < li ? r : r + 1;
                    const bool is_new = r == n - 1;  // (its counts are the candidate's, still in registers)
                    Cnt8<T> mc = mcounts(is_new ? 0u : old);
                    mc.take(w.c, is_new);
                    const double mrt = is_new ? rt : s_rt[old];
                    float t4[2] = {0.0f, 0.0f};
#pragma unroll
                    for (int j = 0; j < H_NB; j++) {
                        double u = (sn[j] - double(mc.at(j)) * mrt) * rdiv;
                        if (u <= DVS_EPS) u = 0.0;
                        acc[2 * q + 1] += u;
                        const float y = u == 0.0 ? 1e-30f : float(u);
                        t4[j >> 2] += y * __builtin_amdgcn_logf(y);
                    }
                    acc[2 * q] = -(double(t4[0]) + double(t4[1]));
                }
            }
            h_block_sums_many<2 * H_LB>(acc, scratch);
#pragma unroll
            for (int q = 0; q < H_LB; q++) {
                const uint32_t r = r0 + q;
                if (r < n) {
                    const uint32_t old = r < li ? r : r + 1;
                    const double mh = r == n - 1 ? w.hrow : s_mH[old];
                    risky |= sum_risky(acc[2 * q + 1], B);
                    if (tid == 0) s_res[r] = tj - (acc[2 * q] - (sumH_n - mh) * rdiv);
                }
            }
        }
        if (risky) return false;
        __syncthreads();
        H_STAMP(4);
        // members that could be the reference's argmin: within the tier's band (both ways) and the
        // reference's own rounding band of the smallest value
        double best = 1e6;
        for (uint32_t r = 0; r < n; r++) best = fmin(best, s_res[r]);
        const double reach = best + 2.0 * cband + band_n;
        uint32_t ncand = 0, lowest = 0;
        for (uint32_t r = 0; r < n; r++)
            if (s_res[r] <= reach) {
                if (ncand == 0) lowest = r;
                ncand++;
            }
        if (ncand > 1) {
            // exact f64 for those members only; strict '<' from 1e6, first index (records.rs:231,246-249)
            double dmin = 1e6, dsec = 1e6;
            for (uint32_t r = 0; r < n; r++) {
                if (!(s_res[r] <= reach)) continue;
                const uint32_t old = r < li ? r : r + 1;
                const bool is_new = r == n - 1;
                double ev[2] = {0.0, 0.0};
                if (active) {
                    Cnt8<T> mc = mcounts(is_new ? 0u : old);
                    mc.take(w.c, is_new);
                    const double mt = is_new ? tot : s_tot[old], mrt = is_new ? rt : s_rt[old];
#pragma unroll
                    for (int j = 0; j < H_NB; j++) {
                        double u = (sn[j] - exact_div_u32(double(mc.at(j)), mt, mrt)) * rdiv;
                        if (u <= DVS_EPS) u = 0.0;
                        if (u > 0.0) ev[0] -= u * log2_tab(u, s_ltab);
                        ev[1] += u;
                    }
                }
                h_block_sums<2>(ev, scratch);
                const double mh = is_new ? w.hrow : s_mH[old];
                const double de = tj - (ev[0] - (sumH_n - mh) * rdiv);
                if (de < dmin) {
                    dsec = dmin;
                    dmin = de;
                    lowest = r;
                } else if (de < dsec) {
                    dsec = de;
                }
            }
            if (dsec - dmin <= band_n && dsec < 1e6) return false;  // argmin too close to call
        }
        H_STAMP(5);
        // ================= commit
        __syncthreads();  // every thread has read the old member arrays and rows
        const uint32_t row_new = s_row[n], row_old = s_row[li], slot_low = s_slot[li];
        const uint64_t old_pos = s_pos[li];
        __syncthreads();
        if (active) {
            T *dst = Mc + uint64_t(row_new) * B + i0;
            if constexpr (sizeof(T) == 2) {
                *reinterpret_cast<uint4 *>(dst) = w.c.q;
            } else {
                *reinterpret_cast<uint4 *>(dst) = w.c.a;
                *reinterpret_cast<uint4 *>(dst + 4) = w.c.b;
            }
#pragma unroll
            for (int j = 0; j < H_NB; j += 2) *reinterpret_cast<double2 *>(S + i0 + j) = make_double2(sn[j], sn[j + 1]);
        }
        if (wave == 0) {  // Vec::remove(li) + push by one wave (reads before the dependent writes)
            const uint32_t i = li + lane;
            const bool mv = i + 1 < n;
            uint32_t a = 0, rw = 0;
            uint64_t c = 0;
            double b = 0.0, t = 1.0, r1 = 1.0;
            if (mv) {
                a = s_slot[i + 1];
                rw = s_row[i + 1];
                b = s_mH[i + 1];
                c = s_pos[i + 1];
                t = s_tot[i + 1];
                r1 = s_rt[i + 1];
            }
            if (mv) {
                s_slot[i] = a;
                s_row[i] = rw;
                s_mH[i] = b;
                s_pos[i] = c;
                s_tot[i] = t;
                s_rt[i] = r1;
            }
            if (lane == 0) {
                s_slot[n - 1] = slot_low;
                s_row[n - 1] = row_new;
                s_row[n] = row_old;  // the dropped member's row is the spare now
                s_mH[n - 1] = w.hrow;
                s_pos[n - 1] = p;
                s_tot[n - 1] = tot;
                s_rt[n - 1] = rt;
                // identity order, unique ids: label = stream position
                if (old_pos < d.nlabels) d.inset[old_pos] = 0;
                if (p < d.nlabels) d.inset[p] = 1;
                d.evlog_pos[ctl->n_logged + n_accepts] = p;
                d.evlog_kind[ctl->n_logged + n_accepts] = 1;
            }
        }
        __syncthreads();
        n_accepts++;
        li = lowest;
        sumH = sumH_n;
        total_jsd = tj;
        band = band_n;
        thr = tj + DVS_EPS;
        refresh_slf();
        __syncthreads();
        H_STAMP(6);
        return true;
    };

    // Two register blocks of H_D rows take turns: while the COARSE tier runs over one (straight-line code,
    // H_D barriers) the other is being fetched.  A row the tier cannot reject ends the scan: it is copied
    // out, block A is requested again from the position behind it (in flight while the event is handled)
    // and the loop restarts there; what the tier said about the rows behind an event is dropped.
    Row A[H_D], Bk[H_D];
    auto load_block = [&](Row (&blk)[H_D], uint64_t p0) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < H_D; k++) issue(blk[k], p0 + uint64_t(k));
    };
    // COARSE tier over a block; advances p past the rejected rows.  Returns the tier (1, 2) of the first
    // row it could not reject, copied to `cur` with p at its position, or 0 when the block is exhausted.
    auto scan = [&](Row (&blk)[H_D], Row &cur) __attribute__((always_inline)) -> int {
        int st_[H_D];
#pragma unroll
        for (int k = 0; k < H_D; k++) st_[k] = coarse(blk[k]);
        const uint64_t left = stop - p;  // rows of this block inside the stream (>= 1)
        int hit = -1, tier = 0;
        uint32_t scored = 0;
#pragma unroll
        for (int k = 0; k < H_D; k++)
            if (hit < 0 && uint64_t(k) < left) {
                if (st_[k] != 3) scored++;
                if (st_[k] == 1 || st_[k] == 2) {
                    hit = k;
                    tier = st_[k];
                }
            }
        if (tid == 0) nread += scored;
        if (hit < 0) {
            p += left < uint64_t(H_D) ? left : uint64_t(H_D);
            return 0;
        }
        cur.c = blk[0].c;
        cur.tot = blk[0].tot;
        cur.hrow = blk[0].hrow;
#pragma unroll
        for (int k = 1; k < H_D; k++) {
            cur.c.take(blk[k].c, hit == k);
            cur.tot = hit == k ? blk[k].tot : cur.tot;
            cur.hrow = hit == k ? blk[k].hrow : cur.hrow;
        }
        p += uint64_t(hit);
        return tier;
    };
    load_block(A, p);
#pragma unroll 1
    while (p < stop) {
        Row cur;
        load_block(Bk, p + uint64_t(H_D));
        int tier = scan(A, cur);
        if (tier == 0) {
            if (p >= stop) break;
            load_block(A, p + uint64_t(H_D));
            tier = scan(Bk, cur);
            if (tier == 0) continue;  // (A holds the rows from p again)
        }
        load_block(A, p + 1);  // in flight while the event is handled
        if (!heavy(cur, tier)) {
            bail = true;
            break;
        }
        p++;
    }

    H_STAMP(1);
    // ---- the mirror: what resolve_kernel leaves behind, leave-one-out + finalize pending (ev_kind 1)
    __syncthreads();
    if (active) {
#pragma unroll
        for (int j = 0; j < H_NB; j++) d.S[i0 + j] = S[i0 + j];
        for (uint32_t r = 0; r < n; r++) {
            double *mrow = d.M + uint64_t(s_slot[r]) * B + i0;
            double fm[H_NB];
            mfreqs(r, fm);
#pragma unroll
            for (int j = 0; j < H_NB; j++) mrow[j] = fm[j];
        }
    }
    if (tid < int(n)) {
        const uint32_t slot = s_slot[tid];
        d.ord[tid] = slot;
        d.mH[slot] = s_mH[tid];
        d.mPos[slot] = s_pos[tid];
        d.mLabel[slot] = uint32_t(s_pos[tid]);
    }
    if (tid == 0) {
        ctl->cursor = p;
        ctl->sum_entropy = sumH;
        ctl->total_jsd = total_jsd;
        ctl->lowest = li;
        if (n_accepts) ctl->s_is_resum = 0;
        ctl->n_logged += n_accepts;
        ctl->n_accepts += n_accepts;
        ctl->n_events += n_events;
        ctl->n_windows += n_accepts + 1;
        ctl->rows_scored += nread;
        ctl->rows_rechecked += nprecise;
        ctl->rows_coarse_passed += nmid;
        ctl->head_rows += nread;
        ctl->head_accepts += n_accepts;
        ctl->head_bailed = bail ? 1u : 0u;
        ctl->event_pos = SEL_NONE;
        ctl->ev_kind = 1;
        ctl->ev_n = n;
        ctl->ev_risky = 0;
    }
    H_STAMP(7);
}

}  // namespace

// Whether the head of this selection's stream can run on one workgroup, and up to which position.
int dvs_head_setup(dvs_ctx *ctx, dvs_select *s) {
    s->head_stop = 0;
    if (getenv("DVS_NO_HEAD")) return DVS_OK;
    const uint64_t B = s->dev.B;
    const uint32_t n = s->params.n_seed;
    if (s->params.mode != DVS_MODE_NMOST || !s->h_order.empty() || !s->h_labels.empty() || s->mat_kind == 1 ||
        (s->params.flags & DVS_SELECT_STEPWISE) || s->params.window || n < 2 || n > H_MAXN || B > H_MAXB || (B & 7))
        return DVS_OK;
    const size_t lds = s->mat_kind == 2 ? head_lds_bytes<uint16_t>(B, n) : head_lds_bytes<uint32_t>(B, n);
    if (lds > ctx->lds_per_block) return DVS_OK;
    // the head runs while an accept comes every `gap` rows or sooner (accept probability ~ n / i)
    uint64_t gap = 64;
    if (const char *e = getenv("DVS_HEAD_GAP")) gap = uint64_t(atoll(e));
    uint64_t stop = std::min<uint64_t>(s->npos, uint64_t(n) * gap);
    if (stop <= uint64_t(n) + 8) return DVS_OK;
    const void *fn = s->mat_kind == 2 ? reinterpret_cast<const void *>(head_nmost_kernel<uint16_t>)
                                      : reinterpret_cast<const void *>(head_nmost_kernel<uint32_t>);
    const int rc = dvs_raise_dyn_lds(ctx, fn, lds);
    if (rc) return rc;
    s->head_stop = stop;
    s->head_lds = lds;
    return DVS_OK;
}

// enqueues the head kernel on `stream` (the set-up kernels of the selection in front of it)
int dvs_head_launch(dvs_ctx *ctx, dvs_select *s, hipStream_t stream) {
    if (s->mat_kind == 2)
        hipLaunchKernelGGL((head_nmost_kernel<uint16_t>), dim3(1), dim3(H_THREADS), s->head_lds, stream, s->dev,
                           static_cast<const uint16_t *>(s->mat->d_counts16), s->head_stop);
    else
        hipLaunchKernelGGL((head_nmost_kernel<uint32_t>), dim3(1), dim3(H_THREADS), s->head_lds, stream, s->dev,
                           static_cast<const uint32_t *>(s->mat->d_counts), s->head_stop);
    DVS_HIP(ctx, hipGetLastError());
    return DVS_OK;
}
