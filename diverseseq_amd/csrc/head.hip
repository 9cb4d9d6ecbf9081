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
constexpr int H_D = 4;              // rows requested ahead
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

// LDS: [S B f64][slf B f32][Mc (H_MAXN + 1) rows of B counts][scratch 64 + 2 * H_MAXN * H_WAVES f64]
//      [s_mH, s_tot, s_rt, s_res (H_MAXN + 1) f64 each][s_pos (H_MAXN + 1) u64][s_slot, s_row (H_MAXN + 1) u32]
//      [log2 table 128 x double2]
template <typename T>
__host__ __device__ constexpr size_t head_lds_bytes(uint64_t B, uint32_t n) {
    return B * 8 + B * 4 + size_t(n + 1) * B * sizeof(T) + (64 + 2 * H_MAXN * H_WAVES) * 8 + 4 * (H_MAXN + 1) * 8 +
           (H_MAXN + 1) * 8 + 2 * (H_MAXN + 1) * 4 + 32 + 128 * 16;
}

template <typename T>
__global__ __launch_bounds__(H_THREADS) void head_nmost_kernel(SelDev d, const T *__restrict__ mat, uint64_t stop_arg) {
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
    double *red = scratch + 2 * H_MAXN * H_WAVES;  // 64 doubles: the per-row partial sums (two parities)
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
    // f64 frequencies of a member's bins from its LDS count row (record.rs:139, correctly rounded)
    auto mfreq = [&](uint32_t r, int j) -> double {
        return exact_div_u32(double(Mc[uint64_t(s_row[r]) * B + i0 + j]), s_tot[r], s_rt[r]);
    };
    auto refresh_slf = [&]() {  // slf = (S - lowest) / n as the COARSE tier wants it
        if (active) {
#pragma unroll
            for (int j = 0; j < H_NB; j++) slf[i0 + j] = coarse_sl(S[i0 + j] - mfreq(li, j), rn);
        }
    };
    refresh_slf();
    __syncthreads();

    uint32_t nread = 0, nprecise = 0, nmid = 0, n_events = 0, n_accepts = 0;
    bool bail = false;

    struct Row {
        Cnt8<T> c;
        uint32_t tot;
        double hrow;
    };
    auto issue = [&](Row &w, uint64_t q) {
        if (active) w.c.load(mat + q * B + i0);
        w.tot = d.totals[q];
        w.hrow = d.rowH[q];
    };
    Row ring[H_D];
#pragma unroll
    for (int k = 0; k < H_D; k++)
        if (p + uint64_t(k) < stop) issue(ring[k], p + uint64_t(k));
    uint32_t par = 0;
    const double cband = coarse_band(B);

    // COARSE tier on one row: 0 = rejected (or no valid k-mers), 1 = the FAST tier must look, 2 = sure event.
    // Small on purpose: it is the only code expanded once per ring slot.
    auto coarse = [&](Row &w) -> int {
        if (w.tot == 0) return 0;  // "No valid k-mers": skipped (records.rs:332-335)
        const double rt = 1.0 / double(w.tot);
        const float rtn = float(rt * rn);
        const dvs_f2 r2 = {rtn, rtn};
        double c0 = 0.0, c1 = 0.0;
        w.c.pin();
        if (active) w.c.coarse(slf + i0, r2, c0, c1);
        const double cs = dvs_wave_sum_dpp(c0 + c1);
        double *slot = red + par * 16;
        par ^= 1;
        if (lane == 0) slot[wave] = cs;
        __syncthreads();
        double hc = 0.0;
#pragma unroll
        for (int q = 0; q < H_WAVES; q++) hc += slot[q];
        const double jf0 = -hc - ((sumH - s_mH[li]) + w.hrow) / dn;
        if (tid == 0) nread++;
        if (!(jf0 > thr - band - cband)) return 0;  // (NaN: a negative bin, rejected as the reference does)
        return jf0 > thr + band + cband ? 2 : 1;
    };

    // Everything past the COARSE tier for the row in `w` at stream position p.  Returns false when the
    // head phase must end here with the event unconsumed (a decision inside the rounding band).
    auto heavy = [&](Row &w, int tier) -> bool {
        const double tot = double(w.tot), rt = 1.0 / tot;
        const double mean_entropy = ((sumH - s_mH[li]) + w.hrow) / dn;
        bool sure = tier == 2;
        if (!sure) {  // ---- FAST
            if (tid == 0) nmid++;
            double v[2] = {0.0, 0.0};  // (sum of -x log2 x, number of negative bins)
            if (active) {
#pragma unroll
                for (int j = 0; j < H_NB; j++) {
                    const double x = fma(double(w.c.at(j)), rt, S[i0 + j] - mfreq(li, j)) * rn;
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
                    e.add(((S[i0 + j] - mfreq(li, j)) + f) * rn, s_ltab);
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
        // ================= replace_lowest (records.rs:94-147), computed before anything is stored
        double fr[H_NB], sn[H_NB];
        const double sumH_n = (sumH - s_mH[li]) + w.hrow;
        double tv[2] = {0.0, 0.0};  // whole set: entropy terms and sum of S' / n
        if (active) {
#pragma unroll
            for (int j = 0; j < H_NB; j++) {
                fr[j] = exact_div_u32(double(w.c.at(j)), tot, rt);
                double v = S[i0 + j] - mfreq(li, j);  // drop_lowest, with its clamp
                if (v <= DVS_EPS) v = 0.0;
                sn[j] = v + fr[j];                     // push
                const double u = sn[j] * rn;
                if (u > 0.0) tv[0] -= u * log2_tab(u, s_ltab);
                tv[1] += u;
            }
        }
        h_block_sums<2>(tv, scratch);
        const double tj = tv[0] - sumH_n / dn;
        if (sum_risky(tv[1], B) || !(tv[0] == tv[0])) return false;
        const double band_n = sel_band(tj + sumH_n / dn, B);
        // ---- leave-one-out, FAST tier (get_lowest_record_index, records.rs:220-252), four members at a
        // time: member r of the NEW order is member r of the old one before the lowest, r + 1 after it;
        // the candidate is member n - 1.  delta_jsd of member r -> s_res[r]; the sums are exact f64 adds.
        bool risky = false;
        for (uint32_t r0 = 0; r0 < n; r0 += 4) {
            double acc[8];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const uint32_t r = r0 + q;
                acc[2 * q] = acc[2 * q + 1] = 0.0;
                if (r < n && active) {
                    const uint32_t old = r < li ? r : r + 1;
                    const bool is_new = r == n - 1;
#pragma unroll
                    for (int j = 0; j < H_NB; j++) {
                        double u = (sn[j] - (is_new ? fr[j] : mfreq(old, j))) * rdiv;  // updated_mean_freqs :276-286
                        if (u <= DVS_EPS) u = 0.0;
                        acc[2 * q] += fast_neg_xlog2x(u);
                        acc[2 * q + 1] += u;
                    }
                }
            }
            h_block_sums<8>(acc, scratch);
#pragma unroll
            for (int q = 0; q < 4; q++) {
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
        // members that could be the reference's argmin: within the FAST band (both ways) and the
        // reference's own rounding band of the smallest FAST value
        double best = 1e6;
        for (uint32_t r = 0; r < n; r++) best = fmin(best, s_res[r]);
        const double reach = best + 2.0 * FAST_BAND + band_n;
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
#pragma unroll
                    for (int j = 0; j < H_NB; j++) {
                        double u = (sn[j] - (is_new ? fr[j] : mfreq(old, j))) * rdiv;
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
        // ================= commit
        __syncthreads();  // every thread has read the old member arrays and rows
        const uint32_t row_new = s_row[n], row_old = s_row[li], slot_low = s_slot[li];
        const uint64_t old_pos = s_pos[li];
        __syncthreads();
        if (active) {
            T *dst = Mc + uint64_t(row_new) * B + i0;
#pragma unroll
            for (int j = 0; j < H_NB; j++) {
                dst[j] = T(w.c.at(j));
                S[i0 + j] = sn[j];
            }
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
        return true;
    };

    // Rows are consumed in stream order; slot k of the ring holds row p when (p - first) % H_D == k.  The
    // unrolled pass runs the COARSE tier slot by slot; a row it cannot reject leaves the pass, is copied
    // out of its slot by a (wave-uniform) branch, and takes the one expansion of `heavy`.
    int start = 0;
#pragma unroll 1
    while (p < stop && !bail) {
        int hit = -1, tier = 0;
#pragma unroll
        for (int k = 0; k < H_D; k++) {
            if (k >= start && hit < 0 && p < stop) {
                const int st_ = coarse(ring[k]);
                if (st_ == 0) {
                    if (p + uint64_t(H_D) < stop) issue(ring[k], p + uint64_t(H_D));
                    p++;
                } else {
                    hit = k;
                    tier = st_;
                }
            }
        }
        start = 0;
        if (hit >= 0) {
            Row cur;
            if (hit == 0) cur = ring[0];
            else if (hit == 1) cur = ring[1];
            else if (hit == 2) cur = ring[2];
            else cur = ring[3];
            static_assert(H_D == 4, "the slot copies above and below list four slots");
            if (!heavy(cur, tier)) {
                bail = true;
            } else {
                if (p + uint64_t(H_D) < stop) {
                    if (hit == 0) issue(ring[0], p + uint64_t(H_D));
                    else if (hit == 1) issue(ring[1], p + uint64_t(H_D));
                    else if (hit == 2) issue(ring[2], p + uint64_t(H_D));
                    else issue(ring[3], p + uint64_t(H_D));
                }
                p++;
                start = (hit + 1) % H_D;
            }
        }
    }

    // ---- the mirror: what resolve_kernel leaves behind, leave-one-out + finalize pending (ev_kind 1)
    __syncthreads();
    if (active) {
#pragma unroll
        for (int j = 0; j < H_NB; j++) d.S[i0 + j] = S[i0 + j];
        for (uint32_t r = 0; r < n; r++) {
            double *mrow = d.M + uint64_t(s_slot[r]) * B + i0;
#pragma unroll
            for (int j = 0; j < H_NB; j++) mrow[j] = mfreq(r, j);
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
