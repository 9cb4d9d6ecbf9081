// Greedy delta-JSD selection over the N x B count matrix, entirely on the GPU.
//
// Replaces SummedRecords and the selectors of the reference (src/records.rs):
//   delta_jsd :70-84, increases_jsd :86-92, drop_lowest :94-109, replace_lowest
//   :111-118, push :120-147, stats :153-172, clone :182-189 (= new, :27-68),
//   get_lowest_record_index :220-252, updated_mean_freqs :276-286,
//   select_nmost_divergent :311-342, select_max_divergent :390-454 and the
//   *_final merges :363-382 / :456-507.
//
// The reference walks the candidates one by one; candidate i is tested against
// the set as modified by every accepted j < i.  Here a WINDOW of candidates is
// scored in parallel against the current state (scan_kernel, the HBM-bound hot
// kernel: one wavefront per 4^k-bin row), the FIRST position whose score clears
// the threshold is the only one acted on, and the scan restarts right after it:
// every candidate before it was rejected under exactly the state the sequential
// code would have used, so the selected ids are the reference's.  The state
// update (resolve -> leave-one-out -> finalize kernels) runs on the device too;
// the cursor, window and event live in device memory, so the host enqueues
// launch batches blindly and only polls the control block between batches.
//
// Decisions the device cannot separate from the reference's own rounding noise
// (|jsd - threshold| <= band, band ~ 4 B eps H) are handed to the host tie
// arbiter (exact_set.cpp), which replays the event log in the reference's exact
// f64 operation order.
#include "dvs_internal.h"
#include "select.h"
#include "select_dev.h"

#include <algorithm>
#include <chrono>
#include <type_traits>
#include <cmath>
#include <cstring>
#include <cstdlib>

namespace {

constexpr int SCAN_THREADS = 512;   // 8 waves share one LDS copy of the state vector
constexpr int WIDE_THREADS = 1024;  // single-block state kernels
constexpr int LOO_THREADS = 256;

// Hot row loop of one wave: identity order, unique ids, B a multiple of 256 * SCAN_CH
// bins.  A row is consumed in batches of SCAN_CH chunks of 1 KiB per wave instruction,
// each batch requested in one burst before any of it is consumed.
template <typename T>
__device__ __forceinline__ void scan_rows_hot(SelCtl *ctl, const T *__restrict__ mat,
                                              const uint32_t *__restrict__ totals,
                                              const double *__restrict__ rowH, const double *bvec,
                                              uint64_t B, const ScanState &st, uint64_t first,
                                              uint64_t stride, uint32_t lane, uint32_t &nread,
                                              uint32_t &nprecise) {
    for (uint64_t r = first; r < st.nrows; r += stride) {
        const uint64_t p = st.cursor + r;
        const T *rp = mat + p * B;
        // the event word and the row's total / entropy in one round trip
        const unsigned long long ev =
            __hip_atomic_load(&ctl->event_pos, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t tot = totals[p];
        const double hrow = rowH[p];
        if (ev < p) break;       // an earlier event already ends this window: later rows are void
        if (tot == 0) continue;  // "No valid k-mers": skipped (src/records.rs:332-335)
        const double rinv = 1.0 / (double(tot) * st.dsize);
        const double mean_entropy = (st.he_base + hrow) / st.dsize;
        nread++;
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0, xmin = 0.0;
        for (uint64_t i0 = 0; i0 < B; i0 += 256 * SCAN_CH) {
            // SCAN_CH chunks (1 KiB per wave instruction each) requested before any is consumed
            Raw4<T> raw[SCAN_CH];
#pragma unroll
            for (int j = 0; j < SCAN_CH; j++) raw[j].load(rp + i0 + uint64_t(j) * 256 + lane * 4);
#pragma unroll
            for (int j = 0; j < SCAN_CH; j++) {
                const uint64_t i = i0 + uint64_t(j) * 256 + lane * 4;
                const double2 b01 = *reinterpret_cast<const double2 *>(bvec + i);
                const double2 b23 = *reinterpret_cast<const double2 *>(bvec + i + 2);
                double v0, v1, v2, v3;
                raw[j].get(v0, v1, v2, v3);
                fast4(b01, b23, v0, v1, v2, v3, rinv, a0, a1, a2, a3, xmin);
            }
        }
        const double hf = dvs_wave_sum((a0 + a1) + (a2 + a3));
        const double mn = dvs_wave_min(xmin);
        // a negative bin is NaN in the reference (rejected); else compare with margin
        const double jf = hf - mean_entropy;
        if (!(mn < 0.0) && jf > st.thr_fast) {
            bool hit = jf > st.thr_sure;  // above the threshold by more than the fast tier's error
            if (!hit) {                   // only the +-FAST_BAND zone pays for the f64 tier
                nprecise++;
                hit = precise_row(rp, bvec, B, rinv, mean_entropy, st.thr_lo, lane);
            }
            if (hit && lane == 0) atomicMin(&ctl->event_pos, (unsigned long long)p);
        }
    }
}

// General row loop (explicit order / labels, any B): correctness first.
template <typename T>
__device__ __forceinline__ void scan_rows_general(SelCtl *ctl, const T *__restrict__ mat,
                                               const uint32_t *__restrict__ totals,
                                               const double *__restrict__ rowH,
                                               const uint32_t *__restrict__ order,
                                               const uint32_t *__restrict__ labels,
                                               const uint8_t *__restrict__ inset, uint32_t nlabels,
                                               const double *bvec, uint64_t B, const ScanState &st,
                                               uint64_t first, uint64_t stride, uint32_t lane,
                                               uint32_t &nread, uint32_t &nprecise) {
    for (uint64_t r = first; r < st.nrows; r += stride) {
        const uint64_t p = st.cursor + r;
        if (__hip_atomic_load(&ctl->event_pos, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < p) break;
        const uint32_t row = order ? order[p] : uint32_t(p);
        if (row == DVS_ROW_REMOTE) continue;  // another rank scores this position
        const uint32_t tot = totals[row];
        if (tot == 0) continue;
        if (labels) {  // ids can only repeat when the caller passed labels (records.rs:87-89)
            const uint32_t lab = labels[p];
            if (lab < nlabels && inset[lab]) continue;
        }
        const T *rp = mat + uint64_t(row) * B;
        const double rinv = 1.0 / (double(tot) * st.dsize);
        const double mean_entropy = (st.he_base + rowH[row]) / st.dsize;
        nread++;
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0, xmin = 0.0;
        if ((B & 255) == 0) {
            for (uint64_t i0 = 0; i0 < B; i0 += 256) {
                const uint64_t i = i0 + lane * 4;
                double v0, v1, v2, v3;
                load4(rp, i, v0, v1, v2, v3);
                const double2 b01 = *reinterpret_cast<const double2 *>(bvec + i);
                const double2 b23 = *reinterpret_cast<const double2 *>(bvec + i + 2);
                fast4(b01, b23, v0, v1, v2, v3, rinv, a0, a1, a2, a3, xmin);
            }
        } else {
            for (uint64_t i = lane; i < B; i += 64) {
                const double x = fma(row_value(rp, i), rinv, bvec[i]);
                a0 += fast_neg_xlog2x(x);
                xmin = fmin(xmin, x);
            }
        }
        const double hf = dvs_wave_sum((a0 + a1) + (a2 + a3));
        const double mn = dvs_wave_min(xmin);
        if (mn < 0.0) continue;
        const double jf = hf - mean_entropy;
        if (!(jf > st.thr_fast)) continue;
        bool hit = jf > st.thr_sure;
        if (!hit) {
            nprecise++;
            hit = precise_row(rp, bvec, B, rinv, mean_entropy, st.thr_lo, lane);
        }
        if (hit && lane == 0) atomicMin(&ctl->event_pos, (unsigned long long)p);
    }
}

// HOT: identity order, unique ids, B a multiple of 256 * SCAN_CH (host-checked)
template <typename T, bool HOT>
__device__ __forceinline__ void scan_body(
    SelCtl *__restrict__ ctl, const T *__restrict__ mat, const uint32_t *__restrict__ totals,
    const double *__restrict__ rowH, const uint32_t *__restrict__ order,
    const uint32_t *__restrict__ labels, const uint8_t *__restrict__ inset, uint32_t nlabels,
    const double *__restrict__ base, uint32_t *__restrict__ wg_rows, uint64_t B, int base_in_lds,
    unsigned char *smem) {
    uint32_t *s_rows = reinterpret_cast<uint32_t *>(smem);  // 16 B header
    double *sb = reinterpret_cast<double *>(smem + 16);
    if (ctl->status != SEL_RUN) return;
    ScanState st;
    st.cursor = ctl->cursor;
    const uint64_t end = umin64(st.cursor + uint64_t(ctl->window), ctl->npos);
    if (st.cursor >= end) return;
    st.nrows = end - st.cursor;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint64_t wpb = SCAN_THREADS / 64;
    if (uint64_t(blockIdx.x) * wpb >= st.nrows) return;  // whole workgroup idle
    st.thr_lo = ctl->thr - ctl->band;
    st.thr_fast = st.thr_lo - FAST_BAND;
    st.thr_sure = ctl->thr + ctl->band + FAST_BAND;
    st.he_base = ctl->he_base;
    st.dsize = double(ctl->size);
    const double *bvec = base;
    if (threadIdx.x < 2) s_rows[threadIdx.x] = 0;
    if (base_in_lds) {
        if ((B & 1) == 0) {
            const double2 *s2 = reinterpret_cast<const double2 *>(base);
            double2 *d2 = reinterpret_cast<double2 *>(sb);
            for (uint64_t i = threadIdx.x; i < B / 2; i += SCAN_THREADS) d2[i] = s2[i];
        } else {
            for (uint64_t i = threadIdx.x; i < B; i += SCAN_THREADS) sb[i] = base[i];
        }
        bvec = sb;
    }
    __syncthreads();
    const uint64_t first = uint64_t(blockIdx.x) * wpb + wave;
    const uint64_t stride = uint64_t(gridDim.x) * wpb;
    uint32_t nread = 0, nprecise = 0;
    if (HOT)
        scan_rows_hot<T>(ctl, mat, totals, rowH, bvec, B, st, first, stride, lane, nread, nprecise);
    else
        scan_rows_general<T>(ctl, mat, totals, rowH, order, labels, inset, nlabels, bvec, B, st, first,
                             stride, lane, nread, nprecise);
    if (lane == 0 && nread) {
        atomicAdd(&s_rows[0], nread);
        if (nprecise) atomicAdd(&s_rows[1], nprecise);
    }
    __syncthreads();
    if (threadIdx.x == 0 && s_rows[0]) {  // summed + cleared by resolve
        wg_rows[2 * blockIdx.x] = s_rows[0];
        wg_rows[2 * blockIdx.x + 1] = s_rows[1];
    }
}

template <typename T, bool HOT>
__global__ __launch_bounds__(SCAN_THREADS, 4) void scan_kernel(
    SelCtl *__restrict__ ctl, const T *__restrict__ mat, const uint32_t *__restrict__ totals,
    const double *__restrict__ rowH, const uint32_t *__restrict__ order,
    const uint32_t *__restrict__ labels, const uint8_t *__restrict__ inset, uint32_t nlabels,
    const double *__restrict__ base, uint32_t *__restrict__ wg_rows, uint64_t B, int base_in_lds) {
    // all LDS in the dynamic region (a static __shared__ in front of it would shift its
    // base off 16 B and every ds_read_b128 of the state vector would be replayed)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    scan_body<T, HOT>(ctl, mat, totals, rowH, order, labels, inset, nlabels, base, wg_rows, B, base_in_lds,
                      smem);
}

// slot of the stepwise exchange: [0] event position as a double (< 0: none), [1] H(row), [2 ..] the
// candidate's frequency row
template <typename T>
__device__ void pack_event_body(const SelDev &d, const T *__restrict__ mat, double *__restrict__ slot) {
    const SelCtl *ctl = d.ctl;
    const unsigned long long p =
        __hip_atomic_load(&ctl->event_pos, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t row = DVS_ROW_REMOTE;
    if (ctl->status == SEL_RUN && p != SEL_NONE) row = d.order ? d.order[p] : uint32_t(p);
    if (row == DVS_ROW_REMOTE) {
        if (threadIdx.x == 0) slot[0] = -1.0;  // (the rest of the slot is not read)
        return;
    }
    const double tot = double(d.totals[row]);
    const T *rp = mat + uint64_t(row) * d.B;
    for (uint64_t i = threadIdx.x; i < d.B; i += blockDim.x) slot[2 + i] = cand_freq(rp, i, tot);
    if (threadIdx.x == 0) {
        slot[0] = double(p);  // (exact below 2^53)
        slot[1] = d.rowH[row];
    }
}

// max |double(v_log_f32(m)) - log2(m)| over every f32 m in [0.5, 1): the hardware
// term of FAST_BAND.  One thread per 2^11 consecutive mantissas.
__global__ void fast_log2_selftest_kernel(double *out) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;  // 4096 threads x 2048 = 2^23
    double worst = 0.0;
    for (uint32_t j = 0; j < 2048; j++) {
        const uint32_t bits = 0x3F000000u + t * 2048u + j;  // [0.5, 1)
        const float m = __uint_as_float(bits);
        const double err = fabs(double(__builtin_amdgcn_logf(m)) - log2(double(m)));
        worst = fmax(worst, err);
    }
    for (int o = 32; o > 0; o >>= 1) worst = fmax(worst, __shfl_xor(worst, o, 64));
    if ((threadIdx.x & 63) == 0) {
        unsigned long long *p = reinterpret_cast<unsigned long long *>(out);
        atomicMax(p, (unsigned long long)__double_as_longlong(worst));  // positive doubles order as ints
    }
}

// ------------------------------------------------------------- state kernels
// JSD of the set with `lowest` swapped for the candidate in d.cand
// (src/records.rs:70-84), all threads of the block get the result.
// *h_out: the entropy of that mean vector; *clamped: some bin of S - lowest is one drop_lowest
// would clamp to zero (records.rs:100-105) -- when none is, the vector is bit for bit the one
// replace_lowest leaves in S / size, so an accept need not evaluate its entropy again.
__device__ double block_delta_jsd(const SelDev &d, const SelCtl *ctl, double cand_H,
                                  double *scratch, double *sum_out, double *h_out, int *clamped) {
    const uint32_t low_slot = d.ord[ctl->lowest];
    const double *low = d.M + uint64_t(low_slot) * d.B;
    const double dsize = double(ctl->size);
    Ent e;
    int cl = 0;
    for (uint64_t i = threadIdx.x; i < d.B; i += blockDim.x) {
        const double v = d.S[i] - low[i];
        cl |= (v <= DVS_EPS && v != 0.0) ? 1 : 0;
        e.add((v + d.cand[i]) / dsize);
    }
    double h = e.h, mn = e.mn, sm = e.sum;
    block_red3(h, mn, sm, scratch);
    *clamped = __syncthreads_or(cl);
    *h_out = (mn < 0.0) ? NAN : h;
    if (sum_out) *sum_out = sm;
    const double mean_entropy = (ctl->sum_entropy - d.mH[low_slot] + cand_H) / dsize;
    return (mn < 0.0) ? NAN : h - mean_entropy;
}

// H(vec / div) over the block, plus sum for the tolerance check
__device__ double block_entropy_div(const double *vec, double div, uint64_t B, double *scratch,
                                    double *sum_out) {
    Ent e;
    for (uint64_t i = threadIdx.x; i < B; i += blockDim.x) e.add(vec[i] / div);
    double h = e.h, mn = e.mn, sm = e.sum;
    block_red3(h, mn, sm, scratch);
    if (sum_out) *sum_out = sm;
    return (mn < 0.0) ? NAN : h;
}


template <typename T>
__device__ void resolve_body(SelDev &d, const T *__restrict__ mat, double *scratch, int &s_action) {
    SelCtl *ctl = d.ctl;
    if (ctl->status != SEL_RUN) return;
    const int tid = threadIdx.x;
    const uint32_t WIDE = blockDim.x;
    const double *ext_row = nullptr, *ext_H = nullptr;
    unsigned long long gathered_p = SEL_NONE;
    // (stepwise re-entry behind the arbiter: the candidate is the one resolve stopped at, still in d.cand --
    // the gathered slots are not looked at: the caller's buffer may have been reused since the step's apply)
    const bool reentry = d.gather_all && (ctl->forced == FORCE_ACCEPT || ctl->forced == FORCE_REJECT);
    if (d.gather_all && !reentry) {
        // stepwise mode: the earliest event among the gathered slots ([pos, H, row]) is this step's
        // event on every rank
        __shared__ int s_best;
        if (tid == 0) {
            int best = -1;
            double bp = 0.0;
            const uint64_t stride = d.B + 2;
            for (uint32_t r = 0; r < d.gather_world; r++) {
                const double q = d.gather_all[uint64_t(r) * stride];
                if (q >= 0.0 && (best < 0 || q < bp)) {
                    best = int(r);
                    bp = q;
                }
            }
            s_best = best;
            ctl->event_pos = best < 0 ? SEL_NONE : (unsigned long long)bp;
        }
        __syncthreads();
        const double *src = d.gather_all + uint64_t(s_best < 0 ? 0 : s_best) * (d.B + 2);
        ext_row = src + 2;
        ext_H = src + 1;
        gathered_p = s_best < 0 ? SEL_NONE : (unsigned long long)src[0];
    }
    const uint64_t p = reentry ? ctl->arb_pos : d.gather_all ? gathered_p : ctl->event_pos;
    if (reentry && tid == 0) ctl->event_pos = p;
    if (ctl->ev_kind != 0) return;  // a finalize is pending (arbiter re-entry)
    if (p == SEL_NONE) {
        if (tid == 0) {
            const uint64_t end = umin64(ctl->cursor + uint64_t(ctl->window), ctl->npos);
            ctl->n_windows++;
            ctl->cursor = end;
            ctl->mb_stuck = 0;
            if (end >= ctl->npos) ctl->status = SEL_DONE;
            ctl_next_window(ctl);
        }
        return;
    }
    const uint32_t row = d.order ? d.order[p] : uint32_t(p);
    uint32_t lab;
    double cand_H;
    if (reentry) {
        lab = (row == DVS_ROW_REMOTE) ? DVS_ROW_REMOTE : (d.labels ? d.labels[p] : row);
        cand_H = ctl->arb_H;
    } else if (ext_row) {  // exchanged candidate (its row may live on another rank)
        lab = (row == DVS_ROW_REMOTE) ? DVS_ROW_REMOTE : (d.labels ? d.labels[p] : row);
        cand_H = *ext_H;
        for (uint64_t i = tid; i < d.B; i += WIDE) d.cand[i] = ext_row[i];
    } else {
        lab = d.labels ? d.labels[p] : row;
        const double tot = double(d.totals[row]);
        cand_H = d.rowH[row];
        const T *rp = mat + uint64_t(row) * d.B;
        for (uint64_t i = tid; i < d.B; i += WIDE) d.cand[i] = cand_freq(rp, i, tot);
    }
    __syncthreads();
    double sm;
    double h_swapped;
    int any_clamped;
    const double jsd = block_delta_jsd(d, ctl, cand_H, scratch, &sm, &h_swapped, &any_clamped);
    if (tid == 0) {
        int action;
        const uint32_t forced = ctl->forced;
        if (forced == FORCE_ACCEPT || forced == FORCE_REJECT) {
            action = forced == FORCE_ACCEPT ? 1 : 0;
            ctl->forced = FORCE_NONE;
        } else if (sum_risky(sm, d.B) || fabs(jsd - ctl->thr) <= ctl->band) {
            ctl->status = SEL_ARBITER;
            ctl->arb_stage = ARB_RESOLVE;
            ctl->arb_pos = p;
            ctl->arb_H = cand_H;
            action = 3;
        } else {
            action = (jsd > ctl->thr) ? 1 : 0;  // NaN -> reject (records.rs:91)
        }
        if (action == 1 && ctl->mode == DVS_MODE_MAX && ctl->size < ctl->max_size) action = 2;
        if (action != 3) {
            ctl->n_windows++;
            ctl->n_events++;
            ctl->cursor = p + 1;
            ctl->event_pos = SEL_NONE;
            ctl->last_jsd = jsd;
            if (action == 0) ctl->mb_stuck = 0;  // (rejected: the row a batch stopped at is behind the cursor now)
            if (action == 0 && ctl->cursor >= ctl->npos) ctl->status = SEL_DONE;
        }
        s_action = action;
    }
    __syncthreads();
    const int action = s_action;
    if (action == 0 || action == 3) return;

    if (action == 1) {
        // replace_lowest = drop_lowest (records.rs:94-109) + push (:120-147)
        const uint32_t li = ctl->lowest, n = ctl->size;
        const uint32_t s = d.ord[li];
        double *mrow = d.M + uint64_t(s) * d.B;
        const uint32_t log_at = ctl->n_logged;  // (read by every thread before thread 0 moves it on, below)
        double *logrow = d.rowlog ? d.rowlog + uint64_t(log_at % d.rowlog_cap) * d.B : nullptr;  // (a ring: the host drains it at every poll)
        __syncthreads();
        if (tid == 0) ctl->s_is_resum = 0;
        for (uint64_t i = tid; i < d.B; i += WIDE) {
            double v = d.S[i] - mrow[i];
            if (v <= DVS_EPS) v = 0.0;
            const double f = d.cand[i];
            d.S[i] = v + f;
            mrow[i] = f;
            if (logrow) logrow[i] = f;
        }
        __syncthreads();
        if (tid == 0) {
            double sh = ctl->sum_entropy - d.mH[s];
            sh += cand_H;
            ctl->sum_entropy = sh;
            const uint32_t old_lab = d.mLabel[s];
            if (old_lab < d.nlabels) d.inset[old_lab] = 0;
            if (lab < d.nlabels) d.inset[lab] = 1;
            d.mH[s] = cand_H;
            d.mLabel[s] = lab;
            d.mPos[s] = p;
            ctl->n_accepts++;
            d.evlog_pos[ctl->n_logged] = p;
            d.evlog_kind[ctl->n_logged] = 1;
            ctl->n_logged++;
        }
        // Vec::remove(li) + push: the member order moves up by one from li, a block-wide chunk at a
        // time (read, barrier, write; thread 0 doing it alone was a chain of n dependent round trips)
        for (uint32_t b0 = li; b0 + 1 < n; b0 += WIDE) {
            const uint32_t i = b0 + tid;
            const uint32_t o = (i + 1 < n) ? d.ord[i + 1] : 0u;
            __syncthreads();
            if (i + 1 < n) d.ord[i] = o;
        }
        if (tid == 0) d.ord[n - 1] = s;  // (slot n-1 is read by no chunk after the one that wrote n-2)
        __syncthreads();
        // H(S / n) of the new set: what delta_jsd evaluated, unless drop_lowest's clamp changed a bin
        double sm2 = sm, hm = h_swapped;
        if (any_clamped) hm = block_entropy_div(d.S, double(n), d.B, scratch, &sm2);
        if (tid == 0) {
            ctl->total_jsd = hm - ctl->sum_entropy / double(n);
            ctl->ev_kind = 1;
            ctl->ev_n = n;
            if (sum_risky(sm2, d.B) || !(hm == hm)) ctl->ev_risky = 1;
        }
    } else {
        // MODE_MAX with room: clone (= new over the members, records.rs:27-68,182-189)
        // then push the candidate; kept only if the stat rises (finalize decides)
        // The clone's sums are the members' rows (entropies) added up in member order.  While the
        // set has only grown by pushes -- the rule in this mode until max_size is reached, after
        // which nothing is cloned any more -- the running S / sum_entropy ARE those sums, bit for
        // bit: rebuild_kernel formed them in member order and every kept push appended its
        // candidate at the end of both the order and the addition chain.
        const uint32_t n = ctl->size;
        const bool resum = ctl->s_is_resum == 0;
        double *mrow = d.M + uint64_t(n) * d.B;  // slot n is free
        for (uint64_t i = tid; i < d.B; i += WIDE) {
            double acc;
            if (resum) {
                acc = 0.0;
                for (uint32_t r = 0; r < n; r++) acc += d.M[uint64_t(d.ord[r]) * d.B + i];
            } else {
                acc = d.S[i];
            }
            const double f = d.cand[i];
            d.Stmp[i] = acc + f;
            mrow[i] = f;
        }
        if (tid == 0) {
            double sh;
            if (resum) {
                sh = 0.0;
                for (uint32_t r = 0; r < n; r++) sh += d.mH[d.ord[r]];
            } else {
                sh = ctl->sum_entropy;
            }
            sh += cand_H;
            ctl->t_sum_entropy = sh;
            d.mH[n] = cand_H;
            d.mLabel[n] = lab;
            d.mPos[n] = p;
        }
        __syncthreads();
        double sm2;
        const double hm = block_entropy_div(d.Stmp, double(n + 1), d.B, scratch, &sm2);
        if (tid == 0) {
            ctl->t_total_jsd = hm - ctl->t_sum_entropy / double(n + 1);
            ctl->ev_kind = 2;
            ctl->ev_n = n + 1;
            if (sum_risky(sm2, d.B) || !(hm == hm)) ctl->ev_risky = 1;
        }
    }
}

// Initial set: S = sum of the member rows in member order, sumH likewise,
// total_jsd (SummedRecords::new, records.rs:36-47).  Members were written by
// seed_kernel.  Two launches: the sum bin by bin over the whole grid (every bin's adds in member order; as
// part of the one-block kernel below it was 0.5 ms at 4^7 bins and 100 members -- one CU pulling 12.8 MB),
// then one block for the entropy of the sum, whose order of additions is the block's.
constexpr int RSUM_THREADS = 256;
__global__ __launch_bounds__(RSUM_THREADS) void rebuild_sum_kernel(SelDev d) {
    __shared__ uint32_t s_ord[RSUM_THREADS];
    const uint32_t n = d.ctl->size;
    const uint64_t i = uint64_t(blockIdx.x) * RSUM_THREADS + threadIdx.x;
    double acc = 0.0;
    for (uint32_t base = 0; base < n; base += RSUM_THREADS) {
        const uint32_t cnt = n - base < RSUM_THREADS ? n - base : RSUM_THREADS;
        __syncthreads();
        if (threadIdx.x < cnt) s_ord[threadIdx.x] = d.ord[base + threadIdx.x];
        __syncthreads();
        if (i < d.B) {
            uint32_t r = 0;
            for (; r + 8 <= cnt; r += 8) {  // eight rows requested together, added in member order
                double v[8];
#pragma unroll
                for (int q = 0; q < 8; q++) v[q] = d.M[uint64_t(s_ord[r + q]) * d.B + i];
#pragma unroll
                for (int q = 0; q < 8; q++) acc += v[q];
            }
            for (; r < cnt; r++) acc += d.M[uint64_t(s_ord[r]) * d.B + i];
        }
    }
    if (i < d.B) d.S[i] = acc;
}

__global__ __launch_bounds__(WIDE_THREADS) void rebuild_kernel(SelDev d) {
    __shared__ double scratch[48];
    // the members' entropies go through LDS first: read straight from global memory thread 0's whole
    // entropy sum is a chain of dependent round trips
    __shared__ double s_mh[WIDE_THREADS];
    SelCtl *ctl = d.ctl;
    const uint32_t n = ctl->size;
    double sh = 0.0;  // (thread 0's, in member order)
    for (uint32_t base = 0; base < n; base += WIDE_THREADS) {
        const uint32_t cnt = n - base < WIDE_THREADS ? n - base : WIDE_THREADS;
        __syncthreads();
        if (threadIdx.x < cnt) s_mh[threadIdx.x] = d.mH[d.ord[base + threadIdx.x]];
        __syncthreads();
        if (threadIdx.x == 0)
            for (uint32_t r = 0; r < cnt; r++) sh += s_mh[r];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        ctl->sum_entropy = sh;
        ctl->s_is_resum = 1;
    }
    __syncthreads();
    double sm;
    const double hm = block_entropy_div(d.S, double(n), d.B, scratch, &sm);
    if (threadIdx.x == 0) {
        ctl->total_jsd = hm - ctl->sum_entropy / double(n);
        ctl->ev_kind = 1;
        ctl->ev_n = n;
        if (sum_risky(sm, d.B) || !(hm == hm)) ctl->ev_risky = 1;
    }
}

// member j <- matrix row of stream position seed_pos[j]
template <typename T>
__global__ __launch_bounds__(LOO_THREADS) void seed_kernel(SelDev d, const T *__restrict__ mat,
                                                         const uint64_t *__restrict__ seed_pos) {
    const uint32_t j = blockIdx.x;
    const uint64_t p = seed_pos[j];
    const uint32_t row = d.order ? d.order[p] : uint32_t(p);
    const uint32_t lab = d.labels ? d.labels[p] : row;
    const double tot = double(d.totals[row]);
    const T *rp = mat + uint64_t(row) * d.B;
    double *mrow = d.M + uint64_t(j) * d.B;
    for (uint64_t i = threadIdx.x; i < d.B; i += LOO_THREADS) mrow[i] = cand_freq(rp, i, tot);
    if (threadIdx.x == 0) {
        d.ord[j] = j;
        d.mH[j] = d.rowH[row];
        d.mLabel[j] = lab;
        d.mPos[j] = p;
        if (lab < d.nlabels) d.inset[lab] = 1;
    }
}

// Leave-one-out pass (get_lowest_record_index, records.rs:220-252): block r
// computes delta_jsd of the r-th member of the (possibly tentative) set.
__device__ void loo_body(const SelDev &d, uint32_t r, double *scratch) {
    const SelCtl *ctl = d.ctl;
    const uint32_t kind = ctl->ev_kind;
    if (kind == 0 || ctl->status != SEL_RUN) return;
    const uint32_t n = ctl->ev_n;
    if (r >= n) return;
    const bool tent = kind == 2;
    const uint32_t slot = (tent && r == n - 1) ? n - 1 : d.ord[r];
    const double *Sv = tent ? d.Stmp : d.S;
    const double sumH = tent ? ctl->t_sum_entropy : ctl->sum_entropy;
    const double tj = tent ? ctl->t_total_jsd : ctl->total_jsd;
    const double div = double(n) - 1.0;
    const double *mrow = d.M + uint64_t(slot) * d.B;
    Ent e;
    for (uint64_t i = threadIdx.x; i < d.B; i += blockDim.x) {
        double v = (Sv[i] - mrow[i]) / div;  // updated_mean_freqs, records.rs:276-286
        if (v <= DVS_EPS) v = 0.0;
        e.add(v);
    }
    double h = e.h, mnz = 0.0, sm = e.sum;
    block_red3(h, mnz, sm, scratch);
    if (threadIdx.x == 0) {
        const double mean_entropy = (sumH - d.mH[slot]) / div;
        const double jsd = h - mean_entropy;
        d.dtmp[r] = tj - jsd;
        d.dsum[r] = sm;
    }
}

// argmin / stats / commit-or-rollback / next scan state.  One block.
__device__ void finalize_body(SelDev &d, double *scratch, int &s_go) {
    SelCtl *ctl = d.ctl;
    const uint32_t kind = ctl->ev_kind;
    if (kind == 0 || ctl->status != SEL_RUN) return;
    const int tid = threadIdx.x;
    const uint32_t WIDE = blockDim.x;
    const uint32_t n = ctl->ev_n;

    // argmin with strict '<' from 1e6, earliest index on ties (records.rs:231,246-249);
    // NaN never wins a '<'.  Three fixed-tree reductions: min value, its first
    // index, and the runner-up (for the ambiguity check).
    bool risky = false;
    double best = 1e6;
    for (uint32_t r = tid; r < n; r += WIDE) {
        const double v = d.dtmp[r];
        if (sum_risky(d.dsum[r], d.B)) risky = true;
        if (v < best) best = v;
    }
    const double dmin = dvs_block_min(best, scratch);
    double fi = 4294967295.0;
    for (uint32_t r = tid; r < n; r += WIDE)
        if (dmin < 1e6 && d.dtmp[r] == dmin) fi = fmin(fi, double(r));
    const double dfirst = dvs_block_min(fi, scratch);
    uint32_t lowest = (dfirst < 4294967295.0) ? uint32_t(dfirst) : 0u;  // nothing below 1e6 -> 0
    double second = 1e6;
    for (uint32_t r = tid; r < n; r += WIDE) {
        const double v = d.dtmp[r];
        if (r != lowest && v < second) second = v;
    }
    const double dsecond = dvs_block_min(second, scratch);
    const int any_risky = __syncthreads_or(risky ? 1 : 0);

    // mean / std / cov of delta_jsd (records.rs:156-172)
    double acc = 0.0;
    for (uint32_t r = tid; r < n; r += WIDE) acc += d.dtmp[r];
    const double mean = dvs_block_sum(acc, scratch) / double(n);
    acc = 0.0;
    for (uint32_t r = tid; r < n; r += WIDE) {
        const double t = d.dtmp[r] - mean;
        acc += t * t;
    }
    const double sd = sqrt(dvs_block_sum(acc, scratch) / (double(n) - 1.0));
    const double cov = sd / mean;

    if (tid == 0) {
        int go = 1;  // 1 commit, 0 rollback, -1 stop for the arbiter
        const uint32_t forced = ctl->forced;
        const double band = sel_band(kind == 2 ? ctl->t_total_jsd + ctl->t_sum_entropy / double(n)
                                               : ctl->total_jsd + ctl->sum_entropy / double(n),
                                     d.B);
        const bool forced_fin = forced == FORCE_COMMIT || forced == FORCE_ROLLBACK;
        if (forced_fin) {
            go = forced == FORCE_COMMIT ? 1 : 0;
            if (ctl->forced_lowest != 0xFFFFFFFFu) lowest = ctl->forced_lowest;
            ctl->forced = FORCE_NONE;
            ctl->forced_lowest = 0xFFFFFFFFu;
        } else {
            bool tie = any_risky || ctl->ev_risky;
            if (n > 1 && dsecond - dmin <= band && dsecond < 1e6) tie = true;  // argmin ambiguous
            if (kind == 2) {
                const double a = ctl->stat == DVS_STAT_STDEV ? sd : cov;
                const double b = ctl->stat == DVS_STAT_STDEV ? ctl->std_delta : ctl->cov_delta;
                // every delta_jsd carries an error <= band, so std moves by <= ~band and
                // cov = std / mean by ~ band (1 + |cov|) / |mean|; NaN compares false
                const double mm = fmin(fabs(mean), fabs(ctl->mean_delta));
                const double sband = ctl->stat == DVS_STAT_STDEV
                                         ? 4.0 * band
                                         : 4.0 * band * (1.0 + fmax(fabs(a), fabs(b))) / fmax(mm, 1e-300);
                if (fabs(a - b) <= sband) tie = true;
                go = (a > b) ? 1 : 0;
            }
            if (tie) {
                ctl->status = SEL_ARBITER;
                ctl->arb_stage = ARB_FINALIZE;
                ctl->arb_pos = (kind == 2) ? d.mPos[n - 1] : ctl->cursor - 1;
                go = -1;
            }
        }
        if (go == 1) {
            if (kind == 2) {
                ctl->sum_entropy = ctl->t_sum_entropy;
                ctl->total_jsd = ctl->t_total_jsd;
                d.ord[n - 1] = n - 1;
                const uint32_t lab = d.mLabel[n - 1];
                if (lab < d.nlabels) d.inset[lab] = 1;
                ctl->size = n;
                ctl->n_accepts++;
                d.evlog_pos[ctl->n_logged] = d.mPos[n - 1];
                d.evlog_kind[ctl->n_logged] = 2;
                ctl->n_logged++;
            }
            ctl->lowest = lowest;
            ctl->mean_delta = mean;
            ctl->std_delta = sd;
            ctl->cov_delta = cov;
            ctl->band = band;
        }
        s_go = go;
    }
    __syncthreads();
    const int go = s_go;
    if (go < 0) return;
    if (go == 1) {
        if (kind == 2 && d.rowlog && ctl->n_logged) {  // the kept push's row
            const double *src = d.M + uint64_t(n - 1) * d.B;
            double *dst = d.rowlog + uint64_t((ctl->n_logged - 1) % d.rowlog_cap) * d.B;
            for (uint64_t i = tid; i < d.B; i += WIDE) dst[i] = src[i];
        }
        for (uint32_t r = tid; r < n; r += WIDE) d.mDelta[r] = d.dtmp[r];
        if (kind == 2)
            for (uint64_t i = tid; i < d.B; i += WIDE) d.S[i] = d.Stmp[i];
    }
    __syncthreads();
    // state for the next scan: b_i = (S_i - low_i) / size, thresholds
    const uint32_t sz = ctl->size;
    const uint32_t low_slot = d.ord[ctl->lowest];
    const double *low = d.M + uint64_t(low_slot) * d.B;
    const double dsize = double(sz);
    for (uint64_t i = tid; i < d.B; i += WIDE) d.base[i] = (d.S[i] - low[i]) / dsize;
    if (tid == 0) {
        ctl->he_base = ctl->sum_entropy - d.mH[low_slot];
        ctl->thr = ctl->total_jsd + DVS_EPS;
        ctl->ev_kind = 0;
        ctl->ev_risky = 0;
        ctl->mb_stuck = 0;
        ctl->event_pos = SEL_NONE;
        if (ctl->cursor >= ctl->npos) ctl->status = SEL_DONE;
        ctl_next_window(ctl);
    }
}


// ---- The stepwise (row-sharded) mode's FAST STEP (MODE_NMOST, no caller's labels): three launches per greedy step
// instead of five, and no single-block kernel on the chain.
//   fs_jobs_kernel   (n + 1) K workgroups: the gathered slots' earliest event is this step's event on every rank; each
//                    workgroup works out one part of one leave-one-out job of the set WITH that candidate in place of the
//                    lowest member (resolve_body's S update and loo_body's pass, bin for bin) -- the job of the whole
//                    set also carries the candidate's own score -- and leaves its sums in jobres.  Nothing else is
//                    written: whether the candidate is accepted at all is decided one launch later.
//   fs_step_kernel   every workgroup takes the SAME decisions from the same sums (resolve's candidate test, finalize's
//                    argmin, bands and statistics: identical code, identical bits), builds the next scan vector
//                    (S' - lowest') / n in its own LDS and scans this rank's rows from the event + 1 to the FIRST local
//                    event wherever it is -- no windows: a step is an event, and the kernel's end is the rendezvous.
//                    The last workgroup scans nothing: it writes the state resolve_kernel / finalize_kernel would have
//                    left (the kernels the arbiter re-enters with, get_members, the next step's jobs read it), and only
//                    after every other workgroup has said -- one relaxed add to a monotonic counter, behind the loads
//                    whose values it has consumed -- that it has read what it needs of the old one.
//   pack_event_kernel  as before: the first local event into this rank's slot.
// A decision inside a rounding band stops the step exactly where the old kernels would have stopped (same status,
// stage, pending candidate, half-applied state): dvs_select_step_poll arbitrates and re-enters THEM; the fast step
// resumes with a scan of its own.  Per step at world 1, n = 10, 4^6 bins: ~20 us against ~64.
constexpr int FS_JOB_THREADS = 256;
constexpr int FS_THREADS = 512;
constexpr uint32_t FS_MAXJOBS = 2048;
constexpr uint32_t FS_HIST = 512;  // words of the status history (a 4 KiB pinned block)
constexpr uint32_t FS_RES = 8;  // doubles per job: h, sum, min; the whole set's job: + the score's h, sum, min, clamp flag
struct FsLine {  // a polled word on a line of its own
    unsigned long long v;
    unsigned long long pad[31];
};
struct FsSync {
    // workgroups that have read the old state, all applied steps of the selection so far (monotonic); group g = blockIdx % 8
    // adds to arrived[g] -- 512 additions to one word are performed one after the other at the memory side, ~5 us of them
    FsLine arrived[8];
    unsigned int steps;  // applied steps so far (written by the state-writing workgroup at the end of one, read at the start of the next)
    unsigned int pad1[63];
    // copies of the scan's event word, one per group: 4096 waves looking at ONE word before every row queue up at the
    // memory side (the persistent engine's lesson, NOTES_r01_r04.md 4.2); the finder posts to all eight and to ctl->event_pos
    FsLine hint[8];
    // workgroups of the running launch that are through (the last one packs this rank's slot for the next exchange, and
    // clears the word)
    FsLine finished;
    unsigned long long launches;  // fs_step_kernel launches with apply = 1 so far (workgroup 0 counts; the word of launch L in the host's history is (L << 8) | status)
};

__device__ __forceinline__ bool fs_applies(const SelCtl *ctl, const SelDev &d) {
    return ctl->status == SEL_RUN && ctl->ev_kind == 0 && ctl->forced == FORCE_NONE && ctl->mode == DVS_MODE_NMOST &&
           d.labels == nullptr && ctl->size >= 2;
}
// the earliest event among the gathered slots ([pos, H, row] per rank; pos < 0: none) -- every thread, same answer
__device__ __forceinline__ unsigned long long fs_pick(const SelDev &d, const double **slot_out) {
    const uint64_t stride = d.B + 2;
    int best = -1;
    double bp = 0.0;
    for (uint32_t r = 0; r < d.gather_world; r++) {
        const double q = d.gather_all[uint64_t(r) * stride];
        if (q >= 0.0 && (best < 0 || q < bp)) {
            best = int(r);
            bp = q;
        }
    }
    *slot_out = d.gather_all + uint64_t(best < 0 ? 0 : best) * stride;
    return best < 0 ? SEL_NONE : (unsigned long long)bp;
}

__global__ __launch_bounds__(FS_JOB_THREADS) void fs_jobs_kernel(SelDev d, double *__restrict__ jobres, uint32_t K, FsSync *sync) {
    __shared__ double scratch[48];
    SelCtl *ctl = d.ctl;
    if (!fs_applies(ctl, d)) return;
    const double *slot;
    const unsigned long long p = fs_pick(d, &slot);
    // (the event word and its copies are the NEXT scan's from here on: the step kernel behind this launch posts into them)
    if (blockIdx.x == 0 && threadIdx.x == 0) ctl->event_pos = SEL_NONE;
    if (blockIdx.x == 0 && threadIdx.x < 8) sync->hint[threadIdx.x].v = SEL_NONE;
    if (p == SEL_NONE) return;
    const uint32_t n = ctl->size, li = ctl->lowest;
    const uint32_t job = blockIdx.x, r = job / K, part = job % K;
    if (r > n) return;
    const double *low = d.M + uint64_t(d.ord[li]) * d.B;
    const double *cand = slot + 2;
    const double dn = double(n), div = dn - 1.0;
    // member r of the NEW order (the lowest removed, the candidate appended): an old member, or the candidate
    const double *mrow = (r + 1 < n) ? d.M + uint64_t(d.ord[r < li ? r : r + 1]) * d.B : cand;
    Ent e, es;
    int cl = 0;
    for (uint64_t i = uint64_t(part) * FS_JOB_THREADS + threadIdx.x; i < d.B; i += uint64_t(K) * FS_JOB_THREADS) {
        const double v = d.S[i] - low[i];
        const double f = cand[i];
        double vc = v;
        if (vc <= DVS_EPS) vc = 0.0;  // drop_lowest's clamp (records.rs:100-105)
        const double sp = vc + f;     // S' of the bin, as resolve_body leaves it
        if (r == n) {
            cl |= (v <= DVS_EPS && v != 0.0) ? 1 : 0;
            es.add((v + f) / dn);  // the candidate's score (block_delta_jsd: no clamp, records.rs:78-81)
            e.add(sp / dn);        // H(S' / n) (block_entropy_div)
        } else {
            double u = (sp - mrow[i]) / div;  // updated_mean_freqs, records.rs:276-286 (loo_body)
            if (u <= DVS_EPS) u = 0.0;
            e.add(u);
        }
    }
    double h = e.h, mn = e.mn, sm = e.sum;
    block_red3(h, mn, sm, scratch);
    double hs = es.h, mns = es.mn, sms = es.sum;
    int cl_any = 0;
    if (r == n) {  // (block-uniform)
        block_red3(hs, mns, sms, scratch);
        cl_any = __syncthreads_or(cl);
    }
    if (threadIdx.x == 0) {
        double *o = jobres + uint64_t(job) * FS_RES;
        o[0] = h;
        o[1] = sm;
        o[2] = mn;
        o[3] = hs;
        o[4] = sms;
        o[5] = mns;
        o[6] = cl_any ? 1.0 : 0.0;
    }
}

// ---- the scan of the fast step.  A wave a row, as everywhere; what differs from scan_rows_general:
//  * count rows of 256-bin multiples take the all-f32 COARSE tier first (select_dev.h: a row farther than its proven band from
//    the threshold is decided there, ~5 vector instructions a bin instead of ~17); what it cannot decide, and every other row
//    form, takes the f32-log FAST tier and, inside that one's band, the f64 one;
//  * the event word is polled through eight copies (FsSync::hint), and a finder posts to all of them;
//  * for rows of FS_BURST bins (k = 6 DNA) the WHOLE row is requested at once (16 wave instructions of 512 B / 1 KiB in flight,
//    one round trip a row instead of four), and a wave's FIRST row is requested at the top of the kernel, before the step's
//    decisions are known: the cursor those leave is the event's position + 1 whatever they are, so the matrix streams in
//    while the decisions and the scan vector are made.
constexpr uint64_t FS_BURST = 4096;
// (raw[j] holds chunk (j + rot) % 16 of the row: waves that start together -- all of them, at the top of every step -- do not
// walk their rows in step)
template <typename T>
__device__ __forceinline__ void fs_row_issue(const SelDev &d, const T *__restrict__ mat, uint64_t p, uint32_t lane, uint32_t rot,
                                             Raw4<T> (&raw)[16], uint32_t &row, uint32_t &tot, double &hrow) {
    row = d.order ? d.order[p] : uint32_t(p);
    tot = 0;
    hrow = 0.0;
    if (row == DVS_ROW_REMOTE) return;  // another rank scores this position
    tot = d.totals[row];
    hrow = d.rowH[row];
    const T *rp = mat + uint64_t(row) * FS_BURST;
#pragma unroll
    for (int j = 0; j < 16; j++) raw[j].load(rp + ((j + rot) & 15u) * 256 + lane * 4);
}
struct FsThr {
    double c_lo, c_hi;  // COARSE: sure reject below, sure event above
};
// true: the row's score may clear the threshold (an event, for the next step's exact evaluation to decide)
template <typename T>
__device__ __forceinline__ bool fs_row_score(const Raw4<T> (&raw)[16], uint32_t row, uint32_t tot, double hrow,
                                             const T *__restrict__ mat, const double *sb, const float *slf, const ScanState &st,
                                             const FsThr &ft, uint32_t lane, uint32_t rot, uint32_t &nprecise) {
    const double rinv = 1.0 / (double(tot) * st.dsize);
    const double mean_entropy = (st.he_base + hrow) / st.dsize;
    const float rtn = float(rinv);
    const dvs_f2 r2 = {rtn, rtn};
    double c0 = 0.0, c1 = 0.0;
#pragma unroll
    for (int j = 0; j < 16; j += 2) {
        const uint32_t i0 = ((j + rot) & 15u) * 256 + lane * 4, i1 = ((j + 1 + rot) & 15u) * 256 + lane * 4;
        c0 += double(coarse4(raw[j].c, *reinterpret_cast<const float4 *>(slf + i0), r2));
        c1 += double(coarse4(raw[j + 1].c, *reinterpret_cast<const float4 *>(slf + i1), r2));
    }
    const double jf0 = -dvs_wave_sum(c0 + c1) - mean_entropy;
    if (!(jf0 > ft.c_lo)) return false;  // (NaN: a negative bin, rejected as the reference does)
    if (jf0 > ft.c_hi) return true;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0, xmin = 0.0;
#pragma unroll
    for (int j = 0; j < 16; j++) {
        const uint32_t i = ((j + rot) & 15u) * 256 + lane * 4;
        double v0, v1, v2, v3;
        raw[j].get(v0, v1, v2, v3);
        const double2 b01 = *reinterpret_cast<const double2 *>(sb + i);
        const double2 b23 = *reinterpret_cast<const double2 *>(sb + i + 2);
        fast4(b01, b23, v0, v1, v2, v3, rinv, a0, a1, a2, a3, xmin);
    }
    const double hf = dvs_wave_sum((a0 + a1) + (a2 + a3));
    const double mn = dvs_wave_min(xmin);
    if (mn < 0.0) return false;
    const double jf = hf - mean_entropy;
    if (!(jf > st.thr_fast)) return false;
    if (jf > st.thr_sure) return true;
    nprecise++;
    return precise_row(mat + uint64_t(row) * FS_BURST, sb, FS_BURST, rinv, mean_entropy, st.thr_lo, lane);
}

__device__ __forceinline__ void fs_post(SelCtl *ctl, FsSync *sync, uint32_t lane, uint64_t p) {
    if (lane < 8) atomicMin(&sync->hint[lane].v, (unsigned long long)p);
    if (lane == 8) atomicMin(&ctl->event_pos, (unsigned long long)p);
}

// any other row form
template <typename T>
__device__ __forceinline__ void fs_scan_rows(SelCtl *ctl, FsSync *sync, const SelDev &d, const T *__restrict__ mat,
                                             const double *sb, const float *slf, uint64_t B, const ScanState &st,
                                             double thr, double band, uint64_t first, uint64_t stride, uint32_t lane,
                                             uint32_t &nread, uint32_t &nprecise) {
    const unsigned long long *hintp = &sync->hint[blockIdx.x & 7u].v;
    constexpr bool COUNTS = sizeof(T) <= 4;
    const bool coarse = COUNTS && (B & 1023) == 0;
    const double cband = coarse_band(B);
    const double thr_c_lo = thr - band - cband, thr_c_hi = thr + band + cband;
    for (uint64_t r = first; r < st.nrows; r += stride) {
        const uint64_t p = st.cursor + r;
        const unsigned long long ev = __hip_atomic_load(hintp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t row = d.order ? d.order[p] : uint32_t(p);
        if (ev < p) break;                    // an earlier event ends the scan: later rows are void
        if (row == DVS_ROW_REMOTE) continue;  // another rank scores this position
        const uint32_t tot = d.totals[row];
        if (tot == 0) continue;
        const T *rp = mat + uint64_t(row) * B;
        const double rinv = 1.0 / (double(tot) * st.dsize);
        const double mean_entropy = (st.he_base + d.rowH[row]) / st.dsize;
        nread++;
        if constexpr (COUNTS) {
            if (coarse) {
                const float rtn = float(rinv);
                const dvs_f2 r2 = {rtn, rtn};
                double c0 = 0.0, c1 = 0.0;
                for (uint64_t i0 = 0; i0 < B; i0 += 1024) {  // four chunks requested together
                    Raw4<T> raw[4];
#pragma unroll
                    for (int j = 0; j < 4; j++) raw[j].load(rp + i0 + uint64_t(j) * 256 + lane * 4);
#pragma unroll
                    for (int j = 0; j < 4; j += 2) {
                        const uint64_t i = i0 + uint64_t(j) * 256 + lane * 4;
                        c0 += double(coarse4(raw[j].c, *reinterpret_cast<const float4 *>(slf + i), r2));
                        c1 += double(coarse4(raw[j + 1].c, *reinterpret_cast<const float4 *>(slf + i + 256), r2));
                    }
                }
                const double jf0 = -dvs_wave_sum(c0 + c1) - mean_entropy;
                if (!(jf0 > thr_c_lo)) continue;  // (NaN: a negative bin, rejected as the reference does)
                if (jf0 > thr_c_hi) {
                    fs_post(ctl, sync, lane, p);
                    continue;
                }
            }
        }
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0, xmin = 0.0;
        if ((B & 255) == 0) {
            for (uint64_t i0 = 0; i0 < B; i0 += 256) {
                const uint64_t i = i0 + lane * 4;
                double v0, v1, v2, v3;
                load4(rp, i, v0, v1, v2, v3);
                const double2 b01 = *reinterpret_cast<const double2 *>(sb + i);
                const double2 b23 = *reinterpret_cast<const double2 *>(sb + i + 2);
                fast4(b01, b23, v0, v1, v2, v3, rinv, a0, a1, a2, a3, xmin);
            }
        } else {
            for (uint64_t i = lane; i < B; i += 64) {
                const double x = fma(row_value(rp, i), rinv, sb[i]);
                a0 += fast_neg_xlog2x(x);
                xmin = fmin(xmin, x);
            }
        }
        const double hf = dvs_wave_sum((a0 + a1) + (a2 + a3));
        const double mn = dvs_wave_min(xmin);
        if (mn < 0.0) continue;
        const double jf = hf - mean_entropy;
        if (!(jf > st.thr_fast)) continue;
        bool hit = jf > st.thr_sure;
        if (!hit) {
            nprecise++;
            hit = precise_row(rp, sb, B, rinv, mean_entropy, st.thr_lo, lane);
        }
        if (hit) fs_post(ctl, sync, lane, p);
    }
}

// LDS (dynamic): [B f64: the scan vector | the state writer's new S][B f64: the scan vector in f32 | the state writer's
// copy of the candidate][s_h cap + 1][s_s cap + 1][s_dl cap + 1][s_job 7 x jobs f64 staging]
constexpr int FS_HOLD = 8;  // bins a thread holds in registers across the decisions (FS_THREADS x FS_HOLD = 4096 of them)
template <typename T>
__device__ __forceinline__ void fs_step_body(const SelDev &d, const T *__restrict__ mat, const double *__restrict__ jobres,
                                             uint32_t K, int apply, FsSync *sync, uint32_t cap, unsigned char *fs_smem) {
    __shared__ double s_whole[16];
    constexpr bool COUNTS = sizeof(T) <= 4;
    const uint64_t B = d.B;
    const uint64_t Bp = (B + 1) & ~1ull;
    double *sb = reinterpret_cast<double *>(fs_smem);
    double *cf = sb + Bp;                        // (the state writer's)
    float *slf = reinterpret_cast<float *>(cf);  // (the scanning workgroups')
    double *s_h = cf + Bp;
    double *s_s = s_h + cap + 1;
    double *s_dl = s_s + cap + 1;
    double *s_job = s_dl + cap + 1;  // [jobs][7]
    SelCtl *ctl = d.ctl;
    const int tid = threadIdx.x;
    const uint32_t lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // (a wave's row, its address and its chunk phase: scalars)
    const uint32_t nscan = gridDim.x - 1;  // scanning workgroups 1 .. nscan; workgroup 0 writes the state (it is resident first)
    const bool lead = blockIdx.x == 0;
    if (!fs_applies(ctl, d)) return;  // (stopped, done, or a step the old kernels own: a no-op on every rank alike)
    const unsigned int steps_done = sync->steps;  // (launches that returned above never counted: the counters stay in step)
#ifdef DVS_FS_TRACE
    unsigned long long *trc = reinterpret_cast<unsigned long long *>(const_cast<double *>(jobres) + uint64_t(FS_MAXJOBS) * FS_RES) + uint64_t(blockIdx.x) * 8;
    const bool trace = apply && steps_done == 20 && tid == 0 && blockIdx.x < 1024;
    if (trace) trc[0] = __builtin_amdgcn_s_memrealtime();
#define FS_T(k) do { if (trace) trc[k] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define FS_T(k) do { } while (0)
#endif
    // ---- the old state, read before anything is written
    const uint32_t n = ctl->size, li = ctl->lowest;
    const uint64_t npos = ctl->npos;
    const double sumH = ctl->sum_entropy, thr = ctl->thr, band = ctl->band;
    const uint32_t low_slot = d.ord[li];
    const double dn = double(n), div = dn - 1.0;
    const double *slot = nullptr;
    unsigned long long p = SEL_NONE;
    if (apply) p = fs_pick(d, &slot);
    uint64_t cursor = ctl->cursor;
    if (apply && p == SEL_NONE) {  // no rank found an event behind the cursor: the stream is through
        if (lead && tid == 0) {
            ctl->n_windows++;
            ctl->cursor = npos;
            ctl->status = SEL_DONE;
        }
        return;
    }
    if (p != SEL_NONE) cursor = p + 1;  // (where every outcome that scans on leaves it)
    const double *low = d.M + uint64_t(low_slot) * B;
    const double *cand = p != SEL_NONE ? slot + 2 : nullptr;
    const uint64_t wpb = FS_THREADS / 64;
    const uint64_t first = uint64_t(blockIdx.x - 1) * wpb + wave, stride = uint64_t(nscan) * wpb;
    const uint32_t rot = uint32_t(first >> 2) & 15u;  // (with the row's index mod 4: 64 consecutive waves, 64 different phases)
    // ---- requested now, consumed after the decisions: this wave's first row
    const bool burst = COUNTS && B == FS_BURST;
    Raw4<T> raw0[16];
    uint32_t row0 = DVS_ROW_REMOTE, tot0 = 0;
    double hrow0 = 0.0;
    bool pre = false;
    if (!lead) {
        if constexpr (COUNTS) {
            if (burst && cursor + first < npos) {
                fs_row_issue<T>(d, mat, cursor + first, lane, rot, raw0, row0, tot0, hrow0);
                pre = true;
            }
        }
    }
    // what the decisions below leave: 0 nothing applied (no event: scan only), 1 rejected, 2 accepted,
    // 3 stop at the candidate test (ARB_RESOLVE), 4 stop at the argmin (ARB_FINALIZE)
    int outcome = 0;
    double jsd = 0.0, cand_H = 0.0, sh_new = sumH, total_new = ctl->total_jsd, band_new = band, thr_new = thr;
    double mean_d = ctl->mean_delta, sd_d = ctl->std_delta;
    uint32_t lowest_new = li;
    bool ev_risky = false;
    if (p != SEL_NONE) {
        cand_H = slot[1];
        // the jobs' sums: every job's seven words requested at once (a thread a job), then a member's K parts added in
        // part order by one thread -- as a chain of dependent loads per member this phase was 10 us of every step
        const uint32_t jobs = (n + 1) * K;
        for (uint32_t j = tid; j < jobs; j += FS_THREADS) {
            const double *o = jobres + uint64_t(j) * FS_RES;
            double v[7];
#pragma unroll
            for (int q = 0; q < 7; q++) v[q] = o[q];
#pragma unroll
            for (int q = 0; q < 7; q++) s_job[uint64_t(j) * 7 + q] = v[q];
        }
        // the entropies of the new order's members (member r < n - 1: an old one; n - 1: the candidate)
        for (uint32_t r = tid; r < n; r += FS_THREADS) s_dl[r] = (r + 1 < n) ? d.mH[d.ord[r < li ? r : r + 1]] : cand_H;
        __syncthreads();
        for (uint32_t r = tid; r <= n; r += FS_THREADS) {
            double h = 0.0, sm = 0.0, hs = 0.0, sms = 0.0, mns = 0.0, cl = 0.0;
            for (uint32_t q = 0; q < K; q++) {
                const double *o = s_job + (uint64_t(r) * K + q) * 7;
                h += o[0];
                sm += o[1];
                hs += o[3];
                sms += o[4];
                mns = fmin(mns, o[5]);
                cl += o[6];
            }
            s_h[r] = h;
            s_s[r] = sm;
            if (r == n) {
                s_whole[0] = hs;
                s_whole[1] = sms;
                s_whole[2] = mns;
                s_whole[3] = cl;
            }
        }
        __syncthreads();
        const double hs = s_whole[0], sms = s_whole[1], mns = s_whole[2];
        const bool clamped = s_whole[3] != 0.0;
        const double mH_low = d.mH[low_slot];
        jsd = (mns < 0.0) ? NAN : hs - (sumH - mH_low + cand_H) / dn;  // block_delta_jsd
        if (sum_risky(sms, B) || fabs(jsd - thr) <= band) {
            outcome = 3;
        } else if (!(jsd > thr)) {  // rejected (NaN included, records.rs:91)
            outcome = 1;
        } else {
            outcome = 2;
            sh_new = sumH - mH_low;  // (resolve_body's order of additions)
            sh_new += cand_H;
            const double hm = clamped ? s_h[n] : hs;
            const double sm2 = clamped ? s_s[n] : sms;
            total_new = hm - sh_new / dn;
            ev_risky = sum_risky(sm2, B) || !(hm == hm);
            // finalize_body's delta_jsd of every member of the new order, argmin (strict '<' from 1e6, earliest index),
            // runner-up and statistics, by ONE wave: lane l owns members l, l + 64, ... (the same code in every
            // workgroup: the same bits)
            if (wave == 0) {
                for (uint32_t r = lane; r < n; r += 64) s_dl[r] = total_new - (s_h[r] - (sh_new - s_dl[r]) / div);  // loo_body
                bool risky = false;
                double best = 1e6, acc = 0.0;
                for (uint32_t r = lane; r < n; r += 64) {
                    if (sum_risky(s_s[r], B)) risky = true;
                    acc += s_dl[r];
                    if (s_dl[r] < best) best = s_dl[r];
                }
                const double dmin = dvs_wave_min(best);
                double fi = 4294967295.0;
                for (uint32_t r = lane; r < n; r += 64)
                    if (dmin < 1e6 && s_dl[r] == dmin) fi = fmin(fi, double(r));
                const double dfirst = dvs_wave_min(fi);
                const uint32_t lw = (dfirst < 4294967295.0) ? uint32_t(dfirst) : 0u;
                const double mu = dvs_wave_sum(acc) / dn;
                double second = 1e6, tv = 0.0;
                for (uint32_t r = lane; r < n; r += 64) {
                    if (r != lw && s_dl[r] < second) second = s_dl[r];
                    const double t = s_dl[r] - mu;
                    tv += t * t;
                }
                second = dvs_wave_min(second);
                const double var = dvs_wave_sum(tv);
                const unsigned long long anyr = __ballot(risky);
                if (lane == 0) {
                    s_whole[8] = dmin;
                    s_whole[9] = double(lw);
                    s_whole[10] = second;
                    s_whole[11] = mu;
                    s_whole[12] = sqrt(var / div);
                    s_whole[13] = anyr ? 1.0 : 0.0;
                }
            }
            __syncthreads();
            const double dmin = s_whole[8], dsecond = s_whole[10];
            lowest_new = uint32_t(s_whole[9]);
            mean_d = s_whole[11];
            sd_d = s_whole[12];
            const bool any_risky = s_whole[13] != 0.0;
            band_new = sel_band(total_new + sh_new / dn, B);
            thr_new = total_new + DVS_EPS;
            if (any_risky || ev_risky || (n > 1 && dsecond - dmin <= band_new && dsecond < 1e6)) outcome = 4;
        }
    }
    FS_T(1);
    const bool stop = outcome == 3 || outcome == 4 || cursor >= npos;
    // the new lowest member's row and entropy, in terms of the OLD order (the candidate, if it is the new one)
    const uint32_t nl_old = lowest_new < li ? lowest_new : lowest_new + 1;
    const bool nl_is_cand = !(lowest_new + 1 < n);
    if (!lead) {
        // ---- the scan vector of the state the decisions leave: (S' - lowest') / n, f64 and f32, in this workgroup's LDS
        double he_base = ctl->he_base;
        if (!stop) {
            // (a thread's FS_HOLD bins of every operand requested together: one round trip)
            if (outcome == 2) {
                const double *nl = nl_is_cand ? cand : d.M + uint64_t(d.ord[nl_old]) * B;
                const double mH_nl = nl_is_cand ? cand_H : d.mH[d.ord[nl_old]];
                for (uint64_t i0 = tid; i0 < B; i0 += uint64_t(FS_HOLD) * FS_THREADS) {
                    double h_s[FS_HOLD], h_lo[FS_HOLD], h_cd[FS_HOLD], h_nl[FS_HOLD];
#pragma unroll
                    for (int k = 0; k < FS_HOLD; k++) {
                        const uint64_t i = i0 + uint64_t(k) * FS_THREADS;
                        const bool in = i < B;
                        h_s[k] = in ? d.S[i] : 0.0;
                        h_lo[k] = in ? low[i] : 0.0;
                        h_cd[k] = in ? cand[i] : 0.0;
                        h_nl[k] = in ? nl[i] : 0.0;
                    }
#pragma unroll
                    for (int k = 0; k < FS_HOLD; k++) {
                        const uint64_t i = i0 + uint64_t(k) * FS_THREADS;
                        double v = h_s[k] - h_lo[k];
                        if (v <= DVS_EPS) v = 0.0;
                        const double b = ((v + h_cd[k]) - h_nl[k]) / dn;
                        if (i < B) {
                            sb[i] = b;
                            slf[i] = b == 0.0 ? 1e-30f : float(b);
                        }
                    }
                }
                he_base = sh_new - mH_nl;
            } else {
                for (uint64_t i0 = tid; i0 < B; i0 += uint64_t(FS_HOLD) * FS_THREADS) {
                    double h_s[FS_HOLD], h_lo[FS_HOLD];
#pragma unroll
                    for (int k = 0; k < FS_HOLD; k++) {
                        const uint64_t i = i0 + uint64_t(k) * FS_THREADS;
                        const bool in = i < B;
                        h_s[k] = in ? d.S[i] : 0.0;
                        h_lo[k] = in ? low[i] : 0.0;
                    }
#pragma unroll
                    for (int k = 0; k < FS_HOLD; k++) {
                        const uint64_t i = i0 + uint64_t(k) * FS_THREADS;
                        const double b = (h_s[k] - h_lo[k]) / dn;
                        if (i < B) {
                            sb[i] = b;
                            slf[i] = b == 0.0 ? 1e-30f : float(b);
                        }
                    }
                }
                he_base = sumH - d.mH[low_slot];
            }
        }
        __syncthreads();
        FS_T(2);
        // ---- arrival: this workgroup has consumed what it reads of the old state
        if (tid == 0 && apply)
            __hip_atomic_fetch_add(&sync->arrived[blockIdx.x & 7u].v, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (stop) return;
        // ---- this rank's rows from the cursor to the first local event (a wave a row; the event word's copies end the scan)
        ScanState st;
        st.cursor = cursor;
        st.nrows = npos - cursor;
        st.thr_lo = thr_new - band_new;
        st.thr_fast = st.thr_lo - FAST_BAND;
        st.thr_sure = thr_new + band_new + FAST_BAND;
        st.he_base = he_base;
        st.dsize = dn;
        uint32_t nread = 0, nprecise = 0;
        bool done = false;
        if constexpr (COUNTS) {
            if (burst) {
                // rows first, first + stride, ...: the first one's data are here already
                const unsigned long long *hintp = &sync->hint[blockIdx.x & 7u].v;
                const double cband = coarse_band(FS_BURST);
                FsThr ft;
                ft.c_lo = thr_new - band_new - cband;
                ft.c_hi = thr_new + band_new + cband;
                uint64_t r = first;
                if (pre) {
#ifdef DVS_FS_TRACE
                    if (trace) {
                        trc[5] = __builtin_amdgcn_s_memrealtime();
                        asm volatile("s_waitcnt vmcnt(0)");
                        trc[3] = __builtin_amdgcn_s_memrealtime();
                    }
#endif
                    if (tot0 != 0) {  // (0: another rank's, or "No valid k-mers", records.rs:332-335)
                        nread++;
                        if (fs_row_score<T>(raw0, row0, tot0, hrow0, mat, sb, slf, st, ft, lane, rot, nprecise)) fs_post(ctl, sync, lane, cursor + r);
                    }
                    FS_T(4);
                    r += stride;
                }
                for (; r < st.nrows; r += stride) {
                    const uint64_t q = cursor + r;
                    if (__hip_atomic_load(hintp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < q) break;  // an earlier event ends the scan
                    Raw4<T> raw[16];
                    uint32_t row, tot;
                    double hrow;
                    fs_row_issue<T>(d, mat, q, lane, rot, raw, row, tot, hrow);
                    if (tot == 0) continue;
                    nread++;
                    if (fs_row_score<T>(raw, row, tot, hrow, mat, sb, slf, st, ft, lane, rot, nprecise)) fs_post(ctl, sync, lane, q);
                }
                done = true;
            }
        }
        if (!done) fs_scan_rows<T>(ctl, sync, d, mat, sb, slf, B, st, thr_new, band_new, first, stride, lane, nread, nprecise);
        FS_T(6);
#ifdef DVS_FS_TRACE
        if (trace) trc[7] = nread;
#endif
        // (a workgroup's count, not a wave's: two thousand additions to one word queue up for ~10 ns each at the memory side)
        unsigned int *s_cnt = reinterpret_cast<unsigned int *>(s_whole + 14);
        __syncthreads();
        if (tid < 2) s_cnt[tid] = 0;
        __syncthreads();
        if (lane == 0 && nread) {
            atomicAdd(&s_cnt[0], nread);
            if (nprecise) atomicAdd(&s_cnt[1], nprecise);
        }
        __syncthreads();
        if (tid == 0 && s_cnt[0]) {
            atomicAdd(&ctl->rows_scored, (unsigned long long)s_cnt[0]);
            if (s_cnt[1]) atomicAdd(&ctl->rows_rechecked, (unsigned long long)s_cnt[1]);
        }
        return;
    }
    // ================= the state writer (workgroup 0)
    if (!apply) return;  // (a scan-only launch: nothing to write)
    // ---- before the others have arrived: everything the new state needs of the old one, into LDS and registers
    const uint32_t row = d.order ? d.order[p] : uint32_t(p);
    const uint32_t old_lab = d.mLabel[low_slot];
    const uint32_t log_at = ctl->n_logged;
    // (the counters and the window's constants too: after the wait the lead thread only stores -- a load between two
    // stores to the control block waits for a round trip each, under the scan's traffic)
    const unsigned long long c_windows = ctl->n_windows, c_events = ctl->n_events, c_accepts = ctl->n_accepts;
    const uint32_t c_wmin = ctl->window_min, c_wmax = ctl->window_max;
    const double c_wscale = ctl->wscale;
    const uint32_t nl_slot = nl_is_cand ? low_slot : d.ord[nl_old];  // (the candidate takes the lowest member's slot)
    double he_base_new = 0.0;
    const double *nlrow = nl_is_cand ? cand : d.M + uint64_t(nl_slot) * B;  // (an old member's row: not the one written below)
    if (outcome == 2) he_base_new = sh_new - (nl_is_cand ? cand_H : d.mH[nl_slot]);
    // FS_THREADS x FS_HOLD bins (k = 6 DNA: all of them) go through registers -- the new S, the candidate, finalize's scan
    // vector -- so that what follows the wait is stores in a straight line: a wait for a load in there would wait for every
    // store in front of it (one counter), a round trip each under the scan's traffic.  Bins beyond go through LDS.
    const bool full = B == uint64_t(FS_HOLD) * FS_THREADS;
    double h_sn[FS_HOLD], h_cd[FS_HOLD], h_bn[FS_HOLD];
    if (full) {
        double h_s[FS_HOLD], h_lo[FS_HOLD], h_nl[FS_HOLD];
#pragma unroll
        for (int k = 0; k < FS_HOLD; k++) {
            const uint32_t i = uint32_t(tid) + uint32_t(k) * FS_THREADS;
            h_s[k] = d.S[i];
            h_lo[k] = low[i];
            h_cd[k] = cand[i];
            h_nl[k] = nlrow[i];
        }
#pragma unroll
        for (int k = 0; k < FS_HOLD; k++) {
            double v = h_s[k] - h_lo[k];
            if (v <= DVS_EPS) v = 0.0;
            h_sn[k] = v + h_cd[k];
            h_bn[k] = (h_sn[k] - h_nl[k]) / dn;
        }
    } else if (outcome == 2 || outcome == 4) {
        for (uint64_t i = tid; i < B; i += FS_THREADS) {
            double v = d.S[i] - low[i];
            if (v <= DVS_EPS) v = 0.0;
            sb[i] = v + cand[i];
            cf[i] = cand[i];
        }
    } else if (outcome == 3) {
        for (uint64_t i = tid; i < B; i += FS_THREADS) cf[i] = cand[i];
    }
    // Vec::remove(li) + push on the member order: the new order's slots, staged in s_h's place (the sums are in s_s / s_dl)
    uint32_t *s_ord = reinterpret_cast<uint32_t *>(s_h);
    if (outcome == 2 || outcome == 4)
        for (uint32_t r = tid; r < n; r += FS_THREADS) s_ord[r] = (r + 1 < n) ? d.ord[r < li ? r : r + 1] : low_slot;
    const uint32_t win_new = sel_next_window(p + 1, n, c_wmin, c_wmax, c_wscale);  // (ctl_next_window)
    const double cov_new = sd_d / mean_d;
    const uint32_t log_slot = log_at % d.rowlog_cap;
    // (every loaded word is made to BE in its register here: left to the compiler, the wait for a word first used behind the
    // stores below lands behind them, and one counter counts loads and stores)
    asm volatile("" ::"v"(row), "v"(old_lab), "v"(log_at), "v"(log_slot), "v"(win_new), "v"(c_windows), "v"(c_events), "v"(c_accepts),
                 "v"(he_base_new), "v"(cov_new), "v"(nl_slot));
    FS_T(2);
    if (tid == 0) s_whole[7] = 1.0;
    __syncthreads();
    if (wave == 0) {  // (lanes 0-7 read a copy each: eight loads in flight, not eight round trips a look)
        const unsigned long long target = (unsigned long long)nscan * (steps_done + 1u);
        uint32_t spins = 0;
        for (;;) {
            const unsigned long long mine = lane < 8 ? __hip_atomic_load(&sync->arrived[lane].v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
            unsigned long long got = mine;
#pragma unroll
            for (int o = 4; o > 0; o >>= 1) got += __shfl_xor(got, o, 64);
            got = __shfl(got, 0, 64);
            if (got >= target) break;
            if (++spins > (1u << 22)) {
                if (lane == 0) s_whole[7] = 0.0;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    __syncthreads();
    if (s_whole[7] == 0.0) {  // (never: every workgroup of the grid arrives; an error, not a hang)
        if (tid == 0) ctl->status = SEL_ERROR;
        return;
    }
    if (tid == 0) sync->steps = steps_done + 1u;
    FS_T(3);
    // ---- the state resolve_kernel / finalize_kernel would have left: stores only from here
    if (outcome == 3) {
        if (full) {
#pragma unroll
            for (int k = 0; k < FS_HOLD; k++) d.cand[uint32_t(tid) + uint32_t(k) * FS_THREADS] = h_cd[k];
        } else {
            for (uint64_t i = tid; i < B; i += FS_THREADS) d.cand[i] = cf[i];
        }
        if (tid == 0) {
            ctl->status = SEL_ARBITER;
            ctl->arb_stage = ARB_RESOLVE;
            ctl->arb_pos = p;
            ctl->arb_H = cand_H;
        }
    } else if (outcome == 1) {
        if (tid == 0) {
            ctl->n_windows = c_windows + 1;
            ctl->n_events = c_events + 1;
            ctl->cursor = p + 1;
            ctl->last_jsd = jsd;
            if (p + 1 >= npos) ctl->status = SEL_DONE;
        }
    } else {  // accepted: replace_lowest = drop_lowest (records.rs:94-109) + push (:120-147)
        double *mrow = d.M + uint64_t(low_slot) * B;
        double *logrow = d.rowlog + uint64_t(log_slot) * B;  // (the stepwise mode always keeps the row log)
        for (uint32_t r = tid; r < n; r += FS_THREADS) {  // (LDS reads: in front of the stores)
            d.ord[r] = s_ord[r];
            d.dtmp[r] = s_dl[r];
            d.dsum[r] = s_s[r];
            if (outcome == 2) d.mDelta[r] = s_dl[r];
        }
        FS_T(4);
        if (full) {
#pragma unroll
            for (int k = 0; k < FS_HOLD; k++) {
                const uint32_t i = uint32_t(tid) + uint32_t(k) * FS_THREADS;
                d.S[i] = h_sn[k];
                mrow[i] = h_cd[k];
                logrow[i] = h_cd[k];
            }
            if (outcome == 2) {  // finalize's scan vector (the multi-launch kernels and the persistent engine resume from it)
#pragma unroll
                for (int k = 0; k < FS_HOLD; k++) d.base[uint32_t(tid) + uint32_t(k) * FS_THREADS] = h_bn[k];
            }
        } else {
            for (uint64_t i = tid; i < B; i += FS_THREADS) {
                const double f = cf[i];
                d.S[i] = sb[i];
                mrow[i] = f;
                logrow[i] = f;
            }
            if (outcome == 2)
                for (uint64_t i = tid; i < B; i += FS_THREADS) d.base[i] = (sb[i] - nlrow[i]) / dn;
        }
        FS_T(6);
        if (tid == 0) {
            if (old_lab < d.nlabels) d.inset[old_lab] = 0;
            if (row < d.nlabels) d.inset[row] = 1;  // (no caller's labels: the label of a position is its row, or REMOTE)
            d.mH[low_slot] = cand_H;
            d.mLabel[low_slot] = row;
            d.mPos[low_slot] = p;
            d.evlog_pos[log_at] = p;
            d.evlog_kind[log_at] = 1;
            ctl->n_logged = log_at + 1;
            ctl->n_accepts = c_accepts + 1;
            ctl->n_windows = c_windows + 1;
            ctl->n_events = c_events + 1;
            ctl->cursor = p + 1;
            ctl->last_jsd = jsd;
            ctl->s_is_resum = 0;
            ctl->sum_entropy = sh_new;
            ctl->total_jsd = total_new;
            if (outcome == 4) {  // argmin (or a sum check) too close to call: finalize's stop
                ctl->ev_kind = 1;
                ctl->ev_n = n;
                ctl->ev_risky = ev_risky ? 1 : 0;
                ctl->status = SEL_ARBITER;
                ctl->arb_stage = ARB_FINALIZE;
                ctl->arb_pos = p;
            } else {
                ctl->lowest = lowest_new;
                ctl->mean_delta = mean_d;
                ctl->std_delta = sd_d;
                ctl->cov_delta = cov_new;
                ctl->band = band_new;
                ctl->thr = thr_new;
                ctl->ev_kind = 0;
                ctl->ev_risky = 0;
                ctl->he_base = he_base_new;
                ctl->window = win_new;
                if (p + 1 >= npos) ctl->status = SEL_DONE;
            }
        }
    }
    FS_T(5);
}

// The step's kernel: the body above, then -- in the workgroup that is through last -- this rank's slot of the NEXT
// exchange (pack_event_body's: the first event the scan found, its row's entropy and frequencies), which as a launch of its
// own cost ~6 us of kernel and ~5 us of launch boundary a step.
template <typename T>
__global__ __launch_bounds__(FS_THREADS, 2) void fs_step_kernel(SelDev d, const T *__restrict__ mat,
                                                             const double *__restrict__ jobres, uint32_t K, int apply,
                                                             FsSync *sync, uint32_t cap, double *__restrict__ slot_next,
                                                             unsigned long long *hist) {
    extern __shared__ __attribute__((aligned(16))) unsigned char fs_smem[];
    __shared__ unsigned int s_last;
    fs_step_body<T>(d, mat, jobres, K, apply, sync, cap, fs_smem);
    if (apply && hist && blockIdx.x == 0 && threadIdx.x == 0) {
        // the engine's status behind this launch, where the host can see it without asking (pinned host memory, a word a
        // launch, FS_HIST of them): dvs_select_step_peek reads the word of a launch some way back, so the driver's look at
        // the status neither drains the queue nor differs between ranks
        const unsigned long long L = sync->launches + 1;
        sync->launches = L;
        __hip_atomic_store(&hist[L % FS_HIST], (L << 8) | (unsigned long long)d.ctl->status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (!slot_next) return;
    __syncthreads();  // (every wave's stores and posts are issued and counted)
    if (threadIdx.x == 0) {
        const unsigned long long before =
            __hip_atomic_fetch_add(&sync->finished.v, 1ull, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        s_last = before + 1 == gridDim.x;
        if (s_last) __hip_atomic_store(&sync->finished.v, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (!s_last) return;
    // (every other workgroup's stores -- the state writer's status word, the finders' posts -- are behind the counter)
    const SelCtl *ctl = d.ctl;
    const unsigned long long p = __hip_atomic_load(&ctl->event_pos, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t status = __hip_atomic_load(&ctl->status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t row = DVS_ROW_REMOTE;
    if (status == SEL_RUN && p != SEL_NONE) row = d.order ? d.order[p] : uint32_t(p);
    if (row == DVS_ROW_REMOTE) {
        if (threadIdx.x == 0) slot_next[0] = -1.0;  // (the rest of the slot is not read)
        return;
    }
    const double tot = double(d.totals[row]);
    const double hrow = d.rowH[row];
    const T *rp = mat + uint64_t(row) * d.B;
    for (uint64_t i0 = threadIdx.x; i0 < d.B; i0 += uint64_t(FS_HOLD) * FS_THREADS) {  // (a thread's bins requested together)
        T c[FS_HOLD];
#pragma unroll
        for (int k = 0; k < FS_HOLD; k++) {
            const uint64_t i = i0 + uint64_t(k) * FS_THREADS;
            c[k] = i < d.B ? rp[i] : T(0);
        }
#pragma unroll
        for (int k = 0; k < FS_HOLD; k++) {
            const uint64_t i = i0 + uint64_t(k) * FS_THREADS;
            if (i < d.B) slot_next[2 + i] = cand_freq(&c[k], 0, tot);
        }
    }
    if (threadIdx.x == 0) {
        slot_next[0] = double(p);  // (exact below 2^53)
        slot_next[1] = hrow;
    }
}

// ---- MODE_MAX while the set may grow: a BATCH of rows against the unchanged set.
// A tentative push that is rolled back leaves the set as it was (records.rs:439-450), and so does a row that
// is no event: the rows from the cursor up to the next push that is KEPT all face the same set.  Where
// events come back to back (genome collections: nearly every row is one, one in fifty is kept) the iteration
// above -- five launches and ~130 us per event -- walks them one by one.  These two kernels take MB_ROWS rows
// at once: max_batch_jobs works out, for every row, its own increases_jsd score and the leave-one-out pass
// of the bigger set (one workgroup per row and member), max_batch_decide takes the decisions (the
// arithmetic and the bands of resolve / finalize) and moves the cursor over the rows that are NOT events or
// whose push is clearly rolled back -- nothing else: the first row that would be kept, or that is too close
// to call, stays where it is, and the ordinary iteration behind the pair deals with it (commit, arbiter).
// The persistent engine has the same thing in-kernel (persist.hip); this one has no limits on 4^k or the set.
constexpr uint32_t MB_ROWS = 32;
constexpr int MB_THREADS = 256;
constexpr uint32_t MB_MAX_MEMBERS = 4096;  // sets up to this size (the result block is 3 x MB_ROWS x (this + 3) doubles)

__device__ __forceinline__ bool max_batch_applies(const SelCtl *ctl, const SelDev &d, uint32_t JW) {
    return ctl->status == SEL_RUN && ctl->ev_kind == 0 && ctl->mode == DVS_MODE_MAX && ctl->size < ctl->max_size &&
           ctl->s_is_resum != 0 && ctl->forced == FORCE_NONE && !d.gather_all && ctl->size + 3 <= JW && ctl->size >= 2 &&
           ctl->mb_stuck == 0;
}

// One workgroup: ONE member r (or the whole bigger set, or the score) and MB_RG consecutive rows of the batch --
// S and the member's row are read once for the group, and what differs per row is the candidate's counts.  The
// jobs only steer the cursor over rows that change nothing (anything inside a band stops the batch), so their
// quotients are the exact ones by fma (exact_div_u32) and the means multiply by reciprocals, like the persistent
// engine's batches (persist.hip): two f64 divisions per bin fewer -- the kernel was bound by them (4^7 bins, 117
// members, 32 rows: 150 us, ~100 vector instructions per bin).
constexpr uint32_t MB_RG = 2;
static_assert(MB_ROWS % MB_RG == 0, "whole groups");
template <typename T>
__global__ __launch_bounds__(MB_THREADS) void max_batch_jobs_kernel(SelDev d, const T *__restrict__ mat,
                                                                    double *__restrict__ res, uint32_t JW) {
    __shared__ double scratch[48];
    __shared__ double2 ltab[128];  // log2_tab's table (select_dev.h): ~18 instead of ~33 instructions per logarithm
    const SelCtl *ctl = d.ctl;
    if (!max_batch_applies(ctl, d, JW)) return;
    const uint32_t n = ctl->size, r = blockIdx.x, e0 = blockIdx.y * MB_RG;
    if (r >= n + 3) return;
    if (threadIdx.x < 128) log2_tab_fill(ltab, threadIdx.x);
    __syncthreads();
    // the group's rows (the same answers in every thread: the branches below are uniform)
    const T *rp[MB_RG];
    double tot[MB_RG], rt[MB_RG];
    bool live[MB_RG], any = false;
#pragma unroll
    for (uint32_t q = 0; q < MB_RG; q++) {
        const uint64_t p = ctl->cursor + e0 + q;
        live[q] = false;
        rp[q] = mat;
        tot[q] = rt[q] = 1.0;
        if (p >= ctl->npos) continue;
        const uint32_t row = d.order ? d.order[p] : uint32_t(p);
        const uint32_t lab = d.labels ? d.labels[p] : row;
        if (lab < d.nlabels && d.inset[lab]) continue;  // a member already: no event (records.rs:71-74,87-89)
        const uint32_t toti = d.totals[row];
        if (toti == 0) continue;  // no valid k-mer: the stream skips the row (records.rs:419-423)
        live[q] = any = true;
        rp[q] = mat + uint64_t(row) * d.B;
        tot[q] = double(toti);
        rt[q] = 1.0 / tot[q];
    }
    if (!any) return;
    Ent en[MB_RG];
    // Four bins a thread per pass: S, the second row and the rows' counts are requested together, then the
    // arithmetic -- every row's terms still in bin order (i, i + 256, ...), as the one-bin loop adds them.
    auto run = [&](const double *second, auto term) {
        constexpr int U = 4;
        uint64_t i = threadIdx.x;
        for (; i + uint64_t(U - 1) * MB_THREADS < d.B; i += uint64_t(U) * MB_THREADS) {
            double sv[U], mv[U];
            T c[MB_RG][U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                sv[u] = d.S[i + uint64_t(u) * MB_THREADS];
                mv[u] = second ? second[i + uint64_t(u) * MB_THREADS] : 0.0;
            }
#pragma unroll
            for (uint32_t q = 0; q < MB_RG; q++)
#pragma unroll
                for (int u = 0; u < U; u++)
                    if (live[q]) c[q][u] = rp[q][i + uint64_t(u) * MB_THREADS];
#pragma unroll
            for (int u = 0; u < U; u++)
#pragma unroll
                for (uint32_t q = 0; q < MB_RG; q++)
                    if (live[q]) en[q].add(term(sv[u], mv[u], count_freq_x(c[q][u], tot[q], rt[q])), ltab);
        }
        for (; i < d.B; i += MB_THREADS) {
            const double sv = d.S[i], mv = second ? second[i] : 0.0;
#pragma unroll
            for (uint32_t q = 0; q < MB_RG; q++)
                if (live[q]) en[q].add(term(sv, mv, cand_freq_x(rp[q], i, tot[q], rt[q])), ltab);
        }
    };
    if (r == n + 2) {  // increases_jsd: block_delta_jsd's arithmetic
        const double rn = 1.0 / double(n);
        run(d.M + uint64_t(d.ord[ctl->lowest]) * d.B, [&](double sv, double lowv, double f) { return ((sv - lowv) + f) * rn; });
    } else if (r == n + 1) {  // the bigger set as a whole (resolve_body: Stmp = S + candidate, H(Stmp / (n + 1)))
        const double rn1 = 1.0 / double(n + 1);
        run(nullptr, [&](double sv, double, double f) { return (sv + f) * rn1; });
    } else if (r < n) {  // without member r (loo_body)
        const double rdiv = 1.0 / double(n);
        run(d.M + uint64_t(d.ord[r]) * d.B, [&](double sv, double mv, double f) {
            double v = ((sv + f) - mv) * rdiv;  // updated_mean_freqs, records.rs:276-286
            return v <= DVS_EPS ? 0.0 : v;
        });
    } else {  // r == n: without the candidate itself
        const double rdiv = 1.0 / double(n);
        run(nullptr, [&](double sv, double, double f) {
            double v = ((sv + f) - f) * rdiv;
            return v <= DVS_EPS ? 0.0 : v;
        });
    }
#pragma unroll
    for (uint32_t q = 0; q < MB_RG; q++) {
        if (!live[q]) continue;
        double h = en[q].h, mn = en[q].mn, sm = en[q].sum;
        block_red3(h, mn, sm, scratch);
        if (threadIdx.x == 0) {
            const uint32_t e = e0 + q;
            res[(uint64_t(e) * 3 + 0) * JW + r] = h;
            res[(uint64_t(e) * 3 + 1) * JW + r] = sm;
            res[(uint64_t(e) * 3 + 2) * JW + r] = mn;
        }
    }
}

__global__ __launch_bounds__(WIDE_THREADS) void max_batch_decide_kernel(SelDev d, const double *__restrict__ res, uint32_t JW) {
    __shared__ double s_kind[MB_ROWS], s_counted[MB_ROWS];
    SelCtl *ctl = d.ctl;
    if (!max_batch_applies(ctl, d, JW)) return;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwave = WIDE_THREADS / 64;
    const uint32_t n = ctl->size, n1 = n + 1;
    const uint64_t p0 = ctl->cursor;
    const uint32_t nrows = uint32_t(umin64(MB_ROWS, ctl->npos - p0));
    const double dn = double(n), dn1 = double(n1);
    for (uint32_t e = wave; e < nrows; e += nwave) {
        // kind: 0 the stream moves on (no event, or a push that is clearly rolled back), 2 it stops here (a push
        // that would be kept, or anything too close to call)
        double kind = 0.0, counted = 0.0;
        const uint64_t p = p0 + e;
        const uint32_t row = d.order ? d.order[p] : uint32_t(p);
        const uint32_t lab = d.labels ? d.labels[p] : row;
        const bool dead = (lab < d.nlabels && d.inset[lab]) || d.totals[row] == 0;
        if (!dead) {
            const double H_e = d.rowH[row];
            const double *re = res + uint64_t(e) * 3 * JW;
            const double hs = re[n + 2], ss = re[JW + n + 2], ms = re[2 * JW + n + 2];
            const double jsd = (ms < 0.0) ? NAN : hs - (ctl->he_base + H_e) / dn;  // block_delta_jsd
            counted = 1.0;
            if (sum_risky(ss, d.B) || fabs(jsd - ctl->thr) <= ctl->band) {
                kind = 2.0;
                counted = 0.0;
            } else if (jsd > ctl->thr) {  // an event: the tentative push (resolve_body, loo_body, finalize_body)
                const double sumH_t = ctl->sum_entropy + H_e;
                const double hm = re[n1], svm = re[JW + n1];
                const double tj = hm - sumH_t / dn1;
                const double div = dn1 - 1.0;
                auto delta = [&](uint32_t r) {
                    const double mh = r < n ? d.mH[d.ord[r]] : H_e;
                    return tj - (re[r] - (sumH_t - mh) / div);
                };
                // (a lane's first four members' deltas stay in registers for the second and third pass -- every delta is
                // two dependent global loads, and sets of up to 256 members need no more; the same values either way)
                constexpr uint32_t KEEP = 4;
                double kept[KEEP];
                auto delta_again = [&](uint32_t r) {
                    const uint32_t q = (r - lane) >> 6;
                    double v = 0.0;
                    bool have = false;
#pragma unroll
                    for (uint32_t x = 0; x < KEEP; x++)
                        if (q == x) {
                            v = kept[x];
                            have = true;
                        }
                    return have ? v : delta(r);
                };
                double best = 1e6, acc = 0.0;
                bool risky = sum_risky(svm, d.B) || !(hm == hm);
#pragma unroll
                for (uint32_t x = 0; x < KEEP; x++) kept[x] = 0.0;
                for (uint32_t r = lane; r < n1; r += 64) {
                    const double v = delta(r);
                    const uint32_t q = (r - lane) >> 6;
#pragma unroll
                    for (uint32_t x = 0; x < KEEP; x++)
                        if (q == x) kept[x] = v;
                    risky |= sum_risky(re[JW + r], d.B);
                    acc += v;
                    if (v < best) best = v;
                }
                const double dmin = dvs_wave_min(best);
                double fi = 4294967295.0;
                for (uint32_t r = lane; r < n1; r += 64)
                    if (dmin < 1e6 && delta_again(r) == dmin) fi = fmin(fi, double(r));
                const double dfirst = dvs_wave_min(fi);
                const uint32_t lowest = (dfirst < 4294967295.0) ? uint32_t(dfirst) : 0u;
                const double mean = dvs_wave_sum(acc) / dn1;
                double second = 1e6, tv = 0.0;
                for (uint32_t r = lane; r < n1; r += 64) {
                    const double v = delta_again(r);
                    if (r != lowest && v < second) second = v;
                    const double t = v - mean;
                    tv += t * t;
                }
                const double dsecond = dvs_wave_min(second);
                const double sd = sqrt(dvs_wave_sum(tv) / (dn1 - 1.0));
                const double cov = sd / mean;
                const bool any_risky = __ballot(risky) != 0ull;
                const double band = sel_band(tj + sumH_t / dn1, d.B);
                const double a = ctl->stat == DVS_STAT_STDEV ? sd : cov;
                const double b = ctl->stat == DVS_STAT_STDEV ? ctl->std_delta : ctl->cov_delta;
                const double mm = fmin(fabs(mean), fabs(ctl->mean_delta));
                const double sband = ctl->stat == DVS_STAT_STDEV
                                         ? 4.0 * band
                                         : 4.0 * band * (1.0 + fmax(fabs(a), fabs(b))) / fmax(mm, 1e-300);
                const bool tie = any_risky || (dsecond - dmin <= band && dsecond < 1e6) || !(fabs(a - b) > sband);
                if (tie || a > b) {  // too close to call, or a push that is kept: the ordinary iteration's
                    kind = 2.0;
                    counted = 0.0;
                }
            }
        }
        if (lane == 0) {
            s_kind[e] = kind;
            s_counted[e] = counted;
        }
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t stop = nrows, events = 0;
        for (uint32_t e = 0; e < nrows; e++) {
            if (s_kind[e] == 2.0) {
                stop = e;
                break;
            }
            events += s_counted[e] != 0.0;
        }
        // (a batch that stopped in front of a row leaves that row to the ordinary iteration; the batch pairs
        // queued behind this one return at once until finalize has dealt with it)
        if (stop < nrows) ctl->mb_stuck = 1;
        if (stop > 0) {
            ctl->cursor = p0 + stop;
            ctl->mb_rows += stop;
            ctl->n_events += events;
            ctl->n_windows++;
            ctl->rows_scored += stop;
            ctl->event_pos = SEL_NONE;
            if (ctl->cursor >= ctl->npos) ctl->status = SEL_DONE;
            ctl_next_window(ctl);
        }
    }
}

// Resolve kernel (one block).  (Resolve + leave-one-out + finalize as ONE single-block launch for
// small sets measured slower -- 10.6 vs 8.9 ms per stepwise selection: one CU issues every
// leave-one-out bin itself -- and was removed.)
template <typename T>
__global__ __launch_bounds__(WIDE_THREADS) void resolve_kernel(SelDev d, const T *__restrict__ mat) {
    __shared__ double scratch[48];
    __shared__ int s_flag;
    resolve_body<T>(d, mat, scratch, s_flag);
}

// Block r < n: the leave-one-out job of member r.  The LAST block adds up the rows the scan
// launch read (per-workgroup counters, cleared for the next launch) beside them, off the event's
// chain of dependent round trips.
__global__ __launch_bounds__(LOO_THREADS) void loo_kernel(SelDev d, uint32_t scan_grid) {
    __shared__ double scratch[48];
    if (blockIdx.x + 1 == gridDim.x) {
        double cnt = 0.0, cnt2 = 0.0, mnz = 0.0;
        for (uint32_t i = threadIdx.x; i < scan_grid; i += LOO_THREADS) {
            cnt += double(d.wg_rows[2 * i]);
            cnt2 += double(d.wg_rows[2 * i + 1]);
            d.wg_rows[2 * i] = 0;
            d.wg_rows[2 * i + 1] = 0;
        }
        block_red3(cnt, mnz, cnt2, scratch);
        if (threadIdx.x == 0 && (cnt != 0.0 || cnt2 != 0.0)) {  // (finalize_kernel, next in the stream, writes other words)
            d.ctl->rows_scored += (unsigned long long)cnt;
            d.ctl->rows_rechecked += (unsigned long long)cnt2;
        }
        return;
    }
    loo_body(d, blockIdx.x, scratch);
}

__global__ __launch_bounds__(WIDE_THREADS) void finalize_kernel(SelDev d) {
    __shared__ double scratch[48];
    __shared__ int s_go;
    finalize_body(d, scratch, s_go);
}

// delta_jsd of every query row against the set (SummedRecordsWrapper.delta_jsd,
// src/records_py.rs:111-120 -> records.rs:70-84).  One block per query.
template <typename T>
__global__ __launch_bounds__(LOO_THREADS) void score_kernel(SelDev d, const T *__restrict__ qmat,
                                                          const uint32_t *__restrict__ qtotals,
                                                          const double *__restrict__ qH,
                                                          const uint32_t *__restrict__ qlabels,
                                                          double *__restrict__ out) {
    __shared__ double scratch[32];
    const SelCtl *ctl = d.ctl;
    const uint32_t q = blockIdx.x;
    const uint32_t tot_u = qtotals[q];
    if (tot_u == 0) {
        if (threadIdx.x == 0) out[q] = NAN;
        return;
    }
    if (qlabels) {
        const uint32_t lab = qlabels[q];
        if (lab < d.nlabels && d.inset[lab]) {
            if (threadIdx.x == 0) out[q] = 0.0;
            return;
        }
    }
    const double tot = double(tot_u);
    const T *rp = qmat + uint64_t(q) * d.B;
    const uint32_t low_slot = d.ord[ctl->lowest];
    const double *low = d.M + uint64_t(low_slot) * d.B;
    const double dsize = double(ctl->size);
    Ent e;
    for (uint64_t i = threadIdx.x; i < d.B; i += LOO_THREADS)
        e.add((d.S[i] - low[i] + cand_freq(rp, i, tot)) / dsize);
    const double h = dvs_block_sum(e.h, scratch);
    const double mn = dvs_block_min(e.mn, scratch);
    if (threadIdx.x == 0) {
        const double mean_entropy = (ctl->sum_entropy - d.mH[low_slot] + qH[q]) / dsize;
        out[q] = (mn < 0.0) ? NAN : h - mean_entropy;
    }
}

// members in set order -> a dense device buffer (for the device-side chunk merge)
__global__ __launch_bounds__(LOO_THREADS) void gather_members_kernel(SelDev d, double *__restrict__ rows,
                                                                    double *__restrict__ meta) {
    const uint32_t r = blockIdx.x, n = d.ctl->size;
    double *out = rows + uint64_t(r) * d.B;
    if (r >= n) {
        for (uint64_t i = threadIdx.x; i < d.B; i += LOO_THREADS) out[i] = 0.0;
        if (threadIdx.x == 0) meta[2 * r] = meta[2 * r + 1] = 0.0;
        return;
    }
    const uint32_t slot = d.ord[r];
    const double *src = d.M + uint64_t(slot) * d.B;
    for (uint64_t i = threadIdx.x; i < d.B; i += LOO_THREADS) out[i] = src[i];
    if (threadIdx.x == 0) {
        meta[2 * r] = double(d.mPos[slot]);  // stream position (exact below 2^53)
        meta[2 * r + 1] = 1.0;
    }
}

// ---- stepwise (distributed) helpers.  One exchange per greedy step: every rank packs its own first
// event of the window -- position, row entropy and the candidate's 4^k frequency row -- into its slot
// of an all_gather; afterwards every rank holds all the slots, takes the smallest position (a stream
// position is scored by exactly one rank) and resolves that candidate against its replica of the set.
// (pick: the head of resolve_body)
template <typename T>
__global__ __launch_bounds__(LOO_THREADS) void pack_event_kernel(SelDev d, const T *__restrict__ mat,
                                                                double *__restrict__ slot) {
    pack_event_body<T>(d, mat, slot);
}

}  // namespace

// ------------------------------------------------------------------ host side
static void sel_free(dvs_select *s) {
    if (!s) return;
#ifdef DVS_FS_TRACE
    if (s->d_jobres && getenv("DVS_FS_DEBUG")) {
        std::vector<unsigned long long> t(8 * 1024);
        (void)hipDeviceSynchronize();
        if (hipMemcpy(t.data(), reinterpret_cast<char *>(s->d_jobres) + size_t(2048) * 8 * 8, t.size() * 8, hipMemcpyDeviceToHost) == hipSuccess) {
            unsigned long long t0 = ~0ull;
            for (uint32_t b = 0; b < s->fs_grid; b++) if (t[b * 8]) t0 = std::min(t0, t[b * 8]);
            auto us = [&](unsigned long long x) { return x ? (double(x) - double(t0)) / 100.0 : -1.0; };
            double mx[8] = {0}, mn[8]; for (int k = 0; k < 8; k++) mn[k] = 1e9;
            for (uint32_t b = 1; b < s->fs_grid; b++) for (int k = 0; k < 7; k++) if (t[b * 8 + k]) { mx[k] = std::max(mx[k], us(t[b * 8 + k])); mn[k] = std::min(mn[k], us(t[b * 8 + k])); }
            fprintf(stderr, "[fs trace] scanning workgroups (us, min/max): start %.1f/%.1f decided %.1f/%.1f base+arrived %.1f/%.1f scan end %.1f/%.1f\n", mn[0], mx[0], mn[1], mx[1], mn[2], mx[2], mn[6], mx[6]);
            std::vector<double> e, r1, c1, sc; double posted = -1; unsigned long long nr = 0, nrmax = 0;
            for (uint32_t b = 1; b < s->fs_grid; b++) { if (t[b * 8 + 6]) e.push_back(us(t[b * 8 + 6])); if (t[b * 8 + 3]) r1.push_back(us(t[b * 8 + 3])); if (t[b * 8 + 4]) c1.push_back(us(t[b * 8 + 4]));
                if (t[b * 8 + 5]) sc.push_back(us(t[b * 8 + 5])); nr += t[b * 8 + 7]; nrmax = std::max(nrmax, t[b * 8 + 7]); }
            std::sort(e.begin(), e.end()); std::sort(r1.begin(), r1.end()); std::sort(c1.begin(), c1.end()); std::sort(sc.begin(), sc.end());
            auto pc = [&](std::vector<double> &v, double q) { return v.empty() ? -1.0 : v[size_t(q * (v.size() - 1))]; };
            fprintf(stderr, "[fs trace] wave 0s: scan loop entered p10/50/90/max %.1f %.1f %.1f %.1f; first row's data there %.1f %.1f %.1f %.1f; its score %.1f %.1f %.1f %.1f; scan end %.1f %.1f %.1f %.1f; first post (wave 0s only) %.1f; rows read sum %llu max %llu\n",
                    pc(sc, .1), pc(sc, .5), pc(sc, .9), pc(sc, 1), pc(r1, .1), pc(r1, .5), pc(r1, .9), pc(r1, 1), pc(c1, .1), pc(c1, .5), pc(c1, .9), pc(c1, 1), pc(e, .1), pc(e, .5), pc(e, .9), pc(e, 1), posted, nr, nrmax);
            const uint32_t L = 0;
            fprintf(stderr, "[fs trace] state writer: start %.1f decided %.1f loaded %.1f all arrived %.1f small stores issued %.1f rows issued %.1f done %.1f\n", us(t[L * 8]), us(t[L * 8 + 1]), us(t[L * 8 + 2]), us(t[L * 8 + 3]), us(t[L * 8 + 4]), us(t[L * 8 + 6]), us(t[L * 8 + 5]));
        }
    }
#endif
    // Work this selection queued on the context's side streams (the head phase's sync block, the seed
    // list, set-up kernels) may still be pending on an error path; the pool only orders reuse on the
    // context's own stream, so those streams are drained before their blocks go back to it.
    if (s->used_side_streams && s->ctx) {
        if (s->ctx->stream_head) (void)hipStreamSynchronize(s->ctx->stream_head);
        if (s->ctx->stream2) (void)hipStreamSynchronize(s->ctx->stream2);
    }
    void *ptrs[] = {s->dev.ctl, s->dev.S, s->dev.Stmp, s->dev.base, s->dev.cand, s->dev.M,
                    s->dev.mH, s->dev.mDelta, s->dev.dtmp, s->dev.dsum, s->dev.mLabel, s->dev.mPos,
                    s->dev.ord, s->dev.inset, s->dev.wg_rows, s->dev.evlog_pos, s->dev.evlog_kind, s->dev.rowlog,
                    (void *)s->dev.order, (void *)s->dev.labels};
    for (void *p : ptrs)
        dvs_dev_free(s->ctx, p);
    // (step kernels still queued -- a driver that gave up half-way -- write their status words into the pinned history:
    // they are waited for before the block goes back to the cache, where the next selection's control mirror may take it)
    if (s->h_fshist && s->fs_launches && s->ctx) (void)hipStreamSynchronize(s->ctx->stream);
    dvs_pinned_put(s->ctx, s->h_ctl);
    if (s->h_fshist) dvs_pinned_put(s->ctx, s->h_fshist);
    for (hipEvent_t e : s->ev_pool) dvs_event_put(s->ctx, e);
    dvs_dev_free(s->ctx, s->d_mbres);
    dvs_dev_free(s->ctx, s->d_jobres);
    dvs_dev_free(s->ctx, s->d_fsync);
    dvs_dev_free(s->ctx, s->psync);
    dvs_dev_free(s->ctx, s->ppart);
    dvs_dev_free(s->ctx, s->psync_head);
    dvs_dev_free(s->ctx, s->ppart_head);
    if (!s->seed_list_in_ctl) dvs_dev_free(s->ctx, s->d_seed_list);
    if (s->ev_side_done) dvs_event_put(s->ctx, s->ev_side_done);
    dvs_select_arbiter_free(s);
    dvs_ctx_release(s->ctx);
    delete s;
}

// stage 0: scan + resolve + loo + finalize; 1: resolve + loo + finalize; 2: loo + finalize
template <typename T>
static void launch_iteration(dvs_ctx *ctx, dvs_select *s, const T *mat, int stage, hipStream_t on = nullptr) {
    const SelDev &d = s->dev;
    const hipStream_t stream = on ? on : ctx->stream;
    if (stage == 0) {
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
            (void)hipEventRecord(e0, stream);
        }
        if (s->scan_hot)
            hipLaunchKernelGGL((scan_kernel<T, true>), dim3(s->scan_grid), dim3(SCAN_THREADS),
                               s->scan_lds, stream, d.ctl, mat, d.totals, d.rowH, d.order, d.labels,
                               d.inset, d.nlabels, d.base, d.wg_rows, d.B, s->base_in_lds ? 1 : 0);
        else
            hipLaunchKernelGGL((scan_kernel<T, false>), dim3(s->scan_grid), dim3(SCAN_THREADS),
                               s->scan_lds, stream, d.ctl, mat, d.totals, d.rowH, d.order, d.labels,
                               d.inset, d.nlabels, d.base, d.wg_rows, d.B, s->base_in_lds ? 1 : 0);
        if (s->time_scan) (void)hipEventRecord(e1, stream);
    }
    if (stage <= 1)
        hipLaunchKernelGGL((resolve_kernel<T>), dim3(1), dim3(WIDE_THREADS), 0, stream, d, mat);
    hipLaunchKernelGGL(loo_kernel, dim3(s->loo_grid + 1), dim3(LOO_THREADS), 0, stream, d, s->scan_grid);
    hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(WIDE_THREADS), 0, stream, d);
}

// the batch pair in front of an ordinary iteration (MODE_MAX while the set may grow and events are dense);
// size_bound: no set this launch can meet is larger
template <typename T>
static int launch_max_batch(dvs_ctx *ctx, dvs_select *s, const T *mat, uint32_t size_bound) {
    if (!s->d_mbres) {
        s->mb_jw = std::min<uint32_t>(s->cap, MB_MAX_MEMBERS) + 3;
        int rc = dvs_dev_alloc(ctx, (void **)&s->d_mbres, size_t(3) * MB_ROWS * s->mb_jw * sizeof(double), "max batch results");
        if (rc) return rc;
    }
    if (size_bound + 3 > s->mb_jw) return DVS_OK;  // (a set beyond the result block: the ordinary iterations alone)
    hipLaunchKernelGGL((max_batch_jobs_kernel<T>), dim3(size_bound + 3, MB_ROWS / MB_RG), dim3(MB_THREADS), 0, ctx->stream, s->dev, mat,
                       s->d_mbres, s->mb_jw);
    hipLaunchKernelGGL(max_batch_decide_kernel, dim3(1), dim3(WIDE_THREADS), 0, ctx->stream, s->dev, s->d_mbres, s->mb_jw);
    s->mb_launched++;
    return DVS_OK;
}

static int sel_poll(dvs_ctx *ctx, dvs_select *s) {
    DVS_HIP(ctx, hipMemcpyAsync(s->h_ctl, s->dev.ctl, sizeof(SelCtl), hipMemcpyDeviceToHost,
                                ctx->stream));
    DVS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return DVS_OK;
}

template <typename T>
static int sel_seed(dvs_ctx *ctx, dvs_select *s, const T *mat, hipStream_t st, bool light = false);

template <typename T>
static int sel_run_loop(dvs_ctx *ctx, dvs_select *s, const T *mat, bool first_unpolled = false) {
    unsigned long long persist_cursor = 0;
    unsigned persist_launches = 0;
    bool persist_was_last = false;  // nothing but the poll happened since the last persistent launch
    unsigned persist_idle = 0;      // persistent launches in a row that came back where they started
    bool mb_seen = false, mb_dense = true;  // MODE_MAX batches: events per row between the last two looks (dense until seen otherwise)
    unsigned long long mb_cursor = 0, mb_events = 0;
    bool mb_useful = true;
    uint32_t mb_rows_seen = 0, mb_round_pairs = 0, mb_idle_rounds = 0;
    if (s->persist && first_unpolled) {
        // straight behind the set-up kernels, no host round trip in between: the kernel itself
        // returns at once unless the control block says RUN
        persist_cursor = s->params.n_seed;  // the cursor the set-up leaves behind
        persist_launches = 1;
        persist_was_last = true;
        int rc0 = dvs_persist_launch(ctx, s);
        if (rc0 == DVS_ERR_UNSUPPORTED) {  // the runtime refused the grid: the multi-launch engine takes over
            s->persist = false;
            s->persist_fell_back = true;
            persist_launches = 0;
            if (s->seeded_start) {  // (nothing set the initial set up yet: the set-up kernels do)
                s->seeded_start = false;
                s->persist_seeded = false;
                int src = sel_seed<T>(ctx, s, mat, ctx->stream);
                if (src) return src;
            }
        } else if (rc0) {
            return rc0;
        }
    }
    // the loo grid must cover the largest set a batch can reach
    for (;;) {
        int rc = sel_poll(ctx, s);
        if (rc) return rc;
        const SelCtl &c = *s->h_ctl;
        if (s->time_scan && s->ev_used) {  // the poll synchronised the stream: events are complete
            for (size_t i = 0; i + 1 < s->ev_used; i += 2) {
                float ms = 0.f;
                if (hipEventElapsedTime(&ms, s->ev_pool[i], s->ev_pool[i + 1]) == hipSuccess) {
                    s->scan_ms += ms;
                    s->scan_ms_last = ms;
                }
                s->scan_launches++;
            }
            s->ev_used = 0;
        }
        for (int which = 0; which < 2 && s->persist && ctx->knobs.persist_debug; which++) {
            unsigned long long dbg[32];
            void *blk = which ? s->psync : s->psync_head;  // (the head phase's launch first, then the full grid's)
            if (!blk) continue;
            if (hipMemcpy(dbg, static_cast<char *>(blk) + dvs_persist_dbg_offset(), sizeof dbg, hipMemcpyDeviceToHost) == hipSuccess) {
                fprintf(stderr, "[dvs persist] %s launch\n", which ? "full-grid" : "head-phase");
                if (dvs_persist_probe_id()) {  // (a one-interval build: [0] ticks, [1] passes; block 0 then the last scanning block)
                    fprintf(stderr, "[dvs persist probe %d] block 0 (owns a job): %.3f us x %llu; the last scanning block (no job): %.3f us x %llu\n",
                            dvs_persist_probe_id(), dbg[1] ? dbg[0] / 100.0 / double(dbg[1]) : 0.0, dbg[1],
                            dbg[17] ? dbg[16] / 100.0 / double(dbg[17]) : 0.0, dbg[17]);
                    continue;
                }
                for (int w = 0; w < 2; w++)
                    fprintf(stderr, "[dvs persist %s] us: scan %.1f bar1 %.1f resolve %.1f loo %.1f bar2 %.1f | partials %.1f combine %.1f lowest-row fetch %.1f rebuild %.1f\n",
                            w ? "mirror block" : "block 0", dbg[0 + 16 * w] / 100.0, dbg[1 + 16 * w] / 100.0,
                            dbg[2 + 16 * w] / 100.0, dbg[3 + 16 * w] / 100.0, dbg[4 + 16 * w] / 100.0,
                            dbg[6 + 16 * w] / 100.0, dbg[7 + 16 * w] / 100.0, dbg[8 + 16 * w] / 100.0,
                            dbg[5 + 16 * w] / 100.0);
                if (dvs_persist_trace_offset()) {  // four windows' timelines across the grid
                    std::vector<unsigned long long> tr(4 * 8 * 256);
                    if (hipMemcpy(tr.data(), static_cast<char *>(blk) + dvs_persist_trace_offset(), tr.size() * 8, hipMemcpyDeviceToHost) == hipSuccess) {
                        const uint32_t G = which ? s->persist_grid : uint32_t(ctx->head_cus);
                        for (int w_ = 0; w_ < 4; w_++) {
                            const unsigned long long *t0 = &tr[(w_ * 8 + 0) * 256], *t1 = &tr[(w_ * 8 + 1) * 256], *t2 = &tr[(w_ * 8 + 2) * 256], *tg = &tr[(w_ * 8 + 3) * 256];
                            const unsigned long long *t4 = &tr[(w_ * 8 + 4) * 256], *t5 = &tr[(w_ * 8 + 5) * 256], *t6 = &tr[(w_ * 8 + 6) * 256], *t7 = &tr[(w_ * 8 + 7) * 256];
                            if (!tg[2] || G < 4) continue;
                            std::vector<double> top, arr, seen, pub, tot, end_;
                            unsigned long long first_top = ~0ull;
                            for (uint32_t b = 0; b + 2 < G; b++) if (t0[b]) first_top = std::min(first_top, t0[b]);
                            for (uint32_t b = 0; b + 2 < G; b++) {
                                if (!t0[b] || !t1[b] || !t2[b]) continue;
                                top.push_back((t0[b] - first_top) / 100.0);
                                arr.push_back((t1[b] - first_top) / 100.0);
                                seen.push_back((double(t2[b]) - double(tg[2])) / 100.0);
                                if (t4[b] > tg[2] && t5[b] > tg[2] && t6[b] > tg[2]) {  // (the window ended in an accept)
                                    pub.push_back((double(t4[b]) - double(tg[2])) / 100.0);
                                    tot.push_back((double(t5[b]) - double(tg[2])) / 100.0);
                                    end_.push_back((double(t6[b]) - double(tg[2])) / 100.0);
                                }
                            }
                            if (arr.empty()) continue;
                            auto srt = [](std::vector<double> &v) { std::sort(v.begin(), v.end()); };
                            srt(top); srt(arr); srt(seen);
                            auto q_ = [](const std::vector<double> &v, double f) { return v[size_t(f * (v.size() - 1))]; };
                            fprintf(stderr, "[dvs persist trace] window %d: tops 0 / %.2f / %.2f (min/median/max us), records stored %.2f / %.2f / %.2f / %.2f (min/median/90%%/max), "
                                    "gather begun %.2f, last record seen %.2f, release stored %.2f, release seen +%.2f / +%.2f / +%.2f (min/median/max after it was stored)\n",
                                    w_ * 12 + 12, q_(top, 0.5), top.back(), arr.front(), q_(arr, 0.5), q_(arr, 0.9), arr.back(),
                                    (double(tg[0]) - double(first_top)) / 100.0, (double(tg[1]) - double(first_top)) / 100.0, (double(tg[2]) - double(first_top)) / 100.0,
                                    seen.front(), q_(seen, 0.5), seen.back());
                            if (!pub.empty()) {
                                {   // the five workgroups that published last: when each arrived, saw the release, published
                                    std::vector<std::pair<double, uint32_t>> late;
                                    for (uint32_t b = 0; b + 2 < G; b++)
                                        if (t4[b] > tg[2]) late.emplace_back((double(t4[b]) - double(tg[2])) / 100.0, b);
                                    std::sort(late.rbegin(), late.rend());
                                    for (size_t i = 0; i < late.size() && i < 5; i++) {
                                        const uint32_t b = late[i].second;
                                        fprintf(stderr, "[dvs persist trace] window %d: workgroup %u arrived %.2f before the release was stored, saw it +%.2f, handed its speculative job over +%.2f, was through the job loop +%.2f\n",
                                                w_ * 12 + 12, b, (double(tg[2]) - double(t1[b])) / 100.0, (double(t2[b]) - double(tg[2])) / 100.0,
                                                (double(t7[b]) - double(tg[2])) / 100.0, late[i].first);
                                    }
                                }
                                {   // when the speculative jobs were handed over (mailboxes: stored), relative to the release
                                    std::vector<double> sp;
                                    for (uint32_t b = 0; b + 2 < G; b++)
                                        if (t7[b] && t4[b] > tg[2]) sp.push_back((double(t7[b]) - double(tg[2])) / 100.0);
                                    if (!sp.empty()) {
                                        srt(sp);
                                        fprintf(stderr, "[dvs persist trace] window %d: speculative jobs handed over %.2f / %.2f / %.2f / %.2f us after the release was stored "
                                                "(min/median/90%%/max over %zu workgroups; negative: before)\n", w_ * 12 + 12, sp.front(), q_(sp, 0.5), q_(sp, 0.9), sp.back(), sp.size());
                                    }
                                }
                                srt(pub); srt(tot); srt(end_);
                                fprintf(stderr, "[dvs persist trace] window %d, its accept (us after the release was stored; min/median/max): job published %.2f / %.2f / %.2f, "
                                        "totals read %.2f / %.2f / %.2f, rebuild done %.2f / %.2f / %.2f\n", w_ * 12 + 12, pub.front(), q_(pub, 0.5), pub.back(),
                                        tot.front(), q_(tot, 0.5), tot.back(), end_.front(), q_(end_, 0.5), end_.back());
                            }
                        }
                    }
                }
                if (dbg[15] + dbg[14])
                    fprintf(stderr, "[dvs persist block 0] accepts for which its leave-one-out job was ready when the release came: %llu (the candidate's frequencies: %llu)\n", dbg[15], dbg[14]);
                if (dbg[9] + dbg[10] + dbg[11] + dbg[12])
                    fprintf(stderr, "[dvs persist block 0] us inside the phases: window top %.1f own rows scanned %.1f hint look + record %.1f (then: scan = the rest) | behind the rebuild %.1f\n",
                            dbg[9] / 100.0, dbg[10] / 100.0, dbg[11] / 100.0, dbg[12] / 100.0);
                if (dbg[16 + 10] + dbg[16 + 12])
                    fprintf(stderr, "[dvs persist] scan + rendezvous: row-per-workgroup windows %llu (%.1f us, %llu rows), "
                            "row-per-wave windows %llu (%.1f us, %llu rows)\n", dbg[16 + 10], dbg[16 + 9] / 100.0,
                            dbg[16 + 13], dbg[16 + 12], dbg[16 + 11] / 100.0, dbg[16 + 14]);
            }
        }
        const bool fake_error = ctx->knobs.test_persist_fake_error;  // (test knob: needs DVS_TEST_KNOBS=1 as well)
        if (s->persist && persist_launches && (c.status == SEL_ERROR || (fake_error && persist_launches == 1))) {
            // The persistent kernel gave up at a grid barrier: its workgroups were not all resident
            // (a CU mask, a partitioned device, another stream's kernels holding CUs).  The
            // replicas may have stopped mid-update, so the selection starts over from its seeds and
            // the multi-launch engine, which needs no co-residency, serves the request.
            s->persist = false;
            s->persist_fell_back = true;
            if (!fake_error) ctx->persist_timeouts++;  // (three in a row and the context stops trying)
            dvs_select_arbiter_free(s);  // (its replay of the event log belongs to the abandoned run)
            rc = sel_seed<T>(ctx, s, mat, ctx->stream);
            if (rc) return rc;
            persist_launches = 0;
            continue;
        }
        if (s->seeded_start && s->persist && persist_launches == 1 &&
            (c.status == SEL_NEED_SETUP || (c.status == SEL_RUN && c.cursor == s->params.n_seed))) {
            // the seeded launch left the initial set alone (its first argmin too close to call on the
            // device): the set-up kernels after all, and the engine again from the state they leave
            s->seeded_start = false;
            s->persist_seeded = false;
            rc = sel_seed<T>(ctx, s, mat, ctx->stream);
            if (rc) return rc;
            persist_cursor = s->params.n_seed;
            persist_launches = 2;
            persist_was_last = true;
            rc = dvs_persist_launch(ctx, s);
            if (rc == DVS_ERR_UNSUPPORTED) {
                s->persist = false;
                s->persist_fell_back = true;
                persist_launches = 0;
            } else if (rc) {
                return rc;
            }
            continue;
        }
        if (c.status == SEL_DONE) {
            if (ctx->knobs.persist_debug)
                fprintf(stderr, "[dvs persist] launches ended early: replica full %u, sum check %u, push argmin %u, stat comparison %u, "
                        "candidate in band %u, replace argmin %u, state not taken %u / %u\n", c.why[0], c.why[1], c.why[2], c.why[3],
                        c.why[4], c.why[5], c.why[6], c.why[7]);
            if (s->persist && persist_launches) ctx->persist_timeouts = 0;
            return DVS_OK;
        }
        if (c.status == SEL_ARBITER) {
            if (s->params.flags & DVS_SELECT_NO_ARBITER)
                return dvs_set_error(ctx, DVS_ERR_UNSUPPORTED,
                                     "ambiguous decision at stream position %llu (stage %u): "
                                     "|score - threshold| within the rounding band",
                                     (unsigned long long)c.arb_pos, c.arb_stage);
            {
                const auto t_arb = std::chrono::steady_clock::now();
                rc = dvs_select_arbitrate(ctx, s);
                s->arbiter_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_arb).count();
            }
            if (rc) return rc;
            persist_was_last = false;
            launch_iteration<T>(ctx, s, mat, c.arb_stage == ARB_RESOLVE ? 1 : 2);
            DVS_HIP(ctx, hipGetLastError());
            continue;
        }
        if (c.status == SEL_ERROR)
            return dvs_set_error(ctx, DVS_ERR_RUNTIME, "selection engine stopped with an internal error");
        if (c.status != SEL_RUN)
            return dvs_set_error(ctx, DVS_ERR_RUNTIME, "selection engine in state %u", c.status);
        // a `max` set that has outgrown the engine's LDS replica and may still grow: the engine would hand
        // every launch straight back (it did: 152 of 309 launches of a 664-member cov selection) -- the
        // multi-launch kernels keep the stream until the set is full
        const bool replica_full = s->params.mode == DVS_MODE_MAX && c.size + 2 > s->persist_maxn && c.size < c.max_size;
        if (s->persist && !replica_full) {
            // a persistent launch that comes back still RUNNING with the cursor where it was did
            // not take the state it found (e.g. an event left pending by the arbiter's hand-off):
            // the multi-launch kernels, which take any state, carry on -- never a relaunch loop
            if (persist_launches && persist_was_last && c.cursor == persist_cursor) {
                persist_was_last = false;  // multi-launch iterations, then the engine again
                persist_idle++;
            } else {
                if (persist_launches && c.cursor != persist_cursor) persist_idle = 0;
                persist_cursor = c.cursor;
                persist_launches++;
                persist_was_last = true;
                rc = dvs_persist_launch(ctx, s);
                if (rc == DVS_ERR_UNSUPPORTED) {
                    s->persist = false;
                    s->persist_fell_back = true;
                } else if (rc) {
                    return rc;
                } else {
                    continue;
                }
            }
        }
        // (behind a persistent launch that left ONE event for these kernels -- a decision inside its band --
        // a single iteration takes that event; the engine is launched again right after)
        // (an engine that keeps coming back where it started -- a set its replica cannot hold -- gets whole batches)
        if (replica_full) persist_was_last = false;
        const int iters = (s->persist && !replica_full && persist_launches && !persist_was_last && persist_idle < 2) ? 1 : s->batch;
        // MODE_MAX while the set may grow: where at least every second row since the last look was an event,
        // a batch pair goes in front of every iteration (it skips the rows that change nothing)
        if (mb_seen) {
            const unsigned long long rows = c.cursor - mb_cursor, evs = c.n_events - mb_events;
            if (rows) mb_dense = evs * 2 >= rows;
        }
        mb_seen = true;
        mb_cursor = c.cursor;
        mb_events = c.n_events;
        // ... and only while the pairs earn their keep: a stream whose pushes are nearly all KEPT (cov over
        // sequences of very different composition) stops every batch at its first row -- then they are left
        // out, and tried again every eighth look
        // (a probe: pairs in front of the first two iterations of a look only)
        if (mb_round_pairs) mb_useful = (c.mb_rows - mb_rows_seen) >= 2u * mb_round_pairs;
        mb_rows_seen = c.mb_rows;
        mb_round_pairs = 0;
        const bool mb_can = mb_dense && iters > 1 && c.mode == DVS_MODE_MAX && c.size < c.max_size && c.s_is_resum != 0 &&
                            !(s->params.flags & DVS_SELECT_STEPWISE);
        const bool mb_probe = mb_can && !mb_useful && ++mb_idle_rounds >= 4;
        if (mb_probe || mb_useful) mb_idle_rounds = 0;
        const int mb_iters = !mb_can ? 0 : mb_useful ? iters : mb_probe ? 2 : 0;  // (a selection starts with them on)
        for (int i = 0; i < iters; i++) {
            for (int b = 0; i < mb_iters && b < 3; b++) {  // (the second and third return at once where the first one stopped at a row)
                if ((rc = launch_max_batch<T>(ctx, s, mat, c.size + uint32_t(i)))) return rc;
                mb_round_pairs += b == 0;
            }
            launch_iteration<T>(ctx, s, mat, 0);
        }
        DVS_HIP(ctx, hipGetLastError());
    }
}

// The stream a selection's set-up (control block, seed list, set-up kernels) goes to, decided once per
// selection (the label flags are cleared on it ahead of time).  When the rest of the matrix is still being
// built on the context's stream (a split build, kmer_hist.hip: the head rows are finished, the host has their
// totals), the set-up -- which reads only the seed rows -- goes to a side stream and runs beside that launch
// instead of queueing behind it; the engine itself is launched on the first stream behind both.
// HEAD PHASE: that launch runs on the context's stream_rest and leaves the head CUs alone (CU split), so the
// persistent engine starts on them right behind the set-up and walks the rows that are already built -- the
// event-dense head of the stream, a chain of hand-overs that needs no more than a few workgroups -- while
// the histogram runs; the launch over the full grid then carries on from the state it mirrors.  (nmost; a
// set whose leave-one-out jobs fit the head grid one per workgroup.)
static void sel_plan_setup_stream(dvs_ctx *ctx, dvs_select *s) {
    s->setup_side = nullptr;
    s->head_phase = false;
    if (s->mat->head_rows_built && s->params.n_seed <= s->mat->head_rows_built && s->h_order.empty() &&
        !(s->params.flags & DVS_SELECT_STEPWISE)) {
        s->head_phase = s->mat->rest_beside_head && ctx->stream_head && s->persist &&
                        s->params.mode == DVS_MODE_NMOST && s->params.window == 0 &&
                        s->cap + 2 <= uint32_t(ctx->head_cus) && s->npos > 4ull * s->mat->head_rows_built &&
                        s->params.n_seed + 64 <= s->mat->head_rows_built && !ctx->knobs.no_head_phase;
        s->setup_side = s->head_phase ? ctx->stream_head : dvs_ctx_stream2(ctx);
    }
}

template <typename T>
static int sel_start(dvs_ctx *ctx, dvs_select *s, const T *mat) {
    hipStream_t side = s->setup_side;
    const bool head_phase = s->head_phase;
    hipStream_t st = side ? side : ctx->stream;
    if (side) s->used_side_streams = true;
    int rc = sel_seed<T>(ctx, s, mat, st, s->seeded_start);
    if (rc) return rc;
    if (head_phase) {
        rc = dvs_persist_launch_head(ctx, s, uint32_t(ctx->head_cus), s->mat->head_rows_built, side);
        if (rc && rc != DVS_ERR_UNSUPPORTED) return rc;  // (refused: the full-grid launch starts from the seeds)
        if (rc == DVS_ERR_UNSUPPORTED && s->seeded_start) {  // ... which, unseeded, it needs the set-up kernels for
            s->seeded_start = false;
            s->persist_seeded = false;
            rc = sel_seed<T>(ctx, s, mat, st);
            if (rc) return rc;
        }
        // the full-grid launch's sync block and accumulators: behind the histogram on the context's
        // stream, i.e. while the head phase runs, not between the two launches
        rc = dvs_persist_prepare_main(ctx, s);
        if (rc) return rc;
        // (a later selection over the same matrix finds the histogram finished: nothing to run beside)
        const_cast<dvs_matrix *>(s->mat)->rest_beside_head = false;
    }
    if (side) {
        if (!s->ev_side_done) s->ev_side_done = dvs_event_get(ctx);
        DVS_HIP(ctx, hipEventRecord(s->ev_side_done, side));
        DVS_HIP(ctx, hipStreamWaitEvent(ctx->stream, s->ev_side_done, 0));
    }
    if (s->params.flags & DVS_SELECT_STEPWISE) return sel_poll(ctx, s);
    return sel_run_loop<T>(ctx, s, mat, true);
}

// control block of a fresh selection + the initial set from the seed positions (SummedRecords::new
// over the first n usable records, records.rs:288-308), enqueued on `st`; also the way back to a clean
// state when the persistent engine has to be abandoned
// light: only what a SEEDED persistent launch reads -- the control block and the seed list; the initial
// set is worked out by that launch itself (persist.hip) instead of the four kernels below
template <typename T>
static int sel_seed(dvs_ctx *ctx, dvs_select *s, const T *mat, hipStream_t st, bool light) {
    const std::vector<uint64_t> &seeds = s->seed_positions;
    SelDev &d = s->dev;
    SelCtl c = s->ctl0;
    // default window: a quarter of the persistent grid's waves (measured best: early in the
    // stream an accept comes every few hundred rows), else 4096 rows per scan launch
    const uint32_t wdef = s->params.window ? s->params.window : (s->persist ? s->persist_grid * 2u : 4096u);
    c.window_min = wdef;
    c.window_max = std::max<uint32_t>(wdef, s->scan_grid * (SCAN_THREADS / 64) * 8);
    c.window = wdef;
    // (the control block travels through the pinned mirror: a pageable source would have to stay
    // alive until the copy has been performed)
    *s->h_ctl = c;
    // (a restart, or a start on another stream than the planned one: the flags are cleared here)
    if (!s->inset_clean || st != (s->setup_side ? s->setup_side : ctx->stream))
        DVS_HIP(ctx, hipMemsetAsync(d.inset, 0, std::max<size_t>(d.nlabels, 1), st));
    s->inset_clean = false;
    if (!s->d_seed_list) {
        int rc0 = dvs_dev_alloc(ctx, &s->d_seed_list, seeds.size() * sizeof(uint64_t), "seed list");
        if (rc0) return rc0;
    }
    uint64_t *d_seed = static_cast<uint64_t *>(s->d_seed_list);
    if (s->seed_list_in_ctl) {  // control block + seed list: one block, one copy
        std::memcpy(reinterpret_cast<unsigned char *>(s->h_ctl) + SEL_SEEDS_AT, seeds.data(), seeds.size() * sizeof(uint64_t));
        DVS_HIP(ctx, hipMemcpyAsync(d.ctl, s->h_ctl, SEL_SEEDS_AT + seeds.size() * sizeof(uint64_t), hipMemcpyHostToDevice, st));
    } else {
        DVS_HIP(ctx, hipMemcpyAsync(d.ctl, s->h_ctl, sizeof c, hipMemcpyHostToDevice, st));
        DVS_HIP(ctx, hipMemcpyAsync(d_seed, seeds.data(), seeds.size() * sizeof(uint64_t),
                                    hipMemcpyHostToDevice, st));
    }
    if (light) {
        DVS_HIP(ctx, hipGetLastError());
        return DVS_OK;
    }
    hipLaunchKernelGGL((seed_kernel<T>), dim3(uint32_t(seeds.size())), dim3(LOO_THREADS), 0,
                       st, s->dev, mat, d_seed);
    hipLaunchKernelGGL(rebuild_sum_kernel, dim3(uint32_t((s->dev.B + RSUM_THREADS - 1) / RSUM_THREADS)), dim3(RSUM_THREADS), 0, st, s->dev);
    hipLaunchKernelGGL(rebuild_kernel, dim3(1), dim3(WIDE_THREADS), 0, st, s->dev);
    launch_iteration<T>(ctx, s, mat, 2, st);  // loo + finalize of the initial set
    // (one fused launch of one block for sets of <= 16 measured no faster: 2.29 vs 2.27 ms per step)
    DVS_HIP(ctx, hipGetLastError());
    return DVS_OK;
}

extern "C" int dvs_select_run(dvs_ctx *ctx, const dvs_matrix *m, const uint32_t *order,
                              const uint32_t *labels, uint64_t npos,
                              const dvs_select_params *params, dvs_select **out) {
    if (!ctx || !m || !params || !out) return dvs_set_error(ctx, DVS_ERR_VALUE, "null argument");
    *out = nullptr;
    DVS_HIP(ctx, hipSetDevice(ctx->device));
    const uint64_t B = m->nbins;
    if (!order && npos > m->nrows)
        return dvs_set_error(ctx, DVS_ERR_VALUE, "npos %llu exceeds the matrix rows %u",
                             (unsigned long long)npos, m->nrows);
    uint32_t n_seed = params->n_seed;
    uint32_t max_size = params->max_size;
    if (params->mode == DVS_MODE_SET) n_seed = uint32_t(npos);
    // Labels only say which positions are the same sequence id (records.rs:71-73: an id already in
    // the set scores 0.0).  When every label occurs once -- what the reference's callers pass: the
    // ids of a store are unique -- no position can meet its own id in the set, so the engine runs
    // label-free (label = position: the persistent engine qualifies) and the caller's values are
    // put back on the way out (dvs_select_get_members, dvs_select_delta_jsd).
    const uint32_t *caller_labels = nullptr;
    if (labels && !order && params->mode != DVS_MODE_SET) {
        bool distinct = true;
        bool rising = true;  // (the usual call: the positions themselves -- one pass, no copy)
        for (uint64_t p = 1; p < npos && rising; p++) rising = labels[p] > labels[p - 1];
        if (!rising) {
            std::vector<uint32_t> sorted(labels, labels + npos);
            std::sort(sorted.begin(), sorted.end());
            for (uint64_t p = 1; p < npos && distinct; p++)
                distinct = sorted[p] != sorted[p - 1] || sorted[p] == 0xFFFFFFFFu;
        }
        if (distinct) {
            caller_labels = labels;
            labels = nullptr;
        }
    }
    // src/records.rs:323-325, 404-410, 369-371, 464-469
    if (npos < n_seed)
        return dvs_set_error(ctx, DVS_ERR_VALUE, "The number of sequences %llu is < n %u",
                             (unsigned long long)npos, n_seed);
    if (params->mode == DVS_MODE_MAX && npos <= max_size) max_size = uint32_t(npos);  // :412-416
    if (params->mode != DVS_MODE_MAX) max_size = n_seed;

    // usable seeds: rows with at least one valid k-mer (records.rs:299-306)
    uint32_t nlabels = 0;
    if (order || labels) {
        for (uint64_t p = 0; p < npos; p++) {
            const uint32_t row = order ? order[p] : uint32_t(p);
            if (row == DVS_ROW_REMOTE) continue;
            if (row >= m->nrows)
                return dvs_set_error(ctx, DVS_ERR_VALUE, "order[%llu] = %u out of range",
                                     (unsigned long long)p, row);
            const uint32_t lab = labels ? labels[p] : row;
            if (lab != 0xFFFFFFFFu) nlabels = std::max(nlabels, lab + 1);
        }
    } else {
        nlabels = uint32_t(npos);  // label = row = position
    }
    // only the seed rows' totals are needed on the host
    if (order)
        for (uint64_t p = 0; p < n_seed; p++)
            if (order[p] == DVS_ROW_REMOTE)
                return dvs_set_error(ctx, DVS_ERR_VALUE,
                                     "seed position %llu is not local: the first n rows must be replicated",
                                     (unsigned long long)p);
    dvs_select *s = new dvs_select();
    // (a failing HIP call after this point releases the half-built selection)
#define SEL_HIP(call)                                    \
    do {                                                 \
        hipError_t e__ = (call);                         \
        if (e__ != hipSuccess) {                         \
            sel_free(s);                                 \
            return dvs_hip_fail(ctx, e__, #call);        \
        }                                                \
    } while (0)
    s->ctx = ctx;
    dvs_ctx_retain(ctx);
    s->params = *params;
    s->params.n_seed = n_seed;
    s->params.max_size = max_size;
    s->mat = m;
    s->mat_kind = m->kind;
    s->npos = npos;
    s->h_order.assign(order ? order : nullptr, order ? order + npos : nullptr);
    s->h_labels.assign(labels ? labels : nullptr, labels ? labels + npos : nullptr);
    if (caller_labels) s->h_out_labels.assign(caller_labels, caller_labels + npos);
    SelDev &d = s->dev;
    d.B = B;
    d.nlabels = nlabels;
    d.totals = m->d_totals;
    d.rowH = m->d_entropy;
    const uint32_t cap = std::max<uint32_t>(std::max<uint32_t>(max_size, n_seed) + 1, 2);  // (seeds <= n_seed)
    s->cap = cap;
    const size_t need = size_t(cap) * B * 8 + 5 * B * 8 + size_t(npos) * 8 + nlabels + (1 << 20);
    size_t free_b = 0, total_b = 0;
    if (need > (size_t(1) << 30)) SEL_HIP(hipMemGetInfo(&free_b, &total_b));
    if (need > (size_t(1) << 30) && need > free_b + ctx->pool_bytes) {
        sel_free(s);
        return dvs_set_error(ctx, DVS_ERR_NOMEM, "selection state needs %zu bytes, %zu free", need,
                             free_b);
    }
    // launch geometry
    s->base_in_lds = B * 8 <= 128 * 1024 && B * 8 + 1024 <= ctx->lds_per_block;
    s->scan_lds = 16 + (s->base_in_lds ? B * 8 : 0);
    const uint32_t wg_fit = s->base_in_lds ? std::max<uint32_t>(1, uint32_t((160 * 1024) / (B * 8 + 512))) : 4;
    uint32_t wg_per_cu = std::min<uint32_t>(wg_fit, 2);  // 16 waves per CU, 16 KB of loads in flight each
    s->scan_grid = std::max<uint32_t>(1, uint32_t(ctx->n_cu) * wg_per_cu);
    s->loo_grid = cap;
    // measured slower than three launches (one CU does the whole leave-one-out pass): opt-in only
    s->batch = 16;
    s->time_scan = ctx->timing;
    s->scan_hot = B % (256 * SCAN_CH) == 0 && !order && !labels && s->base_in_lds;
    if (s->scan_lds > 48 * 1024) {
        const void *fn = dvs_mat_dispatch(m, [&](auto *mp) -> const void * {
            using T = std::remove_cv_t<std::remove_pointer_t<decltype(mp)>>;
            return s->scan_hot ? reinterpret_cast<const void *>(scan_kernel<T, true>)
                               : reinterpret_cast<const void *>(scan_kernel<T, false>);
        });
        const int lrc = dvs_raise_dyn_lds(ctx, fn, s->scan_lds);
        if (lrc) {
            sel_free(s);
            return lrc;
        }
    }

#define SEL_ALLOC(ptr, bytes)                                        \
    do {                                                             \
        int rc__ = dvs_dev_alloc(ctx, (void **)&(ptr), (bytes), #ptr); \
        if (rc__) {                                                  \
            sel_free(s);                                             \
            return rc__;                                             \
        }                                                            \
    } while (0)
    // (the control block and, behind it, the seed list: ONE upload per selection, sel_seed)
    static_assert(sizeof(SelCtl) <= SEL_SEEDS_AT, "the seed list starts behind the control block");
    SEL_ALLOC(d.ctl, 4096);
    SEL_ALLOC(d.S, B * 8);
    SEL_ALLOC(d.Stmp, B * 8);
    SEL_ALLOC(d.base, B * 8);
    SEL_ALLOC(d.cand, B * 8);
    SEL_ALLOC(d.M, size_t(cap) * B * 8);
    SEL_ALLOC(d.mH, size_t(cap) * 8);
    SEL_ALLOC(d.mDelta, size_t(cap) * 8);
    SEL_ALLOC(d.dtmp, size_t(cap) * 8);
    SEL_ALLOC(d.dsum, size_t(cap) * 8);
    SEL_ALLOC(d.mLabel, size_t(cap) * 4);
    SEL_ALLOC(d.mPos, size_t(cap) * 8);
    SEL_ALLOC(d.ord, size_t(cap) * 4);
    SEL_ALLOC(d.inset, std::max<size_t>(nlabels, 1));
    SEL_ALLOC(d.wg_rows, size_t(s->scan_grid) * 8);
    SEL_ALLOC(d.evlog_pos, size_t(npos - n_seed + 2) * 8);
    SEL_ALLOC(d.evlog_kind, size_t(npos - n_seed + 2) * 4);
    if (s->params.flags & DVS_SELECT_STEPWISE) {
        // The arbiter's row log (rows of accepted candidates may live on other ranks): the frequency row of every
        // accepted event in event-log order.  On the device it is a RING of `ring` rows that dvs_select_step_poll
        // drains into host memory (s->h_rowlog) once it is half full (or the arbiter is about to read it) -- a step logs
        // at most one row, and dvs_select_step_apply refuses to run more than ring / 2 steps behind the last poll -- so the
        // log itself has no cap: a stream ordered by rising divergence accepts thousands of rows where a shuffled one
        // accepts n (1 + ln(N / n)), and the common selection (fewer accepts than half the ring) never pays for a copy.
        uint64_t ring = std::min<uint64_t>(256, std::max<uint64_t>(32, (uint64_t(256) << 20) / (B * 8)));
        if (ctx->knobs.test_rowlog_ring >= 4) ring = ctx->knobs.test_rowlog_ring;  // (tests: a ring that wraps after a few accepts)
        d.rowlog_cap = uint32_t(ring);
        SEL_ALLOC(d.rowlog, size_t(ring) * B * 8);
        // the fast step (fs_jobs_kernel / fs_step_kernel): select_nmost_divergent without a caller's labels
        // (the scan vector in f64 and f32, the members' sums, a staging area for the jobs' words)
        uint32_t fsK = 1;
        while (uint64_t(cap + 1) * fsK * 2 <= 96 && uint64_t(fsK) * 2 * FS_JOB_THREADS <= B) fsK *= 2;
        const size_t fs_lds = ((B + 1) & ~1ull) * 16 + size_t(cap + 1) * 3 * 8 + size_t(cap + 1) * fsK * 7 * 8 + 64;
        if (params->mode == DVS_MODE_NMOST && params->n_seed >= 2 && !labels && !ctx->knobs.no_fast_step && fs_lds <= ctx->lds_per_block &&
            uint64_t(cap + 1) <= FS_MAXJOBS) {
            s->fs_K = fsK;
            s->fs_lds = fs_lds;
            s->fs_grid = uint32_t(ctx->n_cu);  // (one workgroup a CU: 256 registers a lane hold a row in flight across the decisions; all resident at once, the state writer among them)
            SEL_ALLOC(s->d_jobres, size_t(FS_MAXJOBS) * FS_RES * 8 + 8 * 8 * 1024);
            SEL_HIP(hipMemsetAsync(s->d_jobres, 0, size_t(FS_MAXJOBS) * FS_RES * 8 + 8 * 8 * 1024, ctx->stream));
            SEL_ALLOC(s->d_fsync, sizeof(FsSync));
            SEL_HIP(hipMemsetAsync(s->d_fsync, 0, sizeof(FsSync), ctx->stream));
            void *hh = nullptr;
            if (dvs_pinned_get(ctx, &hh) == DVS_OK) {  // (without it the driver polls as before)
                memset(hh, 0, 4096);
                s->h_fshist = static_cast<unsigned long long *>(hh);
            }
            s->fast_step = true;
            s->fs_need_scan = true;
        }
    }
    if (order) {
        SEL_ALLOC(d.order, size_t(npos) * 4);
        SEL_HIP(hipMemcpyAsync((void *)d.order, order, size_t(npos) * 4, hipMemcpyHostToDevice,
                                    ctx->stream));
    }
    if (labels) {
        SEL_ALLOC(d.labels, size_t(npos) * 4);
        SEL_HIP(hipMemcpyAsync((void *)d.labels, labels, size_t(npos) * 4,
                                    hipMemcpyHostToDevice, ctx->stream));
    }
#undef SEL_ALLOC

    static_assert(sizeof(SelCtl) <= 4096, "control block must fit a cached pinned block");
    {
        int prc = dvs_pinned_get(ctx, (void **)&s->h_ctl);
        if (prc) {
            sel_free(s);
            return prc;
        }
    }

    {
        int prc = dvs_persist_setup(ctx, s);
        if (prc) {
            sel_free(s);
            return prc;
        }
    }
    // The label flags are cleared NOW, on the stream the set-up will use (same stream: ordered in front of
    // everything that sets one): nothing about them depends on the seeds, so the fill runs while the matrix's
    // head rows are still being built, not between the host's wake-up and the first launch.
    sel_plan_setup_stream(ctx, s);
    if (s->setup_side) s->used_side_streams = true;
    SEL_HIP(hipMemsetAsync(d.inset, 0, std::max<size_t>(nlabels, 1), s->setup_side ? s->setup_side : ctx->stream));
    s->inset_clean = true;
    // ... and so are the scan workgroups' row counters, which the set-up's own leave-one-out launch adds up (on the
    // context's stream this fill raced with a set-up on a side stream: `rows_scored` of a selection with a set of
    // more than 32 came out 4.4 M too high whenever the pool's block still held another selection's counters)
    SEL_HIP(hipMemsetAsync(d.wg_rows, 0, size_t(s->scan_grid) * 8, s->setup_side ? s->setup_side : ctx->stream));
    // SEEDED start (persist.hip): an nmost selection whose state fits the persistent kernel's register
    // cache is begun by that kernel itself -- S, the entropy sum, the leave-one-out pass and the first
    // lowest member from nothing but the seed positions -- instead of four launches in front of it
    // (sets of up to 32: beyond that the S of the seeds -- a memory round trip per four members -- costs
    // what the launches cost; DVS_PERSIST_NO_SEEDED=1 turns it off)
    // (a STEPWISE selection never launches the persistent kernel: its set-up kernels must run)
    s->seeded_start = s->persist && params->mode == DVS_MODE_NMOST && B <= 4096 && !order && !labels &&
                      !(s->params.flags & DVS_SELECT_STEPWISE) && n_seed >= 2 && n_seed <= 32 && !ctx->knobs.persist_no_seeded;
    s->persist_seeded = s->seeded_start;
    if (size_t(n_seed) * sizeof(uint64_t) <= 4096 - SEL_SEEDS_AT) {
        s->d_seed_list = reinterpret_cast<unsigned char *>(d.ctl) + SEL_SEEDS_AT;
        s->seed_list_in_ctl = true;
    } else if (s->seeded_start) {
        int arc = dvs_dev_alloc(ctx, &s->d_seed_list, size_t(n_seed) * sizeof(uint64_t), "seed list");
        if (arc) {
            sel_free(s);
            return arc;
        }
    }
    // (what the head phase needs besides the seeds goes out before the wait below, on its stream)
    if (s->persist && m->rest_beside_head && ctx->stream_head && m->head_rows_built && params->mode == DVS_MODE_NMOST &&
        !order && !labels) {
        s->used_side_streams = true;
        int prc = dvs_persist_prepare_head(ctx, s, m->head_rows_built, ctx->stream_head);
        if (prc) {
            sel_free(s);
            return prc;
        }
    }
    std::vector<uint64_t> seeds;
    {
        std::vector<uint32_t> h_tot(n_seed);
        // (a matrix whose build is still in flight is waited for HERE, with everything above -- the
        // allocations, the memsets, the engine set-up -- done while its kernels ran)
        {
            int src = dvs_matrix_settle(ctx, m);
            if (src) {
                sel_free(s);
                return src;
            }
        }
        hipError_t ce = hipSuccess;
        if (!order && n_seed && n_seed <= m->h_head_totals.size()) {
            std::copy(m->h_head_totals.begin(), m->h_head_totals.begin() + n_seed, h_tot.begin());  // no round trip
        } else if (!order && n_seed) {
            ce = hipMemcpyAsync(h_tot.data(), m->d_totals, size_t(n_seed) * 4, hipMemcpyDeviceToHost, ctx->stream);
        } else {
            for (uint64_t p = 0; p < n_seed && ce == hipSuccess; p++)
                ce = hipMemcpyAsync(&h_tot[p], m->d_totals + order[p], 4, hipMemcpyDeviceToHost, ctx->stream);
        }
        if ((order || n_seed > m->h_head_totals.size()) &&
            (hipStreamSynchronize(ctx->stream) != hipSuccess || ce != hipSuccess)) {
            sel_free(s);
            return dvs_set_error(ctx, DVS_ERR_RUNTIME, "reading the seed rows' totals failed");
        }
        for (uint64_t p = 0; p < n_seed; p++)
            if (h_tot[p] > 0) seeds.push_back(p);
    }
    if (seeds.size() < 2) {
        sel_free(s);
        return seeds.empty() ? dvs_set_error(ctx, DVS_ERR_VALUE, "records cannot be empty")   // :28-30
                             : dvs_set_error(ctx, DVS_ERR_VALUE, "must have > 1 KmerSeq");     // :227-230
    }

    SelCtl c;
    std::memset(&c, 0, sizeof c);
    c.cursor = n_seed;
    c.npos = npos;
    c.event_pos = SEL_NONE;
    c.status = SEL_RUN;  // finalize flips to DONE when cursor >= npos
    c.size = uint32_t(seeds.size());
    c.mode = params->mode;
    c.max_size = max_size;
    c.stat = params->stat;
    c.forced = FORCE_NONE;
    c.forced_lowest = 0xFFFFFFFFu;
    c.band = 0.0;
    // measured on the north-star shape: {2, 3, 4} x window_min {256..2048}, and power laws
    // wc * gap^0.5..0.75 for the early stream, are all within 3 % of each other
    c.wscale = 4.0;
    if (ctx->knobs.persist_no_events) c.wscale = 1e5;  // (the pure-stream measurement: a few long windows)
    s->ctl0 = c;  // (sel_seed adds the window policy of the engine in charge and uploads it)
    s->seed_positions = seeds;

    int rc = dvs_mat_dispatch(m, [&](auto *mp) { return sel_start(ctx, s, mp); });
    if (rc) {
        sel_free(s);
        return rc;
    }
    *out = s;
    return DVS_OK;
#undef SEL_HIP
}

extern "C" void dvs_select_destroy(dvs_select *s) { sel_free(s); }

extern "C" int dvs_select_get_summary(dvs_ctx *ctx, const dvs_select *s, dvs_select_summary *out) {
    if (!s || !out) return dvs_set_error(ctx, DVS_ERR_VALUE, "null argument");
    const SelCtl &c = *s->h_ctl;
    std::memset(out, 0, sizeof *out);
    out->size = c.size;
    out->lowest_index = c.lowest;
    out->total_jsd = c.total_jsd;
    out->mean_delta_jsd = c.mean_delta;
    out->std_delta_jsd = c.std_delta;
    out->cov_delta_jsd = c.cov_delta;
    out->summed_entropies = c.sum_entropy;
    out->rows_scored = c.rows_scored;
    out->rows_rechecked = c.rows_rechecked;
    out->n_windows = c.n_windows;
    out->n_events = c.n_events;
    out->n_accepts = c.n_accepts;
    out->n_arbitrated = s->n_arbitrated;
    out->scan_ms = s->scan_ms;
    out->scan_launches = s->scan_launches;
    out->engine = s->persist ? 1u : 0u;
    out->rows_coarse_passed = uint32_t(std::min<unsigned long long>(c.rows_coarse_passed, 0xFFFFFFFFull));
    out->scan_ms_last = s->scan_ms_last;
    out->rows_scored_last = s->persist ? c.rows_scored - std::min(c.rows_before_launch, c.rows_scored) : 0;
    out->arbiter_ms = s->arbiter_ms;
    return DVS_OK;
}

extern "C" int dvs_select_get_members(dvs_ctx *ctx, const dvs_select *s, uint64_t *positions,
                                      uint32_t *labels, double *delta_jsd, double *entropy,
                                      double *freqs) {
    if (!s) return dvs_set_error(ctx, DVS_ERR_VALUE, "null argument");
    const uint32_t n = s->h_ctl->size;
    const uint64_t B = s->dev.B;
    std::vector<uint32_t> ord(n), lab(s->cap);
    std::vector<uint64_t> pos(s->cap);
    std::vector<double> mh(s->cap);
    DVS_HIP(ctx, hipMemcpy(ord.data(), s->dev.ord, size_t(n) * 4, hipMemcpyDeviceToHost));
    DVS_HIP(ctx, hipMemcpy(lab.data(), s->dev.mLabel, size_t(s->cap) * 4, hipMemcpyDeviceToHost));
    DVS_HIP(ctx, hipMemcpy(pos.data(), s->dev.mPos, size_t(s->cap) * 8, hipMemcpyDeviceToHost));
    DVS_HIP(ctx, hipMemcpy(mh.data(), s->dev.mH, size_t(s->cap) * 8, hipMemcpyDeviceToHost));
    if (delta_jsd)  // mDelta is indexed by member order already
        DVS_HIP(ctx, hipMemcpy(delta_jsd, s->dev.mDelta, size_t(n) * 8, hipMemcpyDeviceToHost));
    for (uint32_t i = 0; i < n; i++) {
        const uint32_t sl = ord[i];
        if (positions) positions[i] = pos[sl];
        if (labels) labels[i] = s->h_out_labels.empty() ? lab[sl] : s->h_out_labels[pos[sl]];
        if (entropy) entropy[i] = mh[sl];
        if (freqs)
            DVS_HIP(ctx, hipMemcpy(freqs + uint64_t(i) * B, s->dev.M + uint64_t(sl) * B, B * 8,
                                   hipMemcpyDeviceToHost));
    }
    return DVS_OK;
}

extern "C" int dvs_select_delta_jsd(dvs_ctx *ctx, const dvs_select *s, const dvs_matrix *q,
                                    const uint32_t *qlabels, double *out) {
    if (!s || !q || !out) return dvs_set_error(ctx, DVS_ERR_VALUE, "null argument");
    if (q->nbins != s->dev.B)
        return dvs_set_error(ctx, DVS_ERR_VALUE, "query matrix has %llu bins, the set %llu",
                             (unsigned long long)q->nbins, (unsigned long long)s->dev.B);
    if (q->nrows == 0) return DVS_OK;
    DVS_HIP(ctx, hipSetDevice(ctx->device));
    double *d_out = nullptr;
    uint32_t *d_lab = nullptr;
    std::vector<uint32_t> qtrans;
    if (qlabels && !s->h_out_labels.empty()) {
        // label-free engine: a query carrying a member's (caller) label gets that member's engine
        // label (its stream position), any other query none
        const uint32_t n = s->h_ctl->size;
        std::vector<uint64_t> pos(s->cap);
        std::vector<uint32_t> ord(n);
        DVS_HIP(ctx, hipMemcpy(ord.data(), s->dev.ord, size_t(n) * 4, hipMemcpyDeviceToHost));
        DVS_HIP(ctx, hipMemcpy(pos.data(), s->dev.mPos, size_t(s->cap) * 8, hipMemcpyDeviceToHost));
        std::map<uint32_t, uint32_t> member_of;
        for (uint32_t i = 0; i < n; i++) {
            const uint64_t p = pos[ord[i]];
            if (s->h_out_labels[p] != 0xFFFFFFFFu) member_of[s->h_out_labels[p]] = uint32_t(p);
        }
        qtrans.resize(q->nrows);
        for (uint32_t i = 0; i < q->nrows; i++) {
            auto it = member_of.find(qlabels[i]);
            qtrans[i] = it == member_of.end() ? 0xFFFFFFFFu : it->second;
        }
        qlabels = qtrans.data();
    }
    DVS_HIP(ctx, hipMalloc(&d_out, size_t(q->nrows) * 8));
    if (qlabels) {
        DVS_HIP(ctx, hipMalloc(&d_lab, size_t(q->nrows) * 4));
        DVS_HIP(ctx, hipMemcpyAsync(d_lab, qlabels, size_t(q->nrows) * 4, hipMemcpyHostToDevice,
                                    ctx->stream));
    }
    dvs_mat_dispatch(q, [&](auto *qp) {
        using T = std::remove_cv_t<std::remove_pointer_t<decltype(qp)>>;
        hipLaunchKernelGGL((score_kernel<T>), dim3(q->nrows), dim3(LOO_THREADS), 0, ctx->stream, s->dev, qp,
                           q->d_totals, q->d_entropy, d_lab, d_out);
        return 0;
    });
    DVS_HIP(ctx, hipGetLastError());
    DVS_HIP(ctx, hipMemcpyAsync(out, d_out, size_t(q->nrows) * 8, hipMemcpyDeviceToHost, ctx->stream));
    DVS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    (void)hipFree(d_out);
    if (d_lab) (void)hipFree(d_lab);
    return DVS_OK;
}

// max |log2_acc(x) - log2(x)| / max(1, |log2 x|), and the same of log2_tab, over 2^17 mantissas x 80 exponents
__global__ void log2_acc_selftest_kernel(double *out) {
    __shared__ double2 tab[128];
    if (threadIdx.x < 128) log2_tab_fill(tab, threadIdx.x);
    __syncthreads();
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;  // 2^17 threads
    double worst = 0.0;
    for (int e = -78; e <= 1; e++) {
        // mantissas spread over [1, 2), plus a dense cluster around sqrt(2) and 1
        const double m = 1.0 + double(t) / 131072.0;
        const double xs[3] = {ldexp(m, e), ldexp(1.4142135623730951 + (double(t) - 65536.0) * 1e-12, e),
                              ldexp(1.0 + (double(t) - 65536.0) * 2.2e-16, e)};
        for (int q = 0; q < 3; q++) {
            const double ref = log2(xs[q]);
            const double err = fabs(log2_acc(xs[q]) - ref) / fmax(1.0, fabs(ref));
            const double err_t = fabs(log2_tab(xs[q], tab) - ref) / fmax(1.0, fabs(ref));
            worst = fmax(worst, fmax(err, err_t));
        }
    }
    for (int o = 32; o > 0; o >>= 1) worst = fmax(worst, __shfl_xor(worst, o, 64));
    if ((threadIdx.x & 63) == 0)
        atomicMax(reinterpret_cast<unsigned long long *>(out), (unsigned long long)__double_as_longlong(worst));
}

// COARSE tier (select_dev.h): max over EVERY f32 y in [2^-101, 2) of
// |v_log_f32(y) - log2 y| / (2^-23 max(1, |log2 y|)), the k of its error bound; and the exact
// quotient by fma against the division, over 2^26 (count, total) pairs per launch
__global__ void log2_f32_selftest_kernel(double *out) {
    double worst = 0.0;
    const uint64_t first = uint64_t(26) << 23, last = uint64_t(128) << 23;  // biased exponents 26 .. 127
    for (uint64_t b = first + uint64_t(blockIdx.x) * blockDim.x + threadIdx.x; b < last;
         b += uint64_t(gridDim.x) * blockDim.x) {
        const float y = __uint_as_float(uint32_t(b));
        const double ref = log2(double(y));
        const double err = fabs(double(__builtin_amdgcn_logf(y)) - ref) / (0x1p-23 * fmax(1.0, fabs(ref)));
        worst = fmax(worst, err);
    }
    for (int o = 32; o > 0; o >>= 1) worst = fmax(worst, __shfl_xor(worst, o, 64));
    if ((threadIdx.x & 63) == 0)
        atomicMax(reinterpret_cast<unsigned long long *>(out), (unsigned long long)__double_as_longlong(worst));
}

__global__ void exact_div_selftest_kernel(unsigned long long *bad) {
    const uint64_t t = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    unsigned long long nbad = 0;
    uint64_t x = (t + 1) * 0x9E3779B97F4A7C15ull;
    for (int it = 0; it < 64; it++) {
        x ^= x << 13;
        x ^= x >> 7;
        x ^= x << 17;
        uint32_t tot = uint32_t(x >> 32), c = uint32_t(x);
        if (it & 1) tot >>= (x >> 27) & 31;  // short totals too
        if (!tot) tot = 1;
        if (it & 2) c %= tot;  // frequencies proper (c <= tot), and any pair
        const double dt = double(tot);
        if (exact_div_u32(double(c), dt, 1.0 / dt) != double(c) / dt) nbad++;
    }
    // every count up to a small total, exhaustively
    const uint32_t tot = uint32_t(t % 8192) + 1, c0 = uint32_t(t / 8192);
    if (c0 <= tot) {
        const double dt = double(tot);
        if (exact_div_u32(double(c0), dt, 1.0 / dt) != double(c0) / dt) nbad++;
    }
    if (nbad) atomicAdd(bad, nbad);
}

extern "C" int dvs_selftest_log2_f32(dvs_ctx *ctx, double *max_ulps) {
    if (!ctx || !max_ulps) return dvs_set_error(ctx, DVS_ERR_VALUE, "null argument");
    DVS_HIP(ctx, hipSetDevice(ctx->device));
    double *d = nullptr;
    DVS_HIP(ctx, hipMalloc(&d, 8));
    DVS_HIP(ctx, hipMemsetAsync(d, 0, 8, ctx->stream));
    hipLaunchKernelGGL(log2_f32_selftest_kernel, dim3(4096), dim3(256), 0, ctx->stream, d);
    DVS_HIP(ctx, hipGetLastError());
    DVS_HIP(ctx, hipMemcpyAsync(max_ulps, d, 8, hipMemcpyDeviceToHost, ctx->stream));
    DVS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    (void)hipFree(d);
    return DVS_OK;
}

extern "C" int dvs_selftest_exact_div(dvs_ctx *ctx, uint64_t *mismatches) {
    if (!ctx || !mismatches) return dvs_set_error(ctx, DVS_ERR_VALUE, "null argument");
    DVS_HIP(ctx, hipSetDevice(ctx->device));
    unsigned long long *d = nullptr;
    DVS_HIP(ctx, hipMalloc(&d, 8));
    DVS_HIP(ctx, hipMemsetAsync(d, 0, 8, ctx->stream));
    hipLaunchKernelGGL(exact_div_selftest_kernel, dim3(8192 * 8192 / 256), dim3(256), 0, ctx->stream, d);
    DVS_HIP(ctx, hipGetLastError());
    DVS_HIP(ctx, hipMemcpyAsync(mismatches, d, 8, hipMemcpyDeviceToHost, ctx->stream));
    DVS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    (void)hipFree(d);
    return DVS_OK;
}

extern "C" int dvs_selftest_log2_acc(dvs_ctx *ctx, double *max_rel_err) {
    if (!ctx || !max_rel_err) return dvs_set_error(ctx, DVS_ERR_VALUE, "null argument");
    DVS_HIP(ctx, hipSetDevice(ctx->device));
    double *d = nullptr;
    DVS_HIP(ctx, hipMalloc(&d, 8));
    DVS_HIP(ctx, hipMemsetAsync(d, 0, 8, ctx->stream));
    hipLaunchKernelGGL(log2_acc_selftest_kernel, dim3(512), dim3(256), 0, ctx->stream, d);
    DVS_HIP(ctx, hipGetLastError());
    DVS_HIP(ctx, hipMemcpyAsync(max_rel_err, d, 8, hipMemcpyDeviceToHost, ctx->stream));
    DVS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    (void)hipFree(d);
    return DVS_OK;
}

extern "C" int dvs_selftest_fast_log2(dvs_ctx *ctx, double *max_abs_err) {
    if (!ctx || !max_abs_err) return dvs_set_error(ctx, DVS_ERR_VALUE, "null argument");
    DVS_HIP(ctx, hipSetDevice(ctx->device));
    double *d = nullptr;
    DVS_HIP(ctx, hipMalloc(&d, 8));
    DVS_HIP(ctx, hipMemsetAsync(d, 0, 8, ctx->stream));
    hipLaunchKernelGGL(fast_log2_selftest_kernel, dim3(16), dim3(256), 0, ctx->stream, d);
    DVS_HIP(ctx, hipGetLastError());
    DVS_HIP(ctx, hipMemcpyAsync(max_abs_err, d, 8, hipMemcpyDeviceToHost, ctx->stream));
    DVS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    (void)hipFree(d);
    return DVS_OK;
}

// ---- stepwise driving (one process per GPU; the exchange between the steps is the host
// framework's: ONE all_gather of every rank's slot per greedy step)
extern "C" int dvs_select_step_pack(dvs_ctx *ctx, dvs_select *s, double *d_slot) {
    if (!ctx || !s || !d_slot) return dvs_set_error(ctx, DVS_ERR_VALUE, "null argument");
    if (s->fast_step) {
        // the scan ran behind the previous step's apply, and that launch's last workgroup packed this slot (fs_step_kernel):
        // nothing to launch, unless there was no such launch (selection start, behind an arbitration) or it packed elsewhere
        s->fs_slot = d_slot;
        if (!s->fs_need_scan && s->fs_packed == d_slot) return DVS_OK;
        int lrc = DVS_OK;
        dvs_mat_dispatch(s->mat, [&](auto *mp) {
            using T = std::remove_cv_t<std::remove_pointer_t<decltype(mp)>>;
            if (s->fs_need_scan) {
                lrc = dvs_raise_dyn_lds(ctx, reinterpret_cast<const void *>(fs_step_kernel<T>), s->fs_lds);
                // (no jobs kernel before this launch: the event word's copies are cleared here -- SEL_NONE is all ones)
                if (!lrc && hipMemsetAsync(&static_cast<FsSync *>(s->d_fsync)->hint, 0xFF, sizeof(FsLine) * 8, ctx->stream) != hipSuccess)
                    lrc = dvs_set_error(ctx, DVS_ERR_RUNTIME, "hipMemsetAsync of the event word's copies failed");
                if (!lrc)
                    hipLaunchKernelGGL((fs_step_kernel<T>), dim3(s->fs_grid), dim3(FS_THREADS), s->fs_lds, ctx->stream, s->dev, mp,
                                       s->d_jobres, s->fs_K, 0, static_cast<FsSync *>(s->d_fsync), s->cap, d_slot, (unsigned long long *)nullptr);
            } else {
                hipLaunchKernelGGL((pack_event_kernel<T>), dim3(1), dim3(LOO_THREADS), 0, ctx->stream, s->dev, mp, d_slot);
            }
            return 0;
        });
        if (lrc) return lrc;
        s->fs_need_scan = false;
        s->fs_packed = d_slot;
        DVS_HIP(ctx, hipGetLastError());
        return DVS_OK;
    }
    dvs_mat_dispatch(s->mat, [&](auto *mp) {
        using T = std::remove_cv_t<std::remove_pointer_t<decltype(mp)>>;
        // (scan and pack as ONE launch -- the last workgroup to arrive at a counter packs -- measured
        // slower: 8.79 vs 7.57 ms per selection; every workgroup's arrival is a same-address atomic)
        const SelDev &d = s->dev;
        hipLaunchKernelGGL((scan_kernel<T, false>), dim3(s->scan_grid), dim3(SCAN_THREADS), s->scan_lds,
                           ctx->stream, d.ctl, mp, d.totals, d.rowH, d.order, d.labels, d.inset, d.nlabels,
                           d.base, d.wg_rows, d.B, s->base_in_lds ? 1 : 0);
        hipLaunchKernelGGL((pack_event_kernel<T>), dim3(1), dim3(LOO_THREADS), 0, ctx->stream, s->dev, mp,
                           d_slot);
        return 0;
    });
    DVS_HIP(ctx, hipGetLastError());
    return DVS_OK;
}

extern "C" int dvs_select_step_apply(dvs_ctx *ctx, dvs_select *s, const double *d_all, uint32_t world) {
    if (!ctx || !s || !d_all || !world) return dvs_set_error(ctx, DVS_ERR_VALUE, "null argument");
    if (s->dev.rowlog && ++s->steps_since_poll > s->dev.rowlog_cap / 2)
        return dvs_set_error(ctx, DVS_ERR_VALUE, "dvs_select_step_poll must be called at least every %u steps (it drains the "
                             "accepted rows' log)", s->dev.rowlog_cap / 2);
    s->dev.gather_all = d_all;
    s->dev.gather_world = world;
    if (s->fast_step) {
        int lrc = DVS_OK;
        dvs_mat_dispatch(s->mat, [&](auto *mp) {
            using T = std::remove_cv_t<std::remove_pointer_t<decltype(mp)>>;
            lrc = dvs_raise_dyn_lds(ctx, reinterpret_cast<const void *>(fs_step_kernel<T>), s->fs_lds);
            if (lrc) return 0;
            // (a set of ctl->size <= cap - 1 members: at most cap * K jobs; a workgroup beyond the set's jobs leaves at once)
            hipLaunchKernelGGL(fs_jobs_kernel, dim3(s->cap * s->fs_K), dim3(FS_JOB_THREADS), 0, ctx->stream, s->dev, s->d_jobres, s->fs_K,
                               static_cast<FsSync *>(s->d_fsync));
            hipLaunchKernelGGL((fs_step_kernel<T>), dim3(s->fs_grid), dim3(FS_THREADS), s->fs_lds, ctx->stream, s->dev, mp,
                               s->d_jobres, s->fs_K, 1, static_cast<FsSync *>(s->d_fsync), s->cap, s->fs_slot, s->h_fshist);
            s->fs_launches++;
            s->fs_packed = s->fs_slot;  // (the caller's slot of the last dvs_select_step_pack: the next one finds it packed)
            return 0;
        });
        if (lrc) return lrc;
        DVS_HIP(ctx, hipGetLastError());
        return DVS_OK;
    }
    dvs_mat_dispatch(s->mat, [&](auto *mp) {
        launch_iteration(ctx, s, mp, 1);
        return 0;
    });
    DVS_HIP(ctx, hipGetLastError());
    return DVS_OK;
}

extern "C" int dvs_select_step_peek(dvs_ctx *ctx, dvs_select *s, uint32_t lag, uint32_t *status, int *must_poll) {
    if (!ctx || !s || !status || !must_poll) return dvs_set_error(ctx, DVS_ERR_VALUE, "null argument");
    if (!s->fast_step || !s->h_fshist || lag == 0 || lag >= FS_HIST / 2)
        return dvs_set_error(ctx, DVS_ERR_UNSUPPORTED, "this selection keeps no status history (or the lag is out of range): poll");
    *must_poll = s->dev.rowlog && s->steps_since_poll + lag > s->dev.rowlog_cap / 2;
    *status = SEL_RUN;
    if (s->fs_launches <= lag) return DVS_OK;  // (nothing that far back yet)
    const unsigned long long L = s->fs_launches - lag;
    if (L <= s->fs_peek_floor) return DVS_OK;  // (a launch the last poll has already accounted for: no-ops behind a stop among them)
    const volatile unsigned long long *w = s->h_fshist + L % FS_HIST;
    // (a launch that far back has long finished when the driver keeps `lag` steps queued: the loop is for the first looks)
    const auto t0 = std::chrono::steady_clock::now();
    unsigned long long v;
    unsigned spins = 0;
    while (((v = *w) >> 8) != L) {
        if ((++spins & 1023u) == 0) {
            if (hipStreamQuery(ctx->stream) == hipSuccess && ((v = *w) >> 8) != L)
                return dvs_set_error(ctx, DVS_ERR_RUNTIME, "the status history holds no word for launch %llu though the stream is idle", L);
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 30.0)
                return dvs_set_error(ctx, DVS_ERR_RUNTIME, "no status word for launch %llu after 30 s", L);
        }
    }
    *status = uint32_t(v & 0xFFu);
    return DVS_OK;
}

extern "C" int dvs_select_step_poll(dvs_ctx *ctx, dvs_select *s, uint32_t *status, uint64_t *cursor) {
    if (!ctx || !s) return dvs_set_error(ctx, DVS_ERR_VALUE, "null argument");
    int rc = sel_poll(ctx, s);
    if (rc) return rc;
    if (status) *status = s->h_ctl->status;
    if (cursor) *cursor = s->h_ctl->cursor;
    if (s->dev.rowlog) s->steps_since_poll = 0;
    s->fs_peek_floor = s->fs_launches;
    // the rows logged since the last drain: ring -> host (the stream is idle: sel_poll waited).  Put off while less than
    // half the ring is waiting -- at most half a ring's worth of steps, a row each, can pass before the next look.
    if (s->dev.rowlog && (s->h_ctl->n_logged - s->rowlog_have >= s->dev.rowlog_cap / 2 || s->h_ctl->status == SEL_ARBITER)) {
        const uint64_t B = s->dev.B, ring = s->dev.rowlog_cap, logged = s->h_ctl->n_logged;
        if (logged - s->rowlog_have > ring)
            return dvs_set_error(ctx, DVS_ERR_RUNTIME, "the accepted rows' log was overrun (%llu rows since the last poll, ring of %llu)",
                                 (unsigned long long)(logged - s->rowlog_have), (unsigned long long)ring);
        s->h_rowlog.resize(size_t(logged) * B);
        while (s->rowlog_have < logged) {
            const uint64_t at = s->rowlog_have % ring;
            const uint64_t cnt = std::min<uint64_t>(logged - s->rowlog_have, ring - at);
            DVS_HIP(ctx, hipMemcpy(s->h_rowlog.data() + s->rowlog_have * B, s->dev.rowlog + at * B, size_t(cnt) * B * 8,
                                   hipMemcpyDeviceToHost));
            s->rowlog_have += cnt;
        }
    }
    if (s->h_ctl->status == SEL_ARBITER) {
        // A decision inside the rounding band: every rank holds the same replicated state, the same row
        // log and the same pending candidate, so every rank's arbiter reaches the same verdict on its own
        // -- no exchange.  The steps enqueued since the engine stopped were no-ops on every rank alike.
        if (s->params.flags & DVS_SELECT_NO_ARBITER)
            return dvs_set_error(ctx, DVS_ERR_UNSUPPORTED,
                                 "ambiguous decision at stream position %llu (stage %u): "
                                 "|score - threshold| within the rounding band",
                                 (unsigned long long)s->h_ctl->arb_pos, s->h_ctl->arb_stage);
        const uint32_t stage = s->h_ctl->arb_stage;
        {
            const auto t_arb = std::chrono::steady_clock::now();
            rc = dvs_select_arbitrate(ctx, s);
            s->arbiter_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_arb).count();
        }
        if (rc) return rc;
        dvs_mat_dispatch(s->mat, [&](auto *mp) {
            launch_iteration(ctx, s, mp, stage == ARB_RESOLVE ? 1 : 2);
            return 0;
        });
        DVS_HIP(ctx, hipGetLastError());
        s->fs_need_scan = true;  // (the kernels just queued leave no event posted: the fast step scans before it packs)
        s->fs_packed = nullptr;
        if (status) *status = SEL_RUN;
    }
    return DVS_OK;
}

// Measurement aid: ONE scan_kernel launch over every streamed row against the set's current
// state with an unreachable threshold (no events), timed with HIP events on the ctx stream.
// This is the steady-state streaming rate of the scan arithmetic, without the greedy chain's
// per-event latencies.  The selection's state is not touched (a scratch control block is used).
extern "C" int dvs_select_bench_scan(dvs_ctx *ctx, const dvs_select *s, int repeats, double *ms_out,
                                     uint64_t *rows_out) {
    if (!ctx || !s || !ms_out || !rows_out || repeats < 1)
        return dvs_set_error(ctx, DVS_ERR_VALUE, "bad argument");
    DVS_HIP(ctx, hipSetDevice(ctx->device));
    SelCtl c = *s->h_ctl;
    const uint64_t first = s->params.n_seed;
    if (s->npos <= first) return dvs_set_error(ctx, DVS_ERR_VALUE, "nothing to scan");
    c.status = SEL_RUN;
    c.cursor = first;
    c.window = uint32_t(std::min<uint64_t>(s->npos - first, 0xFFFFFFFFull));
    c.event_pos = SEL_NONE;
    c.thr = 1e300;
    c.ev_kind = 0;
    SelCtl *d_c = nullptr;
    uint32_t *d_rows = nullptr;
    int rc = dvs_dev_alloc(ctx, (void **)&d_c, sizeof(SelCtl), "scratch control block");
    if (!rc) rc = dvs_dev_alloc(ctx, (void **)&d_rows, size_t(s->scan_grid) * 8, "scratch row counters");
    if (rc) {
        dvs_dev_free(ctx, d_c);
        return rc;
    }
    hipEvent_t e0 = dvs_event_get(ctx), e1 = dvs_event_get(ctx);
    const SelDev &d = s->dev;
    auto launch = [&]() {
        dvs_mat_dispatch(s->mat, [&](auto *mp) {
            using T = std::remove_cv_t<std::remove_pointer_t<decltype(mp)>>;
            if (s->scan_hot)
                hipLaunchKernelGGL((scan_kernel<T, true>), dim3(s->scan_grid), dim3(SCAN_THREADS), s->scan_lds,
                                   ctx->stream, d_c, mp, d.totals, d.rowH, d.order, d.labels, d.inset, d.nlabels,
                                   d.base, d_rows, d.B, s->base_in_lds ? 1 : 0);
            else
                hipLaunchKernelGGL((scan_kernel<T, false>), dim3(s->scan_grid), dim3(SCAN_THREADS), s->scan_lds,
                                   ctx->stream, d_c, mp, d.totals, d.rowH, d.order, d.labels, d.inset, d.nlabels,
                                   d.base, d_rows, d.B, s->base_in_lds ? 1 : 0);
            return 0;
        });
    };
    hipError_t e = hipMemcpyAsync(d_c, &c, sizeof c, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(d_rows, 0, size_t(s->scan_grid) * 8, ctx->stream);
    launch();  // warm-up
    if (e == hipSuccess) e = hipEventRecord(e0, ctx->stream);
    for (int i = 0; i < repeats; i++) launch();
    if (e == hipSuccess) e = hipEventRecord(e1, ctx->stream);
    if (e == hipSuccess) e = hipEventSynchronize(e1);
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    if (e == hipSuccess) e = hipGetLastError();
    dvs_event_put(ctx, e0);
    dvs_event_put(ctx, e1);
    dvs_dev_free(ctx, d_c);
    dvs_dev_free(ctx, d_rows);
    if (e != hipSuccess) return dvs_hip_fail(ctx, e, "scan benchmark");
    *ms_out = double(ms) / repeats;
    *rows_out = s->npos - first;
    return DVS_OK;
}

// Members in set order into caller-provided DEVICE buffers: rows[cap_rows x nbins] and
// meta[cap_rows x 2] = (stream position, 1.0) -- rows beyond the set's size are zeroed with
// meta (0, 0).  Enqueued on the ctx stream; the caller orders it against its own streams.
extern "C" int dvs_select_gather_members(dvs_ctx *ctx, const dvs_select *s, double *d_rows, double *d_meta,
                                         uint32_t cap_rows) {
    if (!ctx || !s || !d_rows || !d_meta) return dvs_set_error(ctx, DVS_ERR_VALUE, "null argument");
    if (cap_rows < s->h_ctl->size)
        return dvs_set_error(ctx, DVS_ERR_VALUE, "buffer of %u rows for a set of %u", cap_rows, s->h_ctl->size);
    if (!cap_rows) return DVS_OK;
    DVS_HIP(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(gather_members_kernel, dim3(cap_rows), dim3(LOO_THREADS), 0, ctx->stream, s->dev,
                       d_rows, d_meta);
    DVS_HIP(ctx, hipGetLastError());
    return DVS_OK;
}
