// Device helpers shared by the selection kernels (select.hip, persist.hip).
#pragma once

#include "select.h"

namespace {

// sum_{x>0} -x log2 x over the bins a thread owns, plus what the reference's
// entropy() (src/record.rs:86-106) is sensitive to: a negative bin (log2 -> NaN)
// and the sum of the bins (the |sum - 1| <= len * eps check).
// log2 for the f64 ("precise") evaluations: ~32 instructions instead of ocml's ~85
// (which is double-double for <= 1 ulp).  x = m 2^e, m in [sqrt(.5), sqrt(2)),
// s = (m-1)/(m+1), log m = 2s (1 + s^2/3 + ... + s^18/19); |s| <= 0.1716 so the
// series is cut below 1e-17.  Error ~1 ulp of the result (~2e-15 absolute for the
// |log2 x| ~ 12 met here), far inside the decision band (4 B eps H ~ 4e-11 at k=6).
// Only called with x > 0 (normal).
__device__ __forceinline__ double log2_acc(double x) {
    double m = __builtin_amdgcn_frexp_mant(x);  // [0.5, 1)
    int e = __builtin_amdgcn_frexp_exp(x);
    if (m < 0.70710678118654752) {
        m *= 2.0;
        e -= 1;
    }
    const double f = m - 1.0;
    const double s = f / (2.0 + f);
    const double z = s * s;
    double p = 1.0 / 19.0;
    p = fma(p, z, 1.0 / 17.0);
    p = fma(p, z, 1.0 / 15.0);
    p = fma(p, z, 1.0 / 13.0);
    p = fma(p, z, 1.0 / 11.0);
    p = fma(p, z, 1.0 / 9.0);
    p = fma(p, z, 1.0 / 7.0);
    p = fma(p, z, 1.0 / 5.0);
    p = fma(p, z, 1.0 / 3.0);
    const double lm = fma(s * z, p, s);  // log(m) / 2
    return fma(lm, 2.8853900817779268, double(e));  // 2 / ln 2
}

// The same number by table: the mantissa m in [0.5, 1) is split by its top seven fraction bits,
// m = c_i (1 + r) with c_i the middle of interval i (|r| <= 2^-8), so that
//   log2 x = e + log2 c_i + log(1 + r) / ln 2,   log(1 + r) = r - r^2/2 + ... + r^7/7  (r^8/8 < 7e-21).
// tab[i] = (log2 c_i, 1 / c_i): 128 entries, 2 KB of LDS, filled by log2_tab_fill.  No division and
// about half the instructions of log2_acc; the same ~1 ulp of the result (the rounding of
// e + log2 c_i dominates both), measured side by side by dvs_selftest_log2_acc.
__device__ __forceinline__ void log2_tab_fill(double2 *tab, int i) {  // one thread per entry, i < 128
    const double c = 0.5 + (double(i) + 0.5) / 256.0;
    tab[i] = make_double2(log2_acc(c), 1.0 / c);
}
__device__ __forceinline__ double log2_tab(double x, const double2 *tab) {
    const double m = __builtin_amdgcn_frexp_mant(x);  // [0.5, 1)
    const int e = __builtin_amdgcn_frexp_exp(x);
    const uint32_t i = (uint32_t(__double2hiint(m)) >> 13) & 127u;
    const double2 t = tab[i];
    const double r = fma(m, t.y, -1.0);
    double p = 1.0 / 7.0;
    p = fma(p, r, -1.0 / 6.0);
    p = fma(p, r, 1.0 / 5.0);
    p = fma(p, r, -1.0 / 4.0);
    p = fma(p, r, 1.0 / 3.0);
    p = fma(p, r, -1.0 / 2.0);
    p = fma(p, r, 1.0);
    return fma(p * r, 1.4426950408889634, double(e) + t.x);  // 1 / ln 2
}

struct Ent {
    double h = 0.0, sum = 0.0, mn = 0.0;
    __device__ __forceinline__ void add(double x) {
        if (x > 0.0) h -= x * log2_acc(x);
        sum += x;
        mn = fmin(mn, x);
    }
    __device__ __forceinline__ void add(double x, const double2 *tab) {  // the same through log2_tab
        if (x > 0.0) h -= x * log2_tab(x, tab);
        sum += x;
        mn = fmin(mn, x);
    }
};

__device__ __forceinline__ uint64_t umin64(uint64_t a, uint64_t b) { return a < b ? a : b; }

// Width of the zone in which the device's score and the reference's (sequential
// f64 sum over B bins, src/record.rs:92-98) cannot be told apart: the reference's
// own worst-case summation error is ~ B * eps/2 * H.
__device__ __forceinline__ double sel_band(double hmean, uint64_t B) {
    return 4.0 * double(B) * DVS_EPS * fmax(1.0, fabs(hmean));
}

__device__ __forceinline__ double row_value(const uint32_t *row, uint64_t i) { return double(row[i]); }
__device__ __forceinline__ double row_value(const uint16_t *row, uint64_t i) { return double(row[i]); }
__device__ __forceinline__ double row_value(const double *row, uint64_t i) { return row[i]; }

// ---------------------------------------------------------------- scan kernel
// One wavefront per candidate row; lane l owns bins 4*(j*64 + l) .. +3 so that a
// wave instruction reads 1 KiB (16 B per lane) of the row.  The per-state vector
// b_i = (S_i - low_i) / size is staged once per workgroup in LDS (8 waves share it).
//   x_i = b_i + c_i / (total * size)      (reference: (S_i - low_i + f_i) / size)
//   jsd = sum_i -x_i log2 x_i - (sumH - H_low + H_c) / size
//
// Two tiers.  FAST: x in f64, log2 x = exponent + v_log_f32(mantissa) -- the
// mantissa is rounded to f32 (<= 2^-24 relative -> <= 8.6e-8 in log2) and
// v_log_f32 is good to ~1 ulp of a result in [-1, 0] (<= 6e-8), so each log is
// off by < 1.5e-7 and, as sum x_i = 1, so is the row's entropy (FAST_BAND below;
// the bound on v_log_f32 is measured exhaustively by dvs_selftest_fast_log2).
// A row whose fast score clears thr + band + FAST_BAND is an event outright; a
// row within FAST_BAND (+ band) of the threshold is re-evaluated by the same wave
// in full f64 (PRECISE) and is an event <=> precise jsd > thr - band.  Either way
// the resolve kernel re-evaluates the first event in f64 before acting (NaN
// compares false everywhere: the reference rejects too).
constexpr double FAST_BAND = 4e-7;
constexpr int SCAN_CH = 16;  // chunks (1 KiB per wave instruction) requested per batch

__device__ __forceinline__ double fast_neg_xlog2x(double x) {
    const double xm = fmax(x, 1e-300);  // x <= 0 contributes ~0 here; sign handled via min(x)
    const double m = __builtin_amdgcn_frexp_mant(xm);       // [0.5, 1)
    const int e = __builtin_amdgcn_frexp_exp(xm);
    const float l = __builtin_amdgcn_logf(float(m));        // v_log_f32 = log2
    return -xm * (double(e) + double(l));
}

// COARSE tier (count matrices, persistent engine): everything in f32 -- y = c * (1 / (T n)) + sl / n
// by packed fma, v_log_f32 on the whole value, the four products of a chunk added in f32 -- and one
// f64 add per four bins: ~22 issue cycles per bin where the FAST tier needs ~127.  With e = 2^-24
// and H = the entropy of the mean vector (<= log2 B), its score is off by at most
//   3 e (H + 1.45)   y: rounding of sl / n, of 1 / (T n) (or of a count >= 2^24) and of the fma;
//                    d(-y log2 y)/dy = -(log2 y + 1 / ln 2), and sum y = 1
// + 2 k e (H + 1)    v_log_f32: k ulp of a result of magnitude max(1, |log2 y|); k <= 1.5 is asserted
//                    over EVERY f32 in [2^-101, 2) by dvs_selftest_log2_f32
// + e H              the f32 product y * log2 y
// + 2 e H            the two f32 additions that fold four products
// = e ((6 + 2k) H + 4.35 + 2k) <= e (9 log2 B + 7.4); COARSE_BAND adds a quarter on top.
// A row scoring beyond thr +- (band + COARSE_BAND) is decided here; anything closer is scored again
// by the FAST tier.  A bin of sl that is exactly zero is stored as 1e-30 (contributes ~1e-28), so
// log2 never sees a zero; a negative bin makes v_log_f32 return NaN, the score NaN, and the row is
// rejected as the reference rejects it (record.rs:95: log2 of a negative is NaN, NaN > x is false).
typedef float dvs_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double coarse_band(uint64_t B) {
    return 1.25 * 0x1p-24 * (9.0 * log2(double(B)) + 7.4);
}
__device__ __forceinline__ float coarse_sl(double v, double rn) { return v == 0.0 ? 1e-30f : float(v * rn); }
// sum of y log2 y over 4 consecutive bins (<= 0)
__device__ __forceinline__ float coarse4(const uint4 c, const float4 b, const dvs_f2 r2) {
    const dvs_f2 c01 = {float(c.x), float(c.y)}, c23 = {float(c.z), float(c.w)};
    const dvs_f2 y01 = __builtin_elementwise_fma(c01, r2, (dvs_f2){b.x, b.y});
    const dvs_f2 y23 = __builtin_elementwise_fma(c23, r2, (dvs_f2){b.z, b.w});
    const dvs_f2 l01 = {__builtin_amdgcn_logf(y01.x), __builtin_amdgcn_logf(y01.y)};
    const dvs_f2 l23 = {__builtin_amdgcn_logf(y23.x), __builtin_amdgcn_logf(y23.y)};
    const dvs_f2 s = y01 * l01 + y23 * l23;
    return s.x + s.y;
}

// the same for a row of 16-bit counts: four bins in two words
__device__ __forceinline__ float coarse4(const uint2 c, const float4 b, const dvs_f2 r2) {
    const dvs_f2 c01 = {float(c.x & 0xFFFFu), float(c.x >> 16)}, c23 = {float(c.y & 0xFFFFu), float(c.y >> 16)};
    const dvs_f2 y01 = __builtin_elementwise_fma(c01, r2, (dvs_f2){b.x, b.y});
    const dvs_f2 y23 = __builtin_elementwise_fma(c23, r2, (dvs_f2){b.z, b.w});
    const dvs_f2 l01 = {__builtin_amdgcn_logf(y01.x), __builtin_amdgcn_logf(y01.y)};
    const dvs_f2 l23 = {__builtin_amdgcn_logf(y23.x), __builtin_amdgcn_logf(y23.y)};
    const dvs_f2 s = y01 * l01 + y23 * l23;
    return s.x + s.y;
}

// eight consecutive 16-bit counts from ONE 16-byte load (a wave instruction reads 1 KiB of the row, as
// it does for 32-bit counts): two groups of four, each folded in f32 and added in f64 like coarse4's
__device__ __forceinline__ void coarse8(const uint4 c, const float *b, const dvs_f2 r2, double &a0, double &a1) {
    a0 += double(coarse4(make_uint2(c.x, c.y), *reinterpret_cast<const float4 *>(b), r2));
    a1 += double(coarse4(make_uint2(c.z, c.w), *reinterpret_cast<const float4 *>(b + 4), r2));
}

// 4 consecutive bins as they sit in memory (converted to f64 only when consumed, so a
// batch of in-flight chunks costs 4 VGPRs each for a count matrix)
template <typename T> struct Raw4;
template <> struct Raw4<uint32_t> {
    uint4 c;
    __device__ __forceinline__ void load(const uint32_t *p) { c = *reinterpret_cast<const uint4 *>(p); }
    // keeps the four registers as loaded up to this point (see the burst loops)
    __device__ __forceinline__ void pin() { asm volatile("" : "+v"(c.x), "+v"(c.y), "+v"(c.z), "+v"(c.w)); }
    __device__ __forceinline__ void get(double &v0, double &v1, double &v2, double &v3) const {
        v0 = double(c.x); v1 = double(c.y); v2 = double(c.z); v3 = double(c.w);
    }
};
template <> struct Raw4<uint16_t> {  // 16-bit counts (rows of whole sequences): 8 bytes per four bins
    uint2 c;
    __device__ __forceinline__ void load(const uint16_t *p) { c = *reinterpret_cast<const uint2 *>(p); }
    __device__ __forceinline__ void pin() { asm volatile("" : "+v"(c.x), "+v"(c.y)); }
    __device__ __forceinline__ void get(double &v0, double &v1, double &v2, double &v3) const {
        v0 = double(c.x & 0xFFFFu); v1 = double(c.x >> 16); v2 = double(c.y & 0xFFFFu); v3 = double(c.y >> 16);
    }
};
template <> struct Raw4<double> {
    double2 a, b;
    __device__ __forceinline__ void load(const double *p) {
        a = *reinterpret_cast<const double2 *>(p);
        b = *reinterpret_cast<const double2 *>(p + 2);
    }
    __device__ __forceinline__ void pin() {}
    __device__ __forceinline__ void get(double &v0, double &v1, double &v2, double &v3) const {
        v0 = a.x; v1 = a.y; v2 = b.x; v3 = b.y;
    }
};

template <typename T>
__device__ __forceinline__ void load4(const T *rp, uint64_t i, double &v0, double &v1, double &v2,
                                      double &v3) {
    if constexpr (sizeof(T) == 4) {
        const uint4 c = *reinterpret_cast<const uint4 *>(rp + i);
        v0 = double(c.x); v1 = double(c.y); v2 = double(c.z); v3 = double(c.w);
    } else if constexpr (sizeof(T) == 2) {
        const uint2 c = *reinterpret_cast<const uint2 *>(rp + i);
        v0 = double(c.x & 0xFFFFu); v1 = double(c.x >> 16); v2 = double(c.y & 0xFFFFu); v3 = double(c.y >> 16);
    } else {
        const double2 a = *reinterpret_cast<const double2 *>(rp + i);
        const double2 b = *reinterpret_cast<const double2 *>(rp + i + 2);
        v0 = a.x; v1 = a.y; v2 = b.x; v3 = b.y;
    }
}

// per-launch constants of the scan, read once from the control block
struct ScanState {
    uint64_t cursor, nrows;
    double thr_lo, thr_fast, thr_sure, he_base, dsize;
};

__device__ __forceinline__ void fast4(const double2 b01, const double2 b23, double v0, double v1,
                                      double v2, double v3, double rinv, double &a0, double &a1,
                                      double &a2, double &a3, double &xmin) {
    const double x0 = fma(v0, rinv, b01.x), x1 = fma(v1, rinv, b01.y);
    const double x2 = fma(v2, rinv, b23.x), x3 = fma(v3, rinv, b23.y);
    a0 += fast_neg_xlog2x(x0);
    a1 += fast_neg_xlog2x(x1);
    a2 += fast_neg_xlog2x(x2);
    a3 += fast_neg_xlog2x(x3);
    xmin = fmin(fmin(xmin, fmin(x0, x1)), fmin(x2, x3));
}

// PRECISE tier for one row (rare): same bins, f64 log2; true if it clears thr - band
template <typename T>
__device__ __forceinline__ bool precise_row(const T *rp, const double *bvec, uint64_t B, double rinv,
                                            double mean_entropy, double thr_lo, uint32_t lane) {
    Ent e;
    if ((B & 255) == 0) {
        for (uint64_t i0 = 0; i0 < B; i0 += 256) {
            const uint64_t i = i0 + lane * 4;
            double v0, v1, v2, v3;
            load4(rp, i, v0, v1, v2, v3);
            e.add(fma(v0, rinv, bvec[i]));
            e.add(fma(v1, rinv, bvec[i + 1]));
            e.add(fma(v2, rinv, bvec[i + 2]));
            e.add(fma(v3, rinv, bvec[i + 3]));
        }
    } else {
        for (uint64_t i = lane; i < B; i += 64) e.add(fma(row_value(rp, i), rinv, bvec[i]));
    }
    const double h = dvs_wave_sum(e.h);
    return h - mean_entropy > thr_lo;
}

__device__ __forceinline__ double cand_freq(const uint32_t *row, uint64_t i, double tot) {
    return double(row[i]) / tot;  // record.rs:139, correctly rounded
}
__device__ __forceinline__ double cand_freq(const uint16_t *row, uint64_t i, double tot) { return double(row[i]) / tot; }
__device__ __forceinline__ double cand_freq(const double *row, uint64_t i, double) { return row[i]; }

// The same quotient without the ~35-instruction f64 division: with rt = RN(1 / tot) (one real
// division per row), q0 = RN(c rt) is within 2 ulp of c / tot, r0 = c - tot q0 is exact in an fma,
// and q0 + r0 rt differs from c / tot by less than 2^-104 relative.  c and tot are integers below
// 2^32, so c / tot is either exactly representable or at least 2^-32 ulp away from every rounding
// midpoint: RN(q0 + r0 rt) = RN(c / tot), the correctly rounded quotient of record.rs:139
// (checked against the division for every c <= tot <= 20000 and 4e8 random pairs on the CPU, and
// on the device by dvs_selftest_exact_div).
__device__ __forceinline__ double exact_div_u32(double c, double tot, double rt) {
    const double q0 = c * rt;
    const double r0 = fma(-tot, q0, c);
    return fma(r0, rt, q0);
}
__device__ __forceinline__ double cand_freq_x(const uint32_t *row, uint64_t i, double tot, double rt) {
    return exact_div_u32(double(row[i]), tot, rt);
}
__device__ __forceinline__ double cand_freq_x(const uint16_t *row, uint64_t i, double tot, double rt) {
    return exact_div_u32(double(row[i]), tot, rt);
}
__device__ __forceinline__ double cand_freq_x(const double *row, uint64_t i, double, double) { return row[i]; }
__device__ __forceinline__ double count_freq_x(uint32_t c, double tot, double rt) { return exact_div_u32(double(c), tot, rt); }
__device__ __forceinline__ double count_freq_x(uint16_t c, double tot, double rt) { return exact_div_u32(double(c), tot, rt); }
__device__ __forceinline__ double count_freq_x(double f, double, double) { return f; }

// (sum, min, sum) over the block in ONE barrier pair; every thread gets the result.
// scratch: >= 3 * 16 doubles.  Fixed tree -> same inputs, same bits.
__device__ __forceinline__ void block_red3(double &h, double &mn, double &sm, double *scratch) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nwave = (blockDim.x + 63) >> 6;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {  // the three butterflies interleaved: one latency chain
        const double t0 = __shfl_xor(h, o, 64), t1 = __shfl_xor(mn, o, 64), t2 = __shfl_xor(sm, o, 64);
        h += t0;
        mn = fmin(mn, t1);
        sm += t2;
    }
    __syncthreads();  // scratch may still be read from a previous call
    if (lane == 0) {
        scratch[wave] = h;
        scratch[16 + wave] = mn;
        scratch[32 + wave] = sm;
    }
    __syncthreads();
    double a = 0.0, b = scratch[16], c = 0.0;
    for (int i = 0; i < nwave; i++) {
        a += scratch[i];
        b = fmin(b, scratch[16 + i]);
        c += scratch[32 + i];
    }
    h = a;
    mn = b;
    sm = c;
}

// accept probability at stream position i is ~ size / i: widen the window as events
// thin out (bounded by what one launch covers)
__device__ __forceinline__ uint32_t sel_next_window(uint64_t cursor, uint32_t size, uint32_t wmin,
                                                    uint32_t wmax, double wscale) {
    uint64_t w = uint64_t(double(cursor) * wscale / double(size ? size : 1u));
    if (w < wmin) w = wmin;
    if (w > wmax) w = wmax;
    return uint32_t(w);
}
__device__ __forceinline__ void ctl_next_window(SelCtl *ctl) {
    ctl->window = sel_next_window(ctl->cursor, ctl->size, ctl->window_min, ctl->window_max, ctl->wscale);
}

// sum-to-one guard of entropy() (record.rs:99-104): the device cannot decide a
// borderline case, so anything past a quarter of the tolerance goes to the arbiter
__device__ __forceinline__ bool sum_risky(double sum, uint64_t B) {
    return !(fabs(sum - 1.0) <= 0.25 * double(B) * DVS_EPS);
}

// Resolves the first event of the window: fetch the candidate, re-evaluate its
// score with the reference's per-bin operation order, decide, and apply
// replace_lowest (or stage a tentative push for MODE_MAX).  One 1024-thread block.

// five values at once: (sum, min, sum, sum, sum); scratch >= 5 * 16 doubles
__device__ __forceinline__ void block_red5(double &a, double &mn, double &b, double &c, double &e,
                                           double *scratch) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nwave = (blockDim.x + 63) >> 6;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double t0 = __shfl_xor(a, o, 64), t1 = __shfl_xor(mn, o, 64), t2 = __shfl_xor(b, o, 64);
        const double t3 = __shfl_xor(c, o, 64), t4 = __shfl_xor(e, o, 64);
        a += t0;
        mn = fmin(mn, t1);
        b += t2;
        c += t3;
        e += t4;
    }
    __syncthreads();
    if (lane == 0) {
        scratch[wave] = a;
        scratch[16 + wave] = mn;
        scratch[32 + wave] = b;
        scratch[48 + wave] = c;
        scratch[64 + wave] = e;
    }
    __syncthreads();
    double ra = 0.0, rm = scratch[16], rb = 0.0, rc = 0.0, re = 0.0;
    for (int i = 0; i < nwave; i++) {
        ra += scratch[i];
        rm = fmin(rm, scratch[16 + i]);
        rb += scratch[32 + i];
        rc += scratch[48 + i];
        re += scratch[64 + i];
    }
    a = ra;
    mn = rm;
    b = rb;
    c = rc;
    e = re;
}

}  // namespace
