// k-mer histogram build: sequences (1 byte/base in HBM) -> N x num_states^k
// uint32 count matrix + per-row total and Shannon entropy.
//
// Replaces count_kmers / count_monomers / SeqRecord::to_kcounts / to_kmerseq /
// entropy of the reference (src/record.rs:31-141).  Semantics restated: bin
// idx(w) = sum_i w[i] * ns^(k-1-i) is incremented for every length-k window w
// of the sequence in which every symbol is < num_states (record.rs:47-64 is an
// incremental way of skipping exactly the windows that contain an invalid
// symbol; record.rs:72-74 is the rolling form of the same index).
//
// Layout / mapping (gfx950):
//   * one 256-thread workgroup per TILE of k-mer end positions of ONE sequence;
//     a sequence up to TILE_LEN bases is a single tile and its row is written
//     once with coalesced 16-B stores; longer sequences (genomes) are split
//     into tiles whose LDS histograms are merged with global u32 atomics (tiles of
//     up to 16 windows per bin for large inputs: dvs_hist_prepare);
//   * each lane owns 16-byte chunks (global_load_dwordx4, 1 KiB per wave
//     instruction) plus the previous chunk as the k-1 halo; for num_states == 4
//     the 32 bases are packed to a 64-bit 2-bit word and a 32-bit invalid mask
//     with word-wide bit tricks, so every k-mer index is two shifts and a mask
//     with no serial dependency between positions;
//   * the histogram lives in LDS (4^k * 4 B: 16 KB at k=6, 64 KB at k=7) and is
//     filled with ds_add_u32; above 64 KB the row itself (L2-resident) takes the
//     atomics;
//   * the row total and entropy are fused into the flush: H = log2 T - (sum c
//     log2 c)/T, c log2 c from a 256-entry LDS table.
#include "dvs_internal.h"

#include <algorithm>
#include <cstring>
#include <cmath>
#include <cstdlib>

namespace {

constexpr int HIST_THREADS = 256;      // threads that flush / reduce a row (fixed: the entropy's bits must not
                                       // depend on the launch geometry)
constexpr int HIST_MAX_THREADS = 512;  // a tile's chunks are spread over up to this many
constexpr uint32_t TILE_LEN = 32768;  // k-mer end positions per workgroup
static_assert(TILE_LEN < 65536, "a whole-sequence tile's counts must fit the packed 16-bit LDS counters");
constexpr int CLOG_TBL = 256;

struct KTile {
    uint64_t begin;      // first k-mer END position (absolute byte offset)
    uint64_t end;        // one past the last END position
    uint64_t seq_begin;  // first byte of the sequence
    uint32_t row;
    uint32_t single;  // 1: the only tile of its row (plain stores + fused stats)
};

__device__ __forceinline__ uint4 load16(const uint8_t *base, uint64_t off, uint64_t nbytes) {
    if (off + 16 <= nbytes) return *reinterpret_cast<const uint4 *>(base + off);
    uint32_t w[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};  // invalid filler
    for (int i = 0; i < 16; i++) {
        if (off + i < nbytes) {
            w[i >> 2] &= ~(0xFFu << (8 * (i & 3)));
            w[i >> 2] |= uint32_t(base[off + i]) << (8 * (i & 3));
        }
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
}

// PK16: two 16-bit counters to a word (a tile holds at most TILE_LEN = 32768 windows, so the low
// half never carries into the high one): half the LDS, twice the workgroups a CU can hold
template <bool LDS_HIST, bool PK16 = false>
__device__ __forceinline__ void bump(uint32_t *hist, uint32_t idx) {
    if (PK16) atomicAdd(&hist[idx >> 1], (idx & 1u) ? 0x10000u : 1u);
    else atomicAdd(&hist[idx], 1u);  // ds_add_u32 / global_atomic_add, no return
}

__device__ __forceinline__ double clog2c(uint32_t c, const double *tbl) {
    if (c < CLOG_TBL) return tbl[c];
    const double d = double(c);
    return d * log2(d);
}

// LDS: [hist B u32 (LDS_HIST)] [tbl 256 f64] [scratch 32 f64]
// tiles == NULL: workgroup b owns sequence b outright (its tile comes from the
// offsets; sequences needing more than one tile are left to the tile-list launch).
// PK16 (whole sequences only, 128 threads, 4 | B): the LDS histogram is packed (see bump) and the
// flush emulates the 256-thread partition of the bins -- thread t stands for threads t and t + 128
// -- so that the row entropy has the same bits as from the unpacked kernel.
// OUT16 (PK16 only): the row leaves as 16-bit counts, the packed LDS words as they are (matrix kind 2)
// PACKED (NS4 only): the sequences are the two planes of the packed form (dvs_packed: `seqs` points at the
// code words, `pmask` at the mask words) -- a lane's 16 bases are one 4-byte and one 2-byte load, and
// the 2-bit pack and the invalid mask the byte form has to work out (pack16 / inv16: ~60 of the loop's
// vector instructions per 16 bases) are simply what was loaded.
template <bool NS4, bool LDS_HIST, bool PK16 = false, bool OUT16 = false, bool PACKED = false>
__global__ __launch_bounds__(HIST_MAX_THREADS) void kmer_hist_kernel(
    const uint8_t *__restrict__ seqs, const uint16_t *__restrict__ pmask, uint64_t nbytes,
    const uint64_t *__restrict__ offsets,
    const KTile *__restrict__ tiles, uint32_t *__restrict__ counts, uint32_t *__restrict__ totals,
    double *__restrict__ entropy, const double *__restrict__ clog_tbl, uint32_t k, uint32_t ns,
    uint64_t B, uint32_t hot_rows, uint32_t row0, uint64_t uni_base, uint64_t uni_stride) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    KTile t;
    if (tiles) {
        t = tiles[blockIdx.x];
    } else {
        // this launch builds rows row0 .. row0 + gridDim.x - 1; rows below hot_rows are built LAST and
        // with ordinary stores (the next reader comes soon)
        const uint32_t r = row0 + (hot_rows ? gridDim.x - 1 - blockIdx.x : blockIdx.x);
        // (offsets == NULL: sequences of one length laid end to end -- nothing was uploaded)
        const uint64_t s0 = offsets ? offsets[r] : uni_base + uint64_t(r) * uni_stride;
        const uint64_t s1 = offsets ? offsets[r + 1] : s0 + uni_stride;
        t.row = r;
        t.single = 1;
        t.seq_begin = s0;
        t.begin = s0 + k - 1;
        t.end = s1;
        if (s1 - s0 >= k && s1 - t.begin > TILE_LEN) return;  // multi-tile: other launch
        if (s1 - s0 < k) t.begin = t.end = s1;               // windows(k) empty: all-zero row
    }
    static_assert(!OUT16 || PK16, "16-bit rows come from the packed histogram");
    static_assert(!PACKED || NS4, "only four-state sequences have a packed form");
    uint32_t *row = OUT16 ? counts + uint64_t(t.row) * (B / 2) : counts + uint64_t(t.row) * B;
    uint32_t *hist = LDS_HIST ? reinterpret_cast<uint32_t *>(smem) : row;
    double *tbl = reinterpret_cast<double *>(smem + (LDS_HIST ? ((B * (PK16 ? 2 : 4) + 15) & ~15ull) : 0));
    double *scratch = tbl + CLOG_TBL;
    const int tid = threadIdx.x;
    const int nthreads = blockDim.x;

    if (LDS_HIST) {
        if ((B & 3) == 0) {
            uint4 *h4 = reinterpret_cast<uint4 *>(hist);
            for (uint64_t i = tid; i < B / (PK16 ? 8 : 4); i += nthreads) h4[i] = make_uint4(0, 0, 0, 0);
            if (PK16 && (B & 7) && tid == 0) reinterpret_cast<uint2 *>(hist)[B / 4 - 1] = make_uint2(0, 0);
        } else {
            for (uint64_t i = tid; i < B; i += nthreads) hist[i] = 0;
        }
    }
    for (int i = tid; i < CLOG_TBL; i += nthreads) tbl[i] = clog_tbl[i];  // c log2 c, c < 256 (clog_tbl_kernel)
    __syncthreads();

    uint32_t nvalid = 0;  // windows this thread counted
    const uint64_t abase = t.begin & ~15ull;
    const uint64_t nchunks = t.end > t.begin ? (t.end - abase + 15) >> 4 : 0;
    const uint32_t bmask = uint32_t(B - 1);  // NS4: B = 4^k, power of two (k = 16 -> 2^32 - 1)

    if (NS4) {
        // Every wave takes a CONTIGUOUS run of 16-byte chunks, 64 at a time (lane l: chunk base + l), so
        // the chunk in front of a lane's own -- the k - 1 halo -- is its neighbour lane's: its 2-bit pack
        // and invalid mask arrive by one DPP shift (wave_shr:1) instead of a second load and a second
        // pack; lane 0 takes lane 63's of the previous step (readlane), and only the first step of a
        // wave loads a chunk it does not own.  When a wave's 1 KiB holds no byte >= 4 -- almost always --
        // the invalid-mask arithmetic is skipped, and when all its 16 x 64 windows count, so are the
        // per-window tests.
        const uint32_t lane = tid & 63, wave = tid >> 6, nwaves = (nthreads + 63) >> 6;
        const uint64_t per_wave = ((nchunks + nwaves - 1) / nwaves + 63) & ~63ull;
        const uint64_t c_lo = uint64_t(wave) * per_wave;
        const uint64_t c_hi = c_lo + per_wave < nchunks ? c_lo + per_wave : nchunks;
        uint32_t carryP = 0, carryI = 0xFFFFu;
        if (c_lo < c_hi) {  // the chunk in front of the wave's first one (uniform address)
            const uint64_t A0 = abase + (c_lo << 4);
            if (A0 >= 16) {
                if constexpr (PACKED) {
                    carryP = reinterpret_cast<const uint32_t *>(seqs)[(A0 >> 4) - 1];
                    carryI = pmask[(A0 >> 4) - 1];
                } else {
                    const uint4 pv = load16(seqs, A0 - 16, nbytes);
                    carryP = dvs_pack16(pv);
                    carryI = dvs_inv16(pv);
                }
            }
        }
        for (uint64_t c0 = c_lo; c0 < c_hi; c0 += 64) {
            const uint64_t c = c0 + lane;
            const bool live = c < c_hi;
            const uint64_t A = abase + ((live ? c : c_hi - 1) << 4);
            uint32_t Pc, Ic = 0;
            if constexpr (PACKED) {  // (A < t.end <= nbytes: word A / 16 exists; positions behind nbytes are flagged in it)
                Pc = reinterpret_cast<const uint32_t *>(seqs)[A >> 4];
                Ic = pmask[A >> 4];
            } else {
                const uint4 cur = load16(seqs, A, nbytes);
                Pc = dvs_pack16(cur);
                const uint32_t hib = (cur.x | cur.y | cur.z | cur.w) & 0xFCFCFCFCu;
                if (__ballot(hib != 0)) Ic = dvs_inv16(cur);  // (wave-uniform branch)
            }
            // the neighbour's pack / mask: lane l <- lane l - 1, lane 0 <- the carry
            const uint32_t Pp = uint32_t(__builtin_amdgcn_update_dpp(int(carryP), int(Pc), 0x138, 0xF, 0xF, false));
            const uint32_t Ip = uint32_t(__builtin_amdgcn_update_dpp(int(carryI), int(Ic), 0x138, 0xF, 0xF, false));
            carryP = uint32_t(__builtin_amdgcn_readlane(int(Pc), 63));
            carryI = uint32_t(__builtin_amdgcn_readlane(int(Ic), 63));
            const uint64_t P = (uint64_t(Pp) << 32) | Pc;
            uint32_t I = (Ip << 16) | Ic;  // bit 31-q: position q = 0..31 <-> absolute A-16+q is invalid
            // bases before the sequence start are invalid
            const int64_t lead = int64_t(t.seq_begin) - (int64_t(A) - 16);
            if (lead > 0) I |= (lead >= 32) ? 0xFFFFFFFFu : ~(0xFFFFFFFFu >> lead);
            // bit 15-j of W: some base of the k-mer ending at A+j is invalid (OR of k bits of I)
            uint32_t W = I;
            if (__ballot(I != 0)) {
                uint32_t cover = 1;
                while (2 * cover <= k) {
                    W |= W >> cover;
                    cover *= 2;
                }
                if (cover < k) W |= W >> (k - cover);
            }
            // bit 15-j of R: A+j lies in [t.begin, t.end)
            const uint32_t lo = t.begin > A ? uint32_t(t.begin - A < 16 ? t.begin - A : 16) : 0u;
            const uint32_t hi = t.end > A ? uint32_t(t.end - A < 16 ? t.end - A : 16) : 0u;
            const uint32_t R = (live && hi > lo) ? ((0xFFFFu >> lo) & ~(0xFFFFu >> hi)) : 0u;
            const uint32_t ok = R & ~W & 0xFFFFu;
            nvalid += __popc(ok);  // (the row total: every counted window, not a second pass over the bins)
            if (__ballot(ok != 0xFFFFu) == 0) {  // every window of the wave's 1 KiB counts
                if constexpr (PK16) {
                    // packed counters, four instructions a window: with P2 = P << 1, t = P2 >> s holds the
                    // bin index times two -- masked to a multiple of four it is the byte address of the
                    // bin's 32-bit word -- and bit 1 of t says which half of the word counts
                    const uint64_t P2 = P << 1;
                    const uint32_t p2lo = uint32_t(P2), p2hi = uint32_t(P2 >> 32);
                    const uint32_t amask = (bmask << 1) & ~3u;
                    unsigned char *hb = smem;  // (PK16 => the histogram is the start of the LDS block)
#pragma unroll
                    for (int j = 0; j < 16; j++) {
                        const uint32_t tq = __builtin_amdgcn_alignbit(p2hi, p2lo, 2 * (15 - j));
                        uint32_t half = __builtin_amdgcn_ubfe(tq, 1, 1);
                        asm volatile("" : "+v"(half));  // (keeps bfe + mad: two instructions, not and + cmp + select)
                        atomicAdd(reinterpret_cast<uint32_t *>(hb + (tq & amask)), __umul24(half, 0xFFFFu) + 1u);  // (v_mad_u32_u24)
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 16; j++) bump<LDS_HIST, PK16>(hist, uint32_t(P >> (2 * (15 - j))) & bmask);
                }
            } else {
#pragma unroll
                for (int j = 0; j < 16; j++) {
                    const uint32_t idx = uint32_t(P >> (2 * (15 - j))) & bmask;
                    if ((ok >> (15 - j)) & 1u) bump<LDS_HIST, PK16>(hist, idx);
                }
            }
        }
    } else {
        for (uint64_t c = tid; c < nchunks; c += nthreads) {
            const uint64_t A = abase + (c << 4);
            const uint4 cur = load16(seqs, A, nbytes);
            const uint4 prev = (A >= 16) ? load16(seqs, A - 16, nbytes)
                                         : make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
            // positions q = 0..31 <-> absolute A-16+q ; bases before the sequence start are invalid
            const int64_t lead = int64_t(t.seq_begin) - (int64_t(A) - 16);
            const uint32_t w[8] = {prev.x, prev.y, prev.z, prev.w, cur.x, cur.y, cur.z, cur.w};
            uint32_t idx = 0, run = 0;
            const uint32_t Bd = uint32_t(B / ns);  // ns^(k-1)
#pragma unroll
            for (int q = 0; q < 32; q++) {
                const uint32_t b = (w[q >> 2] >> (8 * (q & 3))) & 0xFFu;
                const bool ok = (b < ns) && (int64_t(q) >= lead);
                idx = ok ? (idx % Bd) * ns + b : 0u;
                run = ok ? run + 1 : 0u;
                if (q >= 16) {
                    const uint64_t p = A + (q - 16);
                    if (p >= t.begin && p < t.end && run >= k) {
                        bump<LDS_HIST, PK16>(hist, idx);
                        nvalid++;
                    }
                }
            }
        }
    }
    if (!LDS_HIST) __threadfence();  // this thread's global atomics on `row` have been performed
    __syncthreads();

    if constexpr (PK16) {
        const uint2 *h2 = reinterpret_cast<const uint2 *>(hist);  // unit u = bins 4u .. 4u + 3
        uint4 *r4 = reinterpret_cast<uint4 *>(row);
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        double sv[2] = {0.0, 0.0}, tv[2] = {0.0, 0.0};
#pragma unroll
        for (int half = 0; half < 2; half++)  // virtual threads tid and tid + 128 of the 256-thread flush
            for (uint64_t u = uint64_t(tid) + 128u * half; u < B / 4; u += HIST_THREADS) {
                const uint2 w = h2[u];
                const uint4 v = make_uint4(w.x & 0xFFFFu, w.x >> 16, w.y & 0xFFFFu, w.y >> 16);
                if constexpr (OUT16) {
                    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
                    uint2 *r2 = reinterpret_cast<uint2 *>(row);
                    if (t.row < hot_rows) r2[u] = w;
                    else __builtin_nontemporal_store((u32x2){w.x, w.y}, reinterpret_cast<u32x2 *>(r2 + u));
                } else if (t.row < hot_rows) r4[u] = v;
                else __builtin_nontemporal_store((u32x4){v.x, v.y, v.z, v.w}, reinterpret_cast<u32x4 *>(r4 + u));
                // (one test per four bins: a count of 256 or more is what the table does not hold)
                if (((w.x | w.y) & 0xFF00FF00u) == 0) sv[half] += ((tbl[v.x] + tbl[v.y]) + tbl[v.z]) + tbl[v.w];
                else sv[half] += clog2c(v.x, tbl) + clog2c(v.y, tbl) + clog2c(v.z, tbl) + clog2c(v.w, tbl);
            }
        tv[0] = double(nvalid);  // (integers: exact in any order, the same total as the sum over the bins)
        const int lane = tid & 63, wave = tid >> 6;  // (two real waves = virtual waves {0, 2} and {1, 3})
        double sums[2];
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const double a = dvs_wave_sum(q ? tv[0] : sv[0]), b = dvs_wave_sum(q ? tv[1] : sv[1]);
            __syncthreads();
            if (lane == 0) {
                scratch[wave] = a;
                scratch[wave + 2] = b;
            }
            __syncthreads();
            double acc = 0.0;
            for (int i = 0; i < 4; i++) acc += scratch[i];
            sums[q] = acc;
        }
        if (tid == 0) {
            totals[t.row] = uint32_t(sums[1]);
            entropy[t.row] = sums[1] > 0.0 ? log2(sums[1]) - sums[0] / sums[1] : 0.0;
        }
    } else if (t.single) {
        double s = 0.0, tot = 0.0;
        if (tid >= HIST_THREADS) {
            // the row is flushed and reduced by the first HIST_THREADS threads only
        } else if (LDS_HIST) {
            if ((B & 3) == 0) {
                const uint4 *h4 = reinterpret_cast<const uint4 *>(hist);
                uint4 *r4 = reinterpret_cast<uint4 *>(row);
                for (uint64_t i = tid; i < B / 4; i += HIST_THREADS) {
                    const uint4 v = h4[i];
                    // written once, read much later (1.6 GB between): a streaming store keeps it out of
                    // the caches' way (measured: -0.04 ms per 100k x 4^6 build + selection)
                    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                    if (t.row < hot_rows) r4[i] = v;
                    else __builtin_nontemporal_store((u32x4){v.x, v.y, v.z, v.w}, reinterpret_cast<u32x4 *>(r4 + i));
                    s += clog2c(v.x, tbl) + clog2c(v.y, tbl) + clog2c(v.z, tbl) + clog2c(v.w, tbl);
                    tot += double(v.x) + double(v.y) + double(v.z) + double(v.w);
                }
            } else {
                for (uint64_t i = tid; i < B; i += HIST_THREADS) {
                    const uint32_t v = hist[i];
                    row[i] = v;
                    s += clog2c(v, tbl);
                    tot += double(v);
                }
            }
        } else {
            for (uint64_t i = tid; i < B; i += HIST_THREADS) {
                const uint32_t v = __hip_atomic_load(&row[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s += clog2c(v, tbl);
                tot += double(v);
            }
        }
        s = dvs_block_sum(s, scratch);
        tot = dvs_block_sum(tot, scratch);
        if (tid == 0) {
            totals[t.row] = uint32_t(tot);
            entropy[t.row] = tot > 0.0 ? log2(tot) - s / tot : 0.0;
        }
    } else if (LDS_HIST) {
        for (uint64_t i = tid; i < B; i += nthreads) {
            const uint32_t v = hist[i];
            if (v) atomicAdd(&row[i], v);
        }
    }
}

// c log2 c for c < CLOG_TBL, once per context (every histogram workgroup copies it into LDS)
__global__ void clog_tbl_kernel(double *tbl) {
    const int c = threadIdx.x;
    tbl[c] = c ? double(c) * log2(double(c)) : 0.0;
}

// rows listed in `rows`: total and entropy from the finished count row
__global__ __launch_bounds__(HIST_THREADS) void row_stats_kernel(
    const uint32_t *__restrict__ counts, const uint32_t *__restrict__ rows,
    uint32_t *__restrict__ totals, double *__restrict__ entropy, uint64_t B) {
    __shared__ double tbl[CLOG_TBL];
    __shared__ double scratch[32];
    const int tid = threadIdx.x;
    if (tid < CLOG_TBL) tbl[tid] = tid ? double(tid) * log2(double(tid)) : 0.0;
    __syncthreads();
    const uint32_t r = rows[blockIdx.x];
    const uint32_t *row = counts + uint64_t(r) * B;
    double s = 0.0, tot = 0.0;
    for (uint64_t i = tid; i < B; i += HIST_THREADS) {
        const uint32_t v = row[i];
        s += clog2c(v, tbl);
        tot += double(v);
    }
    s = dvs_block_sum(s, scratch);
    tot = dvs_block_sum(tot, scratch);
    if (tid == 0) {
        totals[r] = uint32_t(tot);
        entropy[r] = tot > 0.0 ? log2(tot) - s / tot : 0.0;
    }
}

__global__ __launch_bounds__(HIST_THREADS) void zero_rows_kernel(uint32_t *__restrict__ counts,
                                                                const uint32_t *__restrict__ rows,
                                                                uint64_t B) {
    uint32_t *row = counts + uint64_t(rows[blockIdx.x]) * B;
    for (uint64_t i = threadIdx.x; i < B; i += HIST_THREADS) row[i] = 0;
}

// kind-1 matrices: H(row) = sum_{f != 0} -f log2 f (src/record.rs:86-106)
__global__ __launch_bounds__(HIST_THREADS) void freq_entropy_kernel(
    const double *__restrict__ freqs, double *__restrict__ entropy, uint64_t B) {
    __shared__ double scratch[32];
    const double *row = freqs + uint64_t(blockIdx.x) * B;
    double h = 0.0;
    for (uint64_t i = threadIdx.x; i < B; i += HIST_THREADS) {
        const double f = row[i];
        if (f != 0.0) h += -f * log2(f);
    }
    h = dvs_block_sum(h, scratch);
    if (threadIdx.x == 0) entropy[blockIdx.x] = h;
}

// Rows flagged by meta[2 r + 1] (0: padding) -> the matrix, real rows first in their input order,
// padding behind them (total 0: skipped like a sequence without valid k-mers), so that the first n
// positions of the merge stream are the first n REAL records of the concatenated results
// (get_kmerseqs_and_init_summed_records, records.rs:344-360).  src[dest] = input row.
__global__ __launch_bounds__(HIST_THREADS) void compact_freq_rows_kernel(
    const double *__restrict__ in, const double *__restrict__ meta, double *__restrict__ out,
    uint32_t *__restrict__ totals, uint32_t *__restrict__ src, uint32_t nrows, uint64_t B) {
    __shared__ uint32_t s_cnt[2];
    const uint32_t r = blockIdx.x;
    if (threadIdx.x < 2) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    uint32_t before = 0, all = 0;  // real rows in front of r / in total
    for (uint32_t q = threadIdx.x; q < nrows; q += HIST_THREADS) {
        const uint32_t v = meta[2 * q + 1] != 0.0 ? 1u : 0u;
        all += v;
        if (q < r) before += v;
    }
    atomicAdd(&s_cnt[0], before);
    atomicAdd(&s_cnt[1], all);
    __syncthreads();
    const bool real = meta[2 * r + 1] != 0.0;
    const uint32_t dest = real ? s_cnt[0] : s_cnt[1] + (r - s_cnt[0]);
    const double *a = in + uint64_t(r) * B;
    double *o = out + uint64_t(dest) * B;
    for (uint64_t i = threadIdx.x; i < B; i += HIST_THREADS) o[i] = a[i];
    if (threadIdx.x == 0) {
        totals[dest] = real ? 1u : 0u;
        src[dest] = r;
    }
}

// totals of a frequency-row matrix: 1, or 0 for a padding row (meta[2 r + 1] == 0)
__global__ void freq_totals_kernel(uint32_t *__restrict__ totals, const double *__restrict__ meta,
                                   uint32_t nrows) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < nrows) totals[r] = (!meta || meta[2 * r + 1] != 0.0) ? 1u : 0u;
}

template <typename K>
int set_dyn_lds(dvs_ctx *ctx, K kernel, size_t bytes) {
    return dvs_raise_dyn_lds(ctx, reinterpret_cast<const void *>(kernel), bytes);
}

}  // namespace

uint64_t dvs_pow_u64(uint32_t base, uint32_t exp, bool *overflow) {
    uint64_t r = 1;
    *overflow = false;
    for (uint32_t i = 0; i < exp; i++) {
        if (r > UINT64_MAX / (base ? base : 1)) *overflow = true;
        r *= base;
    }
    return r;
}

int dvs_matrix_settle(dvs_ctx *ctx, const dvs_matrix *cm) {
    dvs_matrix *m = const_cast<dvs_matrix *>(cm);
    if (!m || !m->ev_built) return DVS_OK;
    const hipError_t e = hipEventSynchronize(m->ev_built);
    if (e == hipSuccess) m->h_head_totals.assign(m->h_head_pinned, m->h_head_pinned + m->head_count);
    else m->h_head_totals.clear();
    dvs_event_put(m->ctx, m->ev_built);
    dvs_pinned_put(m->ctx, m->h_head_pinned);
    m->ev_built = nullptr;
    m->h_head_pinned = nullptr;
    if (e != hipSuccess) return dvs_hip_fail(ctx ? ctx : m->ctx, e, "histogram kernels");
    return DVS_OK;
}

void dvs_matrix_free_fields(dvs_matrix *m) {
    if (!m) return;
    (void)dvs_matrix_settle(m->ctx, m);  // the pinned block must not go back to the cache with a copy pending
    if (m->ev_join) dvs_event_put(m->ctx, m->ev_join);
    m->ev_join = nullptr;
    dvs_dev_free(m->ctx, m->d_counts);
    dvs_dev_free(m->ctx, m->d_counts16);
    m->d_counts16 = nullptr;
    dvs_dev_free(m->ctx, m->d_freqs);
    dvs_dev_free(m->ctx, m->d_totals);
    dvs_dev_free(m->ctx, m->d_entropy);
    dvs_dev_free(m->ctx, m->d_src_row);
    m->d_src_row = nullptr;
    m->d_counts = nullptr;
    m->d_freqs = nullptr;
    m->d_totals = nullptr;
    m->d_entropy = nullptr;
    dvs_ctx_release(m->ctx);  // (taken in matrix_alloc; every path deletes the matrix right after this)
    m->ctx = nullptr;
}

// The offsets of a build: validated, the tile lists of genome-length sequences derived, everything
// uploaded -- or found unchanged in the context's cache.  *n_long = sequences needing more than one
// tile (none: every row's counts fit 16 bits and the matrix may be built as kind 2).
int dvs_hist_prepare(dvs_ctx *ctx, const uint64_t *offsets, uint32_t nseq, uint32_t k, uint64_t nbins, uint64_t nbytes,
                     size_t *n_long_out) {
    dvs_ctx::OffsetsCache &oc = ctx->off_cache;
    const size_t n_off = size_t(nseq) + 1;
    // Sequences of ONE length laid end to end (fixed-length reads, amplicons, synthetic sets) need no
    // offsets on the device at all: the kernel derives row r's span from (base, stride).  One read-only
    // pass decides (it stops at the first span of another length), nothing is copied or uploaded.
    oc.uniform = false;
    if (nseq >= 1 && !ctx->knobs.no_uniform_offsets) {
        const uint64_t base = offsets[0], stride = offsets[1] - offsets[0];
        bool uni = offsets[1] >= offsets[0] && stride < uint64_t(TILE_LEN) + k;
        uint64_t diff = 0;
        for (uint32_t r = 0; r < nseq && uni; r += 4096) {  // (blocks: vectorised, with an early way out)
            const uint32_t e = std::min<uint32_t>(nseq, r + 4096);
            for (uint32_t q = r; q < e; q++) diff |= (offsets[q + 1] - offsets[q]) ^ stride;
            uni = diff == 0;
        }
        if (uni && base + uint64_t(nseq) * stride <= nbytes) {
            dvs_dev_free(ctx, oc.d_rows);
            dvs_dev_free(ctx, oc.d_tiles);
            oc.d_rows = oc.d_tiles = nullptr;
            oc.n_long = oc.n_tiles = 0;
            oc.n_off = 0;  // (no entry of the content cache: the next non-uniform build starts afresh)
            oc.uniform = true;
            oc.uni_base = base;
            oc.uni_stride = stride;
            *n_long_out = 0;
            return DVS_OK;
        }
    }
    const bool hit = oc.d_off && oc.h_off && oc.k == k && oc.nbytes == nbytes && oc.n_off == n_off &&
                     !ctx->knobs.no_offsets_cache && std::memcmp(oc.h_off, offsets, n_off * 8) == 0;
    if (!hit) {
        // the pinned block: no upload may still be reading it, and it must be large enough
        if (oc.ev_up) (void)hipEventSynchronize(oc.ev_up);
        if (oc.h_cap < n_off) {
            if (oc.h_off) (void)hipHostFree(oc.h_off);
            oc.h_off = nullptr;
            oc.h_cap = 0;
            const size_t cap = std::max<size_t>(n_off + n_off / 4, 1024);
            const hipError_t he = hipHostMalloc((void **)&oc.h_off, cap * 8, hipHostMallocDefault);
            if (he != hipSuccess) {
                oc.h_off = nullptr;
                return dvs_hip_fail(ctx, he, "pinned offsets block");
            }
            oc.h_cap = cap;
        }
        oc.n_off = 0;  // (the block is being rewritten: no hit on a half-written copy after an error)
        std::vector<KTile> tiles;
        std::vector<uint32_t> long_rows;
        uint64_t *dst = oc.h_off;
        // first pass, branch-free (it vectorises): the pinned copy, whether any pair is out of order or
        // out of range, and the longest sequence; rows are looked at one by one only when something is
        // wrong or some sequence needs more than one tile
        uint64_t bad = 0, longest = 0;
        dst[0] = offsets[0];
        for (uint32_t r = 0; r < nseq; r++) {
            const uint64_t s0 = offsets[r], s1 = offsets[r + 1];
            dst[r + 1] = s1;
            bad |= uint64_t(s1 < s0) | uint64_t(s1 > nbytes);
            const uint64_t len = s1 - s0;
            longest = len > longest ? len : longest;
        }
        if (bad || longest >= uint64_t(TILE_LEN) + k) {
            uint64_t long_windows = 0;
            for (uint32_t r = 0; r < nseq; r++) {
                const uint64_t s0 = offsets[r], s1 = offsets[r + 1];
                if (s1 < s0 || s1 > nbytes)
                    return dvs_set_error(ctx, DVS_ERR_VALUE, "offsets[%u..%u] = %llu..%llu out of range", r,
                                         r + 1, (unsigned long long)s0, (unsigned long long)s1);
                if (s1 - s0 >= k && s1 - (s0 + k - 1) > TILE_LEN) long_windows += s1 - (s0 + k - 1);
            }
            // A tile of a long row ends in one global atomic per non-empty bin, so it should hold many more windows
            // than there are bins (at 4^7 bins a 32768-window tile spent as long merging as counting: 4.35 ms for
            // 1050 genomes of 3 Mb) -- 16 windows a bin, while that still leaves >= 2048 tiles for the grid.
            // (The tiles of a long row count in 32-bit LDS words: no limit from the packed 16-bit counters.)
            uint64_t tile_long = TILE_LEN;
            {
                const uint64_t B_ = nbins ? nbins : 1;  // (num_states^k of this build; any tile length counts correctly)
                while (tile_long < 262144 && tile_long < 16 * B_ && long_windows / (2 * tile_long) >= 2048) tile_long *= 2;
                // (tests: small inputs, long tiles -- at most 2^20 windows, far below what a 32-bit counter holds)
                if (ctx->knobs.test_long_tile >= TILE_LEN) tile_long = std::min<uint64_t>(ctx->knobs.test_long_tile, uint64_t(1) << 20);
            }
            for (uint32_t r = 0; r < nseq; r++) {
                const uint64_t s0 = offsets[r], s1 = offsets[r + 1];
                if (s1 - s0 < k) continue;
                const uint64_t first = s0 + k - 1;
                if (s1 - first <= TILE_LEN) continue;
                long_rows.push_back(r);
                for (uint64_t b = first; b < s1; b += tile_long) {
                    KTile t;
                    t.begin = b;
                    t.end = std::min<uint64_t>(b + tile_long, s1);
                    t.seq_begin = s0;
                    t.row = r;
                    t.single = 0;
                    tiles.push_back(t);
                }
            }
        }
        // (the previous lists go back to the block cache; stream order protects them until the
        // kernels that read them have run; the offsets block is kept while it is large enough)
        dvs_dev_free(ctx, oc.d_rows);
        dvs_dev_free(ctx, oc.d_tiles);
        oc.d_rows = oc.d_tiles = nullptr;
        oc.n_long = oc.n_tiles = 0;
        int arc = DVS_OK;
        if (oc.d_off_cap < n_off) {
            dvs_dev_free(ctx, oc.d_off);
            oc.d_off = nullptr;
            oc.d_off_cap = 0;
            arc = dvs_dev_alloc(ctx, &oc.d_off, (n_off + n_off / 4) * 8, "offsets");
            if (!arc) oc.d_off_cap = n_off + n_off / 4;
        }
        if (!arc && !long_rows.empty()) arc = dvs_dev_alloc(ctx, &oc.d_rows, long_rows.size() * 4, "row list");
        if (!arc && !tiles.empty()) arc = dvs_dev_alloc(ctx, &oc.d_tiles, tiles.size() * sizeof(KTile), "tile list");
        hipError_t ue = hipSuccess;
        // (the other uploads read from the cache entry's own copies, which outlive the call)
        oc.h_tiles.assign(reinterpret_cast<const unsigned char *>(tiles.data()),
                          reinterpret_cast<const unsigned char *>(tiles.data()) + tiles.size() * sizeof(KTile));
        oc.h_long_rows = long_rows;
        if (!arc) ue = hipMemcpyAsync(oc.d_off, oc.h_off, n_off * 8, hipMemcpyHostToDevice, ctx->stream);
        if (!arc && ue == hipSuccess) {
            if (!oc.ev_up) oc.ev_up = dvs_event_get(ctx);
            if (oc.ev_up) ue = hipEventRecord(oc.ev_up, ctx->stream);
        }
        if (!arc && ue == hipSuccess && !long_rows.empty()) {
            ue = hipMemcpyAsync(oc.d_rows, oc.h_long_rows.data(), long_rows.size() * 4, hipMemcpyHostToDevice, ctx->stream);
            if (ue == hipSuccess)
                ue = hipMemcpyAsync(oc.d_tiles, oc.h_tiles.data(), oc.h_tiles.size(), hipMemcpyHostToDevice,
                                    ctx->stream);
        }
        if (arc || ue != hipSuccess) {
            dvs_dev_free(ctx, oc.d_rows);
            dvs_dev_free(ctx, oc.d_tiles);
            oc.d_rows = oc.d_tiles = nullptr;
            return arc ? arc : dvs_hip_fail(ctx, ue, "histogram setup");
        }
        oc.n_off = n_off;
        oc.nbytes = nbytes;
        oc.k = k;
        oc.n_long = long_rows.size();
        oc.n_tiles = tiles.size();
    }
    *n_long_out = oc.n_long;
    return DVS_OK;
}

// a whole-sequence build may leave 16-bit rows (see dvs_matrix): the packed-histogram kernel serves it.
// Up to 4096 bins only: beyond that the selection engines score rows with the f32-log tier alone,
// which is bound by its arithmetic, not by the bytes of a row, and their per-event paths read counts
// one per lane -- measured at 4^7 bins: 22.8 ms per selection with 16-bit rows, 18.3 with 32-bit (round 4,
// after the engine's hand-overs were rewritten: 24.0 against 20.2 ms for C4's whole input).
bool dvs_hist_rows_fit_u16(const dvs_ctx *ctx, uint64_t B, size_t n_long) {
    return n_long == 0 && B <= 4096 && (B & 3) == 0 && !ctx->knobs.counts_u32;
}

// Launches the histogram build for sequences already in HBM (d_seqs, nbytes
// readable) into an allocated matrix, after dvs_hist_prepare for the same offsets.  One launch gives
// every sequence that fits a single tile its own workgroup (tile derived from the offsets on the
// device); genome-length sequences get an explicit tile list and a second launch.
int dvs_matrix_fill_counts(dvs_ctx *ctx, dvs_matrix *m, const dvs_seq_view &sv, const uint64_t *offsets,
                           bool no_wait) {
    // (packed: the kernel's `seqs` argument carries the code words, `pmask` the mask words)
    const bool packed = sv.codes != nullptr;
    const uint8_t *d_seqs = packed ? reinterpret_cast<const uint8_t *>(sv.codes) : sv.seqs;
    const uint16_t *d_pmask = sv.mask;
    const uint64_t nbytes = sv.nbytes;
    const uint32_t nseq = m->nrows, k = m->k, ns = m->num_states;
    if (packed && ns != 4) return dvs_set_error(ctx, DVS_ERR_VALUE, "packed sequences have four states, not %u", ns);
    const uint64_t B = m->nbins;
    const bool lds_hist = B * 4 <= 64 * 1024;
    const bool ns4 = ns == 4;
    (void)offsets;
    dvs_ctx::OffsetsCache &oc = ctx->off_cache;
    uint64_t *d_off = oc.uniform ? nullptr : static_cast<uint64_t *>(oc.d_off);
    const uint64_t uni_base = oc.uni_base, uni_stride = oc.uni_stride;
    uint32_t *d_rows = static_cast<uint32_t *>(oc.d_rows);
    KTile *d_tiles = static_cast<KTile *>(oc.d_tiles);
    const size_t n_long = oc.n_long, n_tiles = oc.n_tiles;
    auto cleanup = [&]() {};  // (the lists belong to the context's cache now)
    int rc = DVS_OK;
    hipError_t e = hipSuccess;
    if (!lds_hist)  // the rows take the atomics directly: all start at zero
        e = hipMemsetAsync(m->d_counts, 0, size_t(nseq) * B * 4, ctx->stream);
    if (e == hipSuccess && n_long && lds_hist)
        hipLaunchKernelGGL(zero_rows_kernel, dim3(uint32_t(n_long)), dim3(HIST_THREADS), 0, ctx->stream,
                           m->d_counts, d_rows, B);
    if (e != hipSuccess) return dvs_hip_fail(ctx, e, "histogram setup");
    if (!ctx->d_clog_tbl) {
        rc = dvs_dev_alloc(ctx, (void **)&ctx->d_clog_tbl, CLOG_TBL * sizeof(double), "c log2 c table");
        if (rc) {
            cleanup();
            return rc;
        }
        hipLaunchKernelGGL(clog_tbl_kernel, dim3(1), dim3(CLOG_TBL), 0, ctx->stream, ctx->d_clog_tbl);
    }
    // 256 threads for whole sequences: with 16-18 KB of LDS each, 8 workgroups (32 waves) fill a CU;
    // measured on 100k x 5 kb: 0.67 ms at 256 threads, 0.78 at 320, 0.86 at 384 and 512
    // (genome tiles: 256 and 512 measure the same at 4^6 bins; a 64 KB histogram -- 4^7 bins -- leaves room for two
    // workgroups a CU, and 512 threads each put sixteen waves on it instead of eight)
    int nthreads = HIST_THREADS;
    const int tile_threads = (lds_hist && B * 4 > 32768) ? HIST_MAX_THREADS : HIST_THREADS;
    const size_t lds = (lds_hist ? ((B * 4 + 15) & ~15ull) : 0) + (CLOG_TBL + 32) * sizeof(double);
#define DVS_LAUNCH_HIST(NS4, LH, PKD, GRID, TILES, NTHR, HOT)                                             \
    do {                                                                                         \
        rc = set_dyn_lds(ctx, kmer_hist_kernel<NS4, LH, false, false, PKD>, lds);                \
        if (!rc)                                                                                 \
            hipLaunchKernelGGL((kmer_hist_kernel<NS4, LH, false, false, PKD>), dim3(GRID), dim3(NTHR), lds,  \
                               ctx->stream, d_seqs, d_pmask, nbytes, d_off, TILES, m->d_counts,  \
                               m->d_totals, m->d_entropy, ctx->d_clog_tbl, k, ns, B, HOT, 0u, uni_base, uni_stride);   \
    } while (0)
#define DVS_LAUNCH_HIST_ANY(GRID, TILES, NTHR, HOT)                            \
    do {                                                                  \
        if (packed && lds_hist) DVS_LAUNCH_HIST(true, true, true, GRID, TILES, NTHR, HOT);   \
        else if (packed) DVS_LAUNCH_HIST(true, false, true, GRID, TILES, NTHR, HOT);         \
        else if (ns4 && lds_hist) DVS_LAUNCH_HIST(true, true, false, GRID, TILES, NTHR, HOT);   \
        else if (ns4) DVS_LAUNCH_HIST(true, false, false, GRID, TILES, NTHR, HOT);         \
        else if (lds_hist) DVS_LAUNCH_HIST(false, true, false, GRID, TILES, NTHR, HOT);    \
        else DVS_LAUNCH_HIST(false, false, false, GRID, TILES, NTHR, HOT);                 \
    } while (0)
    // The rows a selection reads first (its event-dense head is bound by fetch latency) are built
    // last and with ordinary stores, so that they are what the 256 MB memory-side cache still holds
    // when the selection starts; every other row is a streaming store.  Measured on 100k x 4^6:
    // streaming stores -0.04 ms per build + selection, the hot head another -0.01 ms.
    uint32_t hot_rows = uint32_t(std::min<uint64_t>(nseq, (192ull << 20) / (B * (m->kind == 2 ? 2 : 4))));
    // whole sequences: the packed histogram at 128 threads when the row layout allows it
    const bool pk16 = lds_hist && (B & 3) == 0;
    // A build that does not wait for its kernels is cut in two launches: the head of the matrix first
    // (what a selection reads first: its seeds), the totals of those rows on their way to the host right
    // behind it, then everything else.  The selection's set-up kernels run on the context's second
    // stream beside the second launch (select.hip sel_start).
    uint32_t head_rows = 0;
    if (m->kind == 2 && no_wait) {
        uint32_t want = DVS_HEAD_ROWS;
        head_rows = std::min<uint32_t>(nseq, want);
    }
    if (head_rows == nseq) head_rows = 0;  // (nothing left to run beside)
    bool head_event_done = false;
    if (m->kind == 2) {  // 16-bit rows (dvs_hist_rows_fit_u16 held when the matrix was allocated)
        const size_t lds16 = ((B * 2 + 15) & ~15ull) + (CLOG_TBL + 32) * sizeof(double);
        uint32_t *out16 = reinterpret_cast<uint32_t *>(m->d_counts16);
        auto launch16 = [&](uint32_t row0, uint32_t count, uint32_t hot_end, hipStream_t on) {
            if (packed) {
                rc = set_dyn_lds(ctx, kmer_hist_kernel<true, true, true, true, true>, lds16);
                if (!rc)
                    hipLaunchKernelGGL((kmer_hist_kernel<true, true, true, true, true>), dim3(count), dim3(128), lds16, on,
                                       d_seqs, d_pmask, nbytes, d_off, static_cast<const KTile *>(nullptr), out16,
                                       m->d_totals, m->d_entropy, ctx->d_clog_tbl, k, ns, B, hot_end, row0, uni_base, uni_stride);
            } else if (ns4) {
                rc = set_dyn_lds(ctx, kmer_hist_kernel<true, true, true, true>, lds16);
                if (!rc)
                    hipLaunchKernelGGL((kmer_hist_kernel<true, true, true, true>), dim3(count), dim3(128), lds16, on,
                                       d_seqs, d_pmask, nbytes, d_off, static_cast<const KTile *>(nullptr), out16,
                                       m->d_totals, m->d_entropy, ctx->d_clog_tbl, k, ns, B, hot_end, row0, uni_base, uni_stride);
            } else {
                rc = set_dyn_lds(ctx, kmer_hist_kernel<false, true, true, true>, lds16);
                if (!rc)
                    hipLaunchKernelGGL((kmer_hist_kernel<false, true, true, true>), dim3(count), dim3(128), lds16, on,
                                       d_seqs, d_pmask, nbytes, d_off, static_cast<const KTile *>(nullptr), out16,
                                       m->d_totals, m->d_entropy, ctx->d_clog_tbl, k, ns, B, hot_end, row0, uni_base, uni_stride);
            }
        };
        if (head_rows) {
            // With a CU split (dvs_ctx_cu_split) the rest of the matrix is built on the stream that
            // leaves the head CUs alone -- forked here, so that it is ordered behind everything the
            // context's stream holds so far (the sequences' producer, the offsets' upload) and runs
            // beside the head launch; the context's stream joins it again below.
            const bool split = nseq - head_rows >= 16u * head_rows && dvs_ctx_cu_split(ctx);
            hipEvent_t ev_fork = split ? dvs_event_get(ctx) : nullptr;
            bool forked = ev_fork && hipEventRecord(ev_fork, ctx->stream) == hipSuccess &&
                          hipStreamWaitEvent(ctx->stream_rest, ev_fork, 0) == hipSuccess;
            launch16(0, head_rows, head_rows, ctx->stream);
            void *pin = nullptr;
            if (!rc && dvs_pinned_get(ctx, &pin) == DVS_OK) {
                m->head_count = uint32_t(std::min<size_t>(head_rows, 4096 / sizeof(uint32_t)));
                m->h_head_pinned = static_cast<uint32_t *>(pin);
                m->ev_built = dvs_event_get(ctx);
                if (m->ev_built &&
                    hipMemcpyAsync(m->h_head_pinned, m->d_totals, size_t(m->head_count) * 4, hipMemcpyDeviceToHost,
                                   ctx->stream) == hipSuccess &&
                    hipEventRecord(m->ev_built, ctx->stream) == hipSuccess) {
                    head_event_done = true;
                    m->head_rows_built = head_rows;
                } else {
                    if (m->ev_built) dvs_event_put(ctx, m->ev_built);
                    dvs_pinned_put(ctx, pin);
                    m->ev_built = nullptr;
                    m->h_head_pinned = nullptr;
                }
            }
            if (!rc) {
                const uint32_t hot_end = uint32_t(std::min<uint64_t>(nseq, uint64_t(head_rows) + hot_rows));
                forked = forked && head_event_done;
                launch16(head_rows, nseq - head_rows, hot_end, forked ? ctx->stream_rest : ctx->stream);
                if (forked) {
                    hipEvent_t ev_join = dvs_event_get(ctx);
                    if (!ev_join || hipEventRecord(ev_join, ctx->stream_rest) != hipSuccess ||
                        hipStreamWaitEvent(ctx->stream, ev_join, 0) != hipSuccess) {
                        (void)hipGetLastError();
                        (void)hipStreamSynchronize(ctx->stream_rest);  // (no event: the host orders the two streams)
                    } else {
                        m->rest_beside_head = true;
                    }
                    // (kept with the matrix until it goes, for the reason ev_fork is kept below: the
                    // context's stream may not have performed its wait yet when this call returns)
                    m->ev_join = ev_join;
                }
            }
            // (back to the pool only now: an event handed out again and re-recorded while the wait on
            // its first record is still queued would move that wait to the later record)
            if (ev_fork) dvs_event_put(ctx, ev_fork);
        } else {
            launch16(0, nseq, hot_rows, ctx->stream);
        }
    } else if (pk16) {
        const size_t lds16 = ((B * 2 + 15) & ~15ull) + (CLOG_TBL + 32) * sizeof(double);
        if (packed) {
            rc = set_dyn_lds(ctx, kmer_hist_kernel<true, true, true, false, true>, lds16);
            if (!rc)
                hipLaunchKernelGGL((kmer_hist_kernel<true, true, true, false, true>), dim3(nseq), dim3(128), lds16, ctx->stream,
                                   d_seqs, d_pmask, nbytes, d_off, static_cast<const KTile *>(nullptr), m->d_counts,
                                   m->d_totals, m->d_entropy, ctx->d_clog_tbl, k, ns, B, hot_rows, 0u, uni_base, uni_stride);
        } else if (ns4) {
            rc = set_dyn_lds(ctx, kmer_hist_kernel<true, true, true>, lds16);
            if (!rc)
                hipLaunchKernelGGL((kmer_hist_kernel<true, true, true>), dim3(nseq), dim3(128), lds16, ctx->stream,
                                   d_seqs, d_pmask, nbytes, d_off, static_cast<const KTile *>(nullptr), m->d_counts,
                                   m->d_totals, m->d_entropy, ctx->d_clog_tbl, k, ns, B, hot_rows, 0u, uni_base, uni_stride);
        } else {
            rc = set_dyn_lds(ctx, kmer_hist_kernel<false, true, true>, lds16);
            if (!rc)
                hipLaunchKernelGGL((kmer_hist_kernel<false, true, true>), dim3(nseq), dim3(128), lds16, ctx->stream,
                                   d_seqs, d_pmask, nbytes, d_off, static_cast<const KTile *>(nullptr), m->d_counts,
                                   m->d_totals, m->d_entropy, ctx->d_clog_tbl, k, ns, B, hot_rows, 0u, uni_base, uni_stride);
        }
    } else
        DVS_LAUNCH_HIST_ANY(nseq, static_cast<const KTile *>(nullptr), nthreads, hot_rows);
    if (!rc && n_tiles) {
        DVS_LAUNCH_HIST_ANY(uint32_t(n_tiles), d_tiles, tile_threads, 0u);
        if (!rc)
            hipLaunchKernelGGL(row_stats_kernel, dim3(uint32_t(n_long)), dim3(HIST_THREADS), 0,
                               ctx->stream, m->d_counts, d_rows, m->d_totals, m->d_entropy, B);
    }
#undef DVS_LAUNCH_HIST_ANY
#undef DVS_LAUNCH_HIST
    if (!rc && (e = hipGetLastError()) != hipSuccess) rc = dvs_hip_fail(ctx, e, "histogram launch");
    if (!rc && no_wait && head_event_done) return DVS_OK;  // (the head's totals are already on their way)
    if (!rc && no_wait) {
        // Device-resident input: nothing here needs the host to wait.  The first rows' totals are
        // copied to a pinned block behind the kernels and an event marks their arrival; whoever
        // needs them (the selection's seeds) waits on that event -- after doing the rest of its
        // set-up while the histogram kernel runs.
        void *pin = nullptr;
        if (dvs_pinned_get(ctx, &pin) == DVS_OK) {
            m->head_count = uint32_t(std::min<size_t>(nseq, 4096 / sizeof(uint32_t)));
            m->h_head_pinned = static_cast<uint32_t *>(pin);
            m->ev_built = dvs_event_get(ctx);
            if (m->ev_built &&
                hipMemcpyAsync(m->h_head_pinned, m->d_totals, size_t(m->head_count) * 4, hipMemcpyDeviceToHost,
                               ctx->stream) == hipSuccess &&
                hipEventRecord(m->ev_built, ctx->stream) == hipSuccess) {
                cleanup();  // (the lists go back to the cache; stream order protects them)
                return DVS_OK;
            }
            if (m->ev_built) dvs_event_put(ctx, m->ev_built);
            dvs_pinned_put(ctx, pin);
            m->ev_built = nullptr;
            m->h_head_pinned = nullptr;
        }
    }
    // the totals of the first rows travel back in the same synchronisation (seed rows of a selection)
    m->h_head_totals.assign(std::min<size_t>(nseq, 4096), 0u);
    if (!rc && !m->h_head_totals.empty() &&
        hipMemcpyAsync(m->h_head_totals.data(), m->d_totals, m->h_head_totals.size() * 4, hipMemcpyDeviceToHost,
                       ctx->stream) != hipSuccess)
        m->h_head_totals.clear();
    // the lists go back to the cache; stream order protects them until the kernels ran
    if (hipStreamSynchronize(ctx->stream) != hipSuccess && !rc)
        rc = dvs_set_error(ctx, DVS_ERR_RUNTIME, "histogram kernels failed");
    cleanup();
    return rc;
}

int dvs_matrix_fill_freq_entropy(dvs_ctx *ctx, dvs_matrix *m) {
    hipLaunchKernelGGL(freq_entropy_kernel, dim3(m->nrows), dim3(HIST_THREADS), 0, ctx->stream,
                       m->d_freqs, m->d_entropy, m->nbins);
    DVS_HIP(ctx, hipGetLastError());
    return DVS_OK;
}

int dvs_matrix_fill_compacted(dvs_ctx *ctx, dvs_matrix *m, const double *d_in, const double *d_meta) {
    hipLaunchKernelGGL(compact_freq_rows_kernel, dim3(m->nrows), dim3(HIST_THREADS), 0, ctx->stream, d_in, d_meta,
                       m->d_freqs, m->d_totals, m->d_src_row, m->nrows, m->nbins);
    DVS_HIP(ctx, hipGetLastError());
    return DVS_OK;
}

int dvs_matrix_fill_freq_totals(dvs_ctx *ctx, dvs_matrix *m, const double *d_meta) {
    hipLaunchKernelGGL(freq_totals_kernel, dim3((m->nrows + 255) / 256), dim3(256), 0, ctx->stream,
                       m->d_totals, d_meta, m->nrows);
    DVS_HIP(ctx, hipGetLastError());
    return DVS_OK;
}
