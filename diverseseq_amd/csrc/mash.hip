// MinHash bottom-s sketches and pairwise mash / euclidean distances.
//
// Replaces, for batches:
//   murmurhash3_32  src/distance.rs:21-49  (NOT standard murmur3: every BYTE is a
//                   32-bit block, h starts at seed ^ len, no tail, fmix32 finish)
//   hash_kmer       :65-87   (mash-canonical = lexicographic min of k-mer / revcomp)
//   get_kmer_hashes :101-134 (windows holding a symbol >= num_states are skipped)
//   mash_sketch     :151-182 (the sketch_size smallest DISTINCT hashes, ascending)
//   mash_distance   diverse_seq/distance.py:230-291, N x N driver :165-173 and the
//                   strided rows of diverse_seq/cluster.py:640-644
//   euclidean_distance(s) diverse_seq/distance.py:294-336
//
// Device pipeline per sequence:
//   1. hash_filter_kernel: a tile of windows per workgroup, the tile's bytes staged
//      in LDS, one window per lane per step; hashes inside the sequence's current
//      range (lo, hi] that the tile has not produced before (LDS hash set) are
//      appended to the sequence's candidate list.  Hashes are ~uniform, so
//      hi = 2^32 * (1.5 s + 256) / n_windows keeps ~1.5 s candidates and the other
//      ~n_windows hashes never leave registers;
//   2. sort_select_kernel: one workgroup per sequence sorts the candidates in LDS
//      (bitonic), drops duplicates and appends them to the sketch.  If the sketch
//      is still short and hi < 2^32 - 1 the host moves the range up and repeats
//      for that sequence; if a range holds more than SORT_CAP candidates it is
//      halved.  The result is the exact bottom-s set of distinct hashes.
#include "dvs_internal.h"

#include <algorithm>
#include <type_traits>
#include <cmath>

namespace {

constexpr int MASH_THREADS = 256;
constexpr uint32_t MASH_TILE = 8192;       // windows per workgroup
constexpr uint32_t SORT_CAP = 16384;       // candidates one workgroup sorts in LDS (64 KB)
constexpr int MAX_K = 64;

struct MTile {
    uint64_t begin;  // first window START (absolute byte offset)
    uint32_t count;  // windows in this tile
    uint32_t seq;
};

__device__ __forceinline__ uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }

__device__ __forceinline__ uint32_t mix_byte(uint32_t h, uint32_t v) {
    uint32_t k = v * 0xCC9E2D51u;
    k = rotl32(k, 15);
    k *= 0x1B873593u;
    h ^= k;
    h = rotl32(h, 13);
    return h * 5u + 0xE6546B64u;
}
__device__ __forceinline__ uint32_t fmix32(uint32_t h) {
    h ^= h >> 16;
    h *= 0x85EBCA6Bu;
    h ^= h >> 13;
    h *= 0xC2B2AE35u;
    h ^= h >> 16;
    return h;
}

// hash of window w[0..k) (bytes in LDS); valid=false if any symbol >= ns
__device__ __forceinline__ uint32_t hash_window(const uint8_t *w, uint32_t k, uint32_t ns,
                                                bool canonical, bool *valid) {
    bool ok = true;
    bool use_rev = false;
    if (canonical) {
        // first position where kmer and revcomp differ decides (distance.rs:69-78)
        for (uint32_t i = 0; i < k; i++) {
            const uint32_t a = w[i];
            const uint32_t b = (uint32_t(w[k - 1 - i]) + 2u) & 3u;  // (base + 2) % 4, reversed
            if (a < b) break;
            if (a > b) {
                use_rev = true;
                break;
            }
        }
    }
    uint32_t h = 0x9747B28Cu ^ k;
    for (uint32_t i = 0; i < k; i++) {
        const uint32_t a = w[i];
        ok = ok && (a < ns);
        const uint32_t v = use_rev ? ((uint32_t(w[k - 1 - i]) + 2u) & 3u) : a;
        h = mix_byte(h, v);
    }
    // NB canonical decision was made on raw bytes; an invalid window is dropped anyway
    *valid = ok;
    return fmix32(h);
}

// LDS hash set over one tile: returns true if h was not yet present
__device__ __forceinline__ bool tile_insert(uint32_t *tbl, uint32_t mask, uint32_t h) {
    uint32_t slot = (h * 0x9E3779B1u) & mask;
    // (at most one trip round the table: a full table -- more distinct in-range hashes in the tile
    // than it has slots, which only the small table of the DNA kernel can meet -- reports "fresh";
    // the sort stage drops duplicates anyway, the set only keeps repeats from flooding the list)
    for (uint32_t probes = 0; probes <= mask; probes++) {
        const uint32_t old = atomicCAS(&tbl[slot], 0xFFFFFFFFu, h);
        if (old == 0xFFFFFFFFu) return true;
        if (old == h) return false;
        slot = (slot + 1) & mask;
    }
    return true;
}

// Hashes every window of a tile; hashes h with lo < h <= hi (lo as int64, -1 = none)
// that were not seen before IN THIS TILE are appended to the sequence's candidate
// list.  The tile-level dedup bounds the copies of any value by the number of
// tiles, so repeats / homopolymers cannot flood the candidate list.
__global__ __launch_bounds__(MASH_THREADS) void hash_filter_kernel(
    const uint8_t *__restrict__ seqs, const MTile *__restrict__ tiles, uint32_t k, uint32_t ns,
    int canonical, const long long *__restrict__ lo, const uint32_t *__restrict__ hi,
    const uint8_t *__restrict__ active, uint32_t *__restrict__ cand,
    const uint64_t *__restrict__ cand_off, const uint32_t *__restrict__ cand_cap,
    uint32_t *__restrict__ cand_cnt) {
    __shared__ uint8_t sbytes[MASH_TILE + MAX_K + 16];
    __shared__ uint32_t tbl[2 * MASH_TILE];
    __shared__ uint32_t s_max_seen;  // a genuine 0xFFFFFFFF hash (the table's empty marker)
    const MTile t = tiles[blockIdx.x];
    if (!active[t.seq]) return;
    const uint32_t nbytes = t.count + k - 1;
    for (uint32_t i = threadIdx.x; i < nbytes; i += MASH_THREADS) sbytes[i] = seqs[t.begin + i];
    for (uint32_t i = threadIdx.x; i < 2 * MASH_TILE; i += MASH_THREADS) tbl[i] = 0xFFFFFFFFu;
    if (threadIdx.x == 0) s_max_seen = 0;
    __syncthreads();
    const long long lo_q = lo[t.seq];
    const uint32_t hi_q = hi[t.seq];
    const uint32_t cap = cand_cap[t.seq];
    uint32_t *out = cand + cand_off[t.seq];
    for (uint32_t i = threadIdx.x; i < t.count; i += MASH_THREADS) {
        bool ok;
        const uint32_t h = hash_window(sbytes + i, k, ns, canonical != 0, &ok);
        if (ok && (long long)h > lo_q && h <= hi_q) {
            bool fresh;
            if (h == 0xFFFFFFFFu) fresh = atomicExch(&s_max_seen, 1u) == 0u;
            else fresh = tile_insert(tbl, 2 * MASH_TILE - 1, h);
            if (fresh) {
                const uint32_t slot = atomicAdd(&cand_cnt[t.seq], 1u);
                if (slot < cap) out[slot] = h;
            }
        }
    }
}

// ---- DNA fast path (num_states == 4, k <= 32) ------------------------------------------------
// The same hashes from 2-bit packed bases.  The tile's bytes are packed once into LDS words
// (16 bases per word, the earliest base in the top bits, plus a 16-bit invalid mask), a window is
// three LDS words and a funnel shift instead of k byte reads, the mash-canonical choice is one
// integer comparison of the window with its bit-reversed complement, and a hash round has no
// 32-bit multiply left: v * 0xCC9E2D51 -> rotl 15 -> * 0x1B873593 takes four values for v in 0..3
// (selected, not computed) and h * 5 + c is a shift-add.  (v_mul_lo_u32 runs at a quarter of the
// rate of the other integer instructions; the byte-wise kernel spends three per base.)
constexpr uint32_t mash_round_const(uint32_t v) {
    uint32_t k = v * 0xCC9E2D51u;
    k = (k << 15) | (k >> 17);
    return k * 0x1B873593u;
}
__device__ __forceinline__ uint32_t mash_round(uint32_t h, uint32_t v) {
    constexpr uint32_t c1 = mash_round_const(1), c2 = mash_round_const(2), c3 = mash_round_const(3);
    const uint32_t odd = (v & 1u) ? c1 : 0u;
    const uint32_t odd_hi = (v & 1u) ? c3 : c2;
    h ^= (v & 2u) ? odd_hi : odd;
    h = (h << 13) | (h >> 19);
    // h * 5 + c as one shift-add (written (h << 2) + h the compiler turns it back into a
    // v_mad_u64_u32, which issues at a quarter of the rate)
    uint32_t t;
    asm("v_lshl_add_u32 %0, %1, 2, %1" : "=v"(t) : "v"(h));
    return t + 0xE6546B64u;
}

__device__ __forceinline__ uint32_t mash_round_k(uint32_t h, uint32_t kk) {  // the round with its constant given
    h ^= kk;
    h = (h << 13) | (h >> 19);
    uint32_t t;
    asm("v_lshl_add_u32 %0, %1, 2, %1" : "=v"(t) : "v"(h));
    return t + 0xE6546B64u;
}

// TBLW: words of the tile's hash set.  The full 2 x MASH_TILE (64 KB) leaves room for two workgroups
// per CU; when the hash range lets only a few windows per tile through (genomes: ~0.2 %), 2048 words
// do and the CU holds eight workgroups, which hides the serial chain of the rounds much better.
// PACKED: the sequences are the planes of the packed form (dvs_packed: `seqs` points at the code words,
// `pmask` at the mask words), whose words are this kernel's LDS words: staging a tile is two loads per 16 bases.
//
// Round 4: what the counters said (profiles/r04_pmc_hash_*.csv: the SIMDs issue vector instructions 92 % of the
// kernel's cycles) and what was cut.  A lane used to take ONE window per step: three 8-byte LDS reads and a
// 64-bit funnel shift to cut it out of the packed words, a 48-bit mask extraction for its validity, and per
// round an address computation and a 4-byte LDS read for the round's constant.  Now a lane takes the SIXTEEN
// windows that start in one packed word: the three words are read once, the validity of all sixteen is one
// smear of the (almost always zero) mask bits, a window is one v_alignbit with an immediate shift, and the
// constants of FOUR rounds come from one 16-byte LDS read of a 256-entry table indexed by the next four bases
// (its address does not depend on the hash, so the reads run ahead of the serial chain).
template <bool K16, uint32_t TBLW, bool PACKED = false>  // K16: k <= 16, a window fits 32 bits
__global__ __launch_bounds__(MASH_THREADS) void hash_filter_dna_kernel(
    const uint8_t *__restrict__ seqs, const uint16_t *__restrict__ pmask, uint64_t nbytes_all,
    const MTile *__restrict__ tiles, uint32_t ntiles, uint32_t k,
    int canonical, const long long *__restrict__ lo, const uint32_t *__restrict__ hi,
    const uint8_t *__restrict__ active, uint32_t *__restrict__ cand,
    const uint64_t *__restrict__ cand_off, const uint32_t *__restrict__ cand_cap,
    uint32_t *__restrict__ cand_cnt) {
    constexpr uint32_t NW = (MASH_TILE + MAX_K + 15 + 15) / 16 + 3;
    __shared__ uint2 pk[NW];  // x: 16 bases packed, y: their invalid mask
    __shared__ uint32_t tbl[TBLW];
    __shared__ uint32_t s_max_seen;
    // The round constants of FOUR consecutive bases (first base in the top bit pair), 256 entries of 16 bytes: one
    // address computation and one LDS read per four rounds.  The lanes' entries are as good as random, so the
    // reads meet bank conflicts (64 % of the LDS cycles, profiles/r04_pmc_hash_t4_a.csv) -- the conflict-free
    // alternative, sixteen 8-byte entries for two bases, needs twice the address arithmetic and measured slower
    // (9.95 against 8.83 ms for C5, profiles/r04_pmc_hash_t2_a.csv: the vector ALU is what is left to save).
    __shared__ uint4 s_t4[256];
    {
        const uint32_t v = threadIdx.x;  // (MASH_THREADS == 256: one entry a thread)
        s_t4[v] = make_uint4(mash_round_const(v >> 6), mash_round_const((v >> 4) & 3u), mash_round_const((v >> 2) & 3u),
                             mash_round_const(v & 3u));
    }
    // A workgroup takes tiles blockIdx.x, + gridDim.x, ...: a tile is ~13 us of work, and with one tile per
    // workgroup the 366 000 workgroups of 1000 genomes were dispatched no faster than the chip could finish
    // them -- 4.7 of 8 waves a SIMD resident, the shader engines idle 28 % of the kernel
    // (profiles/r04_pmc_hash_before_*.csv).
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const MTile t = tiles[tile];
    if (!active[t.seq]) continue;
    if (tile != blockIdx.x) __syncthreads();  // (the previous tile's words and hash set have been read)
    const uint64_t base_al = t.begin & ~15ull;
    const uint32_t nbytes = t.count + k - 1;
    const uint32_t nwords = uint32_t((t.begin - base_al + nbytes + 15) >> 4);
    for (uint32_t j = threadIdx.x; j < nwords + 3; j += MASH_THREADS) {
        const uint64_t a = base_al + uint64_t(j) * 16;
        if constexpr (PACKED) {
            uint2 w = make_uint2(0u, 0xFFFFu);  // invalid filler
            if (j < nwords && a < nbytes_all)   // (positions behind the end are flagged inside the last word)
                w = make_uint2(reinterpret_cast<const uint32_t *>(seqs)[a >> 4], pmask[a >> 4]);
            pk[j] = w;
            continue;
        }
        uint4 v = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);  // invalid filler
        if (j < nwords) {
            if (a + 16 <= nbytes_all) {
                v = *reinterpret_cast<const uint4 *>(seqs + a);
            } else {
                uint32_t w[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
                for (int i = 0; i < 16; i++)
                    if (a + i < nbytes_all) {
                        w[i >> 2] &= ~(0xFFu << (8 * (i & 3)));
                        w[i >> 2] |= uint32_t(seqs[a + i]) << (8 * (i & 3));
                    }
                v = make_uint4(w[0], w[1], w[2], w[3]);
            }
        }
        pk[j] = make_uint2(dvs_pack16(v), dvs_inv16(v));
    }
    for (uint32_t i = threadIdx.x; i < TBLW; i += MASH_THREADS) tbl[i] = 0xFFFFFFFFu;
    if (threadIdx.x == 0) s_max_seen = 0;
    __syncthreads();
    const long long lo_q = lo[t.seq];
    const uint32_t hi_q = hi[t.seq];
    const uint32_t cap = cand_cap[t.seq];
    uint32_t *out = cand + cand_off[t.seq];
    const uint32_t rel0 = uint32_t(t.begin - base_al);
    const uint32_t rel_end = rel0 + t.count;            // windows start at rel0 .. rel_end - 1 (relative to base_al)
    const uint32_t ngroups = (rel_end + 15) >> 4;       // a group: the sixteen windows starting in one packed word
    const uint64_t kmask = k == 32 ? ~0ull : ((1ull << (2 * k)) - 1);
    const uint32_t h0 = 0x9747B28Cu ^ k;
    const uint32_t k4 = k & ~3u, krem = k & 3u;
    auto emit = [&](uint32_t h) {
        h = fmix32(h);
        if ((long long)h > lo_q && h <= hi_q) {
            bool fresh;
            if (h == 0xFFFFFFFFu) fresh = atomicExch(&s_max_seen, 1u) == 0u;
            else fresh = tile_insert(tbl, TBLW - 1, h);
            if (fresh) {
                const uint32_t slot = atomicAdd(&cand_cnt[t.seq], 1u);
                if (slot < cap) out[slot] = h;
            }
        }
    };
    for (uint32_t g = threadIdx.x; g < ngroups; g += MASH_THREADS) {
        const uint2 w0 = pk[g], w1 = pk[g + 1], w2 = pk[g + 2];
        // bit 15 - sh of ok: the window starting at base sh of this word lies in the tile and holds no invalid base
        const uint32_t sh_lo = g == 0 ? rel0 : 0u;
        const uint32_t left = rel_end - (g << 4);
        const uint32_t sh_hi = left < 16u ? left : 16u;
        uint32_t ok = (0xFFFFu >> sh_lo) & ~(0xFFFFu >> sh_hi) & 0xFFFFu;
        const uint64_t I48 = (uint64_t(w0.y) << 32) | (uint64_t(w1.y) << 16) | w2.y;  // bit 47 - p: base p of the 48 is invalid
        if (I48) {  // bit 47 - p of S: some base of p .. p + k - 1 is invalid
            uint64_t S = I48;
            uint32_t cover = 1;
            while (2 * cover <= k) {
                S |= S << cover;
                cover *= 2;
            }
            if (cover < k) S |= S << (k - cover);
            ok &= ~uint32_t(S >> 32);
        }
        if (!ok) continue;
        // The sixteen hashes first -- pure arithmetic with no side effect, so the compiler interleaves the serial
        // chains of several windows and issues their table reads ahead (a lone chain waits an LDS round trip
        // per four rounds: with ~5 waves a SIMD the counters showed no instruction in flight 39 % of the time) --
        // then the few that fall into the range.  A window that does not count is hashed anyway and dropped.
        uint32_t hs[16];
        // four rounds from one table entry
        auto r4 = [&](uint32_t h, const uint4 q) {
            h = mash_round_k(h, q.x);
            h = mash_round_k(h, q.y);
            h = mash_round_k(h, q.z);
            return mash_round_k(h, q.w);
        };
        auto rrem = [&](uint32_t h, const uint4 q) {  // the last k % 4 rounds
            h = mash_round_k(h, q.x);
            if (krem > 1) h = mash_round_k(h, q.y);
            if (krem > 2) h = mash_round_k(h, q.z);
            return h;
        };
        if constexpr (K16) {
            const uint32_t c0 = w0.x, c1 = w1.x;
            const uint32_t kmask32 = uint32_t(kmask);
#pragma unroll
            for (int sh = 0; sh < 16; sh++) {
                // the window, left-aligned (its first base in the top bit pair; what follows its last base is never looked at)
                uint32_t xt = sh ? __builtin_amdgcn_alignbit(c0, c1, (32 - 2 * sh) & 31) : c0;
                if (canonical) {
                    // reverse complement: pairs in reverse order, each base + 2 mod 4 (= its top bit flipped);
                    // lexicographic order of the bases = numeric order (distance.rs:69-78)
                    const uint32_t K = xt >> (32 - 2 * k);
                    uint32_t R = __brev(K);  // pair order reversed, bits inside a pair swapped
                    R = ((R >> 1) & 0x55555555u) | ((R & 0x55555555u) << 1);
                    R = (R >> (32 - 2 * k)) ^ (0xAAAAAAAAu & kmask32);
                    xt = (R < K ? R : K) << (32 - 2 * k);
                }
                // every table entry the window needs is requested before the first round (the addresses depend on
                // the bases alone); k <= 16: at most four, and none that no round will use
                const uint4 q0 = s_t4[xt >> 24];
                const uint4 q1 = k > 4 ? s_t4[(xt >> 16) & 255u] : q0;
                const uint4 q2 = k > 8 ? s_t4[(xt >> 8) & 255u] : q0;
                const uint4 q3 = k > 12 ? s_t4[xt & 255u] : q0;
                uint32_t h = h0;
                if (k4 >= 4) h = r4(h, q0);
                if (k4 >= 8) h = r4(h, q1);
                if (k4 >= 12) h = r4(h, q2);
                if (k4 >= 16) h = r4(h, q3);
                if (krem) h = rrem(h, k4 == 0 ? q0 : k4 == 4 ? q1 : k4 == 8 ? q2 : q3);
                hs[sh] = h;
            }
        } else {
            const uint32_t c0 = w0.x, c1 = w1.x, c2 = w2.x;
#pragma unroll
            for (int sh = 0; sh < 16; sh++) {
                const uint32_t vh = sh ? __builtin_amdgcn_alignbit(c0, c1, (32 - 2 * sh) & 31) : c0;
                const uint32_t vl = sh ? __builtin_amdgcn_alignbit(c1, c2, (32 - 2 * sh) & 31) : c1;
                uint64_t xt = (uint64_t(vh) << 32) | vl;  // left-aligned, 17 <= k <= 32
                if (canonical) {
                    const uint64_t K = xt >> (64 - 2 * k);
                    uint64_t R = __brevll(K);
                    R = ((R >> 1) & 0x5555555555555555ull) | ((R & 0x5555555555555555ull) << 1);
                    R = (R >> (64 - 2 * k)) ^ (0xAAAAAAAAAAAAAAAAull & kmask);
                    xt = (R < K ? R : K) << (64 - 2 * k);
                }
                uint32_t h = h0;
                for (uint32_t b = 0; b < k4; b += 4) {
                    h = r4(h, s_t4[uint32_t(xt >> 56)]);
                    xt <<= 8;
                }
                if (krem) h = rrem(h, s_t4[uint32_t(xt >> 56)]);
                hs[sh] = h;
            }
        }
#pragma unroll
        for (int sh = 0; sh < 16; sh++)
            if ((ok >> (15 - sh)) & 1u) emit(hs[sh]);
    }
    }  // tiles
}

// One workgroup per listed sequence: bitonic sort of <= SORT_CAP candidates in LDS,
// unique, append after the lens[q] entries already final (they are all smaller).
// status: 0 sketch complete, 1 range exhausted and more needed, 2 overflow.
__global__ __launch_bounds__(1024) void sort_select_kernel(
    const uint32_t *__restrict__ seq_list, const uint32_t *__restrict__ cand,
    const uint64_t *__restrict__ cand_off, const uint32_t *__restrict__ cand_cap,
    const uint32_t *__restrict__ cand_cnt, const uint32_t *__restrict__ hi, uint32_t s,
    uint32_t *__restrict__ sketches, uint32_t *__restrict__ lens, uint32_t *__restrict__ status) {
    extern __shared__ uint32_t keys[];
    __shared__ uint32_t scan[1024];
    const uint32_t q = seq_list[blockIdx.x];
    const uint32_t cnt = cand_cnt[q], cap = cand_cap[q];
    if (cnt > cap) {
        if (threadIdx.x == 0) status[q] = 2;
        return;
    }
    uint32_t n2 = 1;
    while (n2 < cnt) n2 <<= 1;
    const uint32_t *in = cand + cand_off[q];
    // pad with 0xFFFFFFFF: padding and a genuine 0xFFFFFFFF are interchangeable at the tail
    for (uint32_t i = threadIdx.x; i < n2; i += blockDim.x) keys[i] = i < cnt ? in[i] : 0xFFFFFFFFu;
    __syncthreads();
    for (uint32_t size = 2; size <= n2; size <<= 1) {
        for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
            for (uint32_t i = threadIdx.x; i < (n2 >> 1); i += blockDim.x) {
                const uint32_t lo = 2 * i - (i & (stride - 1));
                const uint32_t hi2 = lo + stride;
                const bool up = (lo & size) == 0;
                const uint32_t a = keys[lo], b = keys[hi2];
                if ((a > b) == up) {
                    keys[lo] = b;
                    keys[hi2] = a;
                }
            }
            __syncthreads();
        }
    }
    // stable compaction of first occurrences: per-thread contiguous chunks + block scan
    const uint32_t per = (cnt + blockDim.x - 1) / blockDim.x;
    const uint32_t b0 = min(cnt, threadIdx.x * per), b1 = min(cnt, b0 + per);
    uint32_t mine = 0;
    for (uint32_t i = b0; i < b1; i++) mine += (i == 0 || keys[i] != keys[i - 1]) ? 1u : 0u;
    scan[threadIdx.x] = mine;
    __syncthreads();
    for (uint32_t o = 1; o < blockDim.x; o <<= 1) {
        const uint32_t v = threadIdx.x >= o ? scan[threadIdx.x - o] : 0u;
        __syncthreads();
        scan[threadIdx.x] += v;
        __syncthreads();
    }
    const uint32_t have = lens[q];
    uint32_t pos = have + scan[threadIdx.x] - mine;
    uint32_t *sk = sketches + uint64_t(q) * s;
    for (uint32_t i = b0; i < b1; i++) {
        if (i == 0 || keys[i] != keys[i - 1]) {
            if (pos < s) sk[pos] = keys[i];
            pos++;
        }
    }
    __syncthreads();  // every thread has read lens[q]
    if (threadIdx.x == blockDim.x - 1) {
        const uint32_t total = have + scan[threadIdx.x];
        lens[q] = min(total, s);
        status[q] = (total >= s || hi[q] == 0xFFFFFFFFu) ? 0u : 1u;
    }
}

// mash_distance for the pair (i, j < i); one thread per pair (distance.py:230-291).  A lane's merge takes
// one hash of either sketch per step, so what decides the kernel's speed is how the two sketches reach the
// lane.  Row i is shared by the block's 256 pairs: staged once in LDS and read in place (the lanes of a
// wave stand within a few dozen hashes of each other, so their reads fall into different banks or onto the
// same word).  Row j differs per lane: 64 lanes walk 64 different rows, and the texture path serves one
// lane's 16 bytes per clock however the loads are arranged -- so every byte of row j must be requested
// exactly once.  Each lane keeps a circular WINDOW of 16-byte blocks of its row in LDS, stored word-major
// (word w of lane t at [w][t]: a wave's 64 reads always fall into 64 different banks), and every PAIR_W
// steps -- wave-uniform control flow -- the blocks requested a trip earlier are written into it and the
// ones the lane has used up since are requested (so no step ever waits for memory); the step itself is two
// LDS reads and some fifteen integer instructions, no branch.  The windows are private to their lane: no
// barrier in the loop.
//   geometry: a lane at word ri needs blocks [b, b + AHEAD), b = ri >> 2, for PAIR_W steps; it moves on by at
//   most ADV blocks per trip; so blocks up to b + AHEAD + ADV are requested at every trip and a window of
//   SLOTS >= AHEAD + ADV blocks never overwrites a block at or ahead of the lane's position.
#ifndef DVS_PAIR_W
#define DVS_PAIR_W 4
#endif
constexpr int PAIR_W = DVS_PAIR_W;               // merge steps per trip
constexpr int PAIR_AHEAD = (PAIR_W + 3 + 3) / 4;  // blocks a trip may read
constexpr int PAIR_ADV = (PAIR_W + 3) / 4;        // blocks a trip may leave behind
constexpr int PAIR_SLOTS = PAIR_AHEAD + PAIR_ADV <= 4 ? 4 : 8;  // blocks of the circular window
static_assert(PAIR_AHEAD + PAIR_ADV <= PAIR_SLOTS, "window too small for PAIR_W");
constexpr int PAIR_THREADS = 256;
constexpr uint32_t PAIR_ROW_LDS = 8192;  // hashes of row i staged in LDS at most (longer sketches are read through L1)

__global__ __launch_bounds__(PAIR_THREADS) void mash_pairs_kernel(
    const uint32_t *__restrict__ sketches, const uint32_t *__restrict__ lens, uint32_t nseq,
    uint32_t k, uint32_t s, uint32_t stride, uint32_t row_start, uint32_t row_stride, int symmetric,
    uint32_t row_lds, double *__restrict__ dist, uint32_t *__restrict__ zerodiv) {
    __shared__ uint32_t s_win[PAIR_SLOTS * 4][PAIR_THREADS];
    extern __shared__ uint32_t s_left[];  // row_lds + 4 words
    // grid = (rows, blocks of 256 columns), rows fastest: workgroups go to the XCDs round-robin by their linear
    // index, so with the columns fastest XCD x would get column block x % 4 only -- and column block 0 has
    // work in every row, block 3 in a quarter of them (measured: 3 rounds on two XCDs, half a round on two
    // others).  Longest rows first.
    const uint32_t i = row_start + (gridDim.x - 1 - blockIdx.x) * row_stride;
    const uint32_t j = blockIdx.y * PAIR_THREADS + threadIdx.x;
    if (i >= nseq || blockIdx.y * PAIR_THREADS >= i) return;
    const bool mine = j < i;
    const uint32_t *Lg = sketches + uint64_t(i) * stride;
    const uint32_t *R = sketches + uint64_t(mine ? j : 0u) * stride;
    const uint32_t nl = lens[i], nr = mine ? lens[j] : 0u;
    const bool staged = nl <= row_lds;
    if (staged) {
        for (uint32_t x = threadIdx.x; x < nl; x += PAIR_THREADS) s_left[x] = Lg[x];
        __syncthreads();
    }
    const bool vec = (stride & 3u) == 0;  // rows 16-byte aligned
    uint32_t *win = &s_win[0][threadIdx.x];
    uint32_t inter = 0, uni = 0, li = 0, ri = 0;
    bool run = mine && s > 0 && nl > 0 && nr > 0;
    auto fetch = [&](uint32_t blk) {
        const uint32_t at = blk * 4u;
        uint4 v;
        if (vec && at < stride) {
            v = *reinterpret_cast<const uint4 *>(R + at);
        } else {
            v.x = at + 0u < stride ? R[at + 0u] : 0u;
            v.y = at + 1u < stride ? R[at + 1u] : 0u;
            v.z = at + 2u < stride ? R[at + 2u] : 0u;
            v.w = at + 3u < stride ? R[at + 3u] : 0u;
        }
        return v;
    };
    auto land = [&](uint32_t blk, const uint4 &v) {
        uint32_t *slot = win + (blk & (PAIR_SLOTS - 1)) * 4u * PAIR_THREADS;
        slot[0 * PAIR_THREADS] = v.x;
        slot[1 * PAIR_THREADS] = v.y;
        slot[2 * PAIR_THREADS] = v.z;
        slot[3 * PAIR_THREADS] = v.w;
    };
    // PAIR_W steps.  Row i in LDS: no branch, a lane that has finished keeps reading where it stands and adds
    // zeros (s_left has slack behind the row); row i through L1 (longer than the staging area): reads only while running.
    auto steps_lds = [&]() {
        uint32_t on = run;
#pragma unroll
        for (int t = 0; t < PAIR_W; t++) {
            const uint32_t l = s_left[li];
            const uint32_t r = win[(ri & (PAIR_SLOTS * 4 - 1)) * PAIR_THREADS];
            const uint32_t a = on & uint32_t(l <= r), b = on & uint32_t(r <= l);
            li += a;
            ri += b;
            inter += a & b;
            uni += on;
            on &= uint32_t(uni < s) & uint32_t(li < nl) & uint32_t(ri < nr);
        }
        run = on != 0;
    };
    auto steps_l1 = [&]() {
#pragma unroll
        for (int t = 0; t < PAIR_W; t++) {
            if (run) {
                const uint32_t l = Lg[li];
                const uint32_t r = win[(ri & (PAIR_SLOTS * 4 - 1)) * PAIR_THREADS];
                li += (l <= r);
                ri += (r <= l);
                inter += (l == r);
                uni++;
                run = uni < s && li < nl && ri < nr;
            }
        }
    };
    // blocks [0, have) of row j have been through the window, [have, pf_hi) are on their way in pf[]
    uint32_t have = PAIR_AHEAD, pf_hi = PAIR_AHEAD;
    uint4 pf[PAIR_ADV];
    if (run) {
#pragma unroll
        for (int q = 0; q < PAIR_AHEAD; q++) land(q, fetch(q));
    }
    auto trip_head = [&]() {  // the blocks requested a trip ago into the window, the next ones requested
#pragma unroll
        for (int q = 0; q < PAIR_ADV; q++)
            if (have + q < pf_hi) land(have + q, pf[q]);
        have = pf_hi;
        pf_hi = run ? max(have, (ri >> 2) + PAIR_AHEAD + PAIR_ADV) : have;
#pragma unroll
        for (int q = 0; q < PAIR_ADV; q++)
            if (have + q < pf_hi) pf[q] = fetch(have + q);
    };
    // (a wave without a single pair -- the columns right of the diagonal in a row's last block -- must not walk the
    // full-sketch loop: with sketch_size = 4 000 000 000, which the reference's own ctree tests pass to mean "every
    // k-mer", such a wave idled through a billion trips -- 90 s per call)
    if (staged && __any(run) && __all(!run || (nl >= s && nr >= s))) {
        // Full sketches (the usual case): a pair takes exactly s steps -- li <= uni <= s <= nl, and the same for
        // ri, so neither sketch can run out first -- and a step needs no "still running" mask: two reads, three
        // compares, three adds-with-carry and the two addresses.  (Lanes without a pair idle through it.)
        for (uint32_t done = 0; done < s; done += PAIR_W) {
            trip_head();
            if (run) {
                if (s - done >= uint32_t(PAIR_W)) {
#pragma unroll
                    for (int t = 0; t < PAIR_W; t++) {
                        const uint32_t l = s_left[li];
                        const uint32_t r = win[(ri & (PAIR_SLOTS * 4 - 1)) * PAIR_THREADS];
                        li += (l <= r);
                        ri += (r <= l);
                        inter += (l == r);
                    }
                } else {
                    for (uint32_t t = done; t < s; t++) {
                        const uint32_t l = s_left[li];
                        const uint32_t r = win[(ri & (PAIR_SLOTS * 4 - 1)) * PAIR_THREADS];
                        li += (l <= r);
                        ri += (r <= l);
                        inter += (l == r);
                    }
                }
            }
        }
        uni = run ? s : 0u;  // (= li + ri - inter)
    } else {
        while (__any(run)) {  // distance.py:260-274, PAIR_W steps per trip
            trip_head();
            if (staged) steps_lds();
            else steps_l1();
        }
    }
    if (!mine) return;
    if (uni < s) {  // :276-281
        if (li < nl) uni += nl - li;
        if (ri < nr) uni += nr - ri;
        uni = min(uni, s);
    }
    double d;
    if (uni == 0) {
        atomicExch(zerodiv, 1u);
        d = NAN;
    } else if (inter == uni) {
        d = 0.0;
    } else if (inter == 0) {
        d = 1.0;
    } else {
        const double jac = double(inter) / double(uni);
        d = -log(2.0 * jac / (1.0 + jac)) / double(k);
        if (d > 1.0) d = 1.0;
    }
    dist[uint64_t(i) * nseq + j] = d;
    if (symmetric) dist[uint64_t(j) * nseq + i] = d;
}

// ||f_i - f_j||_2 (diverse_seq/distance.py:335-336), f = counts / total.  Workgroup (i, g) stages row
// i's frequencies in LDS chunk by chunk and its eight waves take the rows j = 8 g .. 8 g + 7 below the
// diagonal, one each: row i is read once per eight pairs, row j streamed by one wave with 16-byte loads
// where the bin count allows.  Only the lower triangle does work; both mirror cells are written.
constexpr int EUC_THREADS = 512;
constexpr uint32_t EUC_CHUNK = 4096;  // bins of row i staged at a time (32 KB)
template <typename T>
__global__ __launch_bounds__(EUC_THREADS) void euclid_kernel(const T *__restrict__ mat,
                                                            const uint32_t *__restrict__ totals, uint64_t B,
                                                            uint32_t n, double *__restrict__ dist) {
    __shared__ double fi[EUC_CHUNK];
    const uint32_t i = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t j = blockIdx.y * (EUC_THREADS / 64) + wave;
    if (blockIdx.y * (EUC_THREADS / 64) >= i) return;  // the whole group is on or above the diagonal
    const T *a = mat + uint64_t(i) * B;
    const bool live = j < i;
    const T *b = mat + uint64_t(live ? j : 0) * B;
    const double ta = double(totals[i]), tb = double(totals[live ? j : 0]);
    double acc = 0.0;
    for (uint64_t c0 = 0; c0 < B; c0 += EUC_CHUNK) {
        const uint32_t cn = uint32_t(B - c0 < EUC_CHUNK ? B - c0 : EUC_CHUNK);
        __syncthreads();
        for (uint32_t x = threadIdx.x; x < cn; x += EUC_THREADS) fi[x] = double(a[c0 + x]) / ta;
        __syncthreads();
        if (live)
            for (uint32_t x = lane; x < cn; x += 64) {
                const double d = fi[x] - double(b[c0 + x]) / tb;
                acc += d * d;
            }
    }
    acc = dvs_wave_sum(acc);
    if (live && lane == 0) {
        const double d = sqrt(acc);
        dist[uint64_t(i) * n + j] = d;
        dist[uint64_t(j) * n + i] = d;
    }
}

// The tile list of a batch, written on the device: tile t belongs to the sequence q with tpre[q] <= t < tpre[q + 1]
// (binary search by the tile's own thread); every tile but a sequence's first begins on a packed word -- an absolute
// position that is a multiple of 16 -- and a full tile ends on one, so its windows are exactly 512 groups of sixteen.
// (The list used to be built on the host and uploaded: 366 000 entries, 8.8 MB from pageable memory, for 1000
// genomes -- 2 ms of every sketch call that no kernel ran beside.)
__global__ __launch_bounds__(256) void mash_tiles_kernel(const uint64_t *__restrict__ off, const uint64_t *__restrict__ tpre,
                                                        uint32_t nseq, uint32_t k, uint64_t ntiles, MTile *__restrict__ tiles) {
    const uint64_t t = uint64_t(blockIdx.x) * 256 + threadIdx.x;
    if (t >= ntiles) return;
    uint32_t lo = 0, hi = nseq;  // tpre[lo] <= t < tpre[hi]
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (tpre[mid] <= t) lo = mid;
        else hi = mid;
    }
    const uint64_t o = off[lo], w = off[lo + 1] - o - (k - 1);  // (a sequence with tiles has at least one window)
    auto mn = [](uint64_t a, uint64_t b) { return a < b ? a : b; };
    const uint64_t first = mn(uint64_t(MASH_TILE) - (o & 15), w);
    const uint64_t j = t - tpre[lo];
    const uint64_t b = j == 0 ? 0 : first + (j - 1) * MASH_TILE;
    MTile m;
    m.begin = o + b;
    m.count = uint32_t(j == 0 ? first : mn(uint64_t(MASH_TILE), w - b));
    m.seq = lo;
    tiles[t] = m;
}

struct PooledBuf {  // a block of the context's cache, handed back on scope exit
    dvs_ctx *ctx;
    void *p = nullptr;
    ~PooledBuf() { dvs_dev_free(ctx, p); }
    template <typename T>
    T *as() { return static_cast<T *>(p); }
};

}  // namespace

// sketches of a batch, left in HBM: nseq x sketch_size uint32 (ascending, first d_lens[i] valid) in
// blocks of the context's cache (the caller hands them back with dvs_dev_free)
static int mash_sketch_view(dvs_ctx *ctx, const dvs_seq_view &sv, const uint64_t *offsets,
                            uint32_t nseq, uint32_t k, uint32_t sketch_size, uint32_t num_states,
                            int mash_canonical, uint32_t **d_sk_out, uint32_t **d_lens_out) {
    *d_sk_out = nullptr;
    *d_lens_out = nullptr;
    DVS_HIP(ctx, hipSetDevice(ctx->device));
    const uint32_t s = sketch_size;
    const uint64_t nbytes = offsets[nseq];
    const bool packed = sv.codes != nullptr;
    if (nbytes > sv.nbytes)
        return dvs_set_error(ctx, DVS_ERR_VALUE, "offsets[%u] = %llu beyond the %llu bases of the batch", nseq,
                             (unsigned long long)nbytes, (unsigned long long)sv.nbytes);
    if (packed && (num_states != 4 || k > 32))
        return dvs_set_error(ctx, DVS_ERR_UNSUPPORTED, "packed sequences are sketched with four states and k <= 32 (k = %u, %u states)",
                             k, num_states);
    const uint8_t *d_seqs = packed ? reinterpret_cast<const uint8_t *>(sv.codes) : sv.seqs;
    const uint16_t *d_pmask = sv.mask;

    // tiles and the first hash range (lo, hi] per sequence: hashes are ~uniform, so
    // hi = 2^32 * (1.5 s + 256) / n_windows holds ~1.5 s candidates; a sequence of
    // at most SORT_CAP windows takes the whole range at once
    std::vector<long long> lo(nseq, -1);
    std::vector<uint32_t> hi(nseq), cap(nseq), nwin(nseq);
    std::vector<uint64_t> coff(nseq + 1, 0), tpre(nseq + 1, 0);
    const uint64_t want = uint64_t(s) + s / 2 + 256;
    for (uint32_t q = 0; q < nseq; q++) {
        const uint64_t len = offsets[q + 1] - offsets[q];
        const uint64_t w = len >= k ? len - k + 1 : 0;
        if (w > 0xFFFFFFFFull)
            return dvs_set_error(ctx, DVS_ERR_UNSUPPORTED, "sequence %u longer than 2^32", q);
        nwin[q] = uint32_t(w);
        cap[q] = uint32_t(std::min<uint64_t>(std::max<uint64_t>(w, 1), SORT_CAP));
        if (w <= SORT_CAP) {
            hi[q] = 0xFFFFFFFFu;
        } else {
            const long double tv = (long double)std::min<uint64_t>(want, SORT_CAP / 2) / (long double)w * 4294967296.0L;
            hi[q] = tv >= 4294967295.0L ? 0xFFFFFFFFu : uint32_t(tv);
        }
        coff[q + 1] = coff[q] + cap[q];
        // tiles of this sequence (mash_tiles_kernel lays them out): a first one up to the next packed word
        // boundary + MASH_TILE windows, then whole tiles
        uint64_t nt = 0;
        if (w) {
            const uint64_t first = std::min<uint64_t>(uint64_t(MASH_TILE) - (offsets[q] & 15), w);
            nt = 1 + (w - first + MASH_TILE - 1) / MASH_TILE;
        }
        tpre[q + 1] = tpre[q] + nt;
    }
    const uint64_t n_tiles = tpre[nseq];
    if (n_tiles > 0xFFFFFFFFull) return dvs_set_error(ctx, DVS_ERR_UNSUPPORTED, "more than 2^32 - 1 tiles in one batch");
    // (every block comes from the context's cache: a sketch call used to hipMalloc and hipFree a dozen buffers)
    PooledBuf d_tiles{ctx}, d_lo{ctx}, d_hi{ctx}, d_cap{ctx}, d_coff{ctx}, d_cnt{ctx}, d_active{ctx}, d_cand{ctx}, d_list{ctx},
        d_status{ctx}, d_off{ctx}, d_tpre{ctx};
    PooledBuf d_sk{ctx}, d_lens{ctx};  // (the two results; released here unless handed over)
    {
        const size_t ntile = std::max<size_t>(size_t(n_tiles), 1);
        int arc = dvs_dev_alloc(ctx, &d_tiles.p, ntile * sizeof(MTile), "sketch tiles");
        if (!arc) arc = dvs_dev_alloc(ctx, &d_lo.p, size_t(nseq) * 8, "hash range (lo)");
        if (!arc) arc = dvs_dev_alloc(ctx, &d_hi.p, size_t(nseq) * 4, "hash range (hi)");
        if (!arc) arc = dvs_dev_alloc(ctx, &d_cap.p, size_t(nseq) * 4, "candidate caps");
        if (!arc) arc = dvs_dev_alloc(ctx, &d_coff.p, (size_t(nseq) + 1) * 8, "candidate offsets");
        if (!arc) arc = dvs_dev_alloc(ctx, &d_cnt.p, size_t(nseq) * 4, "candidate counts");
        if (!arc) arc = dvs_dev_alloc(ctx, &d_active.p, size_t(nseq), "active flags");
        if (!arc) arc = dvs_dev_alloc(ctx, &d_cand.p, std::max<uint64_t>(coff[nseq], 1) * 4, "candidates");
        if (!arc) arc = dvs_dev_alloc(ctx, &d_list.p, size_t(nseq) * 4, "active list");
        if (!arc) arc = dvs_dev_alloc(ctx, &d_status.p, size_t(nseq) * 4, "sort status");
        if (!arc) arc = dvs_dev_alloc(ctx, &d_off.p, (size_t(nseq) + 1) * 8, "sequence offsets");
        if (!arc) arc = dvs_dev_alloc(ctx, &d_tpre.p, (size_t(nseq) + 1) * 8, "tile prefix");
        if (!arc) arc = dvs_dev_alloc(ctx, &d_sk.p, size_t(nseq) * s * 4, "sketches");
        if (!arc) arc = dvs_dev_alloc(ctx, &d_lens.p, size_t(nseq) * 4, "sketch lengths");
        if (arc) return arc;
    }
    // (host vectors are the sources of asynchronous copies: the stream is drained before this function returns on
    // every path -- the round loop below ends each round with a synchronisation -- except the early error
    // returns, which drain it themselves)
    hipError_t ue = hipMemcpyAsync(d_off.p, offsets, (size_t(nseq) + 1) * 8, hipMemcpyHostToDevice, ctx->stream);
    if (ue == hipSuccess) ue = hipMemcpyAsync(d_tpre.p, tpre.data(), (size_t(nseq) + 1) * 8, hipMemcpyHostToDevice, ctx->stream);
    if (ue == hipSuccess) ue = hipMemcpyAsync(d_cap.p, cap.data(), size_t(nseq) * 4, hipMemcpyHostToDevice, ctx->stream);
    if (ue == hipSuccess) ue = hipMemcpyAsync(d_coff.p, coff.data(), (size_t(nseq) + 1) * 8, hipMemcpyHostToDevice, ctx->stream);
    if (ue == hipSuccess) ue = hipMemsetAsync(d_lens.p, 0, size_t(nseq) * 4, ctx->stream);
    if (ue == hipSuccess) ue = hipMemsetAsync(d_sk.p, 0, size_t(nseq) * s * 4, ctx->stream);
    if (ue == hipSuccess && n_tiles) {
        hipLaunchKernelGGL(mash_tiles_kernel, dim3(uint32_t((n_tiles + 255) / 256)), dim3(256), 0, ctx->stream, d_off.as<uint64_t>(),
                           d_tpre.as<uint64_t>(), nseq, k, n_tiles, d_tiles.as<MTile>());
        ue = hipGetLastError();
    }
    if (ue != hipSuccess) {
        (void)hipStreamSynchronize(ctx->stream);
        return dvs_hip_fail(ctx, ue, "sketch set-up");
    }

    std::vector<uint8_t> active(nseq, 1);
    std::vector<uint32_t> status(nseq, 0), lens(nseq, 0);
    for (uint32_t q = 0; q < nseq; q++)
        if (nwin[q] == 0) active[q] = 0;  // L < k: empty sketch (distance.rs:102-104)
    const size_t sort_lds = SORT_CAP * 4;
    {
        const int lrc = dvs_raise_dyn_lds(ctx, reinterpret_cast<const void *>(sort_select_kernel), sort_lds);
        if (lrc) return lrc;
    }

    for (int round = 0;; round++) {
        std::vector<uint32_t> list;
        for (uint32_t q = 0; q < nseq; q++)
            if (active[q]) list.push_back(q);
        if (list.empty()) break;
        if (round >= 200)
            return dvs_set_error(ctx, DVS_ERR_RUNTIME, "mash sketch range search did not converge");
        DVS_HIP(ctx, hipMemcpyAsync(d_lo.p, lo.data(), nseq * 8, hipMemcpyHostToDevice, ctx->stream));
        DVS_HIP(ctx, hipMemcpyAsync(d_hi.p, hi.data(), nseq * 4, hipMemcpyHostToDevice, ctx->stream));
        DVS_HIP(ctx, hipMemcpyAsync(d_active.p, active.data(), nseq, hipMemcpyHostToDevice, ctx->stream));
        DVS_HIP(ctx, hipMemcpyAsync(d_list.p, list.data(), list.size() * 4, hipMemcpyHostToDevice, ctx->stream));
        DVS_HIP(ctx, hipMemsetAsync(d_cnt.p, 0, nseq * 4, ctx->stream));
        if (num_states == 4 && k <= 32 && true) {  // 2-bit packed windows
            // the small hash set when no active sequence lets more than ~512 windows of a tile through
            bool small = true;
            for (uint32_t q : list) {
                const long double frac = ((long double)hi[q] - (long double)lo[q]) / 4294967296.0L;
                if (frac * MASH_TILE > 512.0L) small = false;
            }
            // (sixteen workgroups' worth of tiles per CU at most: every workgroup loops over its share)
            const uint32_t dna_grid = uint32_t(std::min<size_t>(size_t(n_tiles), size_t(ctx->n_cu) * 16));
#define DVS_LAUNCH_DNA(K16, TBLW, PKD)                                                                         \
    hipLaunchKernelGGL((hash_filter_dna_kernel<K16, TBLW, PKD>), dim3(dna_grid), dim3(MASH_THREADS), 0, \
                       ctx->stream, d_seqs, d_pmask, nbytes, d_tiles.as<MTile>(), uint32_t(n_tiles), k, mash_canonical, d_lo.as<long long>(), \
                       d_hi.as<uint32_t>(), d_active.as<uint8_t>(), d_cand.as<uint32_t>(), d_coff.as<uint64_t>(), \
                       d_cap.as<uint32_t>(), d_cnt.as<uint32_t>())
#define DVS_LAUNCH_DNA_ANY(PKD)                                  \
    do {                                                         \
        if (k <= 16 && small) DVS_LAUNCH_DNA(true, 2048, PKD);   \
        else if (k <= 16) DVS_LAUNCH_DNA(true, 2 * MASH_TILE, PKD); \
        else if (small) DVS_LAUNCH_DNA(false, 2048, PKD);        \
        else DVS_LAUNCH_DNA(false, 2 * MASH_TILE, PKD);          \
    } while (0)
            if (packed) DVS_LAUNCH_DNA_ANY(true);
            else DVS_LAUNCH_DNA_ANY(false);
#undef DVS_LAUNCH_DNA_ANY
#undef DVS_LAUNCH_DNA
        } else
            hipLaunchKernelGGL(hash_filter_kernel, dim3(uint32_t(n_tiles)), dim3(MASH_THREADS), 0,
                               ctx->stream, d_seqs, d_tiles.as<MTile>(), k, num_states, mash_canonical,
                               d_lo.as<long long>(), d_hi.as<uint32_t>(), d_active.as<uint8_t>(),
                               d_cand.as<uint32_t>(), d_coff.as<uint64_t>(), d_cap.as<uint32_t>(),
                               d_cnt.as<uint32_t>());
        hipLaunchKernelGGL(sort_select_kernel, dim3(uint32_t(list.size())), dim3(1024), sort_lds,
                           ctx->stream, d_list.as<uint32_t>(), d_cand.as<uint32_t>(),
                           d_coff.as<uint64_t>(), d_cap.as<uint32_t>(), d_cnt.as<uint32_t>(),
                           d_hi.as<uint32_t>(), s, d_sk.as<uint32_t>(), d_lens.as<uint32_t>(),
                           d_status.as<uint32_t>());
        DVS_HIP(ctx, hipGetLastError());
        DVS_HIP(ctx, hipMemcpyAsync(status.data(), d_status.p, nseq * 4, hipMemcpyDeviceToHost, ctx->stream));
        DVS_HIP(ctx, hipMemcpyAsync(lens.data(), d_lens.p, nseq * 4, hipMemcpyDeviceToHost, ctx->stream));
        DVS_HIP(ctx, hipStreamSynchronize(ctx->stream));
        for (uint32_t q : list) {
            const uint64_t width = uint64_t((long long)hi[q] - lo[q]);
            if (status[q] == 0) {
                active[q] = 0;
            } else if (status[q] == 1) {
                // everything in (lo, hi] is final; continue above it with a range sized by
                // the density seen so far (at least doubling)
                const uint64_t found = std::max<uint32_t>(lens[q], 1);
                const uint64_t need = s - lens[q];
                uint64_t next = std::max<uint64_t>(2 * width, uint64_t((long double)(uint64_t(hi[q]) + 1) / found * need * 1.5L));
                lo[q] = hi[q];
                const uint64_t nh = uint64_t(hi[q]) + std::max<uint64_t>(next, 1);
                hi[q] = nh >= 0xFFFFFFFFull ? 0xFFFFFFFFu : uint32_t(nh);
            } else {
                // more than SORT_CAP (tile-distinct) candidates in the range: halve it
                if (width <= 1)
                    return dvs_set_error(ctx, DVS_ERR_UNSUPPORTED,
                                         "sequence %u: one hash value occurs in more than %u tiles", q, SORT_CAP);
                hi[q] = uint32_t(lo[q] + (long long)(width / 2));
            }
        }
    }
    *d_sk_out = d_sk.as<uint32_t>();
    *d_lens_out = d_lens.as<uint32_t>();
    d_sk.p = d_lens.p = nullptr;  // handed over
    return DVS_OK;
}

// the byte form, from either side of PCIe.  Host memory: four-state sequences of k <= 32 cross packed
// (pack.hip: 3/8 of the bytes) and are sketched from the packed words; anything else is copied as it is.
static int mash_sketch_core(dvs_ctx *ctx, const uint8_t *seqs, int seqs_on_device, const uint64_t *offsets,
                            uint32_t nseq, uint32_t k, uint32_t sketch_size, uint32_t num_states,
                            int mash_canonical, uint32_t **d_sk_out, uint32_t **d_lens_out) {
    const uint64_t nbytes = offsets[nseq];
    dvs_seq_view sv;
    sv.nbytes = nbytes;
    if (seqs_on_device) {
        sv.seqs = seqs;
        return mash_sketch_view(ctx, sv, offsets, nseq, k, sketch_size, num_states, mash_canonical, d_sk_out, d_lens_out);
    }
    DVS_HIP(ctx, hipSetDevice(ctx->device));
    if (k <= 32 && dvs_packed_upload_wanted(ctx, num_states, nbytes)) {
        dvs_packed *p = nullptr;
        int rc = dvs_packed_alloc(ctx, nbytes, &p);
        if (!rc) rc = dvs_packed_fill_from_host(ctx, p, seqs);
        if (!rc) {
            sv.codes = p->d_codes;
            sv.mask = p->d_mask;
            rc = mash_sketch_view(ctx, sv, offsets, nseq, k, sketch_size, num_states, mash_canonical, d_sk_out, d_lens_out);
        }
        if (p) {
            (void)hipStreamSynchronize(ctx->stream);
            dvs_packed_destroy(p);
        }
        return rc;
    }
    PooledBuf d_own{ctx};
    int rc = dvs_dev_alloc(ctx, &d_own.p, nbytes ? nbytes : 16, "sequence upload buffer");
    if (rc) return rc;
    if (nbytes) DVS_HIP(ctx, hipMemcpyAsync(d_own.p, seqs, nbytes, hipMemcpyHostToDevice, ctx->stream));
    sv.seqs = d_own.as<uint8_t>();
    rc = mash_sketch_view(ctx, sv, offsets, nseq, k, sketch_size, num_states, mash_canonical, d_sk_out, d_lens_out);
    (void)hipStreamSynchronize(ctx->stream);  // (the upload buffer goes back to the cache)
    return rc;
}

static int mash_check_args(dvs_ctx *ctx, const uint64_t *offsets, uint32_t k) {
    if (!ctx || !offsets) return dvs_set_error(ctx, DVS_ERR_VALUE, "null argument");
    if (k == 0 || k > MAX_K)
        return dvs_set_error(ctx, DVS_ERR_UNSUPPORTED, "mash k = %u outside 1..%d", k, MAX_K);
    return DVS_OK;
}

extern "C" int dvs_mash_sketch(dvs_ctx *ctx, const uint8_t *seqs, int seqs_on_device,
                               const uint64_t *offsets, uint32_t nseq, uint32_t k,
                               uint32_t sketch_size, uint32_t num_states, int mash_canonical,
                               uint32_t *sketches_out, uint32_t *lens_out) {
    if (!sketches_out || !lens_out) return dvs_set_error(ctx, DVS_ERR_VALUE, "null argument");
    int rc = mash_check_args(ctx, offsets, k);
    if (rc) return rc;
    if (sketch_size == 0) {
        for (uint32_t i = 0; i < nseq; i++) lens_out[i] = 0;
        return DVS_OK;
    }
    if (nseq == 0) return DVS_OK;
    uint32_t *d_sk = nullptr, *d_lens = nullptr;
    rc = mash_sketch_core(ctx, seqs, seqs_on_device, offsets, nseq, k, sketch_size, num_states, mash_canonical,
                          &d_sk, &d_lens);
    if (rc) return rc;
    hipError_t e = hipMemcpyAsync(sketches_out, d_sk, size_t(nseq) * sketch_size * 4, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(lens_out, d_lens, size_t(nseq) * 4, hipMemcpyDeviceToHost, ctx->stream);
    const hipError_t se = hipStreamSynchronize(ctx->stream);
    dvs_dev_free(ctx, d_sk);
    dvs_dev_free(ctx, d_lens);
    if (e != hipSuccess) return dvs_hip_fail(ctx, e, "sketch copy");
    if (se != hipSuccess) return dvs_hip_fail(ctx, se, "sketch copy");
    return DVS_OK;
}

// ---- sketches that stay in HBM between the two stages of ctree (sketch, then N x N distances)
struct dvs_sketches {
    dvs_ctx *ctx = nullptr;
    uint32_t nseq = 0, stride = 0;
    uint32_t *d_sk = nullptr, *d_lens = nullptr;
    bool borrowed = false;  // dvs_sketches_from_device: the caller's buffers, not handed back to the cache
};

extern "C" int dvs_sketches_build(dvs_ctx *ctx, const uint8_t *seqs, int seqs_on_device, const uint64_t *offsets,
                                  uint32_t nseq, uint32_t k, uint32_t sketch_size, uint32_t num_states,
                                  int mash_canonical, dvs_sketches **out) {
    if (!out) return dvs_set_error(ctx, DVS_ERR_VALUE, "null argument");
    *out = nullptr;
    int rc = mash_check_args(ctx, offsets, k);
    if (rc) return rc;
    dvs_sketches *sk = new dvs_sketches();
    sk->ctx = ctx;
    sk->nseq = nseq;
    sk->stride = sketch_size;
    dvs_ctx_retain(ctx);
    if (nseq && sketch_size) {
        rc = mash_sketch_core(ctx, seqs, seqs_on_device, offsets, nseq, k, sketch_size, num_states, mash_canonical,
                              &sk->d_sk, &sk->d_lens);
        if (rc) {
            dvs_ctx_release(ctx);
            delete sk;
            return rc;
        }
    }
    *out = sk;
    return DVS_OK;
}

extern "C" int dvs_sketches_build_packed(dvs_ctx *ctx, const dvs_packed *p, const uint64_t *offsets, uint32_t nseq,
                                         uint32_t k, uint32_t sketch_size, int mash_canonical, dvs_sketches **out) {
    if (!out || !p) return dvs_set_error(ctx, DVS_ERR_VALUE, "null argument");
    *out = nullptr;
    int rc = mash_check_args(ctx, offsets, k);
    if (rc) return rc;
    dvs_sketches *sk = new dvs_sketches();
    sk->ctx = ctx;
    sk->nseq = nseq;
    sk->stride = sketch_size;
    dvs_ctx_retain(ctx);
    if (nseq && sketch_size) {
        dvs_seq_view sv;
        sv.codes = p->d_codes;
        sv.mask = p->d_mask;
        sv.nbytes = p->nbases;
        p->async_readers = true;  // (dvs_packed_destroy waits for the kernels enqueued here)
        rc = mash_sketch_view(ctx, sv, offsets, nseq, k, sketch_size, 4, mash_canonical, &sk->d_sk, &sk->d_lens);
        if (rc) {
            dvs_ctx_release(ctx);
            delete sk;
            return rc;
        }
    }
    *out = sk;
    return DVS_OK;
}

extern "C" void dvs_sketches_destroy(dvs_sketches *sk) {
    if (!sk) return;
    if (!sk->borrowed) {
        dvs_dev_free(sk->ctx, sk->d_sk);
        dvs_dev_free(sk->ctx, sk->d_lens);
    }
    dvs_ctx_release(sk->ctx);
    delete sk;
}

extern "C" int dvs_sketches_get(dvs_ctx *ctx, const dvs_sketches *sk, uint32_t *sketches_out, uint32_t *lens_out) {
    if (!ctx || !sk || !lens_out) return dvs_set_error(ctx, DVS_ERR_VALUE, "null argument");
    if (!sk->d_sk) {
        for (uint32_t i = 0; i < sk->nseq; i++) lens_out[i] = 0;
        return DVS_OK;
    }
    if (sketches_out)
        DVS_HIP(ctx, hipMemcpyAsync(sketches_out, sk->d_sk, size_t(sk->nseq) * sk->stride * 4, hipMemcpyDeviceToHost,
                                    ctx->stream));
    DVS_HIP(ctx, hipMemcpyAsync(lens_out, sk->d_lens, size_t(sk->nseq) * 4, hipMemcpyDeviceToHost, ctx->stream));
    DVS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return DVS_OK;
}
extern "C" const void *dvs_sketches_dev(const dvs_sketches *sk) { return sk ? sk->d_sk : nullptr; }
extern "C" const void *dvs_sketches_dev_lens(const dvs_sketches *sk) { return sk ? sk->d_lens : nullptr; }

// the pair kernel over sketches in HBM; dist (host, nseq x nseq) receives the visited cells
static int mash_pairs_device(dvs_ctx *ctx, const uint32_t *d_sk, const uint32_t *d_lens, uint32_t nseq, uint32_t stride,
                             uint32_t k, uint32_t sketch_size, uint32_t row_start, uint32_t row_stride, int symmetric,
                             double *dist) {
    double *d_dist = nullptr;
    uint32_t *d_flag = nullptr;
    int rc = dvs_dev_alloc(ctx, (void **)&d_dist, size_t(nseq) * nseq * 8, "distance matrix");
    if (!rc) rc = dvs_dev_alloc(ctx, (void **)&d_flag, 4, "flag");
    if (rc) {
        dvs_dev_free(ctx, d_dist);
        return rc;
    }
    // Cells this call does not visit keep the caller's values.  A call over the whole triangle that
    // mirrors its cells (the usual one) visits everything but the diagonal: only those nseq cells travel
    // up (one strided copy); any other call starts from a copy of the caller's matrix.
    const bool whole = row_start == 0 && row_stride == 1 && symmetric;
    hipError_t e = whole ? hipMemcpy2DAsync(d_dist, (size_t(nseq) + 1) * 8, dist, (size_t(nseq) + 1) * 8, 8, nseq,
                                            hipMemcpyHostToDevice, ctx->stream)
                         : hipMemcpyAsync(d_dist, dist, size_t(nseq) * nseq * 8, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(d_flag, 0, 4, ctx->stream);
    const uint32_t nrows = (nseq - 1 - row_start) / row_stride + 1;
    uint32_t flag = 0;
    if (e == hipSuccess) {
        const dim3 grid(nrows, (nseq + PAIR_THREADS - 1) / PAIR_THREADS);
        const uint32_t row_lds = std::min(stride, PAIR_ROW_LDS);  // (no sketch is longer than the stride)
        hipLaunchKernelGGL(mash_pairs_kernel, grid, dim3(PAIR_THREADS), (row_lds + 4) * 4, ctx->stream, d_sk, d_lens, nseq, k,
                           sketch_size, stride, row_start, row_stride, symmetric, row_lds, d_dist, d_flag);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(dist, d_dist, size_t(nseq) * nseq * 8, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(&flag, d_flag, 4, hipMemcpyDeviceToHost, ctx->stream);
    const hipError_t se = hipStreamSynchronize(ctx->stream);
    dvs_dev_free(ctx, d_dist);
    dvs_dev_free(ctx, d_flag);
    if (e != hipSuccess) return dvs_hip_fail(ctx, e, "mash distances");
    if (se != hipSuccess) return dvs_hip_fail(ctx, se, "mash distances");
    if (flag) return dvs_set_error(ctx, DVS_ERR_ZERODIV, "division by zero");  // 0 / 0, distance.py:283
    return DVS_OK;
}

extern "C" int dvs_sketches_distances(dvs_ctx *ctx, const dvs_sketches *sk, uint32_t k, uint32_t sketch_size,
                                      uint32_t row_start, uint32_t row_stride, int symmetric, double *dist) {
    if (!ctx || !sk || !dist) return dvs_set_error(ctx, DVS_ERR_VALUE, "null argument");
    if (sk->nseq < 2 || row_start >= sk->nseq) return DVS_OK;
    if (row_stride == 0) row_stride = 1;
    if (k == 0) return dvs_set_error(ctx, DVS_ERR_ZERODIV, "float division by zero");
    DVS_HIP(ctx, hipSetDevice(ctx->device));
    if (!sk->d_sk) return dvs_set_error(ctx, DVS_ERR_ZERODIV, "division by zero");  // every sketch empty
    return mash_pairs_device(ctx, sk->d_sk, sk->d_lens, sk->nseq, sk->stride, k, sketch_size, row_start, row_stride,
                             symmetric, dist);
}

// ---- the sharded ctree (diverse_seq/cluster.py:607-644) without a trip through the host: sketches gathered from the
// other ranks sit in a device buffer of the caller's (an all_gather's output), the strided rows of the lower triangle
// go into a device matrix of the caller's (an all_reduce's input).
extern "C" int dvs_sketches_from_device(dvs_ctx *ctx, const uint32_t *d_sketches, const uint32_t *d_lens, uint32_t nseq,
                                        uint32_t stride, dvs_sketches **out) {
    if (!ctx || !out || (nseq && (!d_lens || (stride && !d_sketches)))) return dvs_set_error(ctx, DVS_ERR_VALUE, "null argument");
    dvs_sketches *sk = new dvs_sketches();
    sk->ctx = ctx;
    sk->nseq = nseq;
    sk->stride = stride;
    sk->d_sk = const_cast<uint32_t *>(d_sketches);
    sk->d_lens = const_cast<uint32_t *>(d_lens);
    sk->borrowed = true;
    dvs_ctx_retain(ctx);
    *out = sk;
    return DVS_OK;
}

extern "C" int dvs_sketches_copy_to_device(dvs_ctx *ctx, const dvs_sketches *sk, uint32_t *d_dst, uint32_t dst_stride,
                                           uint32_t *d_dst_lens) {
    if (!ctx || !sk || !d_dst || !d_dst_lens) return dvs_set_error(ctx, DVS_ERR_VALUE, "null argument");
    if (dst_stride < sk->stride) return dvs_set_error(ctx, DVS_ERR_VALUE, "destination stride %u < sketch stride %u", dst_stride, sk->stride);
    if (!sk->nseq) return DVS_OK;
    DVS_HIP(ctx, hipSetDevice(ctx->device));
    if (!sk->d_sk) {  // (sketch_size 0: every sketch empty)
        DVS_HIP(ctx, hipMemsetAsync(d_dst_lens, 0, size_t(sk->nseq) * 4, ctx->stream));
        return DVS_OK;
    }
    DVS_HIP(ctx, hipMemcpy2DAsync(d_dst, size_t(dst_stride) * 4, sk->d_sk, size_t(sk->stride) * 4, size_t(sk->stride) * 4, sk->nseq,
                                  hipMemcpyDeviceToDevice, ctx->stream));
    DVS_HIP(ctx, hipMemcpyAsync(d_dst_lens, sk->d_lens, size_t(sk->nseq) * 4, hipMemcpyDeviceToDevice, ctx->stream));
    return DVS_OK;
}

// enqueued on the context's stream, nothing waited for: d_dist[nseq x nseq] receives the visited cells (the others keep
// what they hold), *d_zerodiv is set when a visited pair has two empty sketches (distance.py:283)
extern "C" int dvs_sketches_distances_device(dvs_ctx *ctx, const dvs_sketches *sk, uint32_t k, uint32_t sketch_size,
                                             uint32_t row_start, uint32_t row_stride, int symmetric, double *d_dist,
                                             uint32_t *d_zerodiv) {
    if (!ctx || !sk || !d_dist || !d_zerodiv) return dvs_set_error(ctx, DVS_ERR_VALUE, "null argument");
    if (sk->nseq < 2 || row_start >= sk->nseq) return DVS_OK;
    if (row_stride == 0) row_stride = 1;
    if (k == 0) return dvs_set_error(ctx, DVS_ERR_ZERODIV, "float division by zero");
    if (!sk->d_sk) return dvs_set_error(ctx, DVS_ERR_ZERODIV, "division by zero");
    DVS_HIP(ctx, hipSetDevice(ctx->device));
    const uint32_t nseq = sk->nseq;
    const uint32_t nrows = (nseq - 1 - row_start) / row_stride + 1;
    const dim3 grid(nrows, (nseq + PAIR_THREADS - 1) / PAIR_THREADS);
    const uint32_t row_lds = std::min(sk->stride, PAIR_ROW_LDS);
    hipLaunchKernelGGL(mash_pairs_kernel, grid, dim3(PAIR_THREADS), (row_lds + 4) * 4, ctx->stream, sk->d_sk, sk->d_lens, nseq, k,
                       sketch_size, sk->stride, row_start, row_stride, symmetric, row_lds, d_dist, d_zerodiv);
    DVS_HIP(ctx, hipGetLastError());
    return DVS_OK;
}

extern "C" int dvs_mash_distances(dvs_ctx *ctx, const uint32_t *sketches, uint32_t sketch_stride,
                                  const uint32_t *lens, uint32_t nseq, uint32_t k, uint32_t sketch_size,
                                  uint32_t row_start, uint32_t row_stride, int symmetric,
                                  double *dist) {
    if (!ctx || !lens || !dist || (!sketches && sketch_stride))
        return dvs_set_error(ctx, DVS_ERR_VALUE, "null argument");
    for (uint32_t i = 0; i < nseq; i++)
        if (lens[i] > sketch_stride)
            return dvs_set_error(ctx, DVS_ERR_VALUE, "lens[%u] = %u exceeds the sketch stride %u", i,
                                 lens[i], sketch_stride);
    if (nseq < 2 || row_start >= nseq) return DVS_OK;
    if (row_stride == 0) row_stride = 1;
    if (k == 0) return dvs_set_error(ctx, DVS_ERR_ZERODIV, "float division by zero");
    DVS_HIP(ctx, hipSetDevice(ctx->device));
    const uint32_t s = std::max<uint32_t>(sketch_stride, 1);
    uint32_t *d_sk = nullptr, *d_lens = nullptr;
    int rc = dvs_dev_alloc(ctx, (void **)&d_sk, size_t(nseq) * s * 4, "sketches");
    if (!rc) rc = dvs_dev_alloc(ctx, (void **)&d_lens, size_t(nseq) * 4, "sketch lengths");
    hipError_t e = hipSuccess;
    if (!rc && sketch_stride)
        e = hipMemcpyAsync(d_sk, sketches, size_t(nseq) * s * 4, hipMemcpyHostToDevice, ctx->stream);
    if (!rc && e == hipSuccess) e = hipMemcpyAsync(d_lens, lens, size_t(nseq) * 4, hipMemcpyHostToDevice, ctx->stream);
    if (!rc && e != hipSuccess) rc = dvs_hip_fail(ctx, e, "sketch upload");
    if (!rc) rc = mash_pairs_device(ctx, d_sk, d_lens, nseq, s, k, sketch_size, row_start, row_stride, symmetric, dist);
    else (void)hipStreamSynchronize(ctx->stream);
    dvs_dev_free(ctx, d_sk);
    dvs_dev_free(ctx, d_lens);
    return rc;
}

extern "C" int dvs_euclidean_distances(dvs_ctx *ctx, const dvs_matrix *m, double *dist) {
    if (!ctx || !m || !dist) return dvs_set_error(ctx, DVS_ERR_VALUE, "null argument");
    const uint32_t n = m->nrows;
    if (n == 0) return DVS_OK;
    if (n == 1) {
        dist[0] = 0.0;
        return DVS_OK;
    }
    const uint32_t groups = (n + EUC_THREADS / 64 - 1) / (EUC_THREADS / 64);
    if (groups > 65535u)
        return dvs_set_error(ctx, DVS_ERR_UNSUPPORTED, "%u rows: the %u x %u distance matrix is beyond this path", n, n, n);
    DVS_HIP(ctx, hipSetDevice(ctx->device));
    double *d_dist = nullptr;
    int rc = dvs_dev_alloc(ctx, (void **)&d_dist, size_t(n) * n * 8, "distance matrix");
    if (rc) return rc;
    hipError_t e = hipMemsetAsync(d_dist, 0, size_t(n) * n * 8, ctx->stream);  // the diagonal
    const dim3 grid(n, groups);
    if (e == hipSuccess) {
        dvs_mat_dispatch(m, [&](auto *mp) {
            using T = std::remove_cv_t<std::remove_pointer_t<decltype(mp)>>;
            hipLaunchKernelGGL((euclid_kernel<T>), grid, dim3(EUC_THREADS), 0, ctx->stream, mp, m->d_totals, m->nbins,
                               n, d_dist);
            return 0;
        });
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(dist, d_dist, size_t(n) * n * 8, hipMemcpyDeviceToHost, ctx->stream);
    const hipError_t se = hipStreamSynchronize(ctx->stream);
    dvs_dev_free(ctx, d_dist);
    if (e != hipSuccess) return dvs_hip_fail(ctx, e, "euclidean distances");
    if (se != hipSuccess) return dvs_hip_fail(ctx, se, "euclidean distances");
    return DVS_OK;
}
