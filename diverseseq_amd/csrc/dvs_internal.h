// Internal declarations shared by the translation units of libdvs_hip.so.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <map>
#include <string>
#include <vector>

#include "../../include/dvs_hip.h"

#define DVS_EPS 2.220446049250313e-16  // f64::EPSILON
#define DVS_HEAD_ROWS 1024u            // rows of a split build's first launch (kmer_hist.hip)

// Every switch the library takes from the environment, read ONCE when a context is made (and again
// only on dvs_ctx_refresh_knobs, a measurement / test aid): nothing on a per-call path calls getenv.
// All of them are measurement aids or escape hatches; the defaults are the product.
struct dvs_knobs {
    // The library's environment switches, read once per context (and again by dvs_ctx_refresh_knobs).  Every one
    // is listed in INTEGRATION.md with the test or measurement that uses it; there are no others.
    // k-mer matrices (kmer_hist.hip, api.cpp)
    bool no_uniform_offsets = false;  // DVS_NO_UNIFORM_OFFSETS: sequences of one length still get an offsets array on the device
    bool no_offsets_cache = false;    // DVS_NO_OFFSETS_CACHE: every build validates and uploads its offsets
    bool counts_u32 = false;          // DVS_COUNTS_U32: never 16-bit rows
    bool build_wait = false;          // DVS_BUILD_WAIT: device-resident builds wait for their kernels (no split build)
    bool no_packed_upload = false;    // DVS_NO_PACKED_UPLOAD: host sequences cross PCIe one byte per base
    bool cu_mask_set = false;         // HSA_CU_MASK / ROC_GLOBAL_CU_MASK present: the device reports CUs it will not give
    // selection engines (select.hip, persist.hip)
    bool no_fast_step = false;        // DVS_NO_FAST_STEP: the stepwise (row-sharded) mode runs scan / resolve / leave-one-out / finalize per step, as rounds 1-4 did
    bool no_persist = false;          // DVS_NO_PERSIST: the multi-launch engine serves every selection
    bool no_head_phase = false;       // DVS_NO_HEAD_PHASE: no head phase on the CU-masked stream beside the histogram
    bool persist_no_seeded = false;   // DVS_PERSIST_NO_SEEDED: the set-up kernels build the initial set
    bool persist_no_small = false;    // DVS_PERSIST_NO_SMALL: small sets use the general instantiation
    bool persist_no_coarse = false;   // DVS_PERSIST_NO_COARSE: no all-f32 tier in front of the f32-log tier
    bool persist_no_events = false;   // DVS_PERSIST_NO_EVENTS: a threshold no row reaches, long windows (the pure-stream measurement)
    bool persist_debug = false;       // DVS_PERSIST_DEBUG: per-launch report on stderr (phase stamps of a -DDVS_PERSIST_STAMPS build)
    int persist_wg_rounds = -1;       // DVS_PERSIST_WG_ROUNDS: rounds of the grid up to which a window is scanned a row per workgroup (-1: default)
    // ingest
    bool ingest_no_stream = false;    // DVS_INGEST_NO_STREAM: host files are uploaded whole before they are parsed
    // test-only (DVS_TEST_KNOBS=fake_persist_error): the first persistent launch's outcome is read as SEL_ERROR
    bool test_persist_fake_error = false;
    // test-only (DVS_TEST_KNOBS=long_tile_<n>): long rows are cut into tiles of n windows whatever the input's size
    // (the tiles only grow beyond 32768 windows for inputs of 2^27 bases and more: kmer_hist.hip dvs_hist_prepare)
    uint32_t test_long_tile = 0;
    // test-only (DVS_TEST_KNOBS=rowlog_ring_<n>): the stepwise selections' device-side ring of accepted rows holds n rows
    // (>= 4), so that a few dozen accepts wrap it several times before a tie is arbitrated
    uint32_t test_rowlog_ring = 0;
};
void dvs_knobs_from_env(dvs_knobs *k);

struct dvs_ctx {
    dvs_knobs knobs;
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipStream_t stream2 = nullptr;  // second stream (created on first use): a selection's set-up kernels beside
                                    // the histogram of the rest of the matrix
    // The CU split (created on first use, dvs_ctx_cu_split): `stream_head` may only use head_cus CUs
    // (the same number on every XCD), `stream_rest` only the others.  A split build runs the histogram
    // of everything behind the head rows on stream_rest while the selection's set-up and the HEAD
    // PHASE of its persistent engine -- the event-dense first rows of the stream, a chain of
    // hand-overs that keeps no CU busy -- run on stream_head beside it (kmer_hist.hip, select.hip).
    hipStream_t stream_head = nullptr, stream_rest = nullptr;
    int head_cus = 0;
    bool cu_split_tried = false;
    // persistent launches of this context that timed out at a grid barrier in a row (their workgroups
    // were not all resident): after three of them later selections go straight to the multi-launch
    // engine; one that runs to its end clears the count
    int persist_timeouts = 0;
    int n_cu = 0;
    size_t lds_per_block = 0;  // max dynamic LDS a block may ask for
    double *d_clog_tbl = nullptr;  // c log2 c, c < 256 (kmer_hist.hip)
    bool timing = false;
    std::string err;
    // device-memory cache: blocks released by dvs_dev_free are kept by size and
    // handed back by dvs_dev_alloc, so a steady-state loop (build matrix, select,
    // destroy) performs no hipMalloc / hipFree.  Released for real in
    // dvs_ctx_destroy, dvs_ctx_trim or when an allocation fails.
    std::multimap<size_t, void *> pool;
    std::vector<void *> pinned_pool;       // 4 KiB pinned host blocks (control-block mirrors)
    std::vector<hipEvent_t> event_pool;    // recycled HIP events
    std::map<void *, size_t> live;
    size_t pool_bytes = 0;
    // one reference for the owner (dropped by dvs_ctx_destroy) and one per matrix, selection and
    // sequence batch made from this context: their destroy functions hand blocks back to the caches
    // above, so the context outlives them whatever order the caller tears things down in
    int refs = 1;
    bool owner_gone = false;
    std::map<const void *, size_t> lds_raised;  // kernels whose dynamic-LDS limit was raised, and to what
    std::map<std::pair<const void *, size_t>, bool> persist_fits;  // (kernel, LDS bytes) -> a 512-thread workgroup fits a CU
    // The offsets of the last histogram build, on both sides (kmer_hist.hip): a caller that builds
    // again over the same sequences -- same offsets, k, byte count, compared by content -- skips the
    // validation pass, the tile lists and their uploads (~0.1 ms of host time per 100k sequences,
    // during which the GPU would wait for its first launch).
    struct OffsetsCache {
        // the offsets on the host, in PINNED memory: the validation pass writes them there and the upload
        // reads them from there, so it is a real asynchronous copy (a pageable source is staged by the
        // runtime inside the call); ev_up marks the last upload that read the block
        uint64_t *h_off = nullptr;
        size_t h_cap = 0, n_off = 0;  // capacity / valid entries (nseq + 1)
        hipEvent_t ev_up = nullptr;
        // host sources of the other uploads: pageable memory an async copy may still read after the
        // call returned, so they live as long as the cache entry
        std::vector<uint32_t> h_long_rows;
        std::vector<unsigned char> h_tiles;
        uint64_t nbytes = 0;
        uint32_t k = 0;
        void *d_off = nullptr, *d_rows = nullptr, *d_tiles = nullptr;
        size_t d_off_cap = 0;
        size_t n_long = 0, n_tiles = 0;
        // the last build's sequences all had one length and lay end to end: no offsets on the device
        bool uniform = false;
        uint64_t uni_base = 0, uni_stride = 0;
    } off_cache;
    // packed upload of host sequences (pack.hip): the pinned staging block, kept between calls, and the
    // event behind the last copy that read it
    void *h_pack = nullptr;
    size_t h_pack_cap = 0;
    hipEvent_t pack_ev = nullptr;
};
// hipFuncAttributeMaxDynamicSharedMemorySize >= bytes for kernel fn on this context's device
int dvs_raise_dyn_lds(dvs_ctx *ctx, const void *fn, size_t bytes);
void dvs_ctx_retain(dvs_ctx *ctx);
hipStream_t dvs_ctx_stream2(dvs_ctx *ctx);  // NULL when it cannot be created
bool dvs_ctx_cu_split(dvs_ctx *ctx);         // stream_head / stream_rest exist (false: no CU masks here)
void dvs_ctx_release(dvs_ctx *ctx);

// waits for a build that is still in flight (no-op otherwise) and moves the head totals to the vector
struct dvs_matrix;
int dvs_matrix_settle(dvs_ctx *ctx, const dvs_matrix *m);
int dvs_dev_alloc(dvs_ctx *ctx, void **ptr, size_t bytes, const char *what);
void dvs_dev_free(dvs_ctx *ctx, void *ptr);
void dvs_dev_trim(dvs_ctx *ctx);
int dvs_pinned_get(dvs_ctx *ctx, void **ptr);  // 4 KiB pinned block from the ctx cache
void dvs_pinned_put(dvs_ctx *ctx, void *ptr);
hipEvent_t dvs_event_get(dvs_ctx *ctx);
void dvs_event_put(dvs_ctx *ctx, hipEvent_t e);

int dvs_set_error(dvs_ctx *ctx, int code, const char *fmt, ...);
int dvs_hip_fail(dvs_ctx *ctx, hipError_t e, const char *what);

#define DVS_HIP(ctx, call)                                        \
    do {                                                          \
        hipError_t e__ = (call);                                  \
        if (e__ != hipSuccess) return dvs_hip_fail(ctx, e__, #call); \
    } while (0)

// N x B matrix resident in HBM.  kind 0: uint32 counts; kind 1: f64 freqs; kind 2: uint16 counts
// (a build in which every sequence is a single tile of <= 32768 windows: no count can reach 2^16,
// and both streaming phases -- the build's write, the scan's read -- move half the bytes).
struct dvs_matrix {
    int kind = 0;
    uint16_t *d_counts16 = nullptr;  // kind 2
    uint32_t nrows = 0;
    uint64_t nbins = 0;
    uint32_t k = 0, num_states = 0;
    uint32_t *d_counts = nullptr;  // kind 0
    double *d_freqs = nullptr;     // kind 1
    uint32_t *d_totals = nullptr;  // valid k-mers per row (kind 1: 1 for every row)
    double *d_entropy = nullptr;   // H(row freq vector), bits
    uint32_t *d_src_row = nullptr; // kind 1 built from flagged device rows: row r came from input row d_src_row[r]
    // totals of the first rows as the builder left them (copied in the build's own stream sync):
    // the selectors need their seeds' totals on the host and would otherwise pay a round trip
    std::vector<uint32_t> h_head_totals;
    // ... or, for a build that did not wait for its kernels (device-resident input), still on their
    // way: a pinned block the copy lands in and the event recorded behind it
    uint32_t *h_head_pinned = nullptr;
    hipEvent_t ev_built = nullptr;
    hipEvent_t ev_join = nullptr;  // split build: recorded behind the rest-of-matrix launch, waited for by the context's stream
    uint32_t head_count = 0;
    // rows [0, head_rows_built) were built by a launch of their own, finished when ev_built fires; the
    // rest of the matrix may still be in flight on the context's stream after that (0: no such split)
    uint32_t head_rows_built = 0;
    // ... and that rest was launched on the context's stream_rest (CU split): the head CUs are free
    // for a selection's head phase until the context's stream, which waits for it, moves on
    bool rest_beside_head = false;
    int device = 0;
    dvs_ctx *ctx = nullptr;  // owner of the allocations
};

// Four-state sequences in HBM at 3 bits per base (pack.hip, pack_host.cpp): positions index the
// concatenated buffer exactly as in the one-byte form, so a batch's offsets are the same in both.
//   d_codes[w]: bases 16 w .. 16 w + 15, two bits each, base 16 w in bits 31..30
//   d_mask[w]:  bit 15 - i set when base 16 w + i is invalid (>= 4, or behind the end of the buffer)
// nwords = ceil(nbases / 16) words of each plane are valid; the planes are allocated a little longer
// (a multiple of 16 bytes) so that vector loads near the end stay inside them.
struct dvs_packed {
    dvs_ctx *ctx = nullptr;
    uint32_t *d_codes = nullptr;
    uint16_t *d_mask = nullptr;
    uint64_t nbases = 0, nwords = 0;
    // kernels that read the planes were enqueued without being waited for (dvs_matrix_build_packed,
    // dvs_sketches_build_packed -- on the context's stream or, for a split build, its stream_rest): the planes
    // go back to the block cache, which orders reuse on the context's stream only, so dvs_packed_destroy drains
    // those streams first
    mutable bool async_readers = false;
};
// what the kernels take: either the byte form (seqs) or the packed planes
struct dvs_seq_view {
    const uint8_t *seqs = nullptr;  // one byte per base ...
    const uint32_t *codes = nullptr;  // ... or the packed planes (codes != NULL)
    const uint16_t *mask = nullptr;
    uint64_t nbytes = 0;  // readable bases
};
int dvs_packed_alloc(dvs_ctx *ctx, uint64_t nbases, dvs_packed **out);                       // pack.hip
int dvs_packed_fill_from_device(dvs_ctx *ctx, dvs_packed *p, const uint8_t *d_seqs);         // bytes in HBM -> planes
int dvs_packed_fill_from_host(dvs_ctx *ctx, dvs_packed *p, const uint8_t *seqs);             // host threads pack, chunks cross PCIe packed
bool dvs_packed_upload_wanted(const dvs_ctx *ctx, uint32_t num_states, uint64_t nbytes);

// f(typed row pointer) for the matrix's element type
template <typename F>
auto dvs_mat_dispatch(const dvs_matrix *m, F &&f) {
    if (m->kind == 0) return f(static_cast<const uint32_t *>(m->d_counts));
    if (m->kind == 2) return f(static_cast<const uint16_t *>(m->d_counts16));
    return f(static_cast<const double *>(m->d_freqs));
}

// ---- device helpers -------------------------------------------------------
#ifdef __HIPCC__
// 4 bases (little-endian bytes, earliest base in the low byte) -> 8 bits, the earliest base in the TOP two bits
__device__ __forceinline__ uint32_t dvs_pack4(uint32_t w) {
    const uint32_t x = w & 0x03030303u;
    return ((x << 6) | (x >> 4) | (x >> 14) | (x >> 24)) & 0xFFu;
}
// 4 bases -> 4 bits, bit set where the byte is >= 4, earliest base in bit 3
__device__ __forceinline__ uint32_t dvs_inv4(uint32_t w) {
    uint32_t t = w & 0xFCFCFCFCu;
    t |= t >> 4;
    t |= t >> 2;
    t |= t >> 1;
    t &= 0x01010101u;
    return ((t << 3) | (t >> 6) | (t >> 15) | (t >> 24)) & 0xFu;
}
// 16 bases -> a code word / a mask word of the packed form (dvs_packed)
__device__ __forceinline__ uint32_t dvs_pack16(uint4 v) {
    return (dvs_pack4(v.x) << 24) | (dvs_pack4(v.y) << 16) | (dvs_pack4(v.z) << 8) | dvs_pack4(v.w);
}
__device__ __forceinline__ uint32_t dvs_inv16(uint4 v) {
    return (dvs_inv4(v.x) << 12) | (dvs_inv4(v.y) << 8) | (dvs_inv4(v.z) << 4) | dvs_inv4(v.w);
}
__device__ __forceinline__ double dvs_wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// the same sum by DPP (register cross-lane moves, no LDS crossbar): ~23 instructions where the
// shuffle form needs 12 ds_bpermute round trips; every lane gets the total
template <int CTRL>
__device__ __forceinline__ double dvs_dpp_mov(double v) {
    const long long b = __double_as_longlong(v);
    int lo = int(b), hi = int(b >> 32);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
    return __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo));
}
__device__ __forceinline__ double dvs_wave_sum_dpp(double v) {
    v += dvs_dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]
    v += dvs_dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]
    v += dvs_dpp_mov<0x141>(v);  // row_half_mirror
    v += dvs_dpp_mov<0x140>(v);  // row_mirror: every lane holds the sum of its row of 16
    const long long b = __double_as_longlong(v);
    const int lo = int(b), hi = int(b >> 32);
    auto row = [&](int l) {
        return __longlong_as_double((long long)(((unsigned long long)(unsigned)__builtin_amdgcn_readlane(hi, l) << 32) |
                                                (unsigned)__builtin_amdgcn_readlane(lo, l)));
    };
    return (row(0) + row(16)) + (row(32) + row(48));
}
__device__ __forceinline__ double dvs_wave_min(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o, 64));
    return v;
}
// the same minimum by DPP (the minimum does not depend on the order it is taken in: the same bits)
__device__ __forceinline__ double dvs_wave_min_dpp(double v) {
    v = fmin(v, dvs_dpp_mov<0xB1>(v));
    v = fmin(v, dvs_dpp_mov<0x4E>(v));
    v = fmin(v, dvs_dpp_mov<0x141>(v));
    v = fmin(v, dvs_dpp_mov<0x140>(v));
    const long long b = __double_as_longlong(v);
    const int lo = int(b), hi = int(b >> 32);
    auto row = [&](int l) {
        return __longlong_as_double((long long)(((unsigned long long)(unsigned)__builtin_amdgcn_readlane(hi, l) << 32) |
                                                (unsigned)__builtin_amdgcn_readlane(lo, l)));
    };
    return fmin(fmin(row(0), row(16)), fmin(row(32), row(48)));
}
__device__ __forceinline__ unsigned long long dvs_wave_sum_u64(unsigned long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// Sum over the block; every thread gets the result.  scratch: >= 17 doubles of LDS.
// Fixed tree -> the same inputs always give the same bits.
__device__ __forceinline__ double dvs_block_sum(double v, double *scratch) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nwave = (blockDim.x + 63) >> 6;
    v = dvs_wave_sum(v);
    __syncthreads();  // scratch may still be read from a previous call
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    double t = 0.0;
    for (int i = 0; i < nwave; i++) t += scratch[i];
    return t;
}
__device__ __forceinline__ double dvs_block_min(double v, double *scratch) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nwave = (blockDim.x + 63) >> 6;
    v = dvs_wave_min(v);
    __syncthreads();
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    double t = scratch[0];
    for (int i = 1; i < nwave; i++) t = fmin(t, scratch[i]);
    return t;
}
#endif
