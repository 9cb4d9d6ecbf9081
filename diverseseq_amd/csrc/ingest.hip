// Device-side FASTA ingest: raw file bytes in HBM -> one byte per base holding the cogent3
// alphabet index, plus record offsets -- the input format of the histogram kernel -- without a
// host pass over the bases.
//
// Replaces, for the hot path's input, what the reference does on the host per file
// (diverse_seq/io.py:75-104 dvs_load_seqs.main: parse the records, join their sequences with "-",
// diverse_seq/util.py:32-45 str2arr: alphabet.to_indices) before the bytes reach
// ZarrStoreWrapper.write / nmost_divergent.  SURVEY.md 8(f) rank 2.
//
// A byte i of the file is a base of record r when it is not inside a header line (a line that
// starts with '>') and is not white space.  With
//   lastnl(i) = the last '\n' at or before i,  lastgt(i) = the last line-initial '>' at or before i
// a byte is in a header iff lastgt(i) > lastnl(i); r = (number of line-initial '>' up to i) - 1.
// Both are prefix scans (max, max, sum) over the file, and the position of a base in the output
// is a fourth (sum of keep flags).  Each scan is three passes: per-block aggregates (4 KiB of file
// per 256-thread block, 16 bytes per thread), one block scanning the aggregates, and a pass that
// redoes the block locally on top of its carry.  The file is read three times at HBM speed; the
// bases are written once.
//
// A file handed over in HOST memory is streamed: it is uploaded in chunks of 32 MiB on a stream of its
// own and all five passes run chunk by chunk behind their copy -- every scan carries its running state
// from chunk to chunk in a device word -- so the kernels of chunk c overlap the upload of chunk c + 1 and
// the ingest ends a fraction of a millisecond after the last byte has arrived.  (The copies are the
// runtime's own pageable-memory path, which stages through its pinned buffers at ~PCIe rate; a hand-made
// ring of pinned blocks filled by host threads measured slower, 34 against 43 GB/s.)
//
// join_records != 0 restates io.py:100: the records of the file become ONE sequence, a gap symbol
// ('-' = index 4 for DNA, an invalid state, so no k-mer spans two records) between adjacent records.
#include "dvs_internal.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>

namespace {

constexpr int ING_THREADS = 256;
constexpr int ING_CHUNK = 16;                        // bytes per thread
constexpr int ING_TILE = ING_THREADS * ING_CHUNK;    // bytes per block-wide step (4 KiB)
constexpr int ING_TILES = 32;                        // steps per block: the single-block scans of the
                                                     // block aggregates stay short (128 KiB of file per block)
constexpr uint64_t ING_BLOCK = uint64_t(ING_TILE) * ING_TILES;  // bytes per block

struct Agg {       // aggregate of a run of bytes
    long long nl;  // position of the last '\n' (-1: none)
    long long gt;  // position of the last line-initial '>' (-1: none)
    long long ng;  // number of line-initial '>'
};
__device__ __forceinline__ Agg agg_join(const Agg a, const Agg b) {  // a then b
    return Agg{b.nl > a.nl ? b.nl : a.nl, b.gt > a.gt ? b.gt : a.gt, a.ng + b.ng};
}

__device__ __forceinline__ bool is_space(uint8_t c) { return c == '\n' || c == '\r' || c == ' ' || c == '\t' || c == 0; }

// the 16 bytes of this thread (0 beyond the end) and the byte in front of them ('\n' at the file start)
__device__ __forceinline__ void load_chunk(const uint8_t *raw, uint64_t n, uint64_t base, uint8_t (&c)[ING_CHUNK],
                                           uint8_t &prev) {
    if (base + ING_CHUNK <= n) {
        const uint4 v = *reinterpret_cast<const uint4 *>(raw + base);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < ING_CHUNK; i++) c[i] = uint8_t(w[i >> 2] >> (8 * (i & 3)));
    } else {
#pragma unroll
        for (int i = 0; i < ING_CHUNK; i++) c[i] = base + i < n ? raw[base + i] : 0;
    }
    prev = base == 0 ? uint8_t('\n') : (base - 1 < n ? raw[base - 1] : 0);
}

__device__ __forceinline__ Agg chunk_agg(const uint8_t (&c)[ING_CHUNK], uint8_t prev, uint64_t base, uint64_t n) {
    Agg a{-1, -1, 0};
#pragma unroll
    for (int i = 0; i < ING_CHUNK; i++) {
        if (base + i >= n) break;
        const uint8_t p = i ? c[i - 1] : prev;
        if (c[i] == '\n') a.nl = (long long)(base + i);
        if (c[i] == '>' && p == '\n') {
            a.gt = (long long)(base + i);
            a.ng++;
        }
    }
    return a;
}

// scan of one Agg per thread over the block: a shuffle scan inside each wave, the four wave totals
// through LDS; returns the EXCLUSIVE prefix of this thread and the block total in `total`
__device__ __forceinline__ Agg block_scan_agg(Agg mine, Agg *lds, Agg &total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    Agg inc = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const Agg up{__shfl_up(inc.nl, o, 64), __shfl_up(inc.gt, o, 64), __shfl_up(inc.ng, o, 64)};
        if (lane >= o) inc = agg_join(up, inc);
    }
    if (lane == 63) lds[wave] = inc;
    __syncthreads();
    Agg carry{-1, -1, 0}, all{-1, -1, 0};
#pragma unroll
    for (int w = 0; w < ING_THREADS / 64; w++) {
        const Agg t = lds[w];
        if (w < wave) carry = agg_join(carry, t);
        all = agg_join(all, t);
    }
    total = all;
    Agg ex{__shfl_up(inc.nl, 1, 64), __shfl_up(inc.gt, 1, 64), __shfl_up(inc.ng, 1, 64)};
    if (lane == 0) ex = Agg{-1, -1, 0};
    __syncthreads();  // lds is reused by the next scan
    return agg_join(carry, ex);
}

__device__ __forceinline__ unsigned long long block_scan_u64(unsigned long long mine, unsigned long long *lds,
                                                             unsigned long long &total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long inc = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned long long up = __shfl_up(inc, o, 64);
        if (lane >= o) inc += up;
    }
    if (lane == 63) lds[wave] = inc;
    __syncthreads();
    unsigned long long carry = 0, all = 0;
#pragma unroll
    for (int w = 0; w < ING_THREADS / 64; w++) {
        const unsigned long long t = lds[w];
        if (w < wave) carry += t;
        all += t;
    }
    total = all;
    unsigned long long ex = __shfl_up(inc, 1, 64);
    if (lane == 0) ex = 0;
    __syncthreads();
    return carry + ex;
}

// pass 1: aggregate of every block
// (block0: the first block of the chunk this launch covers; n: the whole file)
__global__ __launch_bounds__(ING_THREADS) void ing_agg_kernel(const uint8_t *__restrict__ raw, uint64_t n,
                                                              Agg *__restrict__ blocks, uint64_t block0) {
    __shared__ Agg lds[ING_THREADS];
    Agg run{-1, -1, 0};
    const uint64_t blk = block0 + blockIdx.x;
    for (int tile = 0; tile < ING_TILES; tile++) {
        const uint64_t base = blk * ING_BLOCK + uint64_t(tile) * ING_TILE + uint64_t(threadIdx.x) * ING_CHUNK;
        if (blk * ING_BLOCK + uint64_t(tile) * ING_TILE >= n) break;  // (block-uniform)
        uint8_t c[ING_CHUNK], prev;
        load_chunk(raw, n, base, c, prev);
        Agg total;
        (void)block_scan_agg(chunk_agg(c, prev, base, n), lds, total);
        run = agg_join(run, total);
    }
    if (threadIdx.x == 0) blocks[blk] = run;
}

// pass 2: one block turns the aggregates of nb blocks into exclusive prefixes (carry of every block),
// starting from the running state *running (what precedes them in the file) and leaving the state behind
// them there
__global__ __launch_bounds__(ING_THREADS) void ing_scan_agg_kernel(Agg *__restrict__ blocks, uint64_t nb,
                                                                   Agg *__restrict__ running) {
    __shared__ Agg lds[ING_THREADS];
    Agg carry = *running;
    for (uint64_t b0 = 0; b0 < nb; b0 += ING_THREADS) {
        const uint64_t b = b0 + threadIdx.x;
        const Agg mine = b < nb ? blocks[b] : Agg{-1, -1, 0};
        Agg total;
        const Agg ex = block_scan_agg(mine, lds, total);
        if (b < nb) blocks[b] = agg_join(carry, ex);
        carry = agg_join(carry, total);
    }
    __syncthreads();  // (every thread has read *running)
    if (threadIdx.x == 0) *running = carry;
}

__global__ __launch_bounds__(ING_THREADS) void ing_scan_u64_kernel(unsigned long long *__restrict__ blocks, uint64_t nb,
                                                                   unsigned long long *__restrict__ running) {
    __shared__ unsigned long long lds[ING_THREADS];
    unsigned long long carry = *running;
    for (uint64_t b0 = 0; b0 < nb; b0 += ING_THREADS) {
        const uint64_t b = b0 + threadIdx.x;
        const unsigned long long mine = b < nb ? blocks[b] : 0ull;
        unsigned long long total;
        const unsigned long long ex = block_scan_u64(mine, lds, total);
        if (b < nb) blocks[b] = carry + ex;
        carry += total;
    }
    __syncthreads();
    if (threadIdx.x == 0) *running = carry;
}

// what a byte becomes: 0 dropped, 1 a base, 2 the gap symbol that joins two records
__device__ __forceinline__ int classify(uint8_t ch, uint8_t prevch, uint64_t pos, Agg &run, int join) {
    if (ch == '\n') run.nl = (long long)pos;
    const bool header_start = ch == '>' && prevch == '\n';
    if (header_start) {
        run.gt = (long long)pos;
        run.ng++;
        return (join && run.ng >= 2) ? 2 : 0;
    }
    if (run.gt > run.nl || run.ng == 0 || is_space(ch)) return 0;
    return 1;
}

// passes 3 and 5: WRITE == false counts what each block keeps; WRITE == true writes the codes,
// the start of every record in the output and the file position of its header
template <bool WRITE>
__global__ __launch_bounds__(ING_THREADS) void ing_emit_kernel(const uint8_t *__restrict__ raw, uint64_t n,
                                                               const Agg *__restrict__ carry_agg,
                                                               unsigned long long *__restrict__ keep_blocks,
                                                               const uint8_t *__restrict__ lut, int join,
                                                               uint8_t gap_code, uint8_t *__restrict__ codes,
                                                               unsigned long long *__restrict__ rec_start,
                                                               unsigned long long *__restrict__ hdr_pos,
                                                               uint64_t block0, unsigned long long rec_cap,
                                                               unsigned int *__restrict__ rec_overflow) {
    __shared__ Agg lds[ING_THREADS];
    __shared__ unsigned long long ldk[ING_THREADS];
    __shared__ uint8_t s_lut[256];
    if (WRITE) s_lut[threadIdx.x] = lut[threadIdx.x];
    const uint64_t blk = block0 + blockIdx.x;
    Agg run_block = carry_agg[blk];  // state in front of the tile being processed
    unsigned long long kept_block = WRITE ? keep_blocks[blk] : 0ull;
    for (int tile = 0; tile < ING_TILES; tile++) {
        const uint64_t tile_base = blk * ING_BLOCK + uint64_t(tile) * ING_TILE;
        if (tile_base >= n) break;  // (block-uniform)
        const uint64_t base = tile_base + uint64_t(threadIdx.x) * ING_CHUNK;
        uint8_t c[ING_CHUNK], prev;
        load_chunk(raw, n, base, c, prev);
        Agg total;
        const Agg ex = block_scan_agg(chunk_agg(c, prev, base, n), lds, total);
        const Agg start = agg_join(run_block, ex);  // state in front of this thread's bytes
        Agg run = start;
        unsigned long long kept = 0;
        uint32_t cls = 0;  // 2 bits per byte
#pragma unroll
        for (int i = 0; i < ING_CHUNK; i++) {
            if (base + i >= n) break;
            const int k = classify(c[i], i ? c[i - 1] : prev, base + i, run, join);
            cls |= uint32_t(k) << (2 * i);
            kept += k != 0;
        }
        unsigned long long btotal;
        const unsigned long long kex = block_scan_u64(kept, ldk, btotal);
        if (WRITE) {
            unsigned long long out = kept_block + kex;
            long long ng = start.ng;
#pragma unroll
            for (int i = 0; i < ING_CHUNK; i++) {
                if (base + i >= n) break;
                const uint8_t p = i ? c[i - 1] : prev;
                const uint32_t k = (cls >> (2 * i)) & 3u;
                if (c[i] == '>' && p == '\n') {  // record ng starts here; its bases follow the joining gap, if any
                    if ((unsigned long long)ng < rec_cap) {
                        rec_start[ng] = out + (k == 2 ? 1 : 0);
                        hdr_pos[ng] = base + i;
                    } else {
                        atomicOr(rec_overflow, 1u);  // more records than the arrays hold: the caller sizes them again
                    }
                    ng++;
                }
                if (k == 1) codes[out++] = s_lut[c[i]];
                else if (k == 2) codes[out++] = gap_code;
            }
        }
        run_block = agg_join(run_block, total);
        kept_block += btotal;
    }
    if (!WRITE && threadIdx.x == 0) keep_blocks[blk] = kept_block;
}

}  // namespace

struct dvs_seqbatch {
    dvs_ctx *ctx = nullptr;
    uint8_t *d_codes = nullptr;  // total + 16 bytes (the histogram kernel reads 16-byte chunks)
    uint64_t total = 0;
    uint32_t nseq = 0;
    std::vector<uint64_t> offsets;     // nseq + 1
    std::vector<uint64_t> header_pos;  // file offset of the '>' of every record of the file
    dvs_packed *packed = nullptr;      // after dvs_seqbatch_pack: the bases at 3 bits each (d_codes released)
    // the alphabet the batch was encoded with puts the four bases first (the library's own DNA / RNA table): only
    // then is "symbol >= 4" the same as "not a base" and the batch may be re-stated in the packed form
    bool four_state_alphabet = false;
    // builds over d_codes that were not waited for (dvs_matrix_build_from_seqbatch: the context's stream or, for
    // a split build, its stream_rest): dvs_seqbatch_destroy drains those streams before the block goes back
    mutable bool async_readers = false;
};

extern "C" void dvs_seqbatch_destroy(dvs_seqbatch *b) {
    if (!b) return;
    if (b->packed) dvs_packed_destroy(b->packed);
    if (b->d_codes && b->async_readers && b->ctx) {
        if (b->ctx->stream_rest) (void)hipStreamSynchronize(b->ctx->stream_rest);
        (void)hipStreamSynchronize(b->ctx->stream);
    }
    if (b->d_codes) dvs_dev_free(b->ctx, b->d_codes);
    dvs_ctx_release(b->ctx);
    delete b;
}

// cogent3 moltype "dna" / "rna": most_degen_alphabet() = "TCAG-NRYWSKMBDHV?" (U for T in RNA);
// lower case is folded; any other byte maps to 255 (cogent3 would refuse it)
extern "C" void dvs_default_alphabet_lut(int rna, uint8_t lut[256]) {
    memset(lut, 255, 256);
    const char *order = rna ? "UCAG-NRYWSKMBDHV?" : "TCAG-NRYWSKMBDHV?";
    for (int i = 0; order[i]; i++) {
        const unsigned char ch = (unsigned char)order[i];
        lut[ch] = uint8_t(i);
        if (ch >= 'A' && ch <= 'Z') lut[ch + 32] = uint8_t(i);
    }
    lut[(unsigned char)(rna ? 'T' : 'U')] = 0;  // T and U are the same base, index 0
    lut[(unsigned char)(rna ? 't' : 'u')] = 0;
}

extern "C" int dvs_seqbatch_from_fasta(dvs_ctx *ctx, const uint8_t *raw, int raw_on_device, uint64_t nbytes,
                                       const uint8_t *lut256, int join_records, dvs_seqbatch **out) {
    if (!ctx || !out || (!raw && nbytes)) return dvs_set_error(ctx, DVS_ERR_VALUE, "null argument");
    *out = nullptr;
    DVS_HIP(ctx, hipSetDevice(ctx->device));
    uint8_t lut[256];
    if (lut256) memcpy(lut, lut256, 256);
    else dvs_default_alphabet_lut(0, lut);
    const uint8_t gap = lut[(unsigned char)'-'];
    dvs_seqbatch *b = new dvs_seqbatch;
    b->ctx = ctx;
    {
        uint8_t dna[256], rna[256];
        dvs_default_alphabet_lut(0, dna);
        dvs_default_alphabet_lut(1, rna);
        b->four_state_alphabet = memcmp(lut, dna, 256) == 0 || memcmp(lut, rna, 256) == 0;
    }
    dvs_ctx_retain(ctx);
    const uint64_t nb = (nbytes + ING_BLOCK - 1) / ING_BLOCK;
    uint8_t *d_raw = nullptr, *d_lut = nullptr;
    Agg *d_agg = nullptr;
    unsigned long long *d_keep = nullptr, *d_rec = nullptr, *d_hdr = nullptr;
    struct Running {  // the scans' running state behind the chunks processed so far, and the overflow flag
        Agg agg;
        unsigned long long kept;
        unsigned int overflow, pad;
    } *d_run = nullptr;
    bool own_raw = false;
    hipStream_t scopy = nullptr;
    std::vector<hipEvent_t> ev_copied;
    int rc = DVS_OK;
    auto cleanup = [&]() {
        if (own_raw && d_raw) dvs_dev_free(ctx, d_raw);
        if (d_lut) dvs_dev_free(ctx, d_lut);
        if (d_agg) dvs_dev_free(ctx, d_agg);
        if (d_keep) dvs_dev_free(ctx, d_keep);
        if (d_rec) dvs_dev_free(ctx, d_rec);
        if (d_hdr) dvs_dev_free(ctx, d_hdr);
        if (d_run) dvs_dev_free(ctx, d_run);
        for (hipEvent_t e : ev_copied) dvs_event_put(ctx, e);
        if (scopy) (void)hipStreamDestroy(scopy);
    };
#define ING_TRY(expr)                                              \
    do {                                                           \
        hipError_t e__ = (expr);                                   \
        if (e__ != hipSuccess) {                                   \
            rc = dvs_hip_fail(ctx, e__, #expr);                    \
            (void)hipDeviceSynchronize();                          \
            cleanup();                                             \
            dvs_seqbatch_destroy(b);                               \
            return rc;                                             \
        }                                                          \
    } while (0)
#define ING_RC(expr)                      \
    do {                                  \
        rc = (expr);                      \
        if (rc) {                         \
            (void)hipDeviceSynchronize(); \
            cleanup();                    \
            dvs_seqbatch_destroy(b);      \
            return rc;                    \
        }                                 \
    } while (0)
    // chunks: the whole file at once when it is already in HBM (or small), else 32 MiB pieces
    uint64_t chunk_blocks = nb ? nb : 1;
    const uint64_t CH_BLOCKS = (32ull << 20) / ING_BLOCK;
    const bool streamed = !raw_on_device && nb >= 3 * CH_BLOCKS && !ctx->knobs.ingest_no_stream;
    constexpr int NSLOT = 4;
    if (raw_on_device) {
        d_raw = const_cast<uint8_t *>(raw);
    } else {
        own_raw = true;
        ING_RC(dvs_dev_alloc(ctx, (void **)&d_raw, nbytes + 16, "raw FASTA bytes"));
        if (streamed) {
            chunk_blocks = CH_BLOCKS;
            ING_TRY(hipStreamCreateWithFlags(&scopy, hipStreamNonBlocking));
            for (int i = 0; i < NSLOT; i++) ev_copied.push_back(dvs_event_get(ctx));
        } else {
            ING_TRY(hipMemcpyAsync(d_raw, raw, nbytes, hipMemcpyHostToDevice, ctx->stream));
        }
    }
    ING_RC(dvs_dev_alloc(ctx, (void **)&d_lut, 256, "alphabet table"));
    ING_TRY(hipMemcpyAsync(d_lut, lut, 256, hipMemcpyHostToDevice, ctx->stream));
    ING_RC(dvs_dev_alloc(ctx, (void **)&d_agg, (nb + 1) * sizeof(Agg), "ingest block aggregates"));
    ING_RC(dvs_dev_alloc(ctx, (void **)&d_keep, (nb + 1) * sizeof(unsigned long long), "ingest block counts"));
    ING_RC(dvs_dev_alloc(ctx, (void **)&d_run, sizeof(Running), "ingest running state"));
    const Running run0{Agg{-1, -1, 0}, 0ull, 0u, 0u};
    ING_TRY(hipMemcpyAsync(d_run, &run0, sizeof run0, hipMemcpyHostToDevice, ctx->stream));  // (run0 is const: staged)
    // The number of records is known only at the end; the arrays are sized for records of >= 256 bytes on
    // average (+ 1 M) and the kernel raises a flag instead of writing beyond them -- then the last pass is
    // repeated over the whole file with arrays of the right size.
    unsigned long long rec_cap = nbytes / 256 + (1ull << 20);
    ING_RC(dvs_dev_alloc(ctx, (void **)&d_rec, rec_cap * 8, "record starts"));
    ING_RC(dvs_dev_alloc(ctx, (void **)&d_hdr, rec_cap * 8, "header positions"));
    // the encoded bases: never more than the file's bytes
    ING_RC(dvs_dev_alloc(ctx, (void **)&b->d_codes, nbytes + 16, "encoded sequences"));
    auto passes = [&](uint64_t block0, uint64_t nblk, bool all_five) {
        const dim3 grid{uint32_t(nblk)}, blk{ING_THREADS};
        hipLaunchKernelGGL(ing_agg_kernel, grid, blk, 0, ctx->stream, d_raw, nbytes, d_agg, block0);
        hipLaunchKernelGGL(ing_scan_agg_kernel, dim3(1), blk, 0, ctx->stream, d_agg + block0, nblk, &d_run->agg);
        hipLaunchKernelGGL((ing_emit_kernel<false>), grid, blk, 0, ctx->stream, d_raw, nbytes, d_agg, d_keep, d_lut,
                           join_records, gap, (uint8_t *)nullptr, (unsigned long long *)nullptr,
                           (unsigned long long *)nullptr, block0, 0ull, (unsigned int *)nullptr);
        hipLaunchKernelGGL(ing_scan_u64_kernel, dim3(1), blk, 0, ctx->stream, d_keep + block0, nblk, &d_run->kept);
        if (all_five)
            hipLaunchKernelGGL((ing_emit_kernel<true>), grid, blk, 0, ctx->stream, d_raw, nbytes, d_agg, d_keep, d_lut,
                               join_records, gap, b->d_codes, d_rec, d_hdr, block0, rec_cap, &d_run->overflow);
    };
    if (streamed) {
        const uint64_t cbytes = chunk_blocks * ING_BLOCK;
        uint64_t c = 0;
        for (uint64_t block0 = 0; block0 < nb; block0 += chunk_blocks, c++) {
            const int slot = int(c % NSLOT);
            const uint64_t off = block0 * ING_BLOCK, len = std::min<uint64_t>(cbytes, nbytes - off);
            ING_TRY(hipMemcpyAsync(d_raw + off, raw + off, len, hipMemcpyHostToDevice, scopy));
            ING_TRY(hipEventRecord(ev_copied[slot], scopy));
            ING_TRY(hipStreamWaitEvent(ctx->stream, ev_copied[slot], 0));
            passes(block0, std::min<uint64_t>(chunk_blocks, nb - block0), true);
            ING_TRY(hipGetLastError());
        }
    } else if (nb) {
        passes(0, nb, true);
        ING_TRY(hipGetLastError());
    }
    Running fin = run0;
    if (nb) {
        ING_TRY(hipMemcpyAsync(&fin, d_run, sizeof fin, hipMemcpyDeviceToHost, ctx->stream));
        ING_TRY(hipStreamSynchronize(ctx->stream));
    }
    const uint64_t nrec = uint64_t(fin.agg.ng), total = fin.kept;
    if (nrec > 0xFFFFFFFFull) {
        cleanup();
        dvs_seqbatch_destroy(b);
        return dvs_set_error(ctx, DVS_ERR_UNSUPPORTED, "more than 2^32 - 1 records in one file");
    }
    if (fin.overflow) {  // more (shorter) records than the arrays were sized for: the last pass again, over everything
        dvs_dev_free(ctx, d_rec);
        dvs_dev_free(ctx, d_hdr);
        d_rec = d_hdr = nullptr;
        rec_cap = nrec;
        ING_RC(dvs_dev_alloc(ctx, (void **)&d_rec, rec_cap * 8, "record starts"));
        ING_RC(dvs_dev_alloc(ctx, (void **)&d_hdr, rec_cap * 8, "header positions"));
        hipLaunchKernelGGL((ing_emit_kernel<true>), dim3(uint32_t(nb)), dim3(ING_THREADS), 0, ctx->stream, d_raw, nbytes,
                           d_agg, d_keep, d_lut, join_records, gap, b->d_codes, d_rec, d_hdr, 0ull, rec_cap,
                           &d_run->overflow);
        ING_TRY(hipGetLastError());
    }
    b->total = total;
    ING_TRY(hipMemsetAsync(b->d_codes + total, 0xFF, 16, ctx->stream));  // invalid filler behind the last base
    std::vector<uint64_t> starts(nrec);
    b->header_pos.resize(nrec);
    if (nrec) {
        ING_TRY(hipMemcpyAsync(starts.data(), d_rec, nrec * 8, hipMemcpyDeviceToHost, ctx->stream));
        ING_TRY(hipMemcpyAsync(b->header_pos.data(), d_hdr, nrec * 8, hipMemcpyDeviceToHost, ctx->stream));
    }
    ING_TRY(hipStreamSynchronize(ctx->stream));
#undef ING_TRY
#undef ING_RC
    if (join_records) {
        b->nseq = nrec ? 1u : 0u;
        if (nrec) b->offsets = {0, total};
        else b->offsets = {0};
    } else {
        b->nseq = uint32_t(nrec);
        b->offsets.assign(starts.begin(), starts.end());
        b->offsets.push_back(total);
    }
    cleanup();
    *out = b;
    return DVS_OK;
}

extern "C" int dvs_seqbatch_info(const dvs_seqbatch *b, uint32_t *nseq, uint64_t *total_bases, uint32_t *nrecords) {
    if (!b) return DVS_ERR_VALUE;
    if (nseq) *nseq = b->nseq;
    if (total_bases) *total_bases = b->total;
    if (nrecords) *nrecords = uint32_t(b->header_pos.size());
    return DVS_OK;
}
extern "C" int dvs_seqbatch_offsets(const dvs_seqbatch *b, uint64_t *offsets_out) {
    if (!b || !offsets_out) return DVS_ERR_VALUE;
    memcpy(offsets_out, b->offsets.data(), b->offsets.size() * 8);
    return DVS_OK;
}
extern "C" int dvs_seqbatch_header_positions(const dvs_seqbatch *b, uint64_t *pos_out) {
    if (!b || !pos_out) return DVS_ERR_VALUE;
    memcpy(pos_out, b->header_pos.data(), b->header_pos.size() * 8);
    return DVS_OK;
}
extern "C" const void *dvs_seqbatch_dev_codes(const dvs_seqbatch *b) { return b ? b->d_codes : nullptr; }
extern "C" int dvs_seqbatch_get_codes(dvs_ctx *ctx, const dvs_seqbatch *b, uint8_t *codes_out) {
    if (!ctx || !b || (!codes_out && b->total)) return dvs_set_error(ctx, DVS_ERR_VALUE, "null argument");
    if (b->packed) {  // the planes, expanded on the host: code & 3, or 255 where the mask says invalid
        std::vector<uint32_t> codes(b->packed->nwords);
        std::vector<uint16_t> mask(b->packed->nwords);
        const int rc = dvs_packed_get(ctx, b->packed, codes.data(), mask.data());
        if (rc) return rc;
        for (uint64_t i = 0; i < b->total; i++) {
            const uint32_t sh = uint32_t(i & 15);
            codes_out[i] = ((mask[i >> 4] >> (15 - sh)) & 1u) ? 255 : uint8_t((codes[i >> 4] >> (30 - 2 * sh)) & 3u);
        }
        return DVS_OK;
    }
    DVS_HIP(ctx, hipMemcpyAsync(codes_out, b->d_codes, b->total, hipMemcpyDeviceToHost, ctx->stream));
    DVS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return DVS_OK;
}
extern "C" int dvs_seqbatch_pack(dvs_ctx *ctx, dvs_seqbatch *b) {
    if (!ctx || !b) return dvs_set_error(ctx, DVS_ERR_VALUE, "null argument");
    if (b->packed) return DVS_OK;
    if (!b->four_state_alphabet)
        return dvs_set_error(ctx, DVS_ERR_VALUE,
                             "only a batch encoded with the DNA / RNA alphabet can be packed (2 bits per base): "
                             "another alphabet's symbols >= 4 would all become \"invalid\"");
    dvs_packed *p = nullptr;
    int rc = dvs_packed_alloc(ctx, b->total, &p);
    if (!rc) rc = dvs_packed_fill_from_device(ctx, p, b->d_codes);
    if (rc) {
        if (p) dvs_packed_destroy(p);
        return rc;
    }
    b->packed = p;
    dvs_dev_free(b->ctx, b->d_codes);  // (back to the cache: stream order protects it until the kernel has run)
    b->d_codes = nullptr;
    return DVS_OK;
}
extern "C" const dvs_packed *dvs_seqbatch_packed(const dvs_seqbatch *b) { return b ? b->packed : nullptr; }
extern "C" int dvs_matrix_build_from_seqbatch(dvs_ctx *ctx, const dvs_seqbatch *b, uint32_t k, uint32_t num_states,
                                              dvs_matrix **out) {
    if (!ctx || !b || !out) return dvs_set_error(ctx, DVS_ERR_VALUE, "null argument");
    if (b->packed) {
        if (num_states != 4)
            return dvs_set_error(ctx, DVS_ERR_VALUE, "a packed batch holds four-state sequences, not %u states", num_states);
        return dvs_matrix_build_packed(ctx, b->packed, b->offsets.data(), b->nseq, k, out);
    }
    b->async_readers = true;  // (dvs_seqbatch_destroy waits for the kernels enqueued here)
    return dvs_matrix_build(ctx, b->d_codes, 1, b->offsets.data(), b->nseq, k, num_states, out);
}
extern "C" int dvs_sketches_build_from_seqbatch(dvs_ctx *ctx, const dvs_seqbatch *b, uint32_t k, uint32_t sketch_size,
                                                uint32_t num_states, int mash_canonical, dvs_sketches **out) {
    if (!ctx || !b || !out) return dvs_set_error(ctx, DVS_ERR_VALUE, "null argument");
    if (b->packed) {
        if (num_states != 4)
            return dvs_set_error(ctx, DVS_ERR_VALUE, "a packed batch holds four-state sequences, not %u states", num_states);
        return dvs_sketches_build_packed(ctx, b->packed, b->offsets.data(), b->nseq, k, sketch_size, mash_canonical, out);
    }
    b->async_readers = true;
    return dvs_sketches_build(ctx, b->d_codes, 1, b->offsets.data(), b->nseq, k, sketch_size, num_states, mash_canonical, out);
}
