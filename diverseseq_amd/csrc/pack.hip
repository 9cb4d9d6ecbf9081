// Four-state sequences in HBM at 3 bits per base: 2-bit codes + a 1-bit "invalid" mask (dvs_packed,
// dvs_internal.h).  The reference's convention is one byte per base (src/record.rs:205-209,
// diverse_seq/util.py:32-45); a symbol carries two bits of it and "is this a gap / an ambiguity code".
// The histogram and sketch kernels read the packed words AS THEY ARE (kmer_hist.hip, mash.hip) -- the
// word layout is the one their index arithmetic uses internally -- so
//   * sequences handed over in HOST memory (dvs_matrix_build from a host pointer: what
//     diverse_seq._dvs.nmost_divergent / max_divergent do with a store's sequences, src/lib.rs:59-73)
//     are packed by host threads chunk by chunk into a pinned staging block the context keeps, every
//     chunk is sent as soon as it is packed (copy i + 1 runs beside the packing of chunk i + 2), and
//     that is all: 3/8 of the bytes cross PCIe and nothing is expanded again;
//   * sequences already in HBM one byte per base (a torch tensor, the ingest's output) can be packed
//     once by pack_kernel (dvs_pack_sequences, dvs_seqbatch_pack) and then sit there at 3/8 of the
//     bytes: a genome collection of 31.5 Gbases is 11.8 GB instead of 31.5.
#include "dvs_internal.h"

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <thread>
#include <vector>

extern "C" void dvs_pack_bases(const uint8_t *src, size_t n, uint32_t *codes, uint16_t *mask);  // pack_host.cpp

namespace {

constexpr size_t PACK_CHUNK = size_t(1) << 20;  // bases per host-packed chunk (a multiple of 64)
constexpr size_t PACK_CODE_BYTES = PACK_CHUNK / 4, PACK_MASK_BYTES = PACK_CHUNK / 8;

// bytes in HBM -> the two planes; 16 bases per thread: one 16-byte load, a 4-byte and a 2-byte store
__global__ __launch_bounds__(256) void pack_kernel(const uint8_t *__restrict__ seqs, uint64_t nbytes,
                                                   uint32_t *__restrict__ codes, uint16_t *__restrict__ mask,
                                                   uint64_t nwords) {
    const uint64_t w = uint64_t(blockIdx.x) * 256u + threadIdx.x;
    if (w >= nwords) return;
    const uint64_t a = w * 16;
    uint4 v;
    if (a + 16 <= nbytes) {
        v = *reinterpret_cast<const uint4 *>(seqs + a);
    } else {  // the ragged tail: positions behind the end are invalid
        uint32_t q[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
        for (int i = 0; i < 16; i++)
            if (a + i < nbytes) {
                q[i >> 2] &= ~(0xFFu << (8 * (i & 3)));
                q[i >> 2] |= uint32_t(seqs[a + i]) << (8 * (i & 3));
            }
        v = make_uint4(q[0], q[1], q[2], q[3]);
    }
    codes[w] = dvs_pack16(v);
    mask[w] = uint16_t(dvs_inv16(v));
}

}  // namespace

// host cores this process may really use (the affinity mask capped by the cgroup's CPU quota), at most 16
unsigned dvs_host_threads() {
    static const unsigned n = [] {
        unsigned hw = std::max(1u, std::thread::hardware_concurrency());
        if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
            char q[64] = {0};
            long period = 0;
            if (fscanf(f, "%63s %ld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0)
                hw = std::min<unsigned>(hw, unsigned(std::max(1L, atol(q) / period)));
            fclose(f);
        }
        return std::min(hw, 16u);
    }();
    return n;
}

bool dvs_packed_upload_wanted(const dvs_ctx *ctx, uint32_t num_states, uint64_t nbytes) {
    return num_states == 4 && nbytes >= (uint64_t(32) << 20) && !ctx->knobs.no_packed_upload;
}

int dvs_packed_alloc(dvs_ctx *ctx, uint64_t nbases, dvs_packed **out) {
    *out = nullptr;
    dvs_packed *p = new dvs_packed();
    p->ctx = ctx;
    p->nbases = nbases;
    p->nwords = (nbases + 15) / 16;
    dvs_ctx_retain(ctx);
    // (+ 4 words: the kernels' 16-byte and look-ahead reads near the end stay inside the planes)
    int rc = dvs_dev_alloc(ctx, (void **)&p->d_codes, (p->nwords + 4) * 4, "packed sequences (codes)");
    if (!rc) rc = dvs_dev_alloc(ctx, (void **)&p->d_mask, (p->nwords + 8) * 2, "packed sequences (mask)");
    if (rc) {
        dvs_packed_destroy(p);
        return rc;
    }
    *out = p;
    return DVS_OK;
}

extern "C" void dvs_packed_destroy(dvs_packed *p) {
    if (!p) return;
    if (p->async_readers && p->ctx) {  // (a build that was not waited for may still be reading the planes)
        if (p->ctx->stream_rest) (void)hipStreamSynchronize(p->ctx->stream_rest);
        (void)hipStreamSynchronize(p->ctx->stream);
    }
    dvs_dev_free(p->ctx, p->d_codes);
    dvs_dev_free(p->ctx, p->d_mask);
    dvs_ctx_release(p->ctx);
    delete p;
}

int dvs_packed_fill_from_device(dvs_ctx *ctx, dvs_packed *p, const uint8_t *d_seqs) {
    if (!p->nwords) return DVS_OK;
    hipLaunchKernelGGL(pack_kernel, dim3(uint32_t((p->nwords + 255) / 256)), dim3(256), 0, ctx->stream, d_seqs, p->nbases,
                       p->d_codes, p->d_mask, p->nwords);
    DVS_HIP(ctx, hipGetLastError());
    return DVS_OK;
}

// seqs[0, nbases) (host, one byte per base) -> the planes, enqueued on the context's stream; returns when the
// last chunk's copy has been ENQUEUED (the staging block is protected by ctx->pack_ev).
int dvs_packed_fill_from_host(dvs_ctx *ctx, dvs_packed *p, const uint8_t *seqs) {
    const uint64_t nbytes = p->nbases;
    if (!nbytes) return DVS_OK;
    const size_t nchunks = size_t((nbytes + PACK_CHUNK - 1) / PACK_CHUNK);
    const size_t stage_bytes = nchunks * (PACK_CODE_BYTES + PACK_MASK_BYTES);
    // the staging block: nobody may still be reading it (the previous call's last copy)
    if (ctx->pack_ev) (void)hipEventSynchronize(ctx->pack_ev);
    if (ctx->h_pack_cap < stage_bytes) {
        if (ctx->h_pack) (void)hipHostFree(ctx->h_pack);
        ctx->h_pack = nullptr;
        ctx->h_pack_cap = 0;
        const size_t cap = stage_bytes + stage_bytes / 8;
        const hipError_t he = hipHostMalloc(&ctx->h_pack, cap, hipHostMallocDefault);
        if (he != hipSuccess) {
            ctx->h_pack = nullptr;
            return dvs_hip_fail(ctx, he, "pinned staging block of the packed upload");
        }
        ctx->h_pack_cap = cap;
    }
    // staging layout: every chunk's codes, then every chunk's masks (the planes as they lie on the device)
    uint8_t *stage_codes = static_cast<uint8_t *>(ctx->h_pack);
    uint8_t *stage_mask = stage_codes + nchunks * PACK_CODE_BYTES;
    std::unique_ptr<std::atomic<int>[]> done(new std::atomic<int>[nchunks]);
    for (size_t c = 0; c < nchunks; c++) done[c].store(0, std::memory_order_relaxed);
    std::atomic<size_t> next{0};
    auto pack_chunk = [&](size_t c) {
        const uint64_t a = uint64_t(c) * PACK_CHUNK;
        const size_t n = size_t(std::min<uint64_t>(PACK_CHUNK, nbytes - a));
        dvs_pack_bases(seqs + a, n, reinterpret_cast<uint32_t *>(stage_codes + c * PACK_CODE_BYTES),
                       reinterpret_cast<uint16_t *>(stage_mask + c * PACK_MASK_BYTES));
        done[c].store(1, std::memory_order_release);
    };
    auto work = [&]() {
        for (;;) {
            const size_t c = next.fetch_add(1, std::memory_order_relaxed);
            if (c >= nchunks) return;
            pack_chunk(c);
        }
    };
    const unsigned nthr = unsigned(std::min<size_t>(dvs_host_threads(), nchunks));
    std::vector<std::thread> pool;
    for (unsigned t = 0; t + 1 < nthr; t++) pool.emplace_back(work);  // (this thread sends; with one core it packs too)
    hipError_t e = hipSuccess;
    // Copies of 1, 2, 4, 8, then 16 chunks (the staging block is laid out like the planes, so chunks that follow one
    // another go in one pair of copies): the first copy leaves as soon as one chunk is packed, the later ones are
    // long enough for the link's full rate (MI355X box: 33 GB/s in pieces of 0.8 MB, 52 GB/s in pieces of 6 MB)
    size_t want = 1;
    for (size_t c = 0; c < nchunks && e == hipSuccess;) {
        const size_t m = std::min(want, nchunks - c);
        for (size_t i = 0; i < m;) {
            if (done[c + i].load(std::memory_order_acquire)) {
                i++;
                continue;
            }
            if (nthr <= 1 || next.load(std::memory_order_relaxed) < nchunks) {
                // nothing to send yet: pack a chunk here instead of spinning
                const size_t mine = next.fetch_add(1, std::memory_order_relaxed);
                if (mine < nchunks) {
                    pack_chunk(mine);
                    continue;
                }
            }
            std::this_thread::yield();
        }
        const uint64_t a = uint64_t(c) * PACK_CHUNK;
        const size_t words = size_t((std::min<uint64_t>(uint64_t(m) * PACK_CHUNK, nbytes - a) + 15) / 16);
        e = hipMemcpyAsync(reinterpret_cast<uint8_t *>(p->d_codes) + c * PACK_CODE_BYTES, stage_codes + c * PACK_CODE_BYTES,
                           words * 4, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess)
            e = hipMemcpyAsync(reinterpret_cast<uint8_t *>(p->d_mask) + c * PACK_MASK_BYTES, stage_mask + c * PACK_MASK_BYTES,
                               words * 2, hipMemcpyHostToDevice, ctx->stream);
        c += m;
        want = std::min<size_t>(want * 2, 16);
    }
    for (std::thread &t : pool) t.join();
    if (e == hipSuccess) {
        if (!ctx->pack_ev) (void)hipEventCreateWithFlags(&ctx->pack_ev, hipEventDisableTiming);
        if (ctx->pack_ev) e = hipEventRecord(ctx->pack_ev, ctx->stream);
    }
    if (e != hipSuccess) {
        (void)hipStreamSynchronize(ctx->stream);
        return dvs_hip_fail(ctx, e, "packed sequence upload");
    }
    return DVS_OK;
}

// ---- C ABI (include/dvs_hip.h)
extern "C" int dvs_pack_sequences(dvs_ctx *ctx, const uint8_t *seqs, int seqs_on_device, uint64_t nbases,
                                  dvs_packed **out) {
    if (!ctx || !out || (!seqs && nbases)) return dvs_set_error(ctx, DVS_ERR_VALUE, "null argument");
    *out = nullptr;
    DVS_HIP(ctx, hipSetDevice(ctx->device));
    if (seqs_on_device && (reinterpret_cast<uintptr_t>(seqs) & 15))
        return dvs_set_error(ctx, DVS_ERR_VALUE, "device sequence buffer must be 16-byte aligned");
    dvs_packed *p = nullptr;
    int rc = dvs_packed_alloc(ctx, nbases, &p);
    if (rc) return rc;
    rc = seqs_on_device ? dvs_packed_fill_from_device(ctx, p, seqs) : dvs_packed_fill_from_host(ctx, p, seqs);
    if (!rc && !seqs_on_device && hipStreamSynchronize(ctx->stream) != hipSuccess)  // (the caller's buffer is free again)
        rc = dvs_set_error(ctx, DVS_ERR_RUNTIME, "packed sequence upload failed");
    if (rc) {
        dvs_packed_destroy(p);
        return rc;
    }
    *out = p;
    return DVS_OK;
}

extern "C" int dvs_packed_info(const dvs_packed *p, uint64_t *nbases, uint64_t *nwords) {
    if (!p) return DVS_ERR_VALUE;
    if (nbases) *nbases = p->nbases;
    if (nwords) *nwords = p->nwords;
    return DVS_OK;
}
extern "C" const void *dvs_packed_dev_codes(const dvs_packed *p) { return p ? p->d_codes : nullptr; }
extern "C" const void *dvs_packed_dev_mask(const dvs_packed *p) { return p ? p->d_mask : nullptr; }

extern "C" int dvs_packed_get(dvs_ctx *ctx, const dvs_packed *p, uint32_t *codes_out, uint16_t *mask_out) {
    if (!ctx || !p) return dvs_set_error(ctx, DVS_ERR_VALUE, "null argument");
    if (!p->nwords) return DVS_OK;
    if (codes_out) DVS_HIP(ctx, hipMemcpyAsync(codes_out, p->d_codes, p->nwords * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (mask_out) DVS_HIP(ctx, hipMemcpyAsync(mask_out, p->d_mask, p->nwords * 2, hipMemcpyDeviceToHost, ctx->stream));
    DVS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return DVS_OK;
}
