// Packed upload of four-state sequences handed over in host memory (dvs_matrix_build from a host pointer:
// what diverse_seq._dvs.nmost_divergent / max_divergent do with a store's sequences, src/lib.rs:59-73,
// src/record.rs:205-209).  The boundary's convention is one byte per base; on PCIe that is the whole cost
// of the call (500 MB for 100k x 5 kb: ~10 ms, the selection itself takes 1.5).  So host threads pack the
// stream chunk by chunk -- 2 bits per base + 1 "invalid" bit per base, 3/8 of the bytes -- into a pinned
// staging block the context keeps, every chunk is sent as soon as it is packed (copy i + 1 runs beside the
// packing of chunk i + 2), and one kernel expands the stream again to the one-byte form the histogram
// reads (any symbol >= 4 comes back as 0xFF: the histogram only asks "valid or not").
#include "dvs_internal.h"

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <thread>
#include <vector>

extern "C" void dvs_pack_bases(const uint8_t *src, size_t n, uint8_t *codes, uint8_t *mask);  // pack_host.cpp

namespace {

constexpr size_t PACK_CHUNK = size_t(4) << 20;                   // bases per chunk (a multiple of 32)
constexpr size_t PACK_BLOCK = PACK_CHUNK / 4 + PACK_CHUNK / 8;   // a chunk's packed bytes: its codes, then its mask

// 16 bases per thread: one u32 of codes + one u16 of mask -> one 16-byte store
__global__ __launch_bounds__(256) void unpack_kernel(const uint8_t *__restrict__ packed, uint8_t *__restrict__ out,
                                                     uint64_t ngroups) {
    const uint64_t t = uint64_t(blockIdx.x) * 256u + threadIdx.x;
    if (t >= ngroups) return;
    const uint64_t g = t * 16, c = g / PACK_CHUNK, r = g % PACK_CHUNK;
    const uint8_t *blk = packed + c * PACK_BLOCK;
    const uint32_t codes = *reinterpret_cast<const uint32_t *>(blk + r / 4);
    const uint32_t m = *reinterpret_cast<const uint16_t *>(blk + PACK_CHUNK / 4 + r / 8);
    uint32_t w[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        uint32_t word = 0;
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const int i = q * 4 + b;
            const uint32_t v = ((m >> i) & 1u) ? 0xFFu : ((codes >> (2 * i)) & 3u);
            word |= v << (8 * b);
        }
        w[q] = word;
    }
    *reinterpret_cast<uint4 *>(out + g) = make_uint4(w[0], w[1], w[2], w[3]);
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

// seqs[0, nbytes) (host, one byte per base, four states) -> d_out[0, ceil16(nbytes)) on the context's
// stream; symbols >= 4 and the positions behind nbytes up to the next multiple of 16 become 0xFF.
int dvs_upload_packed(dvs_ctx *ctx, const uint8_t *seqs, uint64_t nbytes, uint8_t *d_out) {
    const size_t nchunks = size_t((nbytes + PACK_CHUNK - 1) / PACK_CHUNK);
    const size_t packed_bytes = nchunks * PACK_BLOCK;
    // the staging block: nobody may still be reading it (the previous call's last copy)
    if (ctx->pack_ev) (void)hipEventSynchronize(ctx->pack_ev);
    if (ctx->h_pack_cap < packed_bytes) {
        if (ctx->h_pack) (void)hipHostFree(ctx->h_pack);
        ctx->h_pack = nullptr;
        ctx->h_pack_cap = 0;
        const size_t cap = packed_bytes + packed_bytes / 8;
        const hipError_t he = hipHostMalloc(&ctx->h_pack, cap, hipHostMallocDefault);
        if (he != hipSuccess) {
            ctx->h_pack = nullptr;
            return dvs_hip_fail(ctx, he, "pinned staging block of the packed upload");
        }
        ctx->h_pack_cap = cap;
    }
    uint8_t *stage = static_cast<uint8_t *>(ctx->h_pack);
    uint8_t *d_packed = nullptr;
    int rc = dvs_dev_alloc(ctx, (void **)&d_packed, packed_bytes, "packed sequences");
    if (rc) return rc;
    std::unique_ptr<std::atomic<int>[]> done(new std::atomic<int>[nchunks]);
    for (size_t c = 0; c < nchunks; c++) done[c].store(0, std::memory_order_relaxed);
    std::atomic<size_t> next{0};
    auto work = [&]() {
        for (;;) {
            const size_t c = next.fetch_add(1, std::memory_order_relaxed);
            if (c >= nchunks) return;
            const uint64_t a = uint64_t(c) * PACK_CHUNK;
            const size_t n = size_t(std::min<uint64_t>(PACK_CHUNK, nbytes - a));
            uint8_t *blk = stage + c * PACK_BLOCK;
            dvs_pack_bases(seqs + a, n, blk, blk + PACK_CHUNK / 4);
            done[c].store(1, std::memory_order_release);
        }
    };
    const unsigned nthr = unsigned(std::min<size_t>(dvs_host_threads(), nchunks));
    std::vector<std::thread> pool;
    for (unsigned t = 0; t + 1 < nthr; t++) pool.emplace_back(work);  // (this thread sends; with one core it packs too)
    hipError_t e = hipSuccess;
    for (size_t c = 0; c < nchunks && e == hipSuccess; c++) {
        while (!done[c].load(std::memory_order_acquire)) {
            if (nthr <= 1 || next.load(std::memory_order_relaxed) < nchunks) {
                // nothing to send yet: pack a chunk here instead of spinning
                const size_t mine = next.fetch_add(1, std::memory_order_relaxed);
                if (mine < nchunks) {
                    const uint64_t a = uint64_t(mine) * PACK_CHUNK;
                    const size_t n = size_t(std::min<uint64_t>(PACK_CHUNK, nbytes - a));
                    uint8_t *blk = stage + mine * PACK_BLOCK;
                    dvs_pack_bases(seqs + a, n, blk, blk + PACK_CHUNK / 4);
                    done[mine].store(1, std::memory_order_release);
                    continue;
                }
            }
            std::this_thread::yield();
        }
        e = hipMemcpyAsync(d_packed + c * PACK_BLOCK, stage + c * PACK_BLOCK, PACK_BLOCK, hipMemcpyHostToDevice, ctx->stream);
    }
    for (std::thread &t : pool) t.join();
    if (e == hipSuccess) {
        if (!ctx->pack_ev) (void)hipEventCreateWithFlags(&ctx->pack_ev, hipEventDisableTiming);
        if (ctx->pack_ev) e = hipEventRecord(ctx->pack_ev, ctx->stream);
    }
    if (e == hipSuccess) {
        const uint64_t ngroups = (nbytes + 15) / 16;
        hipLaunchKernelGGL(unpack_kernel, dim3(uint32_t((ngroups + 255) / 256)), dim3(256), 0, ctx->stream, d_packed, d_out,
                           ngroups);
        e = hipGetLastError();
    }
    dvs_dev_free(ctx, d_packed);  // (back to the pool: stream order protects it until the kernel has run)
    if (e != hipSuccess) {
        (void)hipStreamSynchronize(ctx->stream);
        return dvs_hip_fail(ctx, e, "packed sequence upload");
    }
    return DVS_OK;
}
