// Context, errors, matrix build / accessors of the C ABI (include/dvs_hip.h).
#include "dvs_internal.h"

#include <cmath>
#include <cstring>
#include <mutex>

uint64_t dvs_pow_u64(uint32_t base, uint32_t exp, bool *overflow);
void dvs_matrix_free_fields(dvs_matrix *m);
int dvs_matrix_fill_counts(dvs_ctx *ctx, dvs_matrix *m, const dvs_seq_view &sv, const uint64_t *offsets, bool no_wait);
int dvs_matrix_fill_freq_entropy(dvs_ctx *ctx, dvs_matrix *m);
int dvs_hist_prepare(dvs_ctx *ctx, const uint64_t *offsets, uint32_t nseq, uint32_t k, uint64_t nbins, uint64_t nbytes, size_t *n_long_out);
bool dvs_hist_rows_fit_u16(const dvs_ctx *ctx, uint64_t B, size_t n_long);
int dvs_matrix_fill_freq_totals(dvs_ctx *ctx, dvs_matrix *m, const double *d_meta);
int dvs_matrix_fill_compacted(dvs_ctx *ctx, dvs_matrix *m, const double *d_in, const double *d_meta);

static std::string g_create_err;

int dvs_set_error(dvs_ctx *ctx, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf;
    else g_create_err = buf;
    return code;
}

int dvs_hip_fail(dvs_ctx *ctx, hipError_t e, const char *what) {
    (void)hipGetLastError();  // clear the sticky error
    const int code = (e == hipErrorOutOfMemory) ? DVS_ERR_NOMEM : DVS_ERR_RUNTIME;
    return dvs_set_error(ctx, code, "HIP error %d (%s) in %s", int(e), hipGetErrorString(e), what);
}

int dvs_dev_alloc(dvs_ctx *ctx, void **ptr, size_t bytes, const char *what) {
    const size_t sz = ((bytes ? bytes : 1) + 4095) & ~size_t(4095);
    auto it = ctx->pool.find(sz);
    if (it != ctx->pool.end()) {
        *ptr = it->second;
        ctx->pool.erase(it);
        ctx->pool_bytes -= sz;
        ctx->live[*ptr] = sz;
        return DVS_OK;
    }
    hipError_t e = hipMalloc(ptr, sz);
    if (e != hipSuccess && !ctx->pool.empty()) {
        (void)hipGetLastError();
        dvs_dev_trim(ctx);
        e = hipMalloc(ptr, sz);
    }
    if (e != hipSuccess) {
        *ptr = nullptr;
        return dvs_hip_fail(ctx, e, what);
    }
    ctx->live[*ptr] = sz;
    return DVS_OK;
}

void dvs_dev_free(dvs_ctx *ctx, void *ptr) {
    if (!ptr) return;
    auto it = ctx ? ctx->live.find(ptr) : std::map<void *, size_t>::iterator();
    if (!ctx || it == ctx->live.end()) {
        (void)hipFree(ptr);
        return;
    }
    ctx->pool.emplace(it->second, ptr);
    ctx->pool_bytes += it->second;
    ctx->live.erase(it);
}

int dvs_pinned_get(dvs_ctx *ctx, void **ptr) {
    if (!ctx->pinned_pool.empty()) {
        *ptr = ctx->pinned_pool.back();
        ctx->pinned_pool.pop_back();
        return DVS_OK;
    }
    hipError_t e = hipHostMalloc(ptr, 4096, hipHostMallocDefault);
    if (e != hipSuccess) return dvs_hip_fail(ctx, e, "hipHostMalloc");
    return DVS_OK;
}
void dvs_pinned_put(dvs_ctx *ctx, void *ptr) {
    if (!ptr) return;
    if (ctx) ctx->pinned_pool.push_back(ptr);
    else (void)hipHostFree(ptr);
}
hipEvent_t dvs_event_get(dvs_ctx *ctx) {
    if (!ctx->event_pool.empty()) {
        hipEvent_t e = ctx->event_pool.back();
        ctx->event_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}
void dvs_event_put(dvs_ctx *ctx, hipEvent_t e) {
    if (!e) return;
    if (ctx) ctx->event_pool.push_back(e);
    else (void)hipEventDestroy(e);
}

void dvs_dev_trim(dvs_ctx *ctx) {
    (void)hipStreamSynchronize(ctx->stream);
    for (auto &kv : ctx->pool) (void)hipFree(kv.second);
    ctx->pool.clear();
    ctx->pool_bytes = 0;
}

// the environment's switches, parsed once per context (dvs_internal.h dvs_knobs)
void dvs_knobs_from_env(dvs_knobs *k) {
    auto on = [](const char *name) { return getenv(name) != nullptr; };
    *k = dvs_knobs();
    k->no_uniform_offsets = on("DVS_NO_UNIFORM_OFFSETS");
    k->no_offsets_cache = on("DVS_NO_OFFSETS_CACHE");
    k->counts_u32 = on("DVS_COUNTS_U32");
    k->build_wait = on("DVS_BUILD_WAIT");
    k->no_packed_upload = on("DVS_NO_PACKED_UPLOAD");
    k->cu_mask_set = on("HSA_CU_MASK") || on("ROC_GLOBAL_CU_MASK");
    k->no_persist = on("DVS_NO_PERSIST");
    k->no_fast_step = on("DVS_NO_FAST_STEP");
    k->no_head_phase = on("DVS_NO_HEAD_PHASE");
    k->persist_no_seeded = on("DVS_PERSIST_NO_SEEDED");
    k->persist_no_small = on("DVS_PERSIST_NO_SMALL");
    k->persist_no_coarse = on("DVS_PERSIST_NO_COARSE");
    k->persist_no_events = on("DVS_PERSIST_NO_EVENTS");
    k->persist_debug = on("DVS_PERSIST_DEBUG");
    const char *rounds = getenv("DVS_PERSIST_WG_ROUNDS");
    k->persist_wg_rounds = rounds ? atoi(rounds) : -1;
    k->ingest_no_stream = on("DVS_INGEST_NO_STREAM");
    const char *tk = getenv("DVS_TEST_KNOBS");
    k->test_persist_fake_error = tk && strstr(tk, "fake_persist_error") != nullptr;
    if (const char *lt = tk ? strstr(tk, "long_tile_") : nullptr) k->test_long_tile = uint32_t(strtoul(lt + 10, nullptr, 10));
    k->test_rowlog_ring = 0;
    if (const char *rr = tk ? strstr(tk, "rowlog_ring_") : nullptr) k->test_rowlog_ring = uint32_t(strtoul(rr + 12, nullptr, 10));
}

extern "C" {

int dvs_abi_version(void) { return DVS_ABI_VERSION; }

int dvs_ctx_refresh_knobs(dvs_ctx *ctx) {
    if (!ctx) return dvs_set_error(ctx, DVS_ERR_VALUE, "null argument");
    dvs_knobs_from_env(&ctx->knobs);
    ctx->off_cache.n_off = 0;  // (the tile lists of the last build were made under the old switches)
    return DVS_OK;
}

int dvs_ctx_create(int device, void *stream, dvs_ctx **out) {
    if (!out) return dvs_set_error(nullptr, DVS_ERR_VALUE, "null argument");
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0)
        return dvs_set_error(nullptr, DVS_ERR_RUNTIME,
                             "no HIP device available (%s): libdvs_hip has no CPU fallback",
                             e != hipSuccess ? hipGetErrorString(e) : "device count 0");
    if (device < 0) {
        e = hipGetDevice(&device);
        if (e != hipSuccess) return dvs_hip_fail(nullptr, e, "hipGetDevice");
    }
    if (device >= ndev)
        return dvs_set_error(nullptr, DVS_ERR_VALUE, "device %d out of range (%d devices)", device, ndev);
    e = hipSetDevice(device);
    if (e != hipSuccess) return dvs_hip_fail(nullptr, e, "hipSetDevice");
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) return dvs_hip_fail(nullptr, e, "hipGetDeviceProperties");
    dvs_ctx *ctx = new dvs_ctx();
    dvs_knobs_from_env(&ctx->knobs);
    ctx->device = device;
    ctx->n_cu = prop.multiProcessorCount;
    ctx->lds_per_block = prop.sharedMemPerBlockOptin ? prop.sharedMemPerBlockOptin
                                                     : prop.sharedMemPerBlock;
    if (ctx->lds_per_block < prop.sharedMemPerBlock) ctx->lds_per_block = prop.sharedMemPerBlock;
    if (stream) {
        ctx->stream = static_cast<hipStream_t>(stream);
    } else {
        e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
        if (e != hipSuccess) {
            delete ctx;
            return dvs_hip_fail(nullptr, e, "hipStreamCreate");
        }
        ctx->own_stream = true;
    }
    *out = ctx;
    return DVS_OK;
}

void dvs_ctx_destroy(dvs_ctx *ctx) {
    if (!ctx || ctx->owner_gone) return;
    ctx->owner_gone = true;
    dvs_ctx_release(ctx);
}

}  // extern "C"

int dvs_raise_dyn_lds(dvs_ctx *ctx, const void *fn, size_t bytes) {
    if (bytes <= 48 * 1024) return DVS_OK;
    size_t &have = ctx->lds_raised[fn];
    if (have >= bytes) return DVS_OK;
    DVS_HIP(ctx, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, int(bytes)));
    have = bytes;
    return DVS_OK;
}

hipStream_t dvs_ctx_stream2(dvs_ctx *ctx) {
    if (!ctx->stream2 && hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking) != hipSuccess) {
        (void)hipGetLastError();
        ctx->stream2 = nullptr;
    }
    return ctx->stream2;
}

// Two streams that share no CU.  Mask bit i is CU i / 8 of XCD i % 8 on this part (checked with
// scripts/micro/cu_mask.hip: bits 0..63 = 8 CUs on each of the 8 XCDs, no place shared with the
// complement, cooperative launches accepted), so the first head_cus bits are an even share of every XCD.
bool dvs_ctx_cu_split(dvs_ctx *ctx) {
    if (ctx->cu_split_tried) return ctx->stream_head != nullptr;
    ctx->cu_split_tried = true;
    if (ctx->n_cu < 128 || ctx->n_cu % 8) return false;
    if (ctx->knobs.cu_mask_set) return false;  // (the bit layout below assumes every CU)
    int head = 64;
    head = std::max(16, std::min(ctx->n_cu / 2, head)) & ~7;
    const uint32_t words = uint32_t(ctx->n_cu + 31) / 32;
    std::vector<uint32_t> lo(words, 0u), hi(words, 0u);
    for (int i = 0; i < ctx->n_cu; i++) (i < head ? lo : hi)[i / 32] |= 1u << (i % 32);
    hipStream_t a = nullptr, b = nullptr;
    if (hipExtStreamCreateWithCUMask(&a, words, lo.data()) != hipSuccess ||
        hipExtStreamCreateWithCUMask(&b, words, hi.data()) != hipSuccess) {
        (void)hipGetLastError();
        if (a) (void)hipStreamDestroy(a);
        return false;
    }
    ctx->stream_head = a;
    ctx->stream_rest = b;
    ctx->head_cus = head;
    return true;
}

void dvs_ctx_retain(dvs_ctx *ctx) {
    if (ctx) ctx->refs++;
}

void dvs_ctx_release(dvs_ctx *ctx) {
    if (!ctx || --ctx->refs > 0) return;
    (void)hipSetDevice(ctx->device);
    dvs_dev_free(ctx, ctx->off_cache.d_off);
    dvs_dev_free(ctx, ctx->off_cache.d_rows);
    dvs_dev_free(ctx, ctx->off_cache.d_tiles);
    if (ctx->off_cache.ev_up) (void)hipEventDestroy(ctx->off_cache.ev_up);
    if (ctx->off_cache.h_off) (void)hipHostFree(ctx->off_cache.h_off);
    if (ctx->pack_ev) (void)hipEventDestroy(ctx->pack_ev);
    if (ctx->h_pack) (void)hipHostFree(ctx->h_pack);
    dvs_dev_free(ctx, ctx->d_clog_tbl);
    dvs_dev_trim(ctx);
    for (void *p : ctx->pinned_pool) (void)hipHostFree(p);
    for (hipEvent_t e : ctx->event_pool) (void)hipEventDestroy(e);
    if (ctx->stream2) (void)hipStreamDestroy(ctx->stream2);
    if (ctx->stream_head) (void)hipStreamDestroy(ctx->stream_head);
    if (ctx->stream_rest) (void)hipStreamDestroy(ctx->stream_rest);
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

extern "C" {

const char *dvs_last_error(const dvs_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

int dvs_ctx_trim(dvs_ctx *ctx) {
    if (!ctx) return DVS_ERR_VALUE;
    dvs_dev_trim(ctx);
    return DVS_OK;
}

int dvs_ctx_sync(dvs_ctx *ctx) {
    if (!ctx) return DVS_ERR_VALUE;
    DVS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return DVS_OK;
}

int dvs_ctx_set_timing(dvs_ctx *ctx, int on) {
    if (!ctx) return DVS_ERR_VALUE;
    ctx->timing = on != 0;
    return DVS_OK;
}

int dvs_ctx_device_info(dvs_ctx *ctx, char *name, size_t name_len, int *n_cu, uint64_t *hbm_bytes) {
    if (!ctx) return DVS_ERR_VALUE;
    hipDeviceProp_t prop;
    DVS_HIP(ctx, hipGetDeviceProperties(&prop, ctx->device));
    if (name && name_len) {
        snprintf(name, name_len, "%s (%s)", prop.name, prop.gcnArchName);
    }
    if (n_cu) *n_cu = prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = prop.totalGlobalMem;
    return DVS_OK;
}

static int matrix_alloc(dvs_ctx *ctx, dvs_matrix *m) {
    const size_t cells = size_t(m->nrows) * m->nbins;
    const size_t bytes = cells * (m->kind == 0 ? 4 : m->kind == 2 ? 2 : 8);
    size_t free_b = 0, total_b = 0;
    const bool cached = ctx->pool.count(((bytes ? bytes : 4) + 4095) & ~size_t(4095)) > 0;
    if (!cached) DVS_HIP(ctx, hipMemGetInfo(&free_b, &total_b));
    if (!cached && bytes + size_t(m->nrows) * 12 + (64u << 20) > free_b + ctx->pool_bytes)
        return dvs_set_error(ctx, DVS_ERR_NOMEM,
                             "%u x %llu matrix needs %zu bytes of HBM, %zu free", m->nrows,
                             (unsigned long long)m->nbins, bytes, free_b);
    const size_t nr = m->nrows ? m->nrows : 1;
    m->ctx = ctx;
    dvs_ctx_retain(ctx);
    int rc;
    if (m->kind == 0) rc = dvs_dev_alloc(ctx, (void **)&m->d_counts, bytes ? bytes : 4, "matrix counts");
    else if (m->kind == 2) rc = dvs_dev_alloc(ctx, (void **)&m->d_counts16, bytes ? bytes : 4, "matrix counts (16-bit)");
    else rc = dvs_dev_alloc(ctx, (void **)&m->d_freqs, bytes ? bytes : 8, "matrix freqs");
    if (!rc) rc = dvs_dev_alloc(ctx, (void **)&m->d_totals, nr * 4, "matrix totals");
    if (!rc) rc = dvs_dev_alloc(ctx, (void **)&m->d_entropy, nr * 8, "matrix entropy");
    return rc;
}

// k and num_states of a build: the reference's panics first, then what the device path cannot hold
static int build_shape(dvs_ctx *ctx, uint32_t k, uint32_t num_states, uint64_t *B_out) {
    if (k == 0) return dvs_set_error(ctx, DVS_ERR_VALUE, "k cannot be 0");  // record.rs:126
    if (num_states < 1 || num_states > 255)
        return dvs_set_error(ctx, DVS_ERR_VALUE, "num_states %u outside 1..255", num_states);
    if (k > 17) return dvs_set_error(ctx, DVS_ERR_UNSUPPORTED, "k = %u > 17 is not supported", k);
    bool ovf = false;
    const uint64_t B = dvs_pow_u64(num_states, k, &ovf);
    if (ovf || B > (1ull << 32))
        return dvs_set_error(ctx, DVS_ERR_UNSUPPORTED, "%u^%u bins do not fit a dense count row",
                             num_states, k);
    *B_out = B;
    return DVS_OK;
}

// The build over sequences that are in HBM already, in either form (sv): offsets validated (and the tile
// lists of genome-length sequences derived), matrix allocated, kernels launched.
static int matrix_build_view(dvs_ctx *ctx, const dvs_seq_view &sv, const uint64_t *offsets, uint32_t nseq, uint32_t k,
                             uint32_t num_states, bool no_wait, dvs_matrix **out) {
    *out = nullptr;
    uint64_t B = 0;
    const int arc = build_shape(ctx, k, num_states, &B);
    if (arc) return arc;
    // the offsets first: whether any sequence needs more than one tile decides the width of the rows
    size_t n_long = 0;
    if (nseq) {
        const int prc = dvs_hist_prepare(ctx, offsets, nseq, k, B, sv.nbytes, &n_long);
        if (prc) return prc;
    }
    dvs_matrix *m = new dvs_matrix();
    m->kind = (nseq && dvs_hist_rows_fit_u16(ctx, B, n_long)) ? 2 : 0;
    m->nrows = nseq;
    m->nbins = B;
    m->k = k;
    m->num_states = num_states;
    m->device = ctx->device;
    int rc = matrix_alloc(ctx, m);
    // (a device-resident input needs no host wait: the kernels' completion is an event the consumers
    // of the matrix wait on when they need host-side data, dvs_matrix_settle)
    if (!rc && nseq) rc = dvs_matrix_fill_counts(ctx, m, sv, offsets, no_wait);
    if (rc) {
        dvs_matrix_free_fields(m);
        delete m;
        return rc;
    }
    *out = m;
    return DVS_OK;
}

int dvs_matrix_build(dvs_ctx *ctx, const uint8_t *seqs, int seqs_on_device, const uint64_t *offsets,
                     uint32_t nseq, uint32_t k, uint32_t num_states, dvs_matrix **out) {
    if (!ctx || !offsets || !out || (!seqs && nseq && offsets[nseq] > 0))
        return dvs_set_error(ctx, DVS_ERR_VALUE, "null argument");
    *out = nullptr;
    {
        uint64_t B_ = 0;
        const int arc = build_shape(ctx, k, num_states, &B_);
        if (arc) return arc;
    }
    DVS_HIP(ctx, hipSetDevice(ctx->device));
    const uint64_t nbytes = nseq ? offsets[nseq] : 0;
    dvs_seq_view sv;
    if (seqs_on_device) {
        if (reinterpret_cast<uintptr_t>(seqs) & 15)
            return dvs_set_error(ctx, DVS_ERR_VALUE, "device sequence buffer must be 16-byte aligned");
        sv.seqs = seqs;
        sv.nbytes = nbytes;
        return matrix_build_view(ctx, sv, offsets, nseq, k, num_states, !ctx->knobs.build_wait, out);
    }
    // Host memory.  Four-state sequences cross PCIe packed -- 3 bits per base, packed by host threads
    // beside the copies -- and the histogram reads the packed words as they are (pack.hip); anything
    // else is copied as it is.
    if (nbytes && dvs_packed_upload_wanted(ctx, num_states, nbytes)) {
        dvs_packed *p = nullptr;
        int rc = dvs_packed_alloc(ctx, nbytes, &p);
        if (!rc) rc = dvs_packed_fill_from_host(ctx, p, seqs);
        if (!rc) {
            sv.codes = p->d_codes;
            sv.mask = p->d_mask;
            sv.nbytes = nbytes;
            rc = matrix_build_view(ctx, sv, offsets, nseq, k, num_states, false, out);
        }
        if (p) {
            (void)hipStreamSynchronize(ctx->stream);  // (the planes go back to the cache: nothing may still read them)
            dvs_packed_destroy(p);
        }
        return rc;
    }
    const uint64_t padded = ((nbytes + 15) & ~15ull) + 16;
    uint8_t *d_tmp = nullptr;
    int rc = dvs_dev_alloc(ctx, (void **)&d_tmp, padded, "sequence upload buffer");
    if (rc) return rc;
    hipError_t ue = hipMemsetAsync(d_tmp + (nbytes & ~15ull), 0xFF, padded - (nbytes & ~15ull), ctx->stream);
    if (ue == hipSuccess && nbytes) ue = hipMemcpyAsync(d_tmp, seqs, nbytes, hipMemcpyHostToDevice, ctx->stream);
    if (ue != hipSuccess) {  // a failed upload must not become a silently wrong matrix
        (void)hipStreamSynchronize(ctx->stream);
        dvs_dev_free(ctx, d_tmp);
        return dvs_hip_fail(ctx, ue, "sequence upload");
    }
    sv.seqs = d_tmp;
    sv.nbytes = padded;
    rc = matrix_build_view(ctx, sv, offsets, nseq, k, num_states, false, out);
    (void)hipStreamSynchronize(ctx->stream);
    dvs_dev_free(ctx, d_tmp);
    return rc;
}

int dvs_matrix_build_packed(dvs_ctx *ctx, const dvs_packed *p, const uint64_t *offsets, uint32_t nseq, uint32_t k,
                            dvs_matrix **out) {
    if (!ctx || !p || !offsets || !out) return dvs_set_error(ctx, DVS_ERR_VALUE, "null argument");
    *out = nullptr;
    DVS_HIP(ctx, hipSetDevice(ctx->device));
    dvs_seq_view sv;
    sv.codes = p->d_codes;
    sv.mask = p->d_mask;
    sv.nbytes = p->nbases;
    p->async_readers = true;  // (dvs_packed_destroy waits for the kernels enqueued here)
    return matrix_build_view(ctx, sv, offsets, nseq, k, 4, !ctx->knobs.build_wait, out);
}

int dvs_matrix_from_freqs(dvs_ctx *ctx, const double *freqs, uint32_t nrows, uint64_t nbins,
                          dvs_matrix **out) {
    if (!ctx || !out || (!freqs && nrows)) return dvs_set_error(ctx, DVS_ERR_VALUE, "null argument");
    *out = nullptr;
    if (nbins == 0)  // record.rs:87-89
        return dvs_set_error(ctx, DVS_ERR_VALUE, "cannot calculate entropy as frequency vector empty");
    // KmerSeq::new -> entropy(): sum over the non-zero bins, in order, must be within
    // len * eps of 1 (record.rs:90-104).  Exact on the host: plain sequential adds.
    const double tol = double(nbins) * DVS_EPS;
    for (uint32_t r = 0; r < nrows; r++) {
        const double *row = freqs + uint64_t(r) * nbins;
        double tot = 0.0;
        for (uint64_t i = 0; i < nbins; i++)
            if (row[i] != 0.0) tot += row[i];
        if (!(std::fabs(tot - 1.0) <= tol))
            return dvs_set_error(ctx, DVS_ERR_VALUE,
                                 "cannot calculate entropy as frequency vector total %.17g!=1.0", tot);
    }
    DVS_HIP(ctx, hipSetDevice(ctx->device));
    dvs_matrix *m = new dvs_matrix();
    m->kind = 1;
    m->nrows = nrows;
    m->nbins = nbins;
    m->device = ctx->device;
    int rc = matrix_alloc(ctx, m);
    if (!rc && nrows) {
        std::vector<uint32_t> ones(nrows, 1u);
        hipError_t e = hipMemcpyAsync(m->d_freqs, freqs, size_t(nrows) * nbins * 8,
                                      hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess)
            e = hipMemcpyAsync(m->d_totals, ones.data(), size_t(nrows) * 4, hipMemcpyHostToDevice,
                               ctx->stream);
        if (e != hipSuccess) rc = dvs_hip_fail(ctx, e, "hipMemcpyAsync(freqs)");
        if (!rc) rc = dvs_matrix_fill_freq_entropy(ctx, m);
        if (!rc && hipStreamSynchronize(ctx->stream) != hipSuccess)
            rc = dvs_set_error(ctx, DVS_ERR_RUNTIME, "stream sync failed");
    }
    if (rc) {
        dvs_matrix_free_fields(m);
        delete m;
        return rc;
    }
    *out = m;
    return DVS_OK;
}

// Frequency rows already in HBM (e.g. the all-gathered winners of a chunked run).  meta may be
// NULL; otherwise meta[2 r + 1] == 0 marks row r as padding (it is given total 0 and skipped like
// a sequence without valid k-mers).  The rows are copied; they are trusted to be frequency
// vectors produced by this library (count / total rows always pass the reference's sum check).
int dvs_matrix_from_device_freqs(dvs_ctx *ctx, const double *d_freqs, const double *d_meta, uint32_t nrows,
                                 uint64_t nbins, dvs_matrix **out) {
    if (!ctx || !out || (!d_freqs && nrows)) return dvs_set_error(ctx, DVS_ERR_VALUE, "null argument");
    *out = nullptr;
    if (nbins == 0)
        return dvs_set_error(ctx, DVS_ERR_VALUE, "cannot calculate entropy as frequency vector empty");
    DVS_HIP(ctx, hipSetDevice(ctx->device));
    dvs_matrix *m = new dvs_matrix();
    m->kind = 1;
    m->nrows = nrows;
    m->nbins = nbins;
    m->device = ctx->device;
    int rc = matrix_alloc(ctx, m);
    if (!rc && nrows && d_meta) {
        // flagged rows: real ones first (in input order), padding behind them, in one copy pass
        rc = dvs_dev_alloc(ctx, (void **)&m->d_src_row, size_t(nrows) * 4, "source rows");
        if (!rc) rc = dvs_matrix_fill_compacted(ctx, m, d_freqs, d_meta);
        if (!rc) rc = dvs_matrix_fill_freq_entropy(ctx, m);
    } else if (!rc && nrows) {
        hipError_t e = hipMemcpyAsync(m->d_freqs, d_freqs, size_t(nrows) * nbins * 8, hipMemcpyDeviceToDevice,
                                      ctx->stream);
        if (e != hipSuccess) rc = dvs_hip_fail(ctx, e, "hipMemcpyAsync(freqs)");
        if (!rc) rc = dvs_matrix_fill_freq_totals(ctx, m, nullptr);
        if (!rc) rc = dvs_matrix_fill_freq_entropy(ctx, m);
    }
    if (rc) {
        dvs_matrix_free_fields(m);
        delete m;
        return rc;
    }
    *out = m;
    return DVS_OK;
}

int dvs_matrix_get_source_rows(dvs_ctx *ctx, const dvs_matrix *m, uint32_t *out) {
    if (!ctx || !m || !out) return dvs_set_error(ctx, DVS_ERR_VALUE, "null argument");
    if (!m->d_src_row) {
        for (uint32_t r = 0; r < m->nrows; r++) out[r] = r;
        return DVS_OK;
    }
    DVS_HIP(ctx, hipMemcpyAsync(out, m->d_src_row, size_t(m->nrows) * 4, hipMemcpyDeviceToHost, ctx->stream));
    DVS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return DVS_OK;
}

void dvs_matrix_destroy(dvs_matrix *m) {
    if (!m) return;
    dvs_matrix_free_fields(m);
    delete m;
}

uint32_t dvs_matrix_nrows(const dvs_matrix *m) { return m ? m->nrows : 0; }
uint64_t dvs_matrix_nbins(const dvs_matrix *m) { return m ? m->nbins : 0; }
const void *dvs_matrix_dev_counts(const dvs_matrix *m) {
    return !m ? nullptr : m->kind == 2 ? static_cast<const void *>(m->d_counts16) : static_cast<const void *>(m->d_counts);
}
uint32_t dvs_matrix_count_bytes(const dvs_matrix *m) { return !m ? 0u : m->kind == 0 ? 4u : m->kind == 2 ? 2u : 0u; }
const void *dvs_matrix_dev_totals(const dvs_matrix *m) { return m ? m->d_totals : nullptr; }
const void *dvs_matrix_dev_entropy(const dvs_matrix *m) { return m ? m->d_entropy : nullptr; }

int dvs_matrix_get_counts(dvs_ctx *ctx, const dvs_matrix *m, uint32_t row0, uint32_t nrows,
                          uint32_t *out) {
    if (!ctx || !m || !out) return dvs_set_error(ctx, DVS_ERR_VALUE, "null argument");
    if (m->kind == 1) return dvs_set_error(ctx, DVS_ERR_VALUE, "not a count matrix");
    if (uint64_t(row0) + nrows > m->nrows) return dvs_set_error(ctx, DVS_ERR_VALUE, "row range out of bounds");
    if (!nrows) return DVS_OK;
    if (m->kind == 2) {  // 16-bit rows: copied as they are, widened on the host
        const size_t cells = size_t(nrows) * m->nbins;
        std::vector<uint16_t> tmp(cells);
        DVS_HIP(ctx, hipMemcpyAsync(tmp.data(), m->d_counts16 + uint64_t(row0) * m->nbins, cells * 2,
                                    hipMemcpyDeviceToHost, ctx->stream));
        DVS_HIP(ctx, hipStreamSynchronize(ctx->stream));
        for (size_t i = 0; i < cells; i++) out[i] = tmp[i];
        return DVS_OK;
    }
    DVS_HIP(ctx, hipMemcpyAsync(out, m->d_counts + uint64_t(row0) * m->nbins,
                                size_t(nrows) * m->nbins * 4, hipMemcpyDeviceToHost, ctx->stream));
    DVS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return DVS_OK;
}

int dvs_matrix_get_totals(dvs_ctx *ctx, const dvs_matrix *m, uint32_t *out) {
    if (!ctx || !m || !out) return dvs_set_error(ctx, DVS_ERR_VALUE, "null argument");
    if (!m->nrows) return DVS_OK;
    DVS_HIP(ctx, hipMemcpyAsync(out, m->d_totals, size_t(m->nrows) * 4, hipMemcpyDeviceToHost,
                                ctx->stream));
    DVS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return DVS_OK;
}

int dvs_matrix_get_entropy(dvs_ctx *ctx, const dvs_matrix *m, double *out) {
    if (!ctx || !m || !out) return dvs_set_error(ctx, DVS_ERR_VALUE, "null argument");
    if (!m->nrows) return DVS_OK;
    DVS_HIP(ctx, hipMemcpyAsync(out, m->d_entropy, size_t(m->nrows) * 8, hipMemcpyDeviceToHost,
                                ctx->stream));
    DVS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return DVS_OK;
}

int dvs_kmer_counts(dvs_ctx *ctx, const uint8_t *seqs, const uint64_t *offsets, uint32_t nseq,
                    uint32_t k, uint32_t num_states, uint32_t *counts_out, uint32_t *totals_out,
                    double *entropy_out) {
    dvs_matrix *m = nullptr;
    int rc = dvs_matrix_build(ctx, seqs, 0, offsets, nseq, k, num_states, &m);
    if (rc) return rc;
    if (counts_out) rc = dvs_matrix_get_counts(ctx, m, 0, nseq, counts_out);
    if (!rc && totals_out) rc = dvs_matrix_get_totals(ctx, m, totals_out);
    if (!rc && entropy_out) rc = dvs_matrix_get_entropy(ctx, m, entropy_out);
    dvs_matrix_destroy(m);
    return rc;
}

}  // extern "C"
