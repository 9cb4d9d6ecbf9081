/*
 * dvs_hip.h -- C ABI of libdvs_hip.so: the MI355X (gfx950) implementation of
 * DiverseSeq's k-mer / delta-JSD / mash hot path.
 *
 * This is the drop-in boundary.  It replaces the PyO3 extension module
 * `diverse_seq._dvs` (reference src/lib.rs:175-189): every entry point below
 * names the reference interface it stands in for.  Plain pointers and sizes
 * only; no torch / HIP types in the signatures (a HIP stream crosses as
 * void*).  All functions return 0 (DVS_OK) or a DVS_ERR_* code;
 * dvs_last_error() returns the message (for DVS_ERR_VALUE it is the
 * reference's panic text, which src/lib.rs:36-57 turns into ValueError).
 *
 * Data convention (reference diverse_seq/util.py:41-45, src/distance.rs:6-8):
 * a sequence is one byte per base holding the cogent3 alphabet index
 * (DNA: T0 C1 A2 G3); any byte >= num_states is a gap/ambiguity and
 * invalidates every k-mer window that contains it.  A batch of sequences is
 * one concatenated byte buffer plus nseq+1 uint64 offsets.
 *
 * Threading: one dvs_ctx per process per GPU; a ctx is not thread-safe.
 */
#ifndef DVS_HIP_H
#define DVS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DVS_ABI_VERSION 3

#define DVS_OK 0
#define DVS_ERR_VALUE 1       /* the reference would panic -> python ValueError */
#define DVS_ERR_RUNTIME 2     /* HIP failure / no device */
#define DVS_ERR_NOMEM 3
#define DVS_ERR_UNSUPPORTED 4 /* shape outside what the device path handles */
#define DVS_ERR_ZERODIV 5     /* python ZeroDivisionError (distance.py:283) */

typedef struct dvs_ctx dvs_ctx;       /* device, stream, scratch */
typedef struct dvs_matrix dvs_matrix; /* N x num_states^k count (or freq) matrix in HBM */
typedef struct dvs_select dvs_select; /* a SummedRecords set + the greedy engine */

/* ---- context ------------------------------------------------------------ */
int dvs_abi_version(void);
/* device < 0: current device.  stream: a hipStream_t to launch on, or NULL for
 * a stream owned by the ctx. */
int dvs_ctx_create(int device, void *stream, dvs_ctx **out);
/* Drops the caller's reference.  Matrices, selections and sequence batches made from ctx hold
 * one each, so they may be destroyed after it, in any order (a garbage-collected host language
 * gives none); the caches and the stream go when the last reference does.  ctx itself must not
 * be passed to another call afterwards. */
void dvs_ctx_destroy(dvs_ctx *ctx);
/* message of the last failing call on ctx (ctx == NULL: last ctx_create failure) */
const char *dvs_last_error(const dvs_ctx *ctx);
int dvs_ctx_sync(dvs_ctx *ctx);
/* the ctx caches device allocations released by *_destroy for reuse; this frees them */
int dvs_ctx_trim(dvs_ctx *ctx);
/* name, CU count and HBM bytes of the ctx's device */
int dvs_ctx_device_info(dvs_ctx *ctx, char *name, size_t name_len, int *n_cu,
                        uint64_t *hbm_bytes);

/* ---- k-mer count matrix ------------------------------------------------- *
 * replaces SeqRecord::to_kcounts / to_kmerseq (src/record.rs:124-141),
 * count_kmers (:41-84), count_monomers (:31-39), entropy (:86-106) and
 * LazySeq.get_kcounts / get_kfreqs (:247-262).
 *
 * Builds, on the device, row r = the k-mer histogram of sequence r (uint32),
 * plus per-row total (number of valid k-mers) and Shannon entropy (bits) of
 * the row's frequency vector.  `seqs` is a HOST pointer when seqs_on_device
 * is 0 (it is copied to HBM), else a 16-byte-aligned DEVICE pointer; `offsets` is always a
 * host array.  With a device pointer the call does NOT wait for its kernels: they are ordered on
 * the ctx stream in front of everything that uses the matrix, and the first call that needs
 * host-side data of it (a selection's seeds, dvs_matrix_get_*, dvs_ctx_sync) waits for them -- so
 * the device buffer must stay valid, and unmodified, until then (DVS_BUILD_WAIT=1 in the
 * environment restores a build that returns only when its kernels have finished). */
int dvs_matrix_build(dvs_ctx *ctx, const uint8_t *seqs, int seqs_on_device,
                     const uint64_t *offsets, uint32_t nseq, uint32_t k,
                     uint32_t num_states, dvs_matrix **out);
/* rows are frequency vectors (the members of SummedRecordsResult objects being
 * merged: get_kmerseqs_and_init_summed_records, src/records.rs:344-360).
 * DVS_ERR_VALUE if a row fails the reference's sum-to-one check
 * (src/record.rs:99-104). */
int dvs_matrix_from_freqs(dvs_ctx *ctx, const double *freqs, uint32_t nrows,
                          uint64_t nbins, dvs_matrix **out);
/* the same for rows already in HBM (the all-gathered winners of a chunked multi-GPU run);
 * d_meta may be NULL, else d_meta[2 r + 1] == 0 marks input row r as padding.  Rows are copied, no
 * host round trip; with d_meta the real rows come first in their input order and the padding
 * rows behind them (skipped like sequences without valid k-mers), so that the first n stream
 * positions are the first n real records of the concatenated results, as
 * get_kmerseqs_and_init_summed_records takes them (src/records.rs:344-360);
 * dvs_matrix_get_source_rows gives the input row of every matrix row. */
int dvs_matrix_from_device_freqs(dvs_ctx *ctx, const double *d_freqs, const double *d_meta,
                                 uint32_t nrows, uint64_t nbins, dvs_matrix **out);
int dvs_matrix_get_source_rows(dvs_ctx *ctx, const dvs_matrix *m, uint32_t *out); /* [nrows] */
void dvs_matrix_destroy(dvs_matrix *m);
uint32_t dvs_matrix_nrows(const dvs_matrix *m);
uint64_t dvs_matrix_nbins(const dvs_matrix *m);
/* device pointers (for zero-copy hand-off to the host framework) */
/* counts [nrows x nbins], or NULL for a frequency matrix.  The elements are uint32, or uint16 when
 * every sequence of the build was at most 32768 k-mer windows long and there are at most 4096 bins
 * (no count can reach 2^16; the build's write and the scan's read move half the bytes): dvs_matrix_count_bytes says which (4, 2;
 * 0 for a frequency matrix).  DVS_COUNTS_U32=1 in the environment keeps every build at uint32. */
const void *dvs_matrix_dev_counts(const dvs_matrix *m);
uint32_t dvs_matrix_count_bytes(const dvs_matrix *m);
const void *dvs_matrix_dev_totals(const dvs_matrix *m);  /* uint32 [nrows] */
const void *dvs_matrix_dev_entropy(const dvs_matrix *m); /* double [nrows] */
/* copies to host: counts rows [row0, row0+nrows), widened to uint32 whatever the device width */
int dvs_matrix_get_counts(dvs_ctx *ctx, const dvs_matrix *m, uint32_t row0,
                          uint32_t nrows, uint32_t *out);
int dvs_matrix_get_totals(dvs_ctx *ctx, const dvs_matrix *m, uint32_t *out);
int dvs_matrix_get_entropy(dvs_ctx *ctx, const dvs_matrix *m, double *out);
/* host one-shot: counts for a batch of host sequences (LazySeq.get_kcounts) */
int dvs_kmer_counts(dvs_ctx *ctx, const uint8_t *seqs, const uint64_t *offsets,
                    uint32_t nseq, uint32_t k, uint32_t num_states,
                    uint32_t *counts_out, uint32_t *totals_out, double *entropy_out);

/* ---- packed sequences ------------------------------------------------------ *
 * Four-state sequences at 3 bits per base instead of the reference's one byte per base
 * (src/record.rs:205-209, diverse_seq/util.py:32-45): the form the histogram (count_kmers,
 * src/record.rs:41-84) and sketch (get_kmer_hashes, src/distance.rs:101-134) kernels read from HBM
 * as it is.  Two planes over the CONCATENATED buffer -- positions, and therefore a batch's offsets,
 * are the same as in the byte form:
 *   codes: one uint32 per 16 bases; base 16 w + i in bits (31 - 2 i)..(30 - 2 i), value = index & 3
 *          (the first base of a k-mer is its most significant digit, src/record.rs:18-29)
 *   mask:  one uint16 per 16 bases; bit 15 - i set when base 16 w + i is >= 4 (gap / ambiguity /
 *          behind the end): every window holding such a base is skipped (src/record.rs:47-64)
 * dvs_pack_sequences packs nbases symbols -- a HOST buffer (host threads pack 4 Mi-base chunks beside
 * their copies: 3/8 of the bytes cross PCIe; returns when the buffer may be reused) or a 16-byte-aligned
 * DEVICE buffer (one kernel on the ctx stream, not waited for: keep the buffer until the next sync).
 * Symbols of an alphabet with more than four states cannot be packed: use the byte form. */
typedef struct dvs_packed dvs_packed;
int dvs_pack_sequences(dvs_ctx *ctx, const uint8_t *seqs, int seqs_on_device, uint64_t nbases,
                       dvs_packed **out);
void dvs_packed_destroy(dvs_packed *p);
int dvs_packed_info(const dvs_packed *p, uint64_t *nbases, uint64_t *nwords); /* nwords = ceil(nbases / 16) */
const void *dvs_packed_dev_codes(const dvs_packed *p); /* uint32 [nwords] in HBM */
const void *dvs_packed_dev_mask(const dvs_packed *p);  /* uint16 [nwords] in HBM */
int dvs_packed_get(dvs_ctx *ctx, const dvs_packed *p, uint32_t *codes_out, uint16_t *mask_out);
/* dvs_matrix_build / dvs_sketches_build over a packed batch (num_states is 4 by construction).
 * Neither waits for its kernels; p must outlive them (until the next call that syncs). */
int dvs_matrix_build_packed(dvs_ctx *ctx, const dvs_packed *p, const uint64_t *offsets, uint32_t nseq,
                            uint32_t k, dvs_matrix **out);

/* ---- ingest -------------------------------------- *
 * FASTA file bytes -> the data convention above, on the device: replaces, for the hot path's
 * input, the host-side parse + encode of diverse_seq/io.py:75-104 (dvs_load_seqs.main: records
 * parsed, their sequences joined with "-") and diverse_seq/util.py:32-45 (str2arr:
 * alphabet.to_indices).  `raw` is the file as it is (host pointer, or device pointer when
 * raw_on_device != 0); lut256 maps a file byte to its alphabet index (NULL: the cogent3 "dna"
 * table, dvs_default_alphabet_lut).  join_records != 0: one sequence per file, a gap symbol
 * between adjacent records (io.py:100); 0: one sequence per record.  The encoded bases stay in HBM
 * and feed dvs_matrix_build_from_seqbatch directly. */
typedef struct dvs_seqbatch dvs_seqbatch;
void dvs_default_alphabet_lut(int rna, uint8_t lut[256]);
int dvs_seqbatch_from_fasta(dvs_ctx *ctx, const uint8_t *raw, int raw_on_device, uint64_t nbytes,
                            const uint8_t *lut256, int join_records, dvs_seqbatch **out);
void dvs_seqbatch_destroy(dvs_seqbatch *b);
/* nseq sequences (1 when joined), total encoded symbols, records ('>' lines) in the file */
int dvs_seqbatch_info(const dvs_seqbatch *b, uint32_t *nseq, uint64_t *total_bases, uint32_t *nrecords);
int dvs_seqbatch_offsets(const dvs_seqbatch *b, uint64_t *offsets_out);        /* nseq + 1 */
int dvs_seqbatch_header_positions(const dvs_seqbatch *b, uint64_t *pos_out);   /* nrecords: offset of each '>' */
const void *dvs_seqbatch_dev_codes(const dvs_seqbatch *b);                     /* uint8 [total] in HBM */
int dvs_seqbatch_get_codes(dvs_ctx *ctx, const dvs_seqbatch *b, uint8_t *codes_out);
int dvs_matrix_build_from_seqbatch(dvs_ctx *ctx, const dvs_seqbatch *b, uint32_t k, uint32_t num_states,
                                   dvs_matrix **out);
/* The batch's encoded bases re-stated in the packed form above and the byte form released: a genome
 * collection then sits in HBM at 3/8 of the bytes, and dvs_matrix_build_from_seqbatch /
 * dvs_sketches_build_from_seqbatch (num_states 4) read the packed words.  dvs_seqbatch_dev_codes returns
 * NULL afterwards, dvs_seqbatch_get_codes unpacks (invalid symbols come back as 255), and
 * dvs_seqbatch_packed hands out the planes (NULL before).  DVS_ERR_VALUE for a batch that was encoded
 * with a caller's alphabet table instead of the library's DNA / RNA one (its symbols >= 4 are states, not
 * "invalid"). */
int dvs_seqbatch_pack(dvs_ctx *ctx, dvs_seqbatch *b);
const dvs_packed *dvs_seqbatch_packed(const dvs_seqbatch *b);

/* ---- greedy delta-JSD selection ------------------------------------------ *
 * replaces SummedRecords (src/records.rs:10-216), get_lowest_record_index
 * (:220-252) and the selectors select_nmost_divergent (:311-342),
 * select_nmost_divergent_final (:363-382), select_max_divergent (:390-454),
 * select_max_divergent_final (:456-507), make_summed_records (:509-524), i.e.
 * the bodies of _dvs.nmost_divergent / final_nmost / max_divergent /
 * final_max / get_delta_jsd_calculator (src/lib.rs:59-171).
 *
 * The candidate stream is `npos` positions; position p refers to matrix row
 * order[p] (order == NULL: row p) and carries identity label labels[p]
 * (labels == NULL: label = row index; equal labels == equal seqid, which the
 * reference ignores when already in the set, src/records.rs:71-73,87-89).
 * The first `n_seed` positions seed the set (rows without valid k-mers are
 * skipped, src/records.rs:299-306); the rest are streamed in order. */
#define DVS_MODE_NMOST 0 /* fixed size: replace_lowest on every JSD increase */
#define DVS_MODE_MAX 1   /* grow to max_size while std/cov of delta-JSD rises */
#define DVS_MODE_SET 2   /* build the set from all positions, stream nothing */
#define DVS_STAT_STDEV 0
#define DVS_STAT_COV 1

typedef struct dvs_select_params {
    uint32_t mode;     /* DVS_MODE_* */
    uint32_t n_seed;   /* n (nmost) or min_size (max); ignored for MODE_SET */
    uint32_t max_size; /* MODE_MAX only (already capped to npos by the caller or not) */
    uint32_t stat;     /* DVS_STAT_*, MODE_MAX only */
    uint32_t window;   /* rows scored per scan launch; 0 = library default */
    uint32_t flags;    /* DVS_SELECT_* */
} dvs_select_params;
#define DVS_SELECT_NO_ARBITER 1u /* fail with DVS_ERR_UNSUPPORTED instead of host tie arbitration */
#define DVS_SELECT_STEPWISE 2u   /* dvs_select_run builds the initial set and returns; the caller drives
                                    dvs_select_step_* (row-sharded multi-GPU runs, see below) */
#define DVS_ROW_REMOTE 0xFFFFFFFFu /* order[p]: the row of stream position p lives on another rank */

typedef struct dvs_select_summary {
    uint32_t size;
    uint32_t lowest_index;
    double total_jsd, mean_delta_jsd, std_delta_jsd, cov_delta_jsd, summed_entropies;
    /* engine statistics */
    uint64_t rows_scored;    /* candidate rows read by the scan kernel (re-scans included) */
    uint64_t rows_rechecked; /* rows the scan re-evaluated in full f64 */
    uint32_t n_windows, n_events, n_accepts, n_arbitrated;
    double scan_ms;          /* sum of scan-kernel durations (HIP events) when timing is on, else 0 */
    uint64_t scan_launches;  /* scan-kernel launches the events bracket (no-op launches included) */
    uint32_t engine;         /* 0: one scan launch per window + state kernels; 1: persistent single launch */
    uint32_t rows_coarse_passed; /* persistent engine: rows its all-f32 tier could not decide (scored again by the f32-log tier) */
    /* the LAST scan launch on its own (a selection that starts with a head phase has two very different launches:
     * the event-dense head of the stream on the head CUs, then the full grid): its duration when timing is on, and the
     * rows it scored; the launches in front of it are scan_ms - scan_ms_last and rows_scored - rows_scored_last */
    double scan_ms_last;
    uint64_t rows_scored_last;
    double arbiter_ms;       /* host time spent in tie arbitration (n_arbitrated calls), wall clock */
} dvs_select_summary;

int dvs_select_run(dvs_ctx *ctx, const dvs_matrix *m, const uint32_t *order,
                   const uint32_t *labels, uint64_t npos,
                   const dvs_select_params *params, dvs_select **out);
void dvs_select_destroy(dvs_select *s);
int dvs_select_get_summary(dvs_ctx *ctx, const dvs_select *s, dvs_select_summary *out);
/* members in set order (SummedRecords::get_raw_kseqs, src/records.rs:175-180):
 * stream position, label, delta_jsd, entropy and (if freqs != NULL) the
 * size x nbins frequency rows.  Any output pointer may be NULL. */
int dvs_select_get_members(dvs_ctx *ctx, const dvs_select *s, uint64_t *positions,
                           uint32_t *labels, double *delta_jsd, double *entropy,
                           double *freqs);
/* the same members into caller-provided DEVICE buffers, enqueued on the ctx stream:
 * d_rows[cap_rows x nbins] frequency rows, d_meta[cap_rows x 2] = (stream position, 1.0);
 * rows beyond the set's size are zeroed with meta (0, 0). */
int dvs_select_gather_members(dvs_ctx *ctx, const dvs_select *s, double *d_rows, double *d_meta,
                              uint32_t cap_rows);
/* SummedRecords::delta_jsd (src/records.rs:70-84) for every row of `queries`
 * against the set: 0.0 when qlabels[i] is a member's label, NaN for a row
 * without valid k-mers (the python layer raises, src/records_py.rs:111-120). */
int dvs_select_delta_jsd(dvs_ctx *ctx, const dvs_select *s, const dvs_matrix *queries,
                         const uint32_t *qlabels, double *out);
/* ---- stepwise driving: rows sharded over ranks, set state replicated -------- *
 * The exact multi-GPU form of select_nmost_divergent / select_max_divergent: every
 * rank holds the rows of its share of the stream (order[p] = DVS_ROW_REMOTE for the others; the
 * first n_seed rows replicated everywhere) and an identical copy of the set.  ONE collective per
 * greedy step, everything enqueued on the ctx stream with no host sync:
 *   dvs_select_step_pack   scan this rank's rows of the current window and pack its first local
 *                          event into d_slot[nbins + 2]: position (as a double; < 0: none), the
 *                          row's entropy, the candidate's frequency row
 *   [host framework: all_gather of the slots -- RCCL; world x (nbins + 2) doubles, 8 x 32 KB at k=6]
 *   dvs_select_step_apply  the earliest event among the gathered slots d_all[world][nbins + 2] is the
 *                          step's event (a position is scored by one rank only): resolve +
 *                          leave-one-out + finalize with that candidate, identical arithmetic on
 *                          every rank so the replicas stay bit-identical
 * dvs_select_step_poll syncs and returns the status (0 running, 1 done) / cursor; it must be called at least every 16
 * steps (it also drains the device-side ring of accepted rows, once that is half full, into the host-side log the
 * arbiter replays from, so that log has no cap: dvs_select_step_apply returns DVS_ERR_VALUE when more steps than half
 * the ring holds -- 16 at least -- have passed since the last poll).  When the engine has
 * stopped at a decision inside the rounding band (src/records.rs:86-92,231,246-249) the poll runs the host
 * tie arbiter -- on every rank alike: same seeds, same row log of accepted candidates, same pending
 * candidate, hence the same verdict with no exchange -- and re-enters the step kernels; steps enqueued in
 * between were no-ops.  (DVS_SELECT_NO_ARBITER turns that into an error.) */
int dvs_select_step_pack(dvs_ctx *ctx, dvs_select *s, double *d_slot);
int dvs_select_step_apply(dvs_ctx *ctx, dvs_select *s, const double *d_all, uint32_t world);
int dvs_select_step_poll(dvs_ctx *ctx, dvs_select *s, uint32_t *status, uint64_t *cursor);
/* A look at the status that neither syncs nor drains the queue: *status is the engine's status behind the apply launch
 * `lag` launches before the last one enqueued (0 running while there is none that far back), read from a history the step
 * kernel keeps in pinned host memory.  Every rank sees the same word for the same launch, so a driver that peeks at the same
 * step counts on every rank takes the same decisions (the number of collectives must not depend on timing).  When *status
 * is not 0, or *must_poll is set (half the accepted rows' ring would fill before the next look), call dvs_select_step_poll.
 * DVS_ERR_UNSUPPORTED: this selection keeps no history (not the fast step): poll instead.  (No counterpart in the reference:
 * its greedy loop is one thread, src/records.rs:311-342.) */
int dvs_select_step_peek(dvs_ctx *ctx, dvs_select *s, uint32_t lag, uint32_t *status, int *must_poll);

/* when on, every scan launch is bracketed by a pair of HIP events recorded on the
 * ctx stream (no extra host sync); they are read once the selection has finished
 * and summed into dvs_select_summary.scan_ms / scan_launches */
int dvs_ctx_set_timing(dvs_ctx *ctx, int on);
/* The library's environment switches (DVS_*: measurement aids and escape hatches) are
 * read once, when a context is created; nothing on a per-call path looks at the environment.  This
 * re-reads them for an existing context (tests and A/B measurements that flip a switch in between). */
int dvs_ctx_refresh_knobs(dvs_ctx *ctx);
/* measurement aid: average duration of ONE scan_kernel launch over every streamed row of
 * the selection's stream against its current state, with an unreachable threshold (no
 * events, state untouched): the steady-state streaming rate of the scan arithmetic */
int dvs_select_bench_scan(dvs_ctx *ctx, const dvs_select *s, int repeats, double *ms_out,
                          uint64_t *rows_out);
/* diagnostic: max |v_log_f32(m) - log2(m)| over every f32 m in [0.5, 1), the
 * hardware term of the scan kernel's fast-tier error bound (select.hip FAST_BAND) */
int dvs_selftest_fast_log2(dvs_ctx *ctx, double *max_abs_err);
/* diagnostic: max |log2_acc(x) - log2(x)| / max(1, |log2 x|) of the f64 log2 the precise
 * evaluations use (select_dev.h), over 2^17 mantissas x 80 binades */
int dvs_selftest_log2_acc(dvs_ctx *ctx, double *max_rel_err);
/* diagnostic: the hardware term of the persistent engine's coarse (f32) tier: max over every
 * f32 y in [2^-101, 2) of |v_log_f32(y) - log2 y| in units of 2^-23 max(1, |log2 y|) */
int dvs_selftest_log2_f32(dvs_ctx *ctx, double *max_ulps);
/* diagnostic: count / total by the fma sequence the selection kernels use in place of the f64
 * division (select_dev.h exact_div_u32) against the division itself: every count <= total <=
 * 8192 and 2^32 random pairs; reports the number of mismatches (must be 0) */
int dvs_selftest_exact_div(dvs_ctx *ctx, uint64_t *mismatches);
/* The persistent engine's hand-over words on their own (csrc/persist.hip): `rounds` synthetic windows
 * -- arrival records, hints, listed candidates, the gathering block's release, a use of the never-cleared
 * leave-one-out accumulators -- with contributions every workgroup can recompute and pseudo-random pauses in front of
 * every step; one workgroup per CU.  *failures = workgroup-rounds in which a word read was not the word it must be
 * (+ 2^32 when a bounded spin ran out). */
int dvs_selftest_handover(dvs_ctx *ctx, uint32_t rounds, uint64_t *failures);

/* ---- mash ----------------------------------------------------------------- *
 * dvs_mash_sketch replaces _dvs.mash_sketch (src/distance.rs:136-182) for a
 * batch: sketches_out is nseq x sketch_size (ascending, first lens_out[i]
 * entries valid).  dvs_mash_distances (sketch i at sketches + i * sketch_stride;
 * sketch_size is the value the distance formula uses) replaces diverse_seq/distance.py
 * mash_distance (:230-291) over the pairs (i, j<i) for i = row_start,
 * row_start+row_stride, ... (compute_mash_chunk_distances,
 * diverse_seq/cluster.py:640-644); dist is nseq x nseq row-major, only the
 * visited lower-triangle cells (and their mirror when symmetric != 0) are
 * written.  DVS_ERR_ZERODIV when a visited pair has two empty sketches. */
int dvs_mash_sketch(dvs_ctx *ctx, const uint8_t *seqs, int seqs_on_device,
                    const uint64_t *offsets, uint32_t nseq, uint32_t k,
                    uint32_t sketch_size, uint32_t num_states, int mash_canonical,
                    uint32_t *sketches_out, uint32_t *lens_out);
int dvs_mash_distances(dvs_ctx *ctx, const uint32_t *sketches, uint32_t sketch_stride,
                       const uint32_t *lens, uint32_t nseq, uint32_t k, uint32_t sketch_size,
                       uint32_t row_start, uint32_t row_stride, int symmetric,
                       double *dist);
/* The two stages of ctree (diverse_seq/cluster.py:241-297: sketches, then the N x N distances) with
 * the sketches left in HBM between them: dvs_sketches_build = dvs_mash_sketch without the copy to
 * the host, dvs_sketches_distances = dvs_mash_distances on that handle, dvs_sketches_get copies
 * sketches (may be NULL) and lengths out (what _dvs.mash_sketch returns). */
typedef struct dvs_sketches dvs_sketches;
int dvs_sketches_build(dvs_ctx *ctx, const uint8_t *seqs, int seqs_on_device, const uint64_t *offsets,
                       uint32_t nseq, uint32_t k, uint32_t sketch_size, uint32_t num_states,
                       int mash_canonical, dvs_sketches **out);
/* the same from a packed batch (k <= 32) and from an ingested batch (packed or not) */
int dvs_sketches_build_packed(dvs_ctx *ctx, const dvs_packed *p, const uint64_t *offsets, uint32_t nseq,
                              uint32_t k, uint32_t sketch_size, int mash_canonical, dvs_sketches **out);
int dvs_sketches_build_from_seqbatch(dvs_ctx *ctx, const dvs_seqbatch *b, uint32_t k, uint32_t sketch_size,
                                     uint32_t num_states, int mash_canonical, dvs_sketches **out);
void dvs_sketches_destroy(dvs_sketches *sk);
int dvs_sketches_get(dvs_ctx *ctx, const dvs_sketches *sk, uint32_t *sketches_out, uint32_t *lens_out);
const void *dvs_sketches_dev(const dvs_sketches *sk);      /* uint32 [nseq x sketch_size] in HBM */
const void *dvs_sketches_dev_lens(const dvs_sketches *sk); /* uint32 [nseq] */
int dvs_sketches_distances(dvs_ctx *ctx, const dvs_sketches *sk, uint32_t k, uint32_t sketch_size,
                           uint32_t row_start, uint32_t row_stride, int symmetric, double *dist);
/* The distance stage of the sharded ctree (`dvs_par_ctree`, diverse_seq/cluster.py:607-644: worker g takes rows g,
 * g + G, ... of the lower triangle of ALL sketches) with everything left in HBM: a rank's own sketches are copied
 * into its send buffer (dvs_sketches_copy_to_device: rows dst_stride words apart), the gathered N sketches are
 * wrapped without a copy (dvs_sketches_from_device: the caller's buffers, which must outlive the handle) and the
 * strided rows are written into a device matrix (dvs_sketches_distances_device: enqueued on the context's stream and
 * not waited for; *d_zerodiv is set where dvs_sketches_distances would return DVS_ERR_ZERODIV). */
int dvs_sketches_from_device(dvs_ctx *ctx, const uint32_t *d_sketches, const uint32_t *d_lens, uint32_t nseq,
                             uint32_t stride, dvs_sketches **out);
int dvs_sketches_copy_to_device(dvs_ctx *ctx, const dvs_sketches *sk, uint32_t *d_dst, uint32_t dst_stride,
                                uint32_t *d_dst_lens);
int dvs_sketches_distances_device(dvs_ctx *ctx, const dvs_sketches *sk, uint32_t k, uint32_t sketch_size,
                                  uint32_t row_start, uint32_t row_stride, int symmetric, double *d_dist,
                                  uint32_t *d_zerodiv);
/* euclidean_distances (diverse_seq/distance.py:294-336): ||f_i - f_j||_2 over
 * the rows of m, full symmetric nrows x nrows matrix */
int dvs_euclidean_distances(dvs_ctx *ctx, const dvs_matrix *m, double *dist);

#ifdef __cplusplus
}
#endif
#endif /* DVS_HIP_H */
