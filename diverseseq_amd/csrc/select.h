// Shared between select.hip (device engine) and exact_set.cpp (host tie arbiter).
#pragma once

#include "dvs_internal.h"

#define SEL_NONE 0xFFFFFFFFFFFFFFFFull
#define SEL_SEEDS_AT 512  // byte offset of the seed list inside the 4 KiB control allocation (and its pinned mirror)

enum : uint32_t { SEL_RUN = 0, SEL_DONE = 1, SEL_ARBITER = 2, SEL_ERROR = 3,
                  SEL_NEED_SETUP = 4 };  // a seeded persistent launch left the initial set to the set-up kernels
enum : uint32_t { ARB_RESOLVE = 1, ARB_FINALIZE = 2 };
enum : uint32_t { FORCE_NONE = 0, FORCE_ACCEPT = 1, FORCE_REJECT = 2, FORCE_COMMIT = 3, FORCE_ROLLBACK = 4 };

// Device-resident control block: the engine's cursor, the pending event and the
// scalar part of the SummedRecords state (src/records.rs:10-24).
struct SelCtl {
    unsigned long long cursor;     // next stream position to score
    unsigned long long npos;       // stream length
    unsigned long long event_pos;  // first position of the window whose score clears thr - band
    unsigned long long arb_pos;
    unsigned long long rows_scored, rows_rechecked;
    unsigned long long rows_coarse_passed;  // rows the persistent engine's COARSE tier handed to the FAST tier
    unsigned long long rows_before_launch;  // rows_scored as the latest persistent launch found it (its own share = the difference)
    uint32_t window, window_min, window_max;
    uint32_t status, arb_stage, forced, forced_lowest;
    uint32_t size, lowest, mode, max_size, stat;
    uint32_t ev_kind;   // 0 none, 1 set changed (replace / initial), 2 tentative push (MODE_MAX)
    uint32_t ev_n;      // members taking part in the pending leave-one-out pass
    uint32_t ev_risky;  // a sum-to-one check is too close to call on the device
    uint32_t s_is_resum;  // S still equals the members' rows added up in member order (only pushes so far)
    uint32_t mb_rows;     // MODE_MAX batches: rows the batch pairs have moved the cursor over
    uint32_t mb_stuck;    // MODE_MAX batches: the row at the cursor is the ordinary iteration's (a push to keep, a call too close)
    uint32_t n_windows, n_events, n_accepts;
    uint32_t n_logged;  // entries of the event log (accepted set changes, for the arbiter)
    double total_jsd, sum_entropy;      // records.rs: total_jsd, summed_entropies
    double thr, band;                   // total_jsd + eps ; width of the undecidable zone
    double he_base;                     // summed_entropies - H(lowest)
    double mean_delta, std_delta, cov_delta;
    double t_total_jsd, t_sum_entropy;  // tentative (MODE_MAX push) values
    double last_jsd;
    double arb_H;   // entropy of the candidate a resolve stopped at (stepwise re-entry takes it from here)
    double wscale;  // next window = cursor * wscale / size (expected rows to the next accept ~ cursor / size)
    // why persistent launches ended early (DVS_PERSIST_DEBUG prints them): 0 replica full, 1 a sum check not
    // sure, 2 argmin of a tentative push too close, 3 stat comparison too close, 4 candidate test in band,
    // 5 argmin after a replace too close, 6 state not taken (s_is_resum / pending event)
    uint32_t why[8];
};

// Device pointers of one selection (passed to kernels by value).
struct SelDev {
    SelCtl *ctl = nullptr;
    uint64_t B = 0;
    uint32_t nlabels = 0;
    // candidate stream
    const uint32_t *order = nullptr;   // position -> matrix row (NULL: identity)
    const uint32_t *labels = nullptr;  // position -> identity label (NULL: row)
    const uint32_t *totals = nullptr;  // matrix row totals
    const double *rowH = nullptr;      // matrix row entropies
    // set state
    double *S = nullptr;       // summed_kfreqs
    double *Stmp = nullptr;    // tentative summed_kfreqs (MODE_MAX push)
    double *base = nullptr;    // (S - lowest) / size, what the scan adds candidates to
    double *cand = nullptr;    // frequency row of the candidate being resolved
    double *M = nullptr;       // member frequency rows, cap x B, by slot
    double *mH = nullptr;      // member entropies, by slot
    double *mDelta = nullptr;  // member delta_jsd, by member order
    double *dtmp = nullptr;    // leave-one-out results of the pending event, by member order
    double *dsum = nullptr;    // sum of each leave-one-out mean vector (tolerance check)
    uint32_t *mLabel = nullptr;
    unsigned long long *mPos = nullptr;  // stream position each member came from, by slot
    uint32_t *ord = nullptr;             // member order -> slot (Vec::remove / push order)
    uint8_t *inset = nullptr;            // label -> currently a member
    uint32_t *wg_rows = nullptr;         // rows actually read by each scan workgroup (last launch)
    unsigned long long *evlog_pos = nullptr;  // accepted events in order: stream position ...
    uint32_t *evlog_kind = nullptr;           // ... and kind (1 replace_lowest, 2 kept push)
    // stepwise / distributed use: every rank's gathered slot ([pos, H, row] x world) -- resolve picks
    // the earliest event itself and takes the candidate from there instead of the local matrix
    const double *gather_all = nullptr;
    uint32_t gather_world = 0;
    // ... and the frequency row of every accepted event, in event-log order, as a RING of rowlog_cap rows that the
    // host drains at every poll (dvs_select::h_rowlog holds the whole log): what the
    // tie arbiter replays from when the rows themselves live on other ranks
    double *rowlog = nullptr;
    uint32_t rowlog_cap = 0;
};

struct dvs_select {
    dvs_ctx *ctx = nullptr;
    dvs_select_params params{};
    const dvs_matrix *mat = nullptr;
    int mat_kind = 0;
    uint64_t npos = 0;
    uint32_t cap = 0;
    SelDev dev;
    SelCtl *h_ctl = nullptr;  // pinned mirror
    SelCtl ctl0{};            // the control block of the fresh selection (sel_seed)
    std::vector<uint32_t> h_order, h_labels;
    // labels that were all distinct: the engine runs label-free (label = stream position, the same
    // decisions) and the caller's values are put back on the way out
    std::vector<uint32_t> h_out_labels;
    std::vector<uint64_t> seed_positions;
    // launch geometry
    uint32_t scan_grid = 0, loo_grid = 0;
    size_t scan_lds = 0;
    bool base_in_lds = true;
    bool scan_hot = false;
    // persistent single-launch engine (persist.hip)
    bool persist = false;
    bool persist_fell_back = false;  // the persistent kernel gave up (not co-resident): multi-launch engine from the seeds
    uint32_t persist_grid = 0, persist_maxn = 0, persist_maxjobs = 0;
    bool persist_small = false;       // the SMALL instantiation: member count rows in every workgroup's LDS
    uint32_t persist_small_rows = 0;  // ... that many of them
    size_t persist_lds = 0;
    void *psync = nullptr;
    std::vector<unsigned char> h_psync;  // host image of the sync block (source of its upload)
    std::vector<unsigned char> h_psync_head;  // ... and the head phase's
    void *ppart = nullptr;
    double *d_mbres = nullptr;  // MODE_MAX batches (select.hip: max_batch_*): the jobs' results, 3 x MB_ROWS x mb_jw
    uint32_t mb_jw = 0;
    uint32_t mb_launched = 0;   // batch pairs launched (summary / tests)
    void *psync_head = nullptr, *ppart_head = nullptr;  // the head phase's own blocks
    bool persist_seeded = false;    // the next persistent launch starts from the seed positions (no set-up kernels ran)
    bool seeded_start = false;      // ... this selection began that way (sel_run_loop: a launch may hand the set-up back)
    bool head_prepared = false;     // ... and psync_head / ppart_head for the head phase
    bool persist_prepared = false;  // psync / ppart already hold a fresh image for the next full-grid launch
    bool used_side_streams = false;   // work of this selection was queued on ctx->stream_head / stream2 (sel_free waits)
    hipEvent_t ev_side_done = nullptr;  // the set-up kernels on the context's second stream have run
    void *d_seed_list = nullptr;  // the seed positions on the device (kept until the selection goes: two streams read it)
    bool seed_list_in_ctl = false;  // ... inside the control allocation, behind the control block (one upload for both)
    bool inset_clean = false;       // the label flags are still as they were cleared at allocation time
    hipStream_t setup_side = nullptr;  // the side stream the set-up goes to (NULL: the context's stream), sel_plan_setup_stream
    bool head_phase = false;           // ... and whether the engine starts with a head phase on the head CUs
    int batch = 16;
    // timing
    bool time_scan = false;
    std::vector<hipEvent_t> ev_pool;  // pairs (start, stop), one per scan launch
    size_t ev_used = 0;
    double scan_ms = 0.0;
    double scan_ms_last = 0.0;  // ... the last launch's own
    uint64_t scan_launches = 0;
    uint32_t n_arbitrated = 0;
    double arbiter_ms = 0.0;    // host wall clock inside dvs_select_arbitrate
    // stepwise selections, MODE_NMOST without labels: the fast step (select.hip fs_jobs_kernel / fs_step_kernel)
    bool fast_step = false;
    bool fs_need_scan = false;   // the next dvs_select_step_pack must scan first (selection start, behind an arbitration)
    unsigned long long *h_fshist = nullptr;  // pinned: the status word of every apply launch (FS_HIST of them, dvs_select_step_peek)
    unsigned long long fs_launches = 0;      // apply launches of the fast step so far
    unsigned long long fs_peek_floor = 0;    // ... of them, those enqueued before the last dvs_select_step_poll
    double *fs_slot = nullptr;   // the caller's slot of the last dvs_select_step_pack (the next step's kernel packs into it)
    double *fs_packed = nullptr; // ... and the slot the last fs_step_kernel launch packed (nullptr: none)
    double *d_jobres = nullptr;  // the leave-one-out jobs' sums of the current step
    void *d_fsync = nullptr;     // FsSync
    uint32_t fs_K = 1, fs_jobs = 0, fs_grid = 0;
    size_t fs_lds = 0;
    // stepwise selections: the accepted rows' log on the host (drained from the device ring by dvs_select_step_poll)
    std::vector<double> h_rowlog;
    uint64_t rowlog_have = 0;       // rows of it that are on the host
    uint32_t steps_since_poll = 0;  // dvs_select_step_apply calls since the last poll (bounded by the ring)
    void *arbiter = nullptr;  // ExactSet*, created on first use
};

// exact_set.cpp: resolve the decision the device stopped at (h_ctl is current)
// and write the forced outcome + status RUN back to the device control block.
int dvs_select_arbitrate(dvs_ctx *ctx, dvs_select *s);
void dvs_select_arbiter_free(dvs_select *s);

// persist.hip
int dvs_persist_setup(dvs_ctx *ctx, dvs_select *s);
int dvs_persist_launch(dvs_ctx *ctx, dvs_select *s);
int dvs_persist_launch_head(dvs_ctx *ctx, dvs_select *s, uint32_t grid, uint32_t stop_at, hipStream_t on);
int dvs_persist_prepare_main(dvs_ctx *ctx, dvs_select *s);
int dvs_persist_prepare_head(dvs_ctx *ctx, dvs_select *s, uint32_t stop_at, hipStream_t on);
size_t dvs_persist_dbg_offset(void);
size_t dvs_persist_trace_offset(void);  // 0 unless built with -DDVS_PERSIST_STAMPS
int dvs_persist_probe_id(void);         // the one interval a -DDVS_PROBE=k build measures (0: none)
