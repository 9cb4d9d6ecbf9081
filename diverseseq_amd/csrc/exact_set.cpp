// Host tie arbiter of the selection engine.
//
// The device scores candidates with tree reductions and its own log2; the
// reference sums 4^k terms sequentially with libm's log2 (src/record.rs:92-98).
// Both are the same number up to rounding noise of ~ B * eps * H, so a decision
// whose margin is inside that band (select.hip: sel_band) cannot be attributed
// on the device.  Those -- and only those -- decisions come here: the accepted
// events logged by the device are replayed in the reference's exact f64
// operation order (this file restates src/records.rs:27-147,153-189,220-286 for
// that purpose), the open decision is evaluated exactly, and the outcome is
// written back as a forced decision.  Degenerate inputs (identical sequences,
// k = 1 toy sets) are where this runs; on the benchmark workloads it never does
// (dvs_select_summary.n_arbitrated counts it).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <memory>
#include <thread>

#include "select.h"

#include <cstddef>

unsigned dvs_host_threads();  // pack.hip: host cores this process may use (cgroup quota respected), at most 16

namespace {

// A frequency row: written once when it is fetched, read-only afterwards -- copies (the reference's clone,
// records.rs:182-189, copies every record of the set) share the one buffer.
class RowVec {
    std::shared_ptr<std::vector<double>> v;

   public:
    void resize(size_t n) { v = std::make_shared<std::vector<double>>(n); }
    double *data() { return v->data(); }
    const double *data() const { return v->data(); }
    double &operator[](size_t i) { return (*v)[i]; }
    const double &operator[](size_t i) const { return (*v)[i]; }
};

struct ExactRow {
    uint64_t pos = 0;
    uint32_t label = 0;
    RowVec f;
    double H = 0.0;
    double delta = 0.0;
};

class ExactSet {
   public:
    size_t B = 0;
    std::vector<ExactRow> recs;
    std::vector<double> S, work;
    double sumH = 0.0, total_jsd = 0.0;
    uint32_t lowest = 0;
    std::string err;

    // src/record.rs:86-106
    bool entropy(const double *f, size_t n, double &out) { return entropy_into(f, n, out, err); }
    static bool entropy_into(const double *f, size_t n, double &out, std::string &err) {
        if (n == 0) {
            err = "cannot calculate entropy as frequency vector empty";
            return false;
        }
        double e = 0.0, tot = 0.0;
        for (size_t i = 0; i < n; i++) {
            const double x = f[i];
            if (x == 0.0) continue;
            e += -x * std::log2(x);
            tot += x;
        }
        if (std::fabs(tot - 1.0) > double(n) * DVS_EPS) {
            char buf[128];
            snprintf(buf, sizeof buf, "cannot calculate entropy as frequency vector total %.17g!=1.0", tot);
            err = buf;
            return false;
        }
        out = e;
        return true;
    }

    bool contains(uint32_t label) const {
        if (label == 0xFFFFFFFFu) return false;  // (no identity: a row that lives on another rank)
        for (const ExactRow &r : recs)
            if (r.label == label) return true;
        return false;
    }

    // src/records.rs:220-252 (+ updated_mean_freqs :276-286).  Every member's pass is the reference's,
    // term by term in its order; the members are independent of one another, so large sets spread them
    // over host threads (a genome-scale `max` set: 100+ members x 4^k libm logarithms per arbitration).
    // The argmin is taken afterwards in member order (strict '<' from 1e6: the earliest wins ties).
    bool lowest_index() {
        const double div = double(recs.size()) - 1.0;
        if (div <= 0.0) {
            err = "must have > 1 KmerSeq";
            return false;
        }
        const size_t n = recs.size();
        std::vector<std::string> errs(n);
        auto member = [&](size_t i, std::vector<double> &buf) {
            ExactRow &r = recs[i];
            const double mean_entropy = (sumH - r.H) / div;
            for (size_t j = 0; j < B; j++) {
                buf[j] = (S[j] - r.f[j]) / div;
                if (buf[j] <= DVS_EPS) buf[j] = 0.0;
            }
            double eom;
            std::string e;
            if (!entropy_into(buf.data(), B, eom, e)) {
                errs[i] = e;
                return;
            }
            r.delta = total_jsd - (eom - mean_entropy);
        };
        const size_t nthr = (n * B >= (size_t(1) << 18)) ? std::min<size_t>(n, size_t(dvs_host_threads())) : 1;
        if (nthr <= 1) {
            for (size_t i = 0; i < n; i++) member(i, work);
        } else {
            std::vector<std::thread> pool;
            for (size_t t = 0; t < nthr; t++)
                pool.emplace_back([&, t]() {
                    std::vector<double> buf(B);
                    for (size_t i = t; i < n; i += nthr) member(i, buf);
                });
            for (std::thread &th : pool) th.join();
        }
        double min_delta = 1e6;
        uint32_t low = 0;
        for (size_t i = 0; i < n; i++) {
            if (!errs[i].empty()) {  // (the reference would have panicked at the first such member)
                err = errs[i];
                return false;
            }
            if (recs[i].delta < min_delta) {
                min_delta = recs[i].delta;
                low = uint32_t(i);
            }
        }
        lowest = low;
        return true;
    }

    // src/records.rs:27-68
    bool init(std::vector<ExactRow> &&rows, size_t nbins, bool with_lowest = true) {
        B = nbins;
        recs = std::move(rows);
        if (recs.empty()) {
            err = "records cannot be empty";
            return false;
        }
        S.assign(B, 0.0);
        work.assign(B, 0.0);
        sumH = 0.0;
        for (const ExactRow &r : recs) {
            for (size_t j = 0; j < B; j++) S[j] += r.f[j];
            sumH += r.H;
        }
        const double n = double(recs.size());
        for (size_t j = 0; j < B; j++) work[j] = S[j] / n;
        double eom;
        if (!entropy(work.data(), B, eom)) return false;
        total_jsd = eom - sumH / n;
        return with_lowest ? lowest_index() : true;
    }

    // src/records.rs:70-84
    bool delta_jsd(const ExactRow &c, double &out) {
        if (contains(c.label)) {
            out = 0.0;
            return true;
        }
        const ExactRow &low = recs[lowest];
        const double n = double(recs.size());
        const double mean_entropy = (sumH - low.H + c.H) / n;
        for (size_t j = 0; j < B; j++) work[j] = (S[j] - low.f[j] + c.f[j]) / n;
        double eom;
        if (!entropy(work.data(), B, eom)) return false;
        out = eom - mean_entropy;
        return true;
    }

    // src/records.rs:86-92
    bool increases(const ExactRow &c, bool &out) {
        if (contains(c.label)) {
            out = false;
            return true;
        }
        double jsd;
        if (!delta_jsd(c, jsd)) return false;
        out = jsd > total_jsd + DVS_EPS;
        return true;
    }

    // src/records.rs:120-147
    bool push(ExactRow &&c) {
        if (contains(c.label)) return true;
        sumH += c.H;
        for (size_t j = 0; j < B; j++) S[j] += c.f[j];
        recs.push_back(std::move(c));
        const double n = double(recs.size());
        for (size_t j = 0; j < B; j++) work[j] = S[j] / n;
        double eom;
        if (!entropy(work.data(), B, eom)) return false;
        total_jsd = eom - sumH / n;
        return lowest_index();
    }

    // src/records.rs:94-118
    bool replace_lowest(ExactRow &&c) {
        if (contains(c.label)) return true;
        ExactRow old = std::move(recs[lowest]);
        recs.erase(recs.begin() + lowest);
        sumH -= old.H;
        for (size_t j = 0; j < B; j++) {
            S[j] -= old.f[j];
            if (S[j] <= DVS_EPS) S[j] = 0.0;
        }
        return push(std::move(c));
    }

    // src/records.rs:182-189: clone re-runs new().  for_push: the clone is pushed to at once
    // (records.rs:438-440) and push's own leave-one-out pass overwrites every delta and the argmin, so
    // the clone's pass -- n x 4^k logarithms whose results nobody reads -- is left out.
    bool clone_from(const ExactSet &o, bool for_push = false) {
        std::vector<ExactRow> rows = o.recs;
        for (ExactRow &r : rows) r.delta = 0.0;
        return init(std::move(rows), o.B, !for_push);
    }

    // src/records.rs:156-172
    double mean_delta() const {
        double s = 0.0;
        for (const ExactRow &r : recs) s += r.delta;
        return s / double(recs.size());
    }
    double std_delta() const {
        const double m = mean_delta();
        double s = 0.0;
        for (const ExactRow &r : recs) {
            const double d = r.delta - m;
            s += d * d;
        }
        return std::sqrt(s / (double(recs.size()) - 1.0));
    }
    double cov_delta() const { return std_delta() / mean_delta(); }
};

struct Arbiter {
    ExactSet set;
    bool built = false;
    uint32_t replayed = 0;  // event-log entries already applied
};

// frequency row of stream position p, exactly as the reference builds it
int fetch_row(dvs_ctx *ctx, const dvs_select *s, uint64_t p, ExactSet &scratch, ExactRow &out) {
    const dvs_matrix *m = s->mat;
    const uint64_t B = m->nbins;
    const uint32_t row = s->h_order.empty() ? uint32_t(p) : s->h_order[p];
    out.pos = p;
    out.label = s->h_labels.empty() ? row : s->h_labels[p];
    out.f.resize(B);
    if (m->kind != 1) {
        std::vector<uint32_t> c(B);
        uint32_t tot = 0;
        if (m->kind == 2) {
            std::vector<uint16_t> c16(B);
            DVS_HIP(ctx, hipMemcpy(c16.data(), m->d_counts16 + uint64_t(row) * B, B * 2, hipMemcpyDeviceToHost));
            for (uint64_t i = 0; i < B; i++) c[i] = c16[i];
        } else {
            DVS_HIP(ctx, hipMemcpy(c.data(), m->d_counts + uint64_t(row) * B, B * 4, hipMemcpyDeviceToHost));
        }
        DVS_HIP(ctx, hipMemcpy(&tot, m->d_totals + row, 4, hipMemcpyDeviceToHost));
        const double total = double(tot);  // record.rs:135-139
        for (uint64_t i = 0; i < B; i++) out.f[i] = double(c[i]) / total;
    } else {
        DVS_HIP(ctx, hipMemcpy(out.f.data(), m->d_freqs + uint64_t(row) * B, B * 8, hipMemcpyDeviceToHost));
    }
    if (!scratch.entropy(out.f.data(), B, out.H))  // KmerSeq::new, record.rs:157-159
        return dvs_set_error(ctx, DVS_ERR_VALUE, "%s", scratch.err.c_str());
    return DVS_OK;
}

// many rows at once (the initial set: a genome-scale `max` starts from 100 seeds -- row by row that is 300
// blocking copies and 100 x 4^k libm logarithms on one thread): one copy per run of consecutive matrix rows,
// then the divisions and the entropies on host threads.  Same values as fetch_row, row for row.
int fetch_rows(dvs_ctx *ctx, const dvs_select *s, const std::vector<uint64_t> &positions, ExactSet &scratch,
               std::vector<ExactRow> &out) {
    const dvs_matrix *m = s->mat;
    const uint64_t B = m->nbins;
    const size_t n = positions.size();
    out.assign(n, ExactRow());
    if (m->kind == 1 || n < 8) {
        for (size_t i = 0; i < n; i++) {
            int rc = fetch_row(ctx, s, positions[i], scratch, out[i]);
            if (rc) return rc;
        }
        return DVS_OK;
    }
    std::vector<uint32_t> rows(n), tots(n);
    for (size_t i = 0; i < n; i++) rows[i] = s->h_order.empty() ? uint32_t(positions[i]) : s->h_order[positions[i]];
    const size_t esz = m->kind == 2 ? 2 : 4;
    std::vector<unsigned char> raw(n * B * esz);
    for (size_t a = 0; a < n;) {
        size_t b = a + 1;
        while (b < n && rows[b] == rows[b - 1] + 1) b++;
        const void *src = m->kind == 2 ? static_cast<const void *>(m->d_counts16 + uint64_t(rows[a]) * B)
                                       : static_cast<const void *>(m->d_counts + uint64_t(rows[a]) * B);
        DVS_HIP(ctx, hipMemcpy(raw.data() + a * B * esz, src, (b - a) * B * esz, hipMemcpyDeviceToHost));
        DVS_HIP(ctx, hipMemcpy(tots.data() + a, m->d_totals + rows[a], (b - a) * 4, hipMemcpyDeviceToHost));
        a = b;
    }
    std::vector<std::string> errs(n);
    auto one = [&](size_t i) {
        ExactRow &r = out[i];
        r.pos = positions[i];
        r.label = s->h_labels.empty() ? rows[i] : s->h_labels[positions[i]];
        r.f.resize(B);
        const double total = double(tots[i]);  // record.rs:135-139
        if (esz == 2) {
            const uint16_t *c = reinterpret_cast<const uint16_t *>(raw.data()) + i * B;
            for (uint64_t j = 0; j < B; j++) r.f[j] = double(c[j]) / total;
        } else {
            const uint32_t *c = reinterpret_cast<const uint32_t *>(raw.data()) + i * B;
            for (uint64_t j = 0; j < B; j++) r.f[j] = double(c[j]) / total;
        }
        (void)ExactSet::entropy_into(r.f.data(), B, r.H, errs[i]);  // KmerSeq::new, record.rs:157-159
    };
    const size_t nthr = std::min<size_t>(n, size_t(dvs_host_threads()));
    std::vector<std::thread> pool;
    for (size_t t = 0; t < nthr; t++)
        pool.emplace_back([&, t]() {
            for (size_t i = t; i < n; i += nthr) one(i);
        });
    for (std::thread &th : pool) th.join();
    for (size_t i = 0; i < n; i++)
        if (!errs[i].empty()) return dvs_set_error(ctx, DVS_ERR_VALUE, "%s", errs[i].c_str());
    return DVS_OK;
}

// the same from a frequency row already on the device (stepwise selections: the row log, the pending
// candidate in d.cand, a tentative member's row) -- the row itself may live on another rank
int fetch_dev_row(dvs_ctx *ctx, const dvs_select *s, uint64_t p, const double *d_row, ExactSet &scratch, ExactRow &out) {
    const uint64_t B = s->mat->nbins;
    const uint32_t row = s->h_order.empty() ? uint32_t(p) : s->h_order[p];
    out.pos = p;
    out.label = row == DVS_ROW_REMOTE ? 0xFFFFFFFFu : (s->h_labels.empty() ? row : s->h_labels[p]);
    out.f.resize(B);
    DVS_HIP(ctx, hipMemcpy(out.f.data(), d_row, B * 8, hipMemcpyDeviceToHost));
    if (!scratch.entropy(out.f.data(), B, out.H)) return dvs_set_error(ctx, DVS_ERR_VALUE, "%s", scratch.err.c_str());
    return DVS_OK;
}

// ... and from a frequency row in host memory (the stepwise selections' log of accepted rows)
int fetch_host_row(dvs_ctx *ctx, const dvs_select *s, uint64_t p, const double *h_row, ExactSet &scratch, ExactRow &out) {
    const uint64_t B = s->mat->nbins;
    const uint32_t row = s->h_order.empty() ? uint32_t(p) : s->h_order[p];
    out.pos = p;
    out.label = row == DVS_ROW_REMOTE ? 0xFFFFFFFFu : (s->h_labels.empty() ? row : s->h_labels[p]);
    out.f.resize(B);
    std::memcpy(out.f.data(), h_row, B * sizeof(double));
    if (!scratch.entropy(out.f.data(), B, out.H)) return dvs_set_error(ctx, DVS_ERR_VALUE, "%s", scratch.err.c_str());
    return DVS_OK;
}

// The decision goes to the control block with blocking copies AND a device synchronisation behind them: a
// small copy from pageable memory returns once the bytes are staged, not once they are in device memory,
// and the context's stream is a non-blocking one -- without the wait the kernels queued next saw, one time in
// a few hundred, the old ARBITER status, returned at once, and the host arbitrated the same decision a
// second time (same answer, twice the work; found by repeating one selection 600 times,
// scripts/micro/c4_repeat.py).
int write_forced(dvs_ctx *ctx, dvs_select *s, uint32_t forced, uint32_t forced_lowest) {
    // status, arb_stage, forced and forced_lowest are neighbours in the control block: one copy on the selection's
    // own stream from the pinned mirror (the host has just read it; arb_stage goes back as it came), and a wait
    // for THAT stream -- the kernels launched behind this call see the decision, and nothing else on the device
    // (a histogram on the side stream, another stream's collective) is waited for
    static_assert(offsetof(SelCtl, forced_lowest) - offsetof(SelCtl, status) == 12, "status .. forced_lowest are contiguous");
    SelCtl *d = s->dev.ctl;
    SelCtl *h = s->h_ctl;
    h->status = SEL_RUN;
    h->forced = forced;
    h->forced_lowest = forced_lowest;
    DVS_HIP(ctx, hipMemcpyAsync(&d->status, &h->status, 16, hipMemcpyHostToDevice, ctx->stream));
    DVS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return DVS_OK;
}

}  // namespace

int dvs_select_arbitrate(dvs_ctx *ctx, dvs_select *s) {
    const SelCtl &c = *s->h_ctl;
    const uint64_t B = s->dev.B;
    if (!s->arbiter) s->arbiter = new Arbiter();
    Arbiter &a = *static_cast<Arbiter *>(s->arbiter);
    ExactSet &set = a.set;
    set.B = B;
    if (set.work.size() != B) set.work.assign(B, 0.0);
    int rc;
    const bool stepwise = (s->params.flags & DVS_SELECT_STEPWISE) != 0;
    if (!a.built) {  // SummedRecords::new over the usable seeds (records.rs:288-308)
        std::vector<ExactRow> rows;
        std::vector<uint64_t> seed_pos(s->seed_positions.begin(), s->seed_positions.end());
        if ((rc = fetch_rows(ctx, s, seed_pos, set, rows))) return rc;
        if (!set.init(std::move(rows), B)) return dvs_set_error(ctx, DVS_ERR_VALUE, "%s", set.err.c_str());
        a.built = true;
    }
    // replay what the device has committed since the last arbitration
    const uint32_t n_logged = c.n_logged;
    if (n_logged > a.replayed) {
        const uint32_t cnt = n_logged - a.replayed;
        std::vector<unsigned long long> pos(cnt);
        std::vector<uint32_t> kind(cnt);
        DVS_HIP(ctx, hipMemcpy(pos.data(), s->dev.evlog_pos + a.replayed, size_t(cnt) * 8, hipMemcpyDeviceToHost));
        DVS_HIP(ctx, hipMemcpy(kind.data(), s->dev.evlog_kind + a.replayed, size_t(cnt) * 4, hipMemcpyDeviceToHost));
        for (uint32_t i = 0; i < cnt; i++) {
            ExactRow r;
            if (stepwise) {
                const uint32_t at = a.replayed + i;
                // (the whole log is on the host: dvs_select_step_poll drained the device ring before it called here)
                if (uint64_t(at) >= s->rowlog_have)
                    return dvs_set_error(ctx, DVS_ERR_RUNTIME, "tie arbitration in the stepwise mode: event %u is not in the "
                                         "accepted rows' log (%llu rows)", at, (unsigned long long)s->rowlog_have);
                rc = fetch_host_row(ctx, s, pos[i], s->h_rowlog.data() + uint64_t(at) * B, set, r);
            } else {
                rc = fetch_row(ctx, s, pos[i], set, r);
            }
            if (rc) return rc;
            bool ok;
            if (kind[i] == 1) {
                ok = set.replace_lowest(std::move(r));
            } else {  // records.rs:438-450: the kept object is clone + push
                ExactSet grown;
                ok = grown.clone_from(set, true) && grown.push(std::move(r));
                if (ok) set = std::move(grown);
                else set.err = grown.err;
            }
            if (!ok) return dvs_set_error(ctx, DVS_ERR_VALUE, "%s", set.err.c_str());
        }
        a.replayed = n_logged;
    }
    s->n_arbitrated++;

    if (c.arb_stage == ARB_RESOLVE) {
        ExactRow cand;
        rc = stepwise ? fetch_dev_row(ctx, s, c.arb_pos, s->dev.cand, set, cand) : fetch_row(ctx, s, c.arb_pos, set, cand);
        if (rc) return rc;
        bool inc;
        if (!set.increases(cand, inc)) return dvs_set_error(ctx, DVS_ERR_VALUE, "%s", set.err.c_str());
        return write_forced(ctx, s, inc ? FORCE_ACCEPT : FORCE_REJECT, 0xFFFFFFFFu);
    }
    // ARB_FINALIZE
    if (c.ev_kind == 1) {
        // the set already changed (its event is in the log and was replayed above, or it is
        // the initial set): only the argmin is open
        if (set.recs.size() != c.ev_n)
            return dvs_set_error(ctx, DVS_ERR_RUNTIME, "arbiter out of sync: %zu members vs %u", set.recs.size(), c.ev_n);
        return write_forced(ctx, s, FORCE_COMMIT, set.lowest);
    }
    // tentative push of the candidate at arb_pos (records.rs:438-450)
    ExactRow cand;
    rc = stepwise ? fetch_dev_row(ctx, s, c.arb_pos, s->dev.M + uint64_t(c.ev_n - 1) * B, set, cand)
                  : fetch_row(ctx, s, c.arb_pos, set, cand);
    if (rc) return rc;
    ExactSet grown;
    if (!grown.clone_from(set, true) || !grown.push(std::move(cand)))
        return dvs_set_error(ctx, DVS_ERR_VALUE, "%s", grown.err.c_str());
    const bool better = (c.stat == DVS_STAT_STDEV) ? (grown.std_delta() > set.std_delta())
                                                   : (grown.cov_delta() > set.cov_delta());
    return write_forced(ctx, s, better ? FORCE_COMMIT : FORCE_ROLLBACK, better ? grown.lowest : 0xFFFFFFFFu);
}

void dvs_select_arbiter_free(dvs_select *s) {
    if (s && s->arbiter) {
        delete static_cast<Arbiter *>(s->arbiter);
        s->arbiter = nullptr;
    }
}
