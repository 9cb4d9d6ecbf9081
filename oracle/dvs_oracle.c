/*
 * dvs_oracle.c -- CPU ORACLE for the DiverseSeq k-mer / delta-JSD / mash path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is a plain-C restatement of the
 * reference's (HuttleyLab/DiverseSeq) algorithm for the hot path.  It is
 * linked/loaded only by tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py -- never by the product (diverseseq_amd/).
 *
 * Parity status: PINNED for counting / entropy / set algebra / selectors
 * against the reference's own Rust unit-test constants (tests/golden/
 * rust_unit_vectors.json, checked in tests/test_oracle.py) and for
 * mash_distance against vectors captured from the reference's pure-Python
 * function (tests/golden/mash_distance_vectors.json).  UNPINNED for the
 * murmur-style hash / sketch values: the reference asserts no hash, sketch or
 * distance number anywhere; the contract there is
 * source fidelity with src/distance.rs:21-49.
 *
 * Every function cites the reference file:line it follows.  All f64
 * arithmetic keeps the reference's operation order; build with
 * -ffp-contract=off and without fast-math (see oracle/Makefile).
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define ORC_OK 0
#define ORC_ERR_PANIC 1    /* the reference would panic -> ValueError */
#define ORC_ERR_NOKMERS 2  /* "No valid k-mers" (record.rs:136-138) */
#define ORC_ERR_ALLOC 3

#define ORC_EPS 2.220446049250313e-16 /* f64::EPSILON */

static __thread char orc_msg[256];

const char *orc_last_error(void) { return orc_msg; }

static int orc_fail(int code, const char *fmt, double a, double b) {
    snprintf(orc_msg, sizeof orc_msg, fmt, a, b);
    return code;
}

/* ---------------------------------------------------------------- counting */

/* src/record.rs:10-15 coord_conversion_coeffs: ns^(k-1) ... ns^0 */
static void orc_coeffs(uint64_t *coeffs, unsigned ns, unsigned k) {
    uint64_t c = 1;
    for (unsigned i = 0; i < k; i++) {
        coeffs[k - 1 - i] = c;
        c *= ns;
    }
}

/* src/record.rs:18-29 kmer_to_index */
uint64_t orc_kmer_to_index(const uint8_t *kmer, unsigned k, unsigned ns,
                           uint64_t max_index) {
    uint64_t coeffs[64];
    orc_coeffs(coeffs, ns, k);
    uint64_t index = 0;
    for (unsigned i = 0; i < k; i++) {
        if (kmer[i] >= ns) {
            index = max_index;
            break;
        }
        index += coeffs[i] * kmer[i];
    }
    return index;
}

/* src/record.rs:31-39 count_monomers */
static void orc_count_monomers(const uint8_t *seq, size_t len, unsigned ns,
                               uint64_t *counts) {
    for (size_t i = 0; i < len; i++)
        if (seq[i] < (uint8_t)ns) counts[seq[i]]++;
}

/* src/record.rs:41-84 count_kmers (k >= 2 in the reference; also correct for 1) */
static void orc_count_kmers_k(const uint8_t *seq, size_t len, unsigned ns,
                              unsigned k, uint64_t *counts, uint64_t size) {
    uint64_t coeffs[64] = {0};
    orc_coeffs(coeffs, ns, k);
    size_t skip_until = 0;
    for (size_t i = 0; i < len && i < k; i++) /* record.rs:47-51 */
        if (seq[i] >= ns) skip_until = i + 1;

    int64_t index = -1;
    const uint8_t nstates = (uint8_t)ns;
    const int64_t biggest = (int64_t)coeffs[0];
    if (len < k) return; /* windows(k) is empty */
    for (size_t i = 0; i + k <= len; i++) { /* record.rs:57 */
        const uint8_t gained = seq[i + k - 1];
        if (gained >= nstates) { /* record.rs:59-64 */
            index = -1;
            skip_until = i + k;
        }
        if (i < skip_until) continue;
        if (index < 0) { /* record.rs:69-70 */
            index = (int64_t)orc_kmer_to_index(seq + i, k, ns, size - 1);
        } else { /* record.rs:72-74 */
            const int64_t dropped = seq[i - 1];
            index = (index - dropped * biggest) * (int64_t)ns + (int64_t)gained;
        }
        if (index < 0) continue;
        counts[index]++;
    }
}

/* src/record.rs:124-131 SeqRecord::to_kcounts.  counts has ns^k entries. */
int orc_count_kmers(const uint8_t *seq, size_t len, unsigned ns, unsigned k,
                    uint64_t *counts) {
    if (k == 0) return orc_fail(ORC_ERR_PANIC, "k cannot be 0", 0, 0);
    uint64_t size = 1;
    for (unsigned i = 0; i < k; i++) size *= ns;
    memset(counts, 0, size * sizeof *counts);
    if (k == 1)
        orc_count_monomers(seq, len, ns, counts);
    else
        orc_count_kmers_k(seq, len, ns, k, counts, size);
    return ORC_OK;
}

/* src/record.rs:86-106 entropy */
int orc_entropy(const double *kfreqs, size_t n, double *out) {
    if (n == 0)
        return orc_fail(ORC_ERR_PANIC,
                        "cannot calculate entropy as frequency vector empty", 0, 0);
    double entropy = 0.0, total_freq = 0.0;
    for (size_t i = 0; i < n; i++) {
        const double f = kfreqs[i];
        if (f == 0.0) continue;
        entropy += -f * log2(f);
        total_freq += f;
    }
    const double tolerance = (double)n * ORC_EPS;
    if (fabs(total_freq - 1.0) > tolerance)
        return orc_fail(ORC_ERR_PANIC,
                        "cannot calculate entropy as frequency vector total %.17g!=1.0",
                        total_freq, 0);
    *out = entropy;
    return ORC_OK;
}

/* src/record.rs:133-141 SeqRecord::to_kmerseq: counts -> freqs (+ entropy,
 * KmerSeq::new record.rs:157-168).  scratch has ns^k u64. */
int orc_to_kfreqs(const uint8_t *seq, size_t len, unsigned ns, unsigned k,
                  uint64_t *scratch, double *kfreqs, size_t B, double *entropy) {
    int rc = orc_count_kmers(seq, len, ns, k, scratch);
    if (rc) return rc;
    uint64_t tot = 0;
    for (size_t i = 0; i < B; i++) tot += scratch[i];
    const double total = (double)tot;
    if (total == 0.0) return orc_fail(ORC_ERR_NOKMERS, "No valid k-mers", 0, 0);
    for (size_t i = 0; i < B; i++) kfreqs[i] = (double)scratch[i] / total;
    return orc_entropy(kfreqs, B, entropy);
}

/* ------------------------------------------------------------ set algebra */

typedef struct {
    uint32_t label; /* stands for the seqid string */
    double *kfreqs;
    double entropy;
    double delta_jsd;
} orc_kseq;

typedef struct orc_set {
    orc_kseq *recs;
    size_t size, cap, B;
    double *summed_kfreqs;
    double summed_entropies;
    double total_jsd;
    uint32_t lowest_index;
    double *work; /* mean_kfreqs scratch */
} orc_set;

static int orc_set_contains(const orc_set *s, uint32_t label) {
    for (size_t i = 0; i < s->size; i++)
        if (s->recs[i].label == label) return 1;
    return 0;
}

/* src/records.rs:276-286 updated_mean_freqs */
static void orc_updated_mean_freqs(double *dest, const double *tot,
                                   const double *rec, double div, size_t B) {
    for (size_t i = 0; i < B; i++) {
        dest[i] = (tot[i] - rec[i]) / div;
        if (dest[i] <= ORC_EPS) dest[i] = 0.0;
    }
}

/* src/records.rs:220-252 get_lowest_record_index */
static int orc_lowest_record_index(orc_set *s, uint32_t *out) {
    const double div = (double)s->size - 1.0;
    if (div <= 0.0) return orc_fail(ORC_ERR_PANIC, "must have > 1 KmerSeq", 0, 0);
    double min_delta = 1e6;
    uint32_t lowest = 0;
    for (size_t i = 0; i < s->size; i++) {
        orc_kseq *r = &s->recs[i];
        const double mean_entropy = (s->summed_entropies - r->entropy) / div;
        orc_updated_mean_freqs(s->work, s->summed_kfreqs, r->kfreqs, div, s->B);
        double eom;
        int rc = orc_entropy(s->work, s->B, &eom);
        if (rc) return rc;
        const double jsd = eom - mean_entropy;
        r->delta_jsd = s->total_jsd - jsd;
        if (r->delta_jsd < min_delta) {
            min_delta = r->delta_jsd;
            lowest = (uint32_t)i;
        }
    }
    *out = lowest;
    return ORC_OK;
}

void orc_set_free(orc_set *s) {
    if (!s) return;
    for (size_t i = 0; i < s->size; i++) free(s->recs[i].kfreqs);
    free(s->recs);
    free(s->summed_kfreqs);
    free(s->work);
    free(s);
}

static int orc_set_reserve(orc_set *s, size_t want) {
    if (want <= s->cap) return ORC_OK;
    size_t cap = s->cap ? s->cap * 2 : 8;
    if (cap < want) cap = want;
    orc_kseq *r = realloc(s->recs, cap * sizeof *r);
    if (!r) return ORC_ERR_ALLOC;
    s->recs = r;
    s->cap = cap;
    return ORC_OK;
}

/* src/records.rs:27-68 SummedRecords::new.  freqs is m x B row-major,
 * entropies[m], labels[m]; rows are copied. */
int orc_set_new(const double *freqs, const double *entropies,
                const uint32_t *labels, size_t m, size_t B, orc_set **out) {
    if (m == 0) return orc_fail(ORC_ERR_PANIC, "records cannot be empty", 0, 0);
    orc_set *s = calloc(1, sizeof *s);
    if (!s) return ORC_ERR_ALLOC;
    s->B = B;
    s->summed_kfreqs = calloc(B, sizeof(double));
    s->work = calloc(B, sizeof(double));
    if (!s->summed_kfreqs || !s->work || orc_set_reserve(s, m)) {
        orc_set_free(s);
        return ORC_ERR_ALLOC;
    }
    for (size_t r = 0; r < m; r++) {
        orc_kseq *rec = &s->recs[r];
        rec->label = labels[r];
        rec->entropy = entropies[r];
        rec->delta_jsd = 0.0;
        rec->kfreqs = malloc(B * sizeof(double));
        if (!rec->kfreqs) {
            s->size = r;
            orc_set_free(s);
            return ORC_ERR_ALLOC;
        }
        memcpy(rec->kfreqs, freqs + r * B, B * sizeof(double));
        s->size = r + 1;
        for (size_t j = 0; j < B; j++) s->summed_kfreqs[j] += rec->kfreqs[j];
        s->summed_entropies += rec->entropy;
    }
    for (size_t j = 0; j < B; j++) s->work[j] = s->summed_kfreqs[j] / (double)s->size;
    double eom;
    int rc = orc_entropy(s->work, B, &eom);
    if (!rc) {
        s->total_jsd = eom - s->summed_entropies / (double)s->size;
        rc = orc_lowest_record_index(s, &s->lowest_index);
    }
    if (rc) {
        orc_set_free(s);
        return rc;
    }
    *out = s;
    return ORC_OK;
}

/* src/records.rs:182-189 clone (re-runs new) */
static int orc_set_clone(const orc_set *s, orc_set **out) {
    double *f = malloc(s->size * s->B * sizeof(double));
    double *h = malloc(s->size * sizeof(double));
    uint32_t *l = malloc(s->size * sizeof(uint32_t));
    if (!f || !h || !l) {
        free(f); free(h); free(l);
        return ORC_ERR_ALLOC;
    }
    for (size_t r = 0; r < s->size; r++) {
        memcpy(f + r * s->B, s->recs[r].kfreqs, s->B * sizeof(double));
        h[r] = s->recs[r].entropy;
        l[r] = s->recs[r].label;
    }
    int rc = orc_set_new(f, h, l, s->size, s->B, out);
    free(f); free(h); free(l);
    return rc;
}

/* src/records.rs:70-84 delta_jsd.  NaN is a legal result (no clamp). */
int orc_set_delta_jsd(orc_set *s, const double *kfreqs, double entropy,
                      uint32_t label, double *out) {
    if (orc_set_contains(s, label)) {
        *out = 0.0;
        return ORC_OK;
    }
    const orc_kseq *low = &s->recs[s->lowest_index];
    const double size = (double)s->size;
    const double mean_entropy = (s->summed_entropies - low->entropy + entropy) / size;
    for (size_t i = 0; i < s->B; i++)
        s->work[i] = (s->summed_kfreqs[i] - low->kfreqs[i] + kfreqs[i]) / size;
    double eom;
    int rc = orc_entropy(s->work, s->B, &eom);
    if (rc) return rc;
    *out = eom - mean_entropy;
    return ORC_OK;
}

/* src/records.rs:86-92 increases_jsd */
static int orc_set_increases_jsd(orc_set *s, const double *kfreqs, double entropy,
                                 uint32_t label, int *out) {
    if (orc_set_contains(s, label)) {
        *out = 0;
        return ORC_OK;
    }
    double jsd;
    int rc = orc_set_delta_jsd(s, kfreqs, entropy, label, &jsd);
    if (rc) return rc;
    *out = jsd > s->total_jsd + ORC_EPS; /* NaN -> false */
    return ORC_OK;
}

/* src/records.rs:94-109 drop_lowest (Vec::remove keeps the order of the rest) */
static void orc_set_drop_lowest(orc_set *s) {
    orc_kseq old = s->recs[s->lowest_index];
    memmove(&s->recs[s->lowest_index], &s->recs[s->lowest_index + 1],
            (s->size - s->lowest_index - 1) * sizeof(orc_kseq));
    s->size--; /* NB: the reference leaves self.size stale until push() */
    s->summed_entropies -= old.entropy;
    for (size_t i = 0; i < s->B; i++) {
        s->summed_kfreqs[i] -= old.kfreqs[i];
        if (s->summed_kfreqs[i] <= ORC_EPS) s->summed_kfreqs[i] = 0.0;
    }
    free(old.kfreqs);
}

/* src/records.rs:120-147 push */
static int orc_set_push(orc_set *s, const double *kfreqs, double entropy,
                        uint32_t label) {
    if (orc_set_contains(s, label)) return ORC_OK;
    if (orc_set_reserve(s, s->size + 1)) return ORC_ERR_ALLOC;
    orc_kseq *rec = &s->recs[s->size];
    rec->kfreqs = malloc(s->B * sizeof(double));
    if (!rec->kfreqs) return ORC_ERR_ALLOC;
    memcpy(rec->kfreqs, kfreqs, s->B * sizeof(double));
    rec->label = label;
    rec->entropy = entropy;
    rec->delta_jsd = 0.0;
    s->summed_entropies += entropy;
    for (size_t i = 0; i < s->B; i++) s->summed_kfreqs[i] += kfreqs[i];
    s->size++;
    for (size_t i = 0; i < s->B; i++) s->work[i] = s->summed_kfreqs[i] / (double)s->size;
    double eom;
    int rc = orc_entropy(s->work, s->B, &eom);
    if (rc) return rc;
    s->total_jsd = eom - s->summed_entropies / (double)s->size;
    return orc_lowest_record_index(s, &s->lowest_index);
}

/* src/records.rs:111-118 replace_lowest */
static int orc_set_replace_lowest(orc_set *s, const double *kfreqs, double entropy,
                                  uint32_t label) {
    if (orc_set_contains(s, label)) return ORC_OK;
    orc_set_drop_lowest(s);
    return orc_set_push(s, kfreqs, entropy, label);
}

/* src/records.rs:156-172 mean/std/cov of delta_jsd */
static double orc_mean_delta(const orc_set *s) {
    double sum = 0.0;
    for (size_t i = 0; i < s->size; i++) sum += s->recs[i].delta_jsd;
    return sum / (double)s->size;
}
static double orc_std_delta(const orc_set *s) {
    const double mean = orc_mean_delta(s);
    double sum = 0.0;
    for (size_t i = 0; i < s->size; i++) {
        const double d = s->recs[i].delta_jsd - mean;
        sum += d * d; /* powi(2) */
    }
    return sqrt(sum / ((double)s->size - 1.0));
}
static double orc_cov_delta(const orc_set *s) { return orc_std_delta(s) / orc_mean_delta(s); }

/* accessors used by the python wrapper */
size_t orc_set_size(const orc_set *s) { return s->size; }
size_t orc_set_nbins(const orc_set *s) { return s->B; }
double orc_set_total_jsd(const orc_set *s) { return s->total_jsd; }
double orc_set_summed_entropies(const orc_set *s) { return s->summed_entropies; }
uint32_t orc_set_lowest_index(const orc_set *s) { return s->lowest_index; }
double orc_set_mean_delta_jsd(const orc_set *s) { return orc_mean_delta(s); }
double orc_set_std_delta_jsd(const orc_set *s) { return orc_std_delta(s); }
double orc_set_cov_delta_jsd(const orc_set *s) { return orc_cov_delta(s); }
void orc_set_summed_kfreqs(const orc_set *s, double *out) {
    memcpy(out, s->summed_kfreqs, s->B * sizeof(double));
}
/* src/records.rs:175-180 get_raw_kseqs: labels, deltas, entropies (+freq rows if non-NULL) */
void orc_set_members(const orc_set *s, uint32_t *labels, double *deltas,
                     double *entropies, double *freqs) {
    for (size_t i = 0; i < s->size; i++) {
        if (labels) labels[i] = s->recs[i].label;
        if (deltas) deltas[i] = s->recs[i].delta_jsd;
        if (entropies) entropies[i] = s->recs[i].entropy;
        if (freqs) memcpy(freqs + i * s->B, s->recs[i].kfreqs, s->B * sizeof(double));
    }
}
int orc_set_push_pub(orc_set *s, const double *f, double h, uint32_t l) { return orc_set_push(s, f, h, l); }
int orc_set_replace_lowest_pub(orc_set *s, const double *f, double h, uint32_t l) { return orc_set_replace_lowest(s, f, h, l); }
int orc_set_increases_jsd_pub(orc_set *s, const double *f, double h, uint32_t l, int *o) { return orc_set_increases_jsd(s, f, h, l, o); }

/* --------------------------------------------------------------- selectors */

typedef struct {
    const uint8_t *seqs;      /* concatenated symbol indices, or NULL */
    const uint64_t *offsets;  /* nrec+1 */
    const double *freqs;      /* nrec x B rows (the final_* merges), or NULL */
    const uint32_t *labels;   /* nrec identity labels (NULL: label = position) */
    size_t nrec, B;
    unsigned ns, k;
    uint64_t *scratch;
    double *row;
} orc_source;

/* record.rs:205-209 LazySeqRecord::to_kmerseq, or KmerSeq::new for a merge row */
static int orc_source_row(orc_source *src, size_t i, const double **f, double *h) {
    if (src->freqs) {
        *f = src->freqs + i * src->B;
        return orc_entropy(*f, src->B, h); /* records.rs:353 KmerSeq::new */
    }
    int rc = orc_to_kfreqs(src->seqs + src->offsets[i],
                           (size_t)(src->offsets[i + 1] - src->offsets[i]), src->ns,
                           src->k, src->scratch, src->row, src->B, h);
    *f = src->row;
    return rc;
}
static uint32_t orc_source_label(const orc_source *src, size_t i) {
    return src->labels ? src->labels[i] : (uint32_t)i;
}

/* src/records.rs:288-308 / :344-360: the first n records seed the set; a
 * sequence with no valid k-mers is skipped (only possible for sequences). */
static int orc_init_set(orc_source *src, size_t n, orc_set **out) {
    double *f = malloc((n ? n : 1) * src->B * sizeof(double));
    double *h = malloc((n ? n : 1) * sizeof(double));
    uint32_t *l = malloc((n ? n : 1) * sizeof(uint32_t));
    if (!f || !h || !l) {
        free(f); free(h); free(l);
        return ORC_ERR_ALLOC;
    }
    size_t m = 0;
    int rc = ORC_OK;
    for (size_t i = 0; i < n; i++) {
        const double *row;
        rc = orc_source_row(src, i, &row, &h[m]);
        if (rc == ORC_ERR_NOKMERS) { rc = ORC_OK; continue; }
        if (rc) break;
        memcpy(f + m * src->B, row, src->B * sizeof(double));
        l[m] = orc_source_label(src, i);
        m++;
    }
    if (!rc) rc = orc_set_new(f, h, l, m, src->B, out);
    free(f); free(h); free(l);
    return rc;
}

/* src/records.rs:311-342 select_nmost_divergent and :363-382 ..._final */
static int orc_run_nmost(orc_source *src, size_t n, orc_set **out, uint64_t *n_accepts) {
    if (src->nrec < n)
        return orc_fail(ORC_ERR_PANIC, "The number of sequences %.0f is < n %.0f",
                        (double)src->nrec, (double)n);
    orc_set *s = NULL;
    int rc = orc_init_set(src, n, &s);
    if (rc) return rc;
    for (size_t i = n; i < src->nrec; i++) {
        const double *row;
        double h;
        rc = orc_source_row(src, i, &row, &h);
        if (rc == ORC_ERR_NOKMERS) { rc = ORC_OK; continue; }
        if (rc) break;
        int inc;
        rc = orc_set_increases_jsd(s, row, h, orc_source_label(src, i), &inc);
        if (rc) break;
        if (inc) {
            rc = orc_set_replace_lowest(s, row, h, orc_source_label(src, i));
            if (rc) break;
            if (n_accepts) (*n_accepts)++;
        }
    }
    if (rc) {
        orc_set_free(s);
        return rc;
    }
    *out = s;
    return ORC_OK;
}

/* src/records.rs:390-454 select_max_divergent and :456-507 ..._final.
 * stat_is_std: 1 = Stat::Std, 0 = Stat::Cov. */
static int orc_run_max(orc_source *src, int stat_is_std, size_t min_size,
                       size_t max_size, orc_set **out) {
    if (src->nrec < min_size)
        return orc_fail(ORC_ERR_PANIC, "The number of sequences %.0f is < n %.0f",
                        (double)src->nrec, (double)min_size);
    if (src->nrec <= max_size) max_size = src->nrec;
    orc_set *s = NULL;
    int rc = orc_init_set(src, min_size, &s);
    if (rc) return rc;
    for (size_t i = min_size; i < src->nrec; i++) {
        const double *row;
        double h;
        rc = orc_source_row(src, i, &row, &h);
        if (rc == ORC_ERR_NOKMERS) { rc = ORC_OK; continue; }
        if (rc) break;
        const uint32_t label = orc_source_label(src, i);
        int inc;
        rc = orc_set_increases_jsd(s, row, h, label, &inc);
        if (rc) break;
        if (!inc) continue;
        if (s->size == (uint32_t)max_size) {
            rc = orc_set_replace_lowest(s, row, h, label);
            if (rc) break;
            continue;
        }
        orc_set *ns = NULL;
        rc = orc_set_clone(s, &ns);
        if (rc) break;
        rc = orc_set_push(ns, row, h, label);
        if (rc) {
            orc_set_free(ns);
            break;
        }
        const int better = stat_is_std ? (orc_std_delta(ns) > orc_std_delta(s))
                                       : (orc_cov_delta(ns) > orc_cov_delta(s));
        if (better) {
            orc_set_free(s);
            s = ns;
        } else {
            orc_set_free(ns);
        }
    }
    if (rc) {
        orc_set_free(s);
        return rc;
    }
    *out = s;
    return ORC_OK;
}

static int orc_source_init_seqs(orc_source *src, const uint8_t *seqs,
                                const uint64_t *offsets, const uint32_t *labels,
                                size_t nseq, unsigned ns, unsigned k) {
    memset(src, 0, sizeof *src);
    if (k == 0) return orc_fail(ORC_ERR_PANIC, "k cannot be 0", 0, 0);
    src->seqs = seqs;
    src->offsets = offsets;
    src->labels = labels;
    src->nrec = nseq;
    src->ns = ns;
    src->k = k;
    src->B = 1;
    for (unsigned i = 0; i < k; i++) src->B *= ns;
    src->scratch = malloc(src->B * sizeof(uint64_t));
    src->row = malloc(src->B * sizeof(double));
    if (!src->scratch || !src->row) return ORC_ERR_ALLOC;
    return ORC_OK;
}
static void orc_source_done(orc_source *src) {
    free(src->scratch);
    free(src->row);
}

int orc_nmost(const uint8_t *seqs, const uint64_t *offsets, const uint32_t *labels,
              size_t nseq, size_t n, unsigned k, unsigned ns, orc_set **out,
              uint64_t *n_accepts) {
    orc_source src;
    int rc = orc_source_init_seqs(&src, seqs, offsets, labels, nseq, ns, k);
    if (!rc) rc = orc_run_nmost(&src, n, out, n_accepts);
    orc_source_done(&src);
    return rc;
}

/* The reference's own parallel scheme for `nmost` (diverse_seq/records.py:225-245 apply_app:
 * contiguous chunks of the id list, diverse_seq/util.py:82-102, one select_nmost per worker,
 * then select_final_nmost over the winners in chunk order, src/records.rs:363-382), with the
 * workers as threads of this process instead of worker processes: what bench.py times as the
 * all-cores CPU baseline.  bounds: nchunks + 1 sequence indices.  winners_out: nchunks * n rows of
 * B doubles (chunk c's members, in member order, from row c * n), sizes_out: members per chunk. */
#include <pthread.h>
typedef struct {
    const uint8_t *seqs;
    const uint64_t *offsets;
    size_t lo, hi, n, B;
    unsigned k, ns;
    double *rows;
    uint32_t *labels;
    size_t size;
    int rc;
} orc_chunk_job;

static void *orc_chunk_worker(void *arg) {
    orc_chunk_job *j = (orc_chunk_job *)arg;
    const size_t m = j->hi - j->lo;
    uint64_t *local = (uint64_t *)malloc((m + 1) * sizeof(uint64_t));
    uint32_t *lab = (uint32_t *)malloc((m ? m : 1) * sizeof(uint32_t));
    orc_set *set = NULL;
    uint64_t acc = 0;
    j->size = 0;
    if (!local || !lab) {
        j->rc = ORC_ERR_ALLOC;
    } else {
        for (size_t i = 0; i <= m; i++) local[i] = j->offsets[j->lo + i] - j->offsets[j->lo];
        for (size_t i = 0; i < m; i++) lab[i] = (uint32_t)(j->lo + i);
        j->rc = orc_nmost(j->seqs + j->offsets[j->lo], local, lab, m, j->n, j->k, j->ns, &set, &acc);
        if (!j->rc) {
            j->size = orc_set_size(set);
            orc_set_members(set, j->labels, NULL, NULL, j->rows);
        }
    }
    if (set) orc_set_free(set);
    free(local);
    free(lab);
    return NULL;
}

int orc_nmost_chunks_mt(const uint8_t *seqs, const uint64_t *offsets, const uint64_t *bounds,
                        size_t nchunks, size_t n, unsigned k, unsigned ns, double *winners_out,
                        uint32_t *labels_out, uint64_t *sizes_out) {
    size_t B = 1;
    for (unsigned i = 0; i < k; i++) B *= ns;
    orc_chunk_job *jobs = (orc_chunk_job *)calloc(nchunks, sizeof(orc_chunk_job));
    pthread_t *th = (pthread_t *)calloc(nchunks, sizeof(pthread_t));
    if (!jobs || !th) {
        free(jobs);
        free(th);
        return orc_fail(ORC_ERR_ALLOC, "out of memory", 0, 0);
    }
    int rc = 0;
    for (size_t c = 0; c < nchunks; c++) {
        orc_chunk_job *j = &jobs[c];
        j->seqs = seqs;
        j->offsets = offsets;
        j->lo = bounds[c];
        j->hi = bounds[c + 1];
        j->n = n;
        j->B = B;
        j->k = k;
        j->ns = ns;
        j->rows = winners_out + c * n * B;
        j->labels = labels_out + c * n;
        if (pthread_create(&th[c], NULL, orc_chunk_worker, j)) {
            orc_chunk_worker(j); /* no thread to be had: run it here */
            th[c] = 0;
        }
    }
    for (size_t c = 0; c < nchunks; c++) {
        if (th[c]) pthread_join(th[c], NULL);
        sizes_out[c] = jobs[c].size;
        if (jobs[c].rc && !rc) rc = jobs[c].rc;
    }
    free(jobs);
    free(th);
    return rc;
}

int orc_max(const uint8_t *seqs, const uint64_t *offsets, const uint32_t *labels,
            size_t nseq, size_t min_size, size_t max_size, int stat_is_std, unsigned k,
            unsigned ns, orc_set **out) {
    orc_source src;
    int rc = orc_source_init_seqs(&src, seqs, offsets, labels, nseq, ns, k);
    if (!rc) rc = orc_run_max(&src, stat_is_std, min_size, max_size, out);
    orc_source_done(&src);
    return rc;
}

int orc_final_nmost(const double *freqs, const uint32_t *labels, size_t nrec, size_t B,
                    size_t n, orc_set **out) {
    orc_source src;
    memset(&src, 0, sizeof src);
    src.freqs = freqs;
    src.labels = labels;
    src.nrec = nrec;
    src.B = B;
    return orc_run_nmost(&src, n, out, NULL);
}

int orc_final_max(const double *freqs, const uint32_t *labels, size_t nrec, size_t B,
                  size_t min_size, size_t max_size, int stat_is_std, orc_set **out) {
    orc_source src;
    memset(&src, 0, sizeof src);
    src.freqs = freqs;
    src.labels = labels;
    src.nrec = nrec;
    src.B = B;
    return orc_run_max(&src, stat_is_std, min_size, max_size, out);
}

/* src/records.rs:509-524 make_summed_records (the delta-JSD calculator's set) */
int orc_make_summed_records(const uint8_t *seqs, const uint64_t *offsets,
                            const uint32_t *labels, size_t nseq, unsigned k, unsigned ns,
                            orc_set **out) {
    orc_source src;
    int rc = orc_source_init_seqs(&src, seqs, offsets, labels, nseq, ns, k);
    if (!rc) rc = orc_init_set(&src, nseq, out);
    orc_source_done(&src);
    return rc;
}

/* -------------------------------------------------------------------- mash */

/* src/distance.rs:17-19 reverse_complement */
void orc_reverse_complement(const uint8_t *kmer, size_t k, uint8_t *out) {
    for (size_t i = 0; i < k; i++) out[k - 1 - i] = (uint8_t)((kmer[i] + 2) % 4);
}

static inline uint32_t orc_rotl32(uint32_t x, unsigned r) { return (x << r) | (x >> (32 - r)); }

/* src/distance.rs:21-49 murmurhash3_32 (per-BYTE blocks, no tail, non-standard) */
uint32_t orc_murmurhash3_32(const uint8_t *data, size_t len, uint32_t seed) {
    if (seed == 0) seed = 0x9747B28Cu;
    uint32_t h = seed ^ (uint32_t)len;
    for (size_t i = 0; i < len; i++) {
        uint32_t k = data[i];
        k *= 0xCC9E2D51u;
        k = orc_rotl32(k, 15);
        k *= 0x1B873593u;
        h ^= k;
        h = orc_rotl32(h, 13);
        h = h * 5u + 0xE6546B64u;
    }
    h ^= h >> 16;
    h *= 0x85EBCA6Bu;
    h ^= h >> 13;
    h *= 0xC2B2AE35u;
    h ^= h >> 16;
    return h;
}

/* src/distance.rs:65-87 hash_kmer */
uint32_t orc_hash_kmer(const uint8_t *kmer, size_t k, int canonical) {
    if (canonical) {
        uint8_t rev[256];
        orc_reverse_complement(kmer, k, rev);
        for (size_t i = 0; i < k; i++) {
            if (kmer[i] < rev[i]) break;
            if (kmer[i] > rev[i]) return orc_murmurhash3_32(rev, k, 0);
        }
    }
    return orc_murmurhash3_32(kmer, k, 0);
}

/* src/distance.rs:101-134 get_kmer_hashes; out has room for len-k+1; returns count */
size_t orc_kmer_hashes(const uint8_t *seq, size_t len, size_t k, unsigned ns,
                       int canonical, uint32_t *out) {
    if (len < k || k == 0) return 0;
    size_t skip_until = 0, n = 0;
    for (size_t i = 0; i < k; i++)
        if (seq[i] >= ns) skip_until = i + 1;
    for (size_t i = 0; i + k <= len; i++) {
        if (seq[i + k - 1] >= ns) skip_until = i + k;
        if (i < skip_until) continue;
        out[n++] = orc_hash_kmer(seq + i, k, canonical);
    }
    return n;
}

static int orc_cmp_u32(const void *a, const void *b) {
    const uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b;
    return (x > y) - (x < y);
}

/* src/distance.rs:151-182 mash_sketch: the sketch_size smallest DISTINCT
 * hashes, ascending (HashSet + max-heap + sort == sort + unique + truncate).
 * out has room for sketch_size; returns the sketch length. */
size_t orc_mash_sketch(const uint8_t *seq, size_t len, size_t k, size_t sketch_size,
                       unsigned ns, int canonical, uint32_t *out) {
    if (len < k || k == 0) return 0;
    uint32_t *h = malloc((len - k + 1) * sizeof *h);
    if (!h) return 0;
    const size_t n = orc_kmer_hashes(seq, len, k, ns, canonical, h);
    qsort(h, n, sizeof *h, orc_cmp_u32);
    size_t m = 0;
    for (size_t i = 0; i < n && m < sketch_size; i++)
        if (i == 0 || h[i] != h[i - 1]) out[m++] = h[i];
    free(h);
    return m;
}

/* diverse_seq/distance.py:230-291 mash_distance.  Returns NaN where the
 * python raises ZeroDivisionError (both sketches empty). */
double orc_mash_distance(const uint32_t *left, size_t nl, const uint32_t *right,
                         size_t nr, unsigned k, size_t sketch_size) {
    size_t inter = 0, uni = 0, li = 0, ri = 0;
    while (uni < sketch_size && li < nl && ri < nr) {
        const uint32_t l = left[li], r = right[ri];
        if (l < r) li++;
        else if (r < l) ri++;
        else { li++; ri++; inter++; }
        uni++;
    }
    if (uni < sketch_size) {
        if (li < nl) uni += nl - li;
        if (ri < nr) uni += nr - ri;
        if (uni > sketch_size) uni = sketch_size;
    }
    if (uni == 0) return NAN;
    const double jaccard = (double)inter / (double)uni;
    if (inter == uni) return 0.0;
    if (inter == 0) return 1.0;
    double d = -log(2.0 * jaccard / (1.0 + jaccard)) / (double)k;
    if (d > 1.0) d = 1.0;
    return d;
}

/* diverse_seq/distance.py:335-336 euclidean_distance = numpy.linalg.norm(a-b)
 * (sqrt of the pairwise-summed squares is numpy's; sequential here -- the
 * reference test tolerance for this mode is 1e-3, tests/test_distance.py:62) */
double orc_euclidean_distance(const double *a, const double *b, size_t n) {
    double s = 0.0;
    for (size_t i = 0; i < n; i++) {
        const double d = a[i] - b[i];
        s += d * d;
    }
    return sqrt(s);
}
