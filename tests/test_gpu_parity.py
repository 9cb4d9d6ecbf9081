"""Parity of the HIP path (through the C ABI) against the CPU oracle.
Bit-exact for k-mer counts, sketches and selected ids; 1e-6 relative (the
north-star tolerance; in practice ~1e-13) for JSD / entropy / distance floats."""
import math
import pickle

import numpy as np
import pytest

import oracle
from conftest import str2arr, synth_seqs

pytestmark = pytest.mark.gpu

RTOL = 1e-6   # BASELINE.json north_star: "within 1e-6 relative for JSD/entropy floats"
TIGHT = 1e-11  # what the f64 kernels actually deliver


@pytest.fixture(scope="module")
def ctx():
    from diverseseq_amd import engine

    return engine.default_context()


def _oracle_counts(seqs, ns, k):
    return np.stack([oracle.count_kmers(s, ns, k) for s in seqs]).astype(np.uint64)


def _check_matrix(ctx, seqs, k, ns=4):
    m = ctx.build_matrix(seqs, k, ns)
    got = m.counts().astype(np.uint64)
    exp = _oracle_counts(seqs, ns, k)
    assert got.shape == exp.shape
    assert (got == exp).all(), f"k-mer counts differ (k={k}, ns={ns})"
    tot = m.totals()
    assert (tot == exp.sum(axis=1)).all()
    H = m.entropy()
    for i, s in enumerate(seqs):
        if tot[i]:
            _, h = oracle.to_kfreqs(s, ns, k)
            assert abs(H[i] - h) <= TIGHT * max(1.0, abs(h)), (i, H[i], h)
    m.close()


# ------------------------------------------------------------------ k-mer counts
@pytest.mark.parametrize("k", [1, 2, 3, 4, 6, 7])
def test_counts_brca1(ctx, brca1, k):
    _check_matrix(ctx, list(brca1.values()), k)


def test_counts_rust_vector(ctx, rust_vectors):
    v = rust_vectors["kmer_count"]
    c, _, _ = ctx.kmer_counts([np.array(v["seq"], dtype=np.uint8)], v["k"], v["num_states"])
    assert c[0].tolist() == v["counts"]


def test_counts_edge_cases(ctx):
    rng = np.random.default_rng(3)
    seqs = synth_seqs(40, 700, 11, invalid_frac=0.01, ragged=True)
    seqs += [np.zeros(0, dtype=np.uint8),                       # empty
             np.array([1, 2], dtype=np.uint8),                  # L < k
             np.full(50, 4, dtype=np.uint8),                    # all invalid
             np.array([4] + [0, 1, 2, 3] * 5 + [4], dtype=np.uint8),  # invalid at both edges
             np.array([0, 1, 2, 3, 0, 1], dtype=np.uint8),      # L == k (k=6)
             rng.integers(0, 4, size=17, dtype=np.uint8),
             rng.integers(0, 4, size=16, dtype=np.uint8),
             rng.integers(0, 4, size=15, dtype=np.uint8)]
    for s in seqs[:10]:
        if s.size > 40:
            s[[0, 5, s.size - 1]] = 4  # invalid symbols at window edges
    for k in (1, 2, 5, 6):
        _check_matrix(ctx, seqs, k)


def test_counts_long_sequences_multi_tile(ctx):
    """genome-like rows: many tiles per sequence merged with global atomics"""
    rng = np.random.default_rng(5)
    seqs = [rng.integers(0, 4, size=n, dtype=np.uint8) for n in (200_000, 32_768 + 5, 32_773, 70_001)]
    seqs[0][rng.integers(0, 200_000, size=200)] = 4
    seqs.append(rng.integers(0, 4, size=900, dtype=np.uint8))
    _check_matrix(ctx, seqs, 6)
    _check_matrix(ctx, seqs, 3)


@pytest.mark.parametrize("tile", [65536, 262144])
def test_counts_long_rows_in_long_tiles(ctx, monkeypatch, tile):
    """long rows are cut into tiles of up to 16 windows per bin once the input is large (2^27 bases and more:
    dvs_hist_prepare); here the tile length is forced (DVS_TEST_KNOBS=long_tile_<n>) so that inputs the oracle counts in
    a second take the same path: rows shorter than, equal to and a few windows beyond whole tiles, invalid symbols
    across tile edges, a short row between them -- counts, totals and entropies against the oracle at 4^6 and 4^7"""
    monkeypatch.setenv("DVS_TEST_KNOBS", f"long_tile_{tile}")
    rng = np.random.default_rng(tile)
    k_edge = 7
    lens = (tile + k_edge - 1, tile + k_edge, 2 * tile + k_edge + 3, 3 * tile // 2, 40_000)
    seqs = [rng.integers(0, 4, size=n, dtype=np.uint8) for n in lens]
    for s in seqs:
        s[rng.integers(0, s.size, size=s.size // 5000 + 1)] = 4
    seqs[2][tile + 1:tile + 9] = 4  # invalid symbols across the first tile's end
    seqs.insert(2, rng.integers(0, 4, size=700, dtype=np.uint8))
    _check_matrix(ctx, seqs, 7)
    _check_matrix(ctx, seqs, 6)


def test_counts_fullest_single_tile(ctx):
    """the longest sequence that is still one tile (32 768 windows), as a homopolymer: one bin takes
    every count -- the packed 16-bit LDS counters of the whole-sequence kernel must not carry (low
    half: index 0; high half: the odd index of poly-C), nor may the neighbours change"""
    k = 6
    for base in (0, 1):
        seqs = [np.full(32768 + k - 1, base, dtype=np.uint8), np.full(32768 + k, base, dtype=np.uint8),
                np.full(40000, base, dtype=np.uint8)]
        seqs[0][-1] = 3 - base  # one other k-mer at the end
        _check_matrix(ctx, seqs, k)


def test_counts_offsets_cache_is_compared_by_content(ctx):
    """the context remembers the last build's offsets and tile lists; the same number of sequences
    with other boundaries, another k, or other bytes of the same total must not reuse them"""
    rng = np.random.default_rng(17)
    a = [rng.integers(0, 4, size=n, dtype=np.uint8) for n in (900, 40_000, 1200, 700)]
    b = [rng.integers(0, 4, size=n, dtype=np.uint8) for n in (40_000, 900, 700, 1200)]  # same total, other cuts
    for seqs, k in ((a, 5), (a, 5), (b, 5), (a, 6), (b, 6), (b, 6), (a, 5)):
        _check_matrix(ctx, seqs, k)


def test_counts_large_k_global_histogram(ctx):
    """4^k * 4 B > 64 KB: the row in L2 takes the atomics (k = 8, 9)"""
    seqs = synth_seqs(6, 5000, 21, invalid_frac=0.002)
    seqs.append(np.random.default_rng(2).integers(0, 4, size=90_000, dtype=np.uint8))
    _check_matrix(ctx, seqs, 8)
    _check_matrix(ctx, seqs[:3], 9)


@pytest.mark.parametrize("ns,k", [(20, 2), (5, 3), (3, 4), (2, 7), (20, 3)])
def test_counts_other_alphabets(ctx, ns, k):
    rng = np.random.default_rng(ns * 10 + k)
    seqs = [rng.integers(0, ns + 1, size=int(rng.integers(5, 3000)), dtype=np.uint8) for _ in range(12)]
    _check_matrix(ctx, seqs, k, ns)


def test_counts_errors(ctx):
    with pytest.raises(ValueError, match="k cannot be 0"):
        ctx.build_matrix([np.array([0, 1, 2, 0], dtype=np.uint8)], 0, 4)


def test_counts_device_resident_input(ctx):
    """sequences already in HBM as a torch tensor (zero-copy hand-off)"""
    torch = pytest.importorskip("torch")
    seqs = synth_seqs(50, 1200, 9, ragged=True)
    data, offs = oracle.concat(seqs)
    t = torch.from_numpy(data).to("cuda:0")
    torch.cuda.synchronize()
    m = ctx.build_matrix_device(t.data_ptr(), offs, 5, 4)
    assert (m.counts().astype(np.uint64) == _oracle_counts(seqs, 4, 5)).all()


@pytest.mark.parametrize("bad_seeds", [(), (1,), (0, 3), (0, 1, 4)])
def test_selection_behind_a_device_resident_build(ctx, bad_seeds):
    """device-resident input (the build does not wait for its kernels; the selection picks the
    first rows' totals up from the pinned block): seed rows without a valid k-mer -- too short,
    all invalid -- are skipped as records.rs:299-306 does"""
    torch = pytest.importorskip("torch")
    seqs = synth_seqs(3000, 700, seed=77, ragged=True)
    for i, r in enumerate(bad_seeds):
        seqs[r] = np.full(40, 4, dtype=np.uint8) if i % 2 else seqs[r][:3].copy()
    exp = oracle.nmost(seqs, 5, 5, 4)
    data, offs = oracle.concat(seqs)
    t = torch.from_numpy(np.concatenate([data, np.zeros(16, np.uint8)])).to("cuda:0")
    torch.cuda.synchronize()
    m = ctx.build_matrix_device(t.data_ptr(), offs, 5, 4)
    sel = m.nmost(5)
    _assert_selection(sel, exp)
    sel.close()
    m.close()


# -------------------------------------------------------------------- selection
def _assert_selection(sel, exp, rtol=RTOL):
    got = sel.members(with_freqs=True)
    s = sel.summary()
    elab, edelta, eent, efreq = exp.members(with_freqs=True)
    assert s.size == exp.size
    assert got.positions.tolist() == elab.tolist(), "selected ids / member order differ"
    np.testing.assert_allclose(got.delta_jsd, edelta, rtol=rtol, atol=1e-13)
    np.testing.assert_allclose(got.entropy, eent, rtol=rtol)
    assert (got.kfreqs == efreq).all(), "member frequency rows must be bit-exact (count / total)"
    for name in ("total_jsd", "mean_delta_jsd", "std_delta_jsd", "cov_delta_jsd"):
        g, e = getattr(s, name), getattr(exp, name)
        if math.isnan(e) or math.isinf(e):  # e.g. cov = std / 0 for identical members
            assert (math.isnan(g) and math.isnan(e)) or g == e, (name, g, e)
            continue
        assert abs(g - e) <= rtol * max(abs(e), 1e-300) + 1e-13, (name, g, e)
    assert s.lowest_index == exp.lowest_index
    return s


def test_set_rust_golden(ctx, rust_vectors):
    """src/records.rs:588-621,676-685 through MODE_SET"""
    v = rust_vectors["summed"]
    seqs = [np.array(s, dtype=np.uint8) for s in v["seqs"].values()]
    m = ctx.build_matrix(seqs, v["k"], v["num_states"])
    sel = m.as_set()
    s = sel.summary()
    mem = sel.members()
    assert s.size == v["size"]
    np.testing.assert_allclose(s.total_jsd, v["total_jsd"], rtol=TIGHT)
    np.testing.assert_allclose(mem.entropy, v["entropies"], rtol=TIGHT)
    np.testing.assert_allclose(s.summed_entropies, v["summed_entropies"], rtol=TIGHT)
    np.testing.assert_allclose(mem.delta_jsd, v["delta_jsds"], rtol=1e-10)
    np.testing.assert_allclose(s.mean_delta_jsd, v["mean_delta_jsd"], rtol=1e-10)
    np.testing.assert_allclose(s.std_delta_jsd, v["std_delta_jsd"], rtol=1e-10)
    # candidate scores: better increases JSD, a member scores 0.0 (records.rs:629-649)
    q = ctx.build_matrix([np.array(v["better"]["seq"], dtype=np.uint8), seqs[0]], 1, 4)
    d = sel.delta_jsd(q, [99, 0])
    assert d[0] > s.total_jsd + 2.3e-16 and d[1] == 0.0
    oset = oracle.SummedRecords.from_seqs(seqs, 1, 4)
    f, h = oracle.to_kfreqs(v["better"]["seq"], 4, 1)
    np.testing.assert_allclose(d[0], oset.delta_jsd(f, h, 99), rtol=TIGHT)


def test_selector_rust_golden(ctx, rust_vectors):
    """src/records.rs:694-812"""
    v = rust_vectors["selector"]
    names = [n for n, _ in v["seqs"]]
    seqs = [np.array(s, dtype=np.uint8) for _, s in v["seqs"]]
    order_in = v["fxhash_consistent_order"]
    stream = [seqs[names.index(n)] for n in order_in]
    m = ctx.build_matrix(stream, 1, 4)
    sel = m.nmost(3)
    assert [order_in[int(p)] for p in sel.members().positions] == v["summed234_order"]
    _assert_selection(sel, oracle.nmost(stream, 3, 1, 4))
    # insertion order, an invalid sequence in the stream, duplicate ids
    inv = np.array(v["invalid"][1], dtype=np.uint8)
    m2 = ctx.build_matrix(seqs + [inv], 1, 4)
    sel2 = m2.nmost(3)
    assert {names[int(p)] for p in sel2.members().positions} == set(v["nmost_n3_members"])
    _assert_selection(sel2, oracle.nmost(seqs + [inv], 3, 1, 4))
    lab = [0, 1, 2, 3, 4, 1]
    m3 = ctx.build_matrix(seqs + [seqs[1]], 1, 4)
    _assert_selection(m3.nmost(3, labels=lab), oracle.nmost(seqs + [seqs[1]], 3, 1, 4, labels=lab))
    with pytest.raises(ValueError, match="The number of sequences 5 is < n 20"):
        ctx.build_matrix(seqs, 1, 4).nmost(20)
    for stat in ("stdev", "cov"):
        sel4 = m2.max_divergent(3, 4, stat)
        _assert_selection(sel4, oracle.max_divergent(seqs + [inv], 3, 4, 1, 4, stat))
    sel5 = m2.max_divergent(3, 10, "stdev")
    _assert_selection(sel5, oracle.max_divergent(seqs + [inv], 3, 10, 1, 4, "stdev"))


def test_seed_panics(ctx):
    inv = np.full(4, 4, dtype=np.uint8)
    ok = np.array([0, 1, 2, 3], dtype=np.uint8)
    with pytest.raises(ValueError, match="records cannot be empty"):
        ctx.build_matrix([inv, inv, ok], 1, 4).nmost(2)
    with pytest.raises(ValueError, match="must have > 1 KmerSeq"):
        ctx.build_matrix([ok, inv, ok], 1, 4).nmost(2)


@pytest.mark.parametrize("k,n", [(4, 10), (2, 5), (6, 7), (1, 5)])
def test_nmost_brca1(ctx, brca1, k, n):
    """config C1 on the device: BRCA1 demo, k=4, n=10 (and neighbours)"""
    seqs = list(brca1.values())
    m = ctx.build_matrix(seqs, k, 4)
    _assert_selection(m.nmost(n), oracle.nmost(seqs, n, k, 4))


@pytest.mark.parametrize("stat", ["stdev", "cov"])
@pytest.mark.parametrize("k,lo,hi", [(3, 5, 30), (4, 3, 8), (2, 7, 55), (5, 10, 12)])
def test_max_brca1(ctx, brca1, stat, k, lo, hi):
    seqs = list(brca1.values())
    rng = np.random.default_rng(k * 100 + lo)
    perm = rng.permutation(len(seqs))
    seqs = [seqs[i] for i in perm]
    m = ctx.build_matrix(seqs, k, 4)
    _assert_selection(m.max_divergent(lo, hi, stat), oracle.max_divergent(seqs, lo, hi, k, 4, stat))


@pytest.mark.parametrize("nseq,length,k,n,window", [(3000, 400, 3, 10, 64), (2000, 600, 6, 20, 256),
                                                   (1500, 300, 4, 50, 0), (4000, 200, 2, 5, 1024),
                                                   (800, 2500, 7, 12, 128)])
def test_nmost_synthetic(ctx, nseq, length, k, n, window):
    seqs = synth_seqs(nseq, length, nseq + k, invalid_frac=0.001, ragged=True)
    m = ctx.build_matrix(seqs, k, 4)
    s = _assert_selection(m.nmost(n, window=window), oracle.nmost(seqs, n, k, 4))
    assert s.n_accepts > 0 and s.rows_scored >= nseq - n - 5


def test_nmost_structured_families(ctx):
    """a few divergent families (paper/nbks/synthetic_known.py:16-25 style) so accepts are common"""
    rng = np.random.default_rng(77)
    seqs = []
    for fam in range(6):
        p = rng.dirichlet(np.ones(4) * 0.7)
        for _ in range(150):
            seqs.append(rng.choice(4, size=int(rng.integers(400, 900)), p=p).astype(np.uint8))
    order = rng.permutation(len(seqs))
    seqs = [seqs[i] for i in order]
    for k, n in ((3, 6), (5, 12)):
        m = ctx.build_matrix(seqs, k, 4)
        _assert_selection(m.nmost(n, window=128), oracle.nmost(seqs, n, k, 4))
        _assert_selection(m.max_divergent(4, 25, "stdev", window=128),
                          oracle.max_divergent(seqs, 4, 25, k, 4, "stdev"))


def test_explicit_order_and_labels(ctx):
    seqs = synth_seqs(300, 500, 4, ragged=True)
    rng = np.random.default_rng(0)
    order = rng.permutation(300).astype(np.uint32)
    order = np.concatenate([order, order[:40]])  # ids repeated later in the stream
    m = ctx.build_matrix(seqs, 4, 4)
    got = m.nmost(9, order=order, labels=order)
    exp = oracle.nmost([seqs[i] for i in order], 9, 4, 4, labels=order)
    gm = got.members(False)
    assert [int(order[p]) for p in gm.positions] == exp.members()[0].tolist()
    np.testing.assert_allclose(gm.delta_jsd, exp.members()[1], rtol=RTOL, atol=1e-13)


def test_final_merges(ctx, brca1):
    """chunk + merge: records.py:225-245 -> records.rs:363-382 / 456-507"""
    seqs = list(brca1.values())
    chunks = [seqs[:20], seqs[20:40], seqs[40:]]
    rows, labels, base = [], [], 0
    for ch in chunks:
        r = oracle.nmost(ch, 6, 3, 4)
        l, _, _, f = r.members(with_freqs=True)
        rows.append(f)
        labels.append(l + base)
        base += len(ch)
    rows, labels = np.vstack(rows), np.concatenate(labels)
    m = ctx.matrix_from_freqs(rows)
    got = m.nmost(6, labels=labels)
    exp = oracle.final_nmost(rows, 6, labels=labels)
    gm = got.members()
    assert [int(labels[p]) for p in gm.positions] == exp.members()[0].tolist()
    np.testing.assert_allclose(gm.delta_jsd, exp.members()[1], rtol=RTOL, atol=1e-13)
    np.testing.assert_allclose(got.summary().total_jsd, exp.total_jsd, rtol=RTOL)
    for stat in ("stdev", "cov"):
        got = m.max_divergent(4, 9, stat, labels=labels)
        exp = oracle.final_max(rows, 4, 9, stat, labels=labels)
        assert [int(labels[p]) for p in got.members().positions] == exp.members()[0].tolist()
    bad = rows.copy()
    bad[3] *= 1.001
    with pytest.raises(ValueError, match="cannot calculate entropy"):
        ctx.matrix_from_freqs(bad)


def test_delta_jsd_calculator(ctx, brca1):
    names = list(brca1)
    refs = [brca1[n][1500:1650] for n in names[:4]]
    queries = [brca1[n][1500:1650] for n in names[4:]]
    m = ctx.build_matrix(refs, 3, 4)
    sel = m.as_set()
    oset = oracle.SummedRecords.from_seqs(refs, 3, 4)
    q = ctx.build_matrix(queries, 3, 4)
    got = sel.delta_jsd(q)
    for i, s in enumerate(queries):
        f, h = oracle.to_kfreqs(s, 4, 3)
        np.testing.assert_allclose(got[i], oset.delta_jsd(f, h), rtol=TIGHT)
    assert sel.delta_jsd(ctx.build_matrix(refs[:1], 3, 4), [0])[0] == 0.0


# ------------------------------------------------------- full BASELINE sizes
def test_config_c2_full_size(ctx):
    """C2: 10k x 2 kb, k=6, nmost -- full comparison with the oracle (~1 s of CPU)"""
    rng = np.random.default_rng(20260423)
    data = rng.integers(0, 4, size=10_000 * 2_000, dtype=np.uint8)
    offs = np.arange(10_001, dtype=np.uint64) * 2_000
    m = ctx.build_matrix_concat(data, offs, 6, 4)
    for n in (10, 100):
        exp, _ = oracle.nmost_concat(data, offs, n, 6, 4)
        s = _assert_selection(m.nmost(n), exp)
        assert s.n_arbitrated == 0
    # size-independent property: every row's counts sum to L - k + 1
    assert (m.totals() == 2_000 - 5).all()


def test_north_star_shape_properties(ctx):
    """100k x 5 kb, k=6: count-matrix checksums + idempotence of the selection"""
    rng = np.random.default_rng(20260424)
    n, L = 100_000, 5_000
    data = rng.integers(0, 4, size=n * L, dtype=np.uint8)
    data[rng.integers(0, n * L, size=n * L // 1000)] = 4  # 0.1 % invalid symbols
    offs = np.arange(n + 1, dtype=np.uint64) * L
    m = ctx.build_matrix_concat(data, offs, 6, 4)
    tot = m.totals()
    sample = rng.integers(0, n, size=64)
    for i in sample:
        c = oracle.count_kmers(data[i * L:(i + 1) * L], 4, 6)
        assert int(tot[i]) == int(c.sum())
        assert (m.counts(int(i), 1)[0] == c).all()
    a = m.nmost(10)
    b = m.nmost(10, window=1 << 14)  # a different window partition must not change the answer
    assert a.members(False).positions.tolist() == b.members(False).positions.tolist()
    np.testing.assert_array_equal(a.members(False).delta_jsd, b.members(False).delta_jsd)
    exp, _ = oracle.nmost_concat(data, offs, 10, 6, 4)
    _assert_selection(a, exp)


# ------------------------------------------------------------------------ mash
def _assert_sketches(seqs, k, s, ns=4, canonical=False):
    from diverseseq_amd import distance

    sk, lens = distance.sketch_batch(seqs, k, s, ns, canonical)
    for i, q in enumerate(seqs):
        exp = oracle.mash_sketch(q, k, s, ns, canonical)
        assert lens[i] == exp.size, (i, lens[i], exp.size)
        assert (sk[i, : lens[i]] == exp).all(), f"sketch {i} differs"
    return sk, lens


@pytest.mark.parametrize("canonical", [False, True])
def test_sketch_brca1(brca1, canonical):
    seqs = list(brca1.values())
    _assert_sketches(seqs, 16, 400, 4, canonical)
    _assert_sketches(seqs[:8], 12, 3000, 4, canonical)       # sketch longer than the sequence
    _assert_sketches(seqs[:8], 5, 4_000_000_000, 4, canonical)  # tests/test_ctree.py: sketch 4e9


def test_sketch_edge_cases():
    rng = np.random.default_rng(8)
    seqs = synth_seqs(20, 900, 5, invalid_frac=0.02, ragged=True)
    seqs += [np.zeros(0, dtype=np.uint8), np.array([1, 2, 3], dtype=np.uint8),
             np.full(40, 4, dtype=np.uint8), np.zeros(300, dtype=np.uint8)]  # homopolymer: 1 distinct
    for k, s in ((8, 50), (4, 10), (21, 100), (1, 3)):
        _assert_sketches(seqs, k, s, 4, False)
        _assert_sketches(seqs, k, s, 4, True)


def test_sketch_long_sequences_threshold_path():
    """genome-like: only hashes under the per-sequence threshold ever leave the hash kernel"""
    rng = np.random.default_rng(13)
    seqs = [rng.integers(0, 4, size=n, dtype=np.uint8) for n in (400_000, 250_000, 20_000)]
    seqs[1][rng.integers(0, 250_000, size=100)] = 4
    seqs.append(np.tile(rng.integers(0, 4, size=500, dtype=np.uint8), 200))  # 100 kb of repeats
    for canonical in (False, True):
        _assert_sketches(seqs, 12, 3000, 4, canonical)
        _assert_sketches(seqs, 16, 400, 4, canonical)


def test_mash_distance_golden(mash_vectors):
    """vectors captured from the reference's pure-python mash_distance"""
    from diverseseq_amd import distance

    for c in mash_vectors["mash_distance"]:
        if c["distance"] == "ZeroDivisionError":
            with pytest.raises(ZeroDivisionError):
                distance.mash_distance(c["left"], c["right"], c["k"], c["sketch_size"])
            continue
        got = distance.mash_distance(c["left"], c["right"], c["k"], c["sketch_size"])
        assert abs(got - c["distance"]) <= RTOL * abs(c["distance"]), c
        assert abs(got - c["distance"]) <= 1e-14 * max(1.0, abs(c["distance"]))


def test_mash_distances_matrix(brca1):
    from diverseseq_amd import distance

    names = ["Human", "Chimpanzee", "Manatee", "Dugong", "Rhesus"]
    seqs = [brca1[n] for n in names]
    d = distance.mash_distances(seqs, 16, 400, 4, True)
    sk = [oracle.mash_sketch(s, 16, 400, 4, True) for s in seqs]
    np.testing.assert_allclose(d, oracle.mash_distances(sk, 16, 400), rtol=1e-13)
    D = {(a, b): d[i, j] for i, a in enumerate(names) for j, b in enumerate(names)}
    # the reference's own assertions (tests/test_distance.py:119-138)
    assert D["Human", "Chimpanzee"] < D["Human", "Dugong"]
    assert D["Human", "Rhesus"] < D["Human", "Manatee"]
    assert D["Chimpanzee", "Rhesus"] < D["Chimpanzee", "Dugong"]
    assert D["Manatee", "Dugong"] < D["Manatee", "Rhesus"]
    # strided chunks (cluster.py:640-644) tile the same lower triangle
    skb, lens = distance.sketch_batch(seqs, 16, 400, 4, True)
    acc = np.zeros((5, 5))
    for start in range(3):
        distance.distances_from_sketches(skb, lens, 16, 400, row_start=start, row_stride=3,
                                         symmetric=False, out=acc)
    np.testing.assert_array_equal(acc + acc.T, d)


def test_euclidean(brca1):
    from diverseseq_amd import distance

    names = ["Human", "Chimpanzee", "Manatee", "Dugong", "Rhesus"]
    seqs = [brca1[n] for n in names]
    d = distance.euclidean_distances(seqs, 5, 4)
    for i in range(5):
        for j in range(5):
            fi, fj = oracle.to_kfreqs(seqs[i], 4, 5)[0], oracle.to_kfreqs(seqs[j], 4, 5)[0]
            np.testing.assert_allclose(d[i, j], np.linalg.norm(fi - fj), rtol=1e-12, atol=1e-15)
    assert d[0, 1] < d[0, 3]  # tests/test_distance.py:37-39


# ------------------------------------------------ every path of the persistent engine
@pytest.mark.parametrize("k,n", [(6, 2), (6, 10), (6, 63), (6, 64), (6, 65), (5, 12), (7, 9), (4, 70)])
def test_persistent_engine_set_sizes_and_bin_counts(ctx, k, n):
    """the persistent engine's variants by set size (one polling wave below 64 members, a grid
    barrier from 64 on), by state size (4^k <= 4096 with the candidate in registers and the f32
    COARSE tier, 4^7 without) and by row mapping (row per workgroup / per wave): selected ids,
    member order and rows bit-exact, floats to 1e-6, against the oracle"""
    seqs = synth_seqs(2500, 900, seed=100 * k + n, ragged=True, invalid_frac=0.002)
    m = ctx.build_matrix(seqs, k, 4)
    sel = m.nmost(n)
    s = _assert_selection(sel, oracle.nmost(seqs, n, k, 4))
    assert s.engine == 1, "the persistent engine should have run"
    sel.close()
    m.close()


@pytest.mark.parametrize("k,min_size,max_size,stat", [(6, 5, 40, "stdev"), (6, 5, 40, "cov"), (5, 3, 12, "stdev"),
                                                      (6, 10, None, "stdev"), (4, 20, 90, "cov")])
def test_persistent_engine_max_mode(ctx, k, min_size, max_size, stat):
    """`dvs max` in the persistent engine: tentative pushes while the set is below max_size (kept iff
    the stat of the members' delta_jsd rose), replace_lowest once it is full; sizes, members, member
    order and floats against the oracle"""
    seqs = synth_seqs(1800, 1000, seed=7 * k + min_size, ragged=True, invalid_frac=0.001)
    mx = len(seqs) if max_size is None else max_size
    m = ctx.build_matrix(seqs, k, 4)
    sel = m.max_divergent(min_size, mx, stat)
    s = _assert_selection(sel, oracle.max_divergent(seqs, min_size, mx, k, 4, stat))
    assert s.engine == 1, "the persistent engine should have run"
    assert s.size > min_size, "the case should exercise kept pushes"
    sel.close()
    m.close()


def _own_composition_seqs(rng, nseq, lo, hi, dead=()):
    """every sequence with a base composition of its own (so nearly every row of a `max` stream is an
    event); the rows listed in `dead` hold no valid k-mer"""
    seqs = []
    for i in range(nseq):
        n = int(rng.integers(lo, hi + 1))
        pr = rng.dirichlet([3.0, 3.0, 3.0, 3.0])
        s = rng.choice(4, size=n, p=pr).astype(np.uint8)
        if i in dead:
            s[:] = 4 if i % 2 else s[:]
            if i % 2 == 0:
                s = s[:3]  # shorter than k
        seqs.append(s)
    return seqs


@pytest.mark.parametrize("k,min_size,max_size,stat", [(6, 30, None, "stdev"), (6, 30, None, "cov"), (5, 8, None, "stdev"),
                                                      (6, 12, 20, "stdev"), (4, 40, 47, "cov"), (6, 100, None, "stdev"),
                                                      (7, 30, None, "stdev"), (7, 9, 40, "cov"), (7, 20, None, "cov")])
def test_max_mode_batches_of_consecutive_events(ctx, k, min_size, max_size, stat):
    """`dvs max` over a stream in which nearly every row is an event (records.rs:390-454): the persistent
    engine takes the rows behind an event along in batches while the set does not change -- rolled-back
    pushes, rows without k-mers inside a batch, a kept push in the middle of one, the switch to
    replace_lowest at max_size, the end of the stream inside a batch -- against the oracle, row for row"""
    rng = np.random.default_rng(900 + 7 * k + min_size)
    # (under cov nearly every push of this stream is kept: the oracle's clone + leave-one-out pass per push grows
    # with the square of the set -- 39 s at 700 rows of 4^7 bins, 4 s at 330)
    nseq = 330 if (k == 7 and stat == "cov" and max_size is None) else 700
    dead = {min_size + 3, min_size + 4, 200, 201, 202, nseq - 50, nseq - 1}
    seqs = _own_composition_seqs(rng, nseq, 3000, 5000, dead)
    mx = nseq if max_size is None else max_size
    m = ctx.build_matrix(seqs, k, 4)
    sel = m.max_divergent(min_size, mx, stat)
    s = _assert_selection(sel, oracle.max_divergent(seqs, min_size, mx, k, 4, stat))
    # 4^7 bins: not the persistent engine's -- the multi-launch kernels' own batches (select.hip: max_batch_*)
    assert s.engine == (1 if k <= 6 else 0)
    if max_size is None and stat == "stdev":  # (under cov nearly every push of this stream is kept: nothing to batch)
        assert s.n_events > 300 and s.n_windows * 4 < s.n_events, (s.n_windows, s.n_events)  # (batches were formed)
    sel.close()
    m.close()


def test_max_mode_batches_with_an_order_and_labels(ctx):
    """the multi-launch batches under a caller's order and labels (duplicated labels: a row whose label is
    already in the set is no event, records.rs:71-74): the oracle's answer"""
    rng = np.random.default_rng(4242)
    seqs = _own_composition_seqs(rng, 400, 2000, 3000, {50, 51, 300})
    order = rng.permutation(400).astype(np.uint32)
    labels = np.arange(400, dtype=np.uint32)
    labels[order[120]] = labels[order[3]]  # two rows share a label with a seed / an early row
    labels[order[121]] = labels[order[40]]
    m = ctx.build_matrix(seqs, 6, 4)
    lab = labels[order]
    sel = m.max_divergent(20, 400, "stdev", order=order, labels=lab)
    exp = oracle.max_divergent([seqs[i] for i in order], 20, 400, 6, 4, "stdev", labels=lab.tolist())
    gm, s = sel.members(False), sel.summary()
    assert s.size == exp.size and s.size > 20
    assert [int(lab[p]) for p in gm.positions] == exp.members()[0].tolist()
    np.testing.assert_allclose(gm.delta_jsd, exp.members()[1], rtol=RTOL, atol=1e-13)
    assert abs(s.total_jsd - exp.total_jsd) <= RTOL * exp.total_jsd
    assert s.engine == 0 and s.n_windows * 3 < s.n_events, (s.n_windows, s.n_events)  # (batches were formed)
    sel.close()
    m.close()


def test_randomised_small_selections(ctx):
    """sixty small random problems -- short sequences (many near-ties: the arbiter and the engines'
    hand-overs get their share), duplicates, invalid symbols, every mode -- against the oracle"""
    rng = np.random.default_rng(20261003)
    engines = set()
    for case in range(60):
        nseq = int(rng.integers(12, 400))
        length = int(rng.integers(24, 320))
        k = int(rng.integers(2, 7))
        n = int(rng.integers(2, min(12, nseq - 1)))
        seqs = synth_seqs(nseq, length, seed=1000 + case, ragged=bool(case & 1), invalid_frac=0.01 if case % 3 == 0 else 0.0)
        if case % 5 == 0:  # a few exact duplicates in the stream
            for _ in range(3):
                seqs[int(rng.integers(0, nseq))] = seqs[int(rng.integers(0, nseq))].copy()
        m = ctx.build_matrix(seqs, k, 4)
        mode = case % 3
        if mode == 0:
            sel, exp = m.nmost(n), oracle.nmost(seqs, n, k, 4)
        else:
            stat = "stdev" if mode == 1 else "cov"
            mx = nseq if case % 2 else min(nseq, n + int(rng.integers(0, 20)))
            sel, exp = m.max_divergent(n, mx, stat), oracle.max_divergent(seqs, n, mx, k, 4, stat)
        s = _assert_selection(sel, exp)
        engines.add(s.engine)
        sel.close()
        m.close()
    assert engines == {0, 1} or engines == {1}


def test_sum_to_one_panic_is_the_references(ctx):
    """k = 1 (tolerance 4 eps): after a few dozen pushes the running mean no longer sums to one
    within the reference's check (record.rs:99-104) and `dvs max` panics there -- so must this
    path, with the same message (case 212 of scripts/micro/stress_selections.py, seed 7)"""
    seqs = synth_seqs(169, 67, seed=251128967)
    with pytest.raises(oracle.OraclePanic) as exp:
        oracle.max_divergent(seqs, 40, 78, 1, 4, "cov")
    m = ctx.build_matrix(seqs, 1, 4)
    with pytest.raises(ValueError) as got:
        m.max_divergent(40, 78, "cov")
    assert str(got.value) == str(exp.value)
    assert "cannot calculate entropy as frequency vector total" in str(got.value)
    m.close()


def test_context_may_be_closed_before_its_objects():
    """matrices, selections and sequence batches hold a reference on their context: closing the
    context first (what interpreter shutdown may do to objects in a cycle) leaves them usable
    and their own close() safe (include/dvs_hip.h, dvs_ctx_destroy)"""
    from diverseseq_amd import engine

    own = engine.Context()
    seqs = synth_seqs(300, 400, seed=5)
    exp = oracle.nmost(seqs, 5, 4, 4)
    m = own.build_matrix(seqs, 4, 4)
    sel = m.nmost(5)
    batch = own.encode_fasta(b">a\nACGT\n>b\nGGTA\n")
    own.close()
    sel.close()
    m.close()
    batch.close()
    # and a fresh context works afterwards
    c2 = engine.Context()
    m2 = c2.build_matrix(seqs, 4, 4)
    sel2 = m2.nmost(5)
    _assert_selection(sel2, exp)
    sel2.close(); m2.close(); c2.close()


@pytest.mark.parametrize("env", [{"DVS_PERSIST_WG_ROUNDS": "0"}, {"DVS_PERSIST_WG_ROUNDS": "100000"},
                                 {"DVS_PERSIST_NO_COARSE": "1"}, {"DVS_NO_PERSIST": "1"}])
def test_engine_knobs_do_not_change_the_answer(ctx, env, monkeypatch):
    """row-per-wave only, row-per-workgroup only, without the COARSE tier, and the multi-launch
    engine: one selection, one answer"""
    seqs = synth_seqs(6000, 1200, seed=4242, ragged=True)
    exp = oracle.nmost(seqs, 10, 6, 4)
    for key, val in env.items():
        monkeypatch.setenv(key, val)
    m = ctx.build_matrix(seqs, 6, 4)
    sel = m.nmost(10)
    _assert_selection(sel, exp)
    sel.close()
    m.close()


# ------------------------------------------------------------ _dvs drop-in level
def test_selection_through_an_on_disk_store(brca1, tmp_path):
    """`dvs nmost` / `dvs max` over a .dvseqsz directory (diverseseq_amd/zarr_store.py): the store a
    `prep` step wrote is reopened read-only and gives what the in-memory store and the oracle give"""
    from diverseseq_amd import _dvs as dvs

    path = str(tmp_path / "brca1.dvseqsz")
    disk = dvs.make_zarr_store(path, mode="w")
    mem = dvs.make_zarr_store()
    for name, arr in brca1.items():
        disk.write(name, arr.tobytes(), {"source": f"brca1:{name}"})
        mem.write(name, arr.tobytes())
    disk.close()
    ro = dvs.make_zarr_store(path, mode="r")
    ids = dvs.get_seqids_from_store(path)
    assert ids == list(brca1)
    a = dvs.nmost_divergent(ro, n=10, k=4, seqids=ids)
    b = dvs.nmost_divergent(mem, n=10, k=4, seqids=ids)
    exp = oracle.nmost([brca1[i] for i in ids], 10, 4, 4)
    assert a.record_names == b.record_names == [ids[i] for i in exp.members()[0]]
    assert a.total_jsd == b.total_jsd
    np.testing.assert_allclose(a.total_jsd, exp.total_jsd, rtol=1e-12)
    c = dvs.max_divergent(ro, min_size=5, max_size=12, k=3, seqids=ids)
    d = dvs.max_divergent(mem, min_size=5, max_size=12, k=3, seqids=ids)
    assert c.record_names == d.record_names and 5 <= c.size <= 12


def test_dvs_module_like_reference_tests(brca1):
    """reference tests/test_records.py through the drop-in module"""
    from diverseseq_amd import _dvs as dvs

    data = {"a": "AAAA", "b": "AAAA", "c": "TTTT", "d": "ACGT"}
    st = dvs.make_zarr_store()
    for n, s in data.items():
        st.write(n, str2arr(s).tobytes())
    assert st.unique_seqids == ["b", "c", "d"]
    for k in (1, 2):  # test_total_jsd
        lz = [st.get_lazyseq(n, num_states=4) for n in st.unique_seqids]
        sr = dvs.get_delta_jsd_calculator([(s.seqid, s.get_seq()) for s in lz], k=k,
                                          num_states=4).get_result()
        freqs = np.array([s.get_kfreqs(k) for s in lz])

        def H(p):
            p = p[p > 0]
            return float(-(p * np.log2(p)).sum())

        np.testing.assert_allclose(sr.total_jsd, H(freqs.mean(0)) - np.mean([H(f) for f in freqs]))
    got = dvs.max_divergent(st, min_size=2, max_size=2, k=1)
    assert got.size == 2
    got = dvs.nmost_divergent(st, n=3, k=1)
    assert got.size == 3 and set(got.record_names) == set(st.unique_seqids)
    assert pickle.loads(pickle.dumps(got)).size == 3
    with pytest.raises(ValueError):
        dvs.nmost_divergent(st, n=30, k=1)
    # brca1: chunk + merge (test_merge_summed_records)
    store = dvs.make_zarr_store()
    for n, s in brca1.items():
        store.write(n, s.tobytes())
    names = store.unique_seqids
    sr1 = dvs.nmost_divergent(store, n=5, k=1, seqids=names[:10])
    sr2 = dvs.nmost_divergent(store, n=5, k=1, seqids=names[10:20])
    merged = dvs.final_nmost([sr1, sr2], n=5)
    assert len(merged.record_names) == 5
    assert (merged.k, merged.num_states) == (4, 1)  # records.rs:353 swaps them
    with pytest.raises(ValueError):
        dvs.final_nmost([sr1, sr2], n=500)
    mx = dvs.max_divergent(store, min_size=4, max_size=5, k=1, seqids=names[:10])
    with pytest.raises(ValueError):
        dvs.final_max([mx], min_size=10, max_size=20, stat="stdev")
    # calculator edge cases (test_jsd_calc*)
    recs = [(n, brca1[n][1500:1650].tobytes()) for n in names[:4]]
    calc = dvs.get_delta_jsd_calculator(recs, k=3, num_states=4)
    assert calc.get_result().total_jsd > 0
    assert np.allclose(calc.delta_jsd(*recs[0]), 0.0)
    with pytest.raises(ValueError, match="failed: No valid k-mers"):
        calc.delta_jsd("blah", b"")
    sk = dvs.mash_sketch(brca1["Human"].tobytes(), 16, 400, 4, True)
    assert sk == oracle.mash_sketch(brca1["Human"], 16, 400, 4, True).tolist()
    lz = store.get_lazyseq("Human", 4)
    assert lz.get_kcounts(3) == oracle.count_kmers(brca1["Human"], 4, 3).tolist()


# ------------------------------------------------------------------ tie arbiter
def test_degenerate_inputs_take_the_arbiter(ctx):
    """identical / duplicated sequences: every decision is an exact tie up to rounding
    noise, so the device defers to the exact host replay and still matches the oracle"""
    rng = np.random.default_rng(42)
    base = rng.integers(0, 4, size=400, dtype=np.uint8)
    same = [base.copy() for _ in range(30)]
    m = ctx.build_matrix(same, 2, 4)
    sel = m.nmost(5)
    s = _assert_selection(sel, oracle.nmost(same, 5, 2, 4))
    uniq = synth_seqs(40, 300, 17, ragged=True)
    dup = [u for u in uniq for _ in range(2)]  # each sequence twice, different ids
    for k, n in ((1, 4), (3, 6)):
        m = ctx.build_matrix(dup, k, 4)
        _assert_selection(m.nmost(n), oracle.nmost(dup, n, k, 4))
        _assert_selection(m.max_divergent(3, 9, "stdev"), oracle.max_divergent(dup, 3, 9, k, 4, "stdev"))
        _assert_selection(m.max_divergent(3, 9, "cov"), oracle.max_divergent(dup, 3, 9, k, 4, "cov"))
    toy = [str2arr(x) for x in ("AAAA", "TTTT", "ACGT", "AAAA", "TTTA", "CCGG", "ACGT", "GGGG")]
    m = ctx.build_matrix(toy, 1, 4)
    for n in (2, 3, 4):
        _assert_selection(m.nmost(n), oracle.nmost(toy, n, 1, 4))
    with pytest.raises(NotImplementedError, match="ambiguous decision"):
        from diverseseq_amd import _lib
        ctx.build_matrix(same, 2, 4).nmost(5, flags=_lib.SELECT_NO_ARBITER)


def test_fast_log2_error_bound(ctx):
    """the hardware term of the scan's fast tier: |v_log_f32(m) - log2 m| over EVERY f32
    mantissa in [0.5, 1) must stay below what FAST_BAND (select.hip, 4e-7) budgets:
    mantissa rounding to f32 (8.6e-8) + this + margin"""
    import ctypes as C

    err = C.c_double()
    ctx.check(ctx._L.dvs_selftest_fast_log2(ctx._h, C.byref(err)))
    assert 0.0 < err.value < 1.5e-7, err.value
    print("max |v_log_f32 - log2| on [0.5,1):", err.value)


def test_log2_acc_error_bound(ctx):
    """the f64 log2 of the precise evaluations agrees with ocml's to ~1 ulp of the result"""
    import ctypes as C

    err = C.c_double()
    ctx.check(ctx._L.dvs_selftest_log2_acc(ctx._h, C.byref(err)))
    print("max rel |log2_acc - log2|:", err.value)
    assert 0.0 <= err.value < 1e-15, err.value


def test_coarse_tier_log2_error_bound(ctx):
    """the hardware term of the persistent engine's f32 tier: v_log_f32 over EVERY f32 in
    [2^-101, 2) stays within the 1.5 ulp (of max(1, |log2 y|)) that COARSE_BAND budgets"""
    import ctypes as C

    k = C.c_double()
    ctx.check(ctx._L.dvs_selftest_log2_f32(ctx._h, C.byref(k)))
    print("max |v_log_f32 - log2| in ulps of the result:", k.value)
    assert 0.0 < k.value <= 1.5, k.value


def test_exact_quotient_by_fma(ctx):
    """count / total by fma (no f64 division) is the correctly rounded quotient, bit for bit"""
    import ctypes as C

    bad = C.c_uint64(1)
    ctx.check(ctx._L.dvs_selftest_exact_div(ctx._h, C.byref(bad)))
    assert bad.value == 0


def test_handover_words_of_the_persistent_engine(ctx):
    """The persistent engine's cross-workgroup words on their own: 200 000 synthetic windows -- arrival
    records, hints, listed candidates, the gathering block's release and a use of the never-cleared leave-one-out
    accumulators each -- with pseudo-random pauses in front of every step and contributions every workgroup can
    recompute: no workgroup ever reads a word that is not the one it must be, and no spin runs out."""
    import ctypes as C

    bad = C.c_uint64(1)
    ctx.check(ctx._L.dvs_selftest_handover(ctx._h, 200_000, C.byref(bad)))
    assert bad.value == 0, (bad.value >> 32, bad.value & 0xFFFFFFFF)


# ------------------------------------------------ genome-scale rows (configs C3 / C5, scaled down)
def test_genome_length_sequences_max_and_sketch(ctx):
    """C3 / C5 shapes at reduced N: ~3 Mb sequences (92 tiles each), `max` min_size 5 and
    k=12 / s=3000 sketches, all against the oracle"""
    from diverseseq_amd import distance

    rng = np.random.default_rng(31)
    seqs = []
    for i in range(16):
        p = rng.dirichlet(np.ones(4) * 5.0)
        s = rng.choice(4, size=int(rng.integers(2_500_000, 3_500_000)), p=p).astype(np.uint8)
        s[rng.integers(0, s.size, size=300)] = 4
        seqs.append(s)
    m = ctx.build_matrix(seqs, 6, 4)
    exp_counts = np.stack([oracle.count_kmers(s, 4, 6) for s in seqs[:3]])
    assert (m.counts(0, 3).astype(np.uint64) == exp_counts).all()
    _assert_selection(m.max_divergent(5, 16, "stdev"), oracle.max_divergent(seqs, 5, 16, 6, 4, "stdev"))
    _assert_selection(m.nmost(4), oracle.nmost(seqs, 4, 6, 4))
    sk, lens = distance.sketch_batch(seqs[:4], 12, 3000, 4, True)
    for i in range(4):
        assert (sk[i, : lens[i]] == oracle.mash_sketch(seqs[i], 12, 3000, 4, True)).all()


def test_config_c4_k7(ctx):
    """C4's bin count (k=7, 16384 bins: 64 KB LDS histogram, 128 KB state vector) at 3000 rows"""
    seqs = synth_seqs(3000, 5000, 44, invalid_frac=0.0005)
    m = ctx.build_matrix(seqs, 7, 4)
    assert (m.counts(17, 2).astype(np.uint64) == _oracle_counts(seqs[17:19], 4, 7)).all()
    sel = m.nmost(25)
    assert sel.summary().engine == 1  # 128 KB of set state in LDS: the persistent engine still fits
    _assert_selection(sel, oracle.nmost(seqs, 25, 7, 4))
    sel = m.nmost(70)  # more members than fit one wave's argmin, candidate row not register-cached
    assert sel.summary().engine == 1
    _assert_selection(sel, oracle.nmost(seqs, 70, 7, 4))


def test_persistent_engine_odd_bin_counts(ctx):
    """bins beyond the register cache and not a multiple of 256 (20 states, k=3: 8000 bins):
    the persistent engine's scalar row path"""
    rng = np.random.default_rng(4)
    prot = [rng.integers(0, 21, size=int(rng.integers(900, 2500)), dtype=np.uint8) for _ in range(700)]
    m = ctx.build_matrix(prot, 3, 20)
    sel = m.nmost(9)
    assert sel.summary().engine == 1
    _assert_selection(sel, oracle.nmost(prot, 9, 3, 20))


# ------------------------------------------------------ exact row-sharded mode
def _degenerate_seqs():
    """sequences in which exact ties abound: blocks of identical copies and near-copies (one base changed)
    of a few short sequences, a handful of unrelated ones in between"""
    rng = np.random.default_rng(77)
    base = [rng.integers(0, 4, size=int(rng.integers(150, 260)), dtype=np.uint8) for _ in range(6)]
    seqs = []
    for i in range(420):
        b = base[int(rng.integers(0, len(base)))].copy()
        r = rng.random()
        if r < 0.35:
            b[int(rng.integers(0, b.size))] = (b[0] + 1) % 4  # a near-copy
        elif r < 0.45:
            b = rng.integers(0, 4, size=int(rng.integers(150, 260)), dtype=np.uint8)
        seqs.append(b)
    return seqs


def _equal_length_seqs(nseq=3000, length=3000, seed=321):
    """i.i.d. uniform sequences of ONE length (the benchmark's kind of stream): every row is as likely as any earlier one to
    be the most divergent so far, so a greedy selection of n keeps accepting -- ~n ln(N / n) events; with ragged lengths the
    shortest sequences (the noisiest k-mer frequencies) win early and hardly anything follows"""
    rng = np.random.default_rng(seed)
    return [rng.integers(0, 4, size=length, dtype=np.uint8) for _ in range(nseq)]


def _exact_worker(rank, world, port, q, degenerate=False, k6=False):
    import os
    import sys

    sys.path.insert(0, str(__import__("conftest").ROOT))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist

    from diverseseq_amd import engine, parallel

    torch.cuda.set_device(0)  # both ranks share the one GPU of the test box; gloo moves the words
    dev = torch.device("cuda", 0)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        ctx = engine.Context(0, stream=stream.cuda_stream)
        seqs = _degenerate_seqs() if degenerate else synth_seqs(1500, 600, 123, invalid_frac=0.001, ragged=True)
        n, k = (7, 3) if degenerate else (12, 5)
        if k6:  # rows of 4096 bins: the fast step's whole-row requests and its all-f32 tier
            seqs = _equal_length_seqs()
            n, k = 10, 6
        owned, order = parallel.shard_order(len(seqs), n, rank, world, block=32)
        local = seqs[:n] + [seqs[int(p)] for p in owned]
        m = ctx.build_matrix(local, k, 4)
        out = [rank]
        sel = parallel.nmost_exact(ctx, m, order, n, dev, world, window=256 * world, poll_every=4)
        mem = sel.members(with_freqs=False)
        s = sel.summary()
        out.append((mem.positions.tolist(), mem.delta_jsd.tolist(), s.total_jsd))
        for stat in (() if k6 else ("stdev", "cov")):  # the same rows, select_max_divergent: the set grows from n
            sel = parallel.max_exact(ctx, m, order, n, 40, stat, dev, world, window=256 * world, poll_every=4)
            mem = sel.members(with_freqs=False)
            s = sel.summary()
            out.append((mem.positions.tolist(), mem.delta_jsd.tolist(), s.total_jsd))
            out.append(int(s.n_arbitrated))
        q.put(tuple(out))
    if world > 1:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [1, 2])
def test_exact_row_sharded_mode(world):
    """rows sharded over ranks + replicated set state + ONE all_gather per greedy step (every rank's
    first event and its candidate row) must give the single-process answer (ids bit-exact), for
    select_nmost_divergent and select_max_divergent (records.rs:311-454)"""
    import socket

    import torch.multiprocessing as mp

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    procs = [mpc.Process(target=_exact_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    seqs = synth_seqs(1500, 600, 123, invalid_frac=0.001, ragged=True)
    exps = [oracle.nmost(seqs, 12, 5, 4), oracle.max_divergent(seqs, 12, 40, 5, 4, "stdev"),
            oracle.max_divergent(seqs, 12, 40, 5, 4, "cov")]
    for r in res:
        for (pos, delta, total), exp in zip((r[1], r[2], r[4]), exps):
            elab, edelta, _, _ = exp.members()
            assert pos == elab.tolist()
            np.testing.assert_allclose(delta, edelta, rtol=RTOL, atol=1e-13)
            np.testing.assert_allclose(total, exp.total_jsd, rtol=RTOL)
    assert res[0][1:] == res[-1][1:]  # replicas are bit-identical


@pytest.mark.parametrize("world", [1, 2])
def test_exact_row_sharded_mode_4096_bins(world):
    """The same at k = 6 (rows of 4096 16-bit counts, n = 10: the north-star's shape in small): the fast step requests a
    wave's whole row at once -- its first row before the step's decisions are known --, scores it in the all-f32 tier
    first, and skips the positions another rank owns; ids bit-exact, delta_jsd and total_jsd within the tolerance."""
    import socket

    import torch.multiprocessing as mp

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    procs = [mpc.Process(target=_exact_worker, args=(r, world, port, q, False, True)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    seqs = _equal_length_seqs()
    exp, acc = oracle.nmost_concat(*__import__("diverseseq_amd.engine", fromlist=["concat"]).concat(seqs), 10, 6, 4)
    elab, edelta, _, _ = exp.members()
    assert acc >= 20  # (the stream held events: the steps ran)
    for r in res:
        pos, delta, total = r[1]
        assert pos == elab.tolist()
        np.testing.assert_allclose(delta, edelta, rtol=RTOL, atol=1e-13)
        np.testing.assert_allclose(total, exp.total_jsd, rtol=RTOL)
    assert res[0][1:] == res[-1][1:]  # replicas are bit-identical


@pytest.mark.parametrize("world", [1, 2])
def test_exact_row_sharded_mode_arbitrates_ties(world):
    """Degenerate input (identical and near-identical sequences: decisions inside the rounding band) in
    the exact row-sharded mode: every rank's host arbiter replays the same event log -- from the row log
    the step kernels keep, since the rows themselves may live on another rank -- and reaches the same
    verdict with no exchange; the answer is the one-process oracle's (records.rs:86-92,231,246-249)."""
    import socket

    import torch.multiprocessing as mp

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    procs = [mpc.Process(target=_exact_worker, args=(r, world, port, q, True)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    seqs = _degenerate_seqs()
    exps = [oracle.nmost(seqs, 7, 3, 4), oracle.max_divergent(seqs, 7, 40, 3, 4, "stdev"),
            oracle.max_divergent(seqs, 7, 40, 3, 4, "cov")]
    for r in res:
        for (pos, delta, total), exp in zip((r[1], r[2], r[4]), exps):
            elab, edelta, _, _ = exp.members()
            assert pos == elab.tolist()
            np.testing.assert_allclose(delta, edelta, rtol=RTOL, atol=1e-13)
            np.testing.assert_allclose(total, exp.total_jsd, rtol=RTOL)
        assert r[3] + r[5] > 0, "no decision went to the arbiter: the input is not degenerate enough"
    assert res[0][1:] == res[-1][1:]


def _mash_shard_worker(rank, world, port, q):
    import os
    import sys

    sys.path.insert(0, str(__import__("conftest").ROOT))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist

    from diverseseq_amd import parallel

    dist.init_process_group("gloo", rank=rank, world_size=world)
    seqs = synth_seqs(41, 3000, 77, invalid_frac=0.002, ragged=True)
    d = parallel.mash_distances_sharded(seqs, 10, 200, rank, world, torch.device("cpu"),
                                        mash_canonical=True)
    q.put((rank, d))
    dist.destroy_process_group()


def _mash_shard_device_worker(rank, world, port, q):
    """the device-resident path (sketches, gathered sketches and the N x N matrix stay in HBM); the two ranks share
    the test box's one card, so the collectives run over gloo through host copies of the device tensors"""
    import os
    import sys

    sys.path.insert(0, str(__import__("conftest").ROOT))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist

    from diverseseq_amd import parallel

    dist.init_process_group("gloo", rank=rank, world_size=world)

    def gather(out, inp):
        o, i = out.cpu(), inp.cpu()
        dist.all_gather_into_tensor(o, i)
        out.copy_(o)

    def reduce_(t, op):
        c = t.cpu()
        dist.all_reduce(c, op=op)
        t.copy_(c)

    seqs = synth_seqs(41, 3000, 77, invalid_frac=0.002, ragged=True)  # (41 sequences: chunks of 21 and 20 -- a padded row)
    d = parallel.mash_distances_sharded(seqs, 10, 200, rank, world, torch.device("cuda:0"), mash_canonical=True,
                                        collectives=(gather, reduce_))
    q.put((rank, d))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [1, 2])
def test_sharded_mash_distances_device_resident(world):
    """the same with nothing but the result crossing PCIe: dvs_sketches_copy_to_device -> [all_gather] ->
    dvs_sketches_from_device -> dvs_sketches_distances_device -> [all_reduce] -> symmetrised on the device"""
    import socket

    import torch.multiprocessing as mp

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    procs = [mpc.Process(target=_mash_shard_device_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=240) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    seqs = synth_seqs(41, 3000, 77, invalid_frac=0.002, ragged=True)
    exp = oracle.mash_distances([oracle.mash_sketch(x, 10, 200, 4, True) for x in seqs], 10, 200)
    for _, d in res:
        np.testing.assert_allclose(d, exp, rtol=1e-12, atol=0)
    np.testing.assert_array_equal(res[0][1], res[-1][1])


@pytest.mark.parametrize("world", [1, 2])
def test_sharded_mash_distances(world):
    """ctree over ranks on the HIP kernels: chunked dvs_mash_sketch, all_gather,
    strided dvs_mash_distances rows, SUM all-reduce == one-process matrix == oracle"""
    import socket

    import torch.multiprocessing as mp

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    procs = [mpc.Process(target=_mash_shard_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=240) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    seqs = synth_seqs(41, 3000, 77, invalid_frac=0.002, ragged=True)
    exp = oracle.mash_distances([oracle.mash_sketch(x, 10, 200, 4, True) for x in seqs], 10, 200)
    for _, d in res:
        np.testing.assert_allclose(d, exp, rtol=1e-12, atol=0)
    np.testing.assert_array_equal(res[0][1], res[-1][1])


def test_final_merge_identity_labels_persistent_engine(ctx, brca1):
    """frequency-row matrix + identity labels: the persistent engine's f64-row instantiation"""
    seqs = list(brca1.values())
    rows = np.vstack([oracle.nmost(seqs[i:i + 11], 4, 4, 4).members(with_freqs=True)[3]
                      for i in range(0, 55, 11)])
    m = ctx.matrix_from_freqs(rows)
    got = m.nmost(4)
    assert got.summary().engine == 1
    exp = oracle.final_nmost(rows, 4)
    assert got.members().positions.tolist() == exp.members()[0].tolist()
    np.testing.assert_allclose(got.members().delta_jsd, exp.members()[1], rtol=RTOL, atol=1e-13)


def test_device_side_chunk_merge(ctx, brca1):
    """the multi-GPU merge without a host round trip: dvs_select_gather_members into the
    all_gather buffer, dvs_matrix_from_device_freqs on the gathered rows (padding rows of a short
    chunk are skipped), final_nmost -- against the oracle's chunk + merge"""
    import torch

    from diverseseq_amd.parallel import _global_ids

    seqs = list(brca1.values())
    bounds = [(0, 25), (25, 28), (28, 55)]  # the middle chunk has fewer than n sequences
    n, cap, k = 5, 5, 4
    B = 4 ** k
    dev = torch.device("cuda:0")
    all_rows = torch.full((3 * cap, B), -7.0, dtype=torch.float64, device=dev)
    all_meta = torch.full((3 * cap, 2), -7.0, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    erows, eids = [], []
    for r, (a, b) in enumerate(bounds):
        m = ctx.build_matrix(seqs[a:b], k, 4)
        sel = m.nmost(min(n, b - a))
        sel.gather_members(all_rows[r * cap:].data_ptr(), all_meta[r * cap:].data_ptr(), cap)
        ctx.sync()
        o = oracle.nmost(seqs[a:b], min(n, b - a), k, 4)
        l, _, _, f = o.members(with_freqs=True)
        erows.append(f)
        eids.append(l + a)
    erows, eids = np.vstack(erows), np.concatenate(eids)
    meta = all_meta.cpu().numpy()
    assert meta[:, 1].tolist() == [1] * 5 + [1] * 3 + [0] * 2 + [1] * 5
    np.testing.assert_array_equal(all_rows.cpu().numpy()[meta[:, 1] != 0], erows)
    assert not all_rows.cpu().numpy()[meta[:, 1] == 0].any()
    gids = _global_ids(all_meta, [a for a, _ in bounds], cap)
    assert gids[gids >= 0].tolist() == eids.tolist()
    mm = ctx.matrix_from_device_freqs(all_rows.data_ptr(), 3 * cap, B, all_meta.data_ptr())
    # the matrix keeps the real rows first, in their gathered order, and the padding behind them
    src = mm.source_rows()
    assert src.tolist() == [0, 1, 2, 3, 4, 5, 6, 7, 10, 11, 12, 13, 14, 8, 9]
    gids = _global_ids(all_meta, [a for a, _ in bounds], cap, src_rows=src)
    got = mm.nmost(n)
    exp = oracle.final_nmost(erows, n, labels=eids)
    gm = got.members()
    assert [int(gids[p]) for p in gm.positions] == exp.members()[0].tolist()
    np.testing.assert_allclose(gm.delta_jsd, exp.members()[1], rtol=RTOL, atol=1e-13)
    np.testing.assert_allclose(got.summary().total_jsd, exp.total_jsd, rtol=RTOL)
    with pytest.raises(ValueError, match="buffer of"):
        sel.gather_members(all_rows.data_ptr(), all_meta.data_ptr(), 2)


def test_device_side_chunk_merge_short_first_chunk(ctx, brca1):
    """the FIRST chunk holds fewer than n sequences: the merge still seeds from the first n real
    records of the concatenated results (get_kmerseqs_and_init_summed_records, records.rs:344-360),
    not from the first n rows of the padded gather"""
    import torch

    from diverseseq_amd.parallel import _global_ids

    seqs = list(brca1.values())
    bounds = [(0, 3), (3, 30), (30, 55)]
    n, cap, k = 5, 5, 4
    B = 4 ** k
    dev = torch.device("cuda:0")
    all_rows = torch.zeros((3 * cap, B), dtype=torch.float64, device=dev)
    all_meta = torch.zeros((3 * cap, 2), dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    erows, eids = [], []
    for r, (a, b) in enumerate(bounds):
        m = ctx.build_matrix(seqs[a:b], k, 4)
        sel = m.nmost(min(n, b - a))
        sel.gather_members(all_rows[r * cap:].data_ptr(), all_meta[r * cap:].data_ptr(), cap)
        ctx.sync()
        l, _, _, f = oracle.nmost(seqs[a:b], min(n, b - a), k, 4).members(with_freqs=True)
        erows.append(f)
        eids.append(l + a)
    erows, eids = np.vstack(erows), np.concatenate(eids)
    mm = ctx.matrix_from_device_freqs(all_rows.data_ptr(), 3 * cap, B, all_meta.data_ptr())
    gids = _global_ids(all_meta, [a for a, _ in bounds], cap, src_rows=mm.source_rows())
    got = mm.nmost(n)
    exp = oracle.final_nmost(erows, n, labels=eids)
    assert got.summary().size == n == exp.size
    gm = got.members()
    assert [int(gids[p]) for p in gm.positions] == exp.members()[0].tolist()
    np.testing.assert_allclose(gm.delta_jsd, exp.members()[1], rtol=RTOL, atol=1e-13)


@pytest.mark.parametrize("stat", ["stdev", "cov"])
def test_device_side_chunk_merge_max(ctx, brca1, stat):
    """final_max (select_max_divergent_final, records.rs:456-507) over winners that never leave HBM:
    three chunks' `max` sets of different sizes gathered into padded device buffers, wrapped by
    dvs_matrix_from_device_freqs and merged on the device -- against the oracle's final_max over the same rows"""
    import torch

    from diverseseq_amd.parallel import _global_ids

    seqs = list(brca1.values())
    bounds = [(0, 20), (20, 24), (24, 55)]  # the middle chunk is smaller than min_size
    lo, hi, cap, k = 5, 9, 9, 4
    B = 4 ** k
    dev = torch.device("cuda:0")
    all_rows = torch.full((3 * cap, B), -3.0, dtype=torch.float64, device=dev)
    all_meta = torch.full((3 * cap, 2), -3.0, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    erows, eids = [], []
    for r, (a, b) in enumerate(bounds):
        m = ctx.build_matrix(seqs[a:b], k, 4)
        sel = m.max_divergent(min(lo, b - a), hi, stat)
        sel.gather_members(all_rows[r * cap:].data_ptr(), all_meta[r * cap:].data_ptr(), cap)
        ctx.sync()
        o = oracle.max_divergent(seqs[a:b], min(lo, b - a), hi, k, 4, stat)
        l, _, _, f = o.members(with_freqs=True)
        erows.append(f)
        eids.append(l + a)
    erows, eids = np.vstack(erows), np.concatenate(eids)
    mm = ctx.matrix_from_device_freqs(all_rows.data_ptr(), 3 * cap, B, all_meta.data_ptr())
    gids = _global_ids(all_meta, [a for a, _ in bounds], cap, src_rows=mm.source_rows())
    got = mm.max_divergent(lo, hi, stat)
    exp = oracle.final_max(erows, lo, hi, stat, labels=eids)
    gm = got.members()
    assert got.summary().size == exp.size
    assert [int(gids[p]) for p in gm.positions] == exp.members()[0].tolist()
    np.testing.assert_allclose(gm.delta_jsd, exp.members()[1], rtol=RTOL, atol=1e-13)
    np.testing.assert_allclose(got.summary().total_jsd, exp.total_jsd, rtol=RTOL)


def test_merge_nmost_over_rccl_world1(ctx, brca1):
    """parallel.merge_nmost end to end on the RCCL backend (world 1): device path == host path"""
    import torch
    import torch.distributed as dist

    from diverseseq_amd.parallel import merge_nmost

    seqs = list(brca1.values())
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    if not dist.is_initialized():
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29617", rank=0, world_size=1)
    try:
        m = ctx.build_matrix(seqs, 4, 4)
        sel = m.nmost(6)
        a = merge_nmost(ctx, sel, 6, 0, 1, 100, dev, chunk_starts=[100])
        b = merge_nmost(ctx, sel, 6, 0, 1, 100, dev)
        pa, pb = a.members(False).positions, b.members(False).positions
        assert [int(a.global_ids[p]) for p in pa] == [int(b.global_ids[p]) for p in pb]
        assert a.summary().total_jsd == b.summary().total_jsd
        exp = oracle.final_nmost(oracle.nmost(seqs, 6, 4, 4).members(with_freqs=True)[3], 6)
        assert pa.tolist() == exp.members()[0].tolist()
    finally:
        dist.destroy_process_group()


def _chunk_merge_worker(rank, world, port, q):
    import os
    import sys

    sys.path.insert(0, str(__import__("conftest").ROOT))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist

    from diverseseq_amd import engine, parallel

    torch.cuda.set_device(0)  # the ranks share the one GPU of the test box; gloo moves the words
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    stream = torch.cuda.Stream()
    out = [rank]
    with torch.cuda.stream(stream):
        ctx = engine.Context(0, stream=stream.cuda_stream)
        seqs = synth_seqs(907, 500, 31, invalid_frac=0.002, ragged=True)
        n, k = 8, 4
        bounds = parallel.chunk_bounds(len(seqs), world)
        lo, hi = bounds[rank]
        buffers = {}
        for _ in range(2):  # the second pass reuses the exchange buffers, as bench.py's steps do
            m = ctx.build_matrix(seqs[lo:hi], k, 4)
            sel = m.nmost(n)
            merged = parallel.merge_nmost(ctx, sel, n, rank, world, lo, dev, chunk_starts=[b[0] for b in bounds],
                                          shared_stream=True, buffers=buffers)
            mem = merged.members(with_freqs=False)
            gids = merged.global_ids
            out.append(([int(gids[p]) for p in mem.positions], mem.delta_jsd.tolist(), merged.summary().total_jsd,
                        (sel.summary().engine, merged.summary().engine)))
            merged.close()
            sel.close()
            m.close()
    q.put(tuple(out))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_chunk_merge_mode_across_ranks(world):
    """what bench.py does with more than one rank (the reference's `-np G`: contiguous chunks, an
    independent selection per rank, final_nmost over the winners): device-side gather, all_gather,
    device-side merge on every rank == the oracle's chunk + merge, twice with the buffers reused"""
    import socket

    import torch.multiprocessing as mp

    from diverseseq_amd import parallel

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    procs = [mpc.Process(target=_chunk_merge_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    seqs = synth_seqs(907, 500, 31, invalid_frac=0.002, ragged=True)
    rows, ids = [], []
    for lo, hi in parallel.chunk_bounds(len(seqs), world):
        lab, _, _, f = oracle.nmost(seqs[lo:hi], 8, 4, 4).members(with_freqs=True)
        rows.append(f)
        ids.append(lab.astype(np.int64) + lo)
    rows, ids = np.vstack(rows), np.concatenate(ids)
    exp = oracle.final_nmost(rows, 8)
    epos, edelta, _, _ = exp.members()
    # (ids exact everywhere; floats to the tolerance: the ranks share ONE card here, and a persistent
    # launch that cannot get its whole grid resident beside another process's hands the selection to
    # the multi-launch engine, whose last bits differ -- the engines used travel with the result)
    for r in res:
        for got_ids, got_delta, got_total, _engines in r[1:]:
            assert got_ids == ids[epos].tolist()
            np.testing.assert_allclose(got_delta, edelta, rtol=RTOL, atol=1e-13)
            np.testing.assert_allclose(got_total, exp.total_jsd, rtol=RTOL)
        if r[1][3] == r[2][3]:
            assert r[1] == r[2], "same engines, same bits"


def test_large_sets_and_other_alphabets(ctx):
    """n beyond what the persistent engine replicates in LDS (multi-launch engine), a 20-state
    alphabet (k=2, 400 bins: not a multiple of 256) and a tiny one (3 states, k=3)"""
    seqs = synth_seqs(1400, 260, 909, ragged=True)
    m = ctx.build_matrix(seqs, 3, 4)
    sel = m.nmost(600)
    assert sel.summary().engine == 0
    _assert_selection(sel, oracle.nmost(seqs, 600, 3, 4))
    rng = np.random.default_rng(21)
    prot = [rng.integers(0, 21, size=int(rng.integers(150, 600)), dtype=np.uint8) for _ in range(500)]
    mp = ctx.build_matrix(prot, 2, 20)
    _assert_selection(mp.nmost(15), oracle.nmost(prot, 15, 2, 20))
    _assert_selection(mp.max_divergent(6, 30, "stdev"), oracle.max_divergent(prot, 6, 30, 2, 20, "stdev"))
    tri = [rng.integers(0, 4, size=int(rng.integers(60, 300)), dtype=np.uint8) for _ in range(400)]
    mt = ctx.build_matrix(tri, 3, 3)
    _assert_selection(mt.nmost(8), oracle.nmost(tri, 8, 3, 3))


@pytest.mark.parametrize("k,ns,n,nseq,length", [(6, 4, 100, 2500, 3000), (6, 4, 2, 1500, 2000), (3, 4, 5, 3000, 300),
                                                 (1, 4, 3, 2000, 200), (2, 20, 6, 2500, 400), (5, 4, 13, 3000, 1500),
                                                 (4, 3, 7, 2000, 500)])
def test_exact_mode_fast_step_shapes(ctx, k, ns, n, nseq, length):
    """The stepwise mode's two-launch step (fs_jobs_kernel / fs_step_kernel) over the shapes its paths split on: rows of
    4096 bins with a set of 100 (one part a job, the members' sums through LDS in several rounds) and of 2 (a
    leave-one-out set of one member); bin counts below a workgroup's threads and not a multiple of 256 (k = 1, 3; 20 and 3
    states): the row-form-agnostic scan and the state writer's LDS path; 1024 bins with 13 members.  ids bit-exact,
    delta_jsd / total_jsd within the tolerance, the accept count the oracle's (records.rs:311-342)."""
    import torch

    from diverseseq_amd import engine, parallel

    rng = np.random.default_rng(1000 * k + n)
    seqs = [rng.integers(0, ns, size=length, dtype=np.uint8) for _ in range(nseq)]
    dev = torch.device("cuda", 0)
    m = ctx.build_matrix(seqs, k, ns)
    _, order = parallel.shard_order(len(seqs), n, 0, 1, block=32)
    sel = parallel.nmost_exact(ctx, m, order, n, dev, 1)
    exp, acc = oracle.nmost_concat(*engine.concat(seqs), n, k, ns)
    summ = sel.summary()
    assert summ.n_accepts == acc and acc >= 1, (summ.n_accepts, acc)
    elab, edelta, _, _ = exp.members()
    mem = sel.members(with_freqs=False)
    assert mem.positions.tolist() == elab.tolist()
    np.testing.assert_allclose(mem.delta_jsd, edelta, rtol=RTOL, atol=1e-13)
    np.testing.assert_allclose(summ.total_jsd, exp.total_jsd, rtol=RTOL)
    sel.close()
    m.close()


def test_exact_mode_status_history(ctx):
    """dvs_select_step_peek: the engine's status behind an earlier apply launch, read from the history the step kernel keeps
    in pinned host memory -- no sync.  Driven one step at a time with a look at the launch before the last: running until the
    launch that found the stream's end, done from there on (steps behind the end are no-ops that still write their word) --
    i.e. after exactly events + 2 launches; a lag longer than the launches so far reads running; a selection the fast
    step does not drive (max_divergent) keeps no history and the driver polls.  (test_exact_row_sharded_mode drives every
    world size through the peeking loop and checks its answer.)"""
    import torch

    from diverseseq_amd import _lib, parallel

    seqs = synth_seqs(1500, 600, 123, invalid_frac=0.001, ragged=True)
    dev = torch.device("cuda", 0)
    m = ctx.build_matrix(seqs, 5, 4)
    _, order = parallel.shard_order(len(seqs), 12, 0, 1, block=32)
    sel = m.select(_lib.MODE_NMOST, 12, order=order, window=256, flags=_lib.SELECT_STEPWISE)
    st = parallel.HipStepper(ctx, sel, m.nbins, dev)
    assert not st.done()
    assert st.peek(4) == (0, False)  # (nothing that far back yet)
    steps = 0
    while True:
        st.apply(st.pack(), 1)
        steps += 1
        assert st.peek(steps + 3) == (0, False)
        status, must = st.peek(1)
        if status == 1:
            break
        assert status == 0 and steps < 2000
        if must:  # (half the accepted rows' ring: the syncing poll drains it)
            assert not st.done()
    assert st.done()
    summ = sel.summary()
    exp = oracle.nmost(seqs, 12, 5, 4)
    assert sel.members(with_freqs=False).positions.tolist() == exp.members()[0].tolist()
    if summ.n_arbitrated == 0:
        assert steps == summ.n_events + 2 and summ.n_events > 10, (steps, summ.n_events)
    sel.close()
    sel = m.select(_lib.MODE_MAX, 5, max_size=20, stat=_lib.STAT_STDEV, order=order, window=256, flags=_lib.SELECT_STEPWISE)
    st = parallel.HipStepper(ctx, sel, m.nbins, dev)
    assert not st.done()
    assert st.peek(4) is None
    sel.close()
    m.close()


def test_exact_mode_row_log_has_no_cap(ctx, monkeypatch):
    """The stepwise (row-sharded) mode keeps the frequency row of every accepted event for the tie arbiter: a ring on
    the device that dvs_select_step_poll drains into a host-side log without a cap.  With a ring of eight rows
    (DVS_TEST_KNOBS=rowlog_ring_8) the degenerate stream's accepts wrap it many times before and between its
    arbitrations: the oracle's answer, arbitrations > 0; and a driver that lets more than half a ring's worth of steps pass
    between two polls (the ring is drained once half full) is refused, not left to a log with holes."""
    import ctypes as C

    import torch

    from diverseseq_amd import _lib, engine, parallel

    monkeypatch.setenv("DVS_TEST_KNOBS", "rowlog_ring_8")  # (the autouse fixture has every live context re-read its switches)
    if True:
        seqs = _degenerate_seqs()
        n, k = 7, 3
        dev = torch.device("cuda", 0)
        _, order = parallel.shard_order(len(seqs), n, 0, 1, block=32)
        m = ctx.build_matrix(seqs, k, 4)
        sel = parallel.nmost_exact(ctx, m, order, n, dev, 1, window=256, poll_every=4)
        s = sel.summary()
        exp, acc = oracle.nmost_concat(*engine.concat(seqs), n, k, 4)
        assert s.n_accepts == acc and acc > 3 * 8, (s.n_accepts, acc)  # (the ring went round several times)
        assert s.n_arbitrated > 0
        elab, edelta, _, _ = exp.members()
        mem = sel.members(with_freqs=False)
        assert mem.positions.tolist() == elab.tolist()
        np.testing.assert_allclose(mem.delta_jsd, edelta, rtol=RTOL, atol=1e-13)
        sel.close()
        # too many steps between two polls: refused
        sel = m.select(_lib.MODE_NMOST, n, order=order, window=256, flags=_lib.SELECT_STEPWISE)
        stepper = parallel.HipStepper(ctx, sel, m.nbins, dev)
        with pytest.raises(ValueError, match="dvs_select_step_poll must be called at least every 4 steps"):
            for _ in range(5):
                stepper.apply(stepper.pack(), 1)
        sel.close()
        m.close()
