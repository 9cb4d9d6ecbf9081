"""Round-2 parity cases on the GPU: the BASELINE.json configurations at their stated workloads
against the oracle, the drop-in module on the persistent engine, the engine's fall-back, and
candidates built to sit inside the decision bands of the f32 tiers."""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

import oracle
from conftest import synth_seqs
from gpu_synth import synth_device
from test_gpu_parity import RTOL, _assert_selection

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from diverseseq_amd import engine

    return engine.default_context()


# ---------------------------------------------------------------- drop-in module, fast engine
def _store_of(seqs):
    from diverseseq_amd import _dvs as dvs

    st = dvs.make_zarr_store()
    for i, s in enumerate(seqs):
        st.write(f"s{i:06d}", s.tobytes())
    return st


def test_dvs_module_runs_the_persistent_engine():
    """reference src/lib.rs:59-73,105-137 through the drop-in: the ids of a store are unique, so the
    labels _dvs passes are all distinct and the selection runs on the persistent engine"""
    from diverseseq_amd import _dvs as dvs

    seqs = synth_seqs(1500, 1200, 77, invalid_frac=0.001)
    st = _store_of(seqs)
    ids = st.unique_seqids
    assert len(ids) == len(seqs)
    got = dvs.nmost_divergent(st, n=12, k=5)
    assert got.stats["engine"] == 1
    exp = oracle.nmost(seqs, 12, 5, 4)
    elab, edelta, _, efreq = exp.members(with_freqs=True)
    assert got.record_names == [ids[i] for i in elab]
    np.testing.assert_allclose([r[2] for r in got.records], edelta, rtol=RTOL, atol=1e-13)
    assert (np.array([r[1] for r in got.records]) == efreq).all()
    assert abs(got.total_jsd - exp.total_jsd) <= RTOL * exp.total_jsd
    # a caller-chosen order of the same ids: still distinct labels, still the persistent engine
    perm = np.random.default_rng(5).permutation(len(ids))
    got = dvs.nmost_divergent(st, n=7, k=4, seqids=[ids[i] for i in perm])
    assert got.stats["engine"] == 1
    exp = oracle.nmost([seqs[i] for i in perm], 7, 4, 4)
    assert got.record_names == [ids[perm[i]] for i in exp.members()[0]]
    for stat in ("stdev", "cov"):
        got = dvs.max_divergent(st, min_size=5, max_size=40, k=5, stat=stat)
        assert got.stats["engine"] == 1
        exp = oracle.max_divergent(seqs, 5, 40, 5, 4, stat)
        assert got.record_names == [ids[i] for i in exp.members()[0]]
        assert abs(got.std_delta_jsd - exp.std_delta_jsd) <= RTOL * abs(exp.std_delta_jsd)
    # the chunk merge: final_nmost over results whose ids are distinct
    a = dvs.nmost_divergent(st, n=6, k=5, seqids=ids[:700])
    b = dvs.nmost_divergent(st, n=6, k=5, seqids=ids[700:])
    merged = dvs.final_nmost([a, b], n=6)
    assert merged.stats["engine"] == 1
    rows = np.array([r[1] for r in a.records] + [r[1] for r in b.records])
    exp = oracle.final_nmost(rows, 6)
    names = a.record_names + b.record_names
    assert merged.record_names == [names[i] for i in exp.members()[0]]
    # ... and final_max (src/lib.rs:139-160) over two max results
    ma = dvs.max_divergent(st, min_size=4, max_size=10, k=5, seqids=ids[:700])
    mb = dvs.max_divergent(st, min_size=4, max_size=10, k=5, seqids=ids[700:])
    mm = dvs.final_max([ma, mb], min_size=4, max_size=12, stat="stdev")
    assert mm.stats["engine"] == 1
    rows = np.array([r[1] for r in ma.records] + [r[1] for r in mb.records])
    exp = oracle.final_max(rows, 4, 12, "stdev")
    names = ma.record_names + mb.record_names
    assert mm.record_names == [names[i] for i in exp.members()[0]]
    assert abs(mm.total_jsd - exp.total_jsd) <= RTOL * exp.total_jsd


def test_repeated_ids_keep_the_label_aware_engine():
    """an id listed twice (records.rs:71-73: a member's own id scores 0.0) needs the labels: the
    multi-launch engine serves it, with the reference's answer"""
    from diverseseq_amd import _dvs as dvs

    seqs = synth_seqs(400, 900, 78)
    st = _store_of(seqs)
    ids = st.unique_seqids
    order = list(range(400)) + list(range(0, 400, 7))
    got = dvs.nmost_divergent(st, n=8, k=4, seqids=[ids[i] for i in order])
    assert got.stats["engine"] == 0
    exp = oracle.nmost([seqs[i] for i in order], 8, 4, 4, labels=np.array(order, dtype=np.uint32))
    assert got.record_names == [ids[i] for i in exp.members()[0]]


def test_caller_labels_come_back_from_the_label_free_engine(ctx):
    """distinct but arbitrary labels: the engine runs without them, the caller gets them back"""
    seqs = synth_seqs(600, 800, 79)
    labels = (np.random.default_rng(1).permutation(600) * 7 + 3).astype(np.uint32)
    m = ctx.build_matrix(seqs, 5, 4)
    sel = m.nmost(9, labels=labels)
    assert sel.summary().engine == 1
    mem = sel.members(False)
    exp = oracle.nmost(seqs, 9, 5, 4)
    assert mem.positions.tolist() == exp.members()[0].tolist()
    assert mem.labels.tolist() == labels[mem.positions.astype(np.int64)].tolist()
    # delta_jsd against that selection speaks the caller's labels too: a member scores 0.0
    q = ctx.build_matrix([seqs[int(mem.positions[0])], seqs[5]], 5, 4)
    d = sel.delta_jsd(q, [int(mem.labels[0]), 0xFFFFFFF0])
    assert d[0] == 0.0 and d[1] != 0.0


def test_persistent_engine_falls_back_to_the_multi_launch_engine(ctx, monkeypatch):
    """a persistent launch that gives up at a grid barrier (workgroups not co-resident) is not an
    error: the selection starts over from its seeds on the multi-launch engine"""
    seqs = synth_seqs(2500, 700, 80)
    m = ctx.build_matrix(seqs, 5, 4)
    exp = oracle.nmost(seqs, 11, 5, 4)
    sel = m.nmost(11)
    assert sel.summary().engine == 1
    _assert_selection(sel, exp)
    monkeypatch.setenv("DVS_TEST_KNOBS", "fake_persist_error")  # (test-only: the first launch's outcome is read as SEL_ERROR)
    sel = m.nmost(11)
    assert sel.summary().engine == 0
    _assert_selection(sel, exp)
    sel = m.max_divergent(4, 30, "stdev")
    assert sel.summary().engine == 0
    _assert_selection(sel, oracle.max_divergent(seqs, 4, 30, 5, 4, "stdev"))


# ---------------------------------------------------------------- both count widths
@pytest.mark.parametrize("u32", [False, True])
def test_count_rows_in_both_widths(ctx, u32, monkeypatch):
    """whole sequences (<= 32768 windows each) build 16-bit count rows, DVS_COUNTS_U32 keeps them at
    32 bits: the same counts, entropies and selections either way (record.rs:41-84, records.rs:311-454),
    on both engines and at 4^6 and 4^7 bins; a genome-length row keeps the matrix at 32 bits"""
    if u32:
        monkeypatch.setenv("DVS_COUNTS_U32", "1")
    width = 4 if u32 else 2
    seqs = synth_seqs(3000, 3000, 55, invalid_frac=0.001, ragged=True)
    seqs[7] = np.full(32768 + 2, 1, dtype=np.uint8)  # a single tile for every k >= 3 below: one bin takes ~32768 counts
    for k, n in ((6, 10), (6, 70), (7, 12), (3, 6), (5, 9)):
        m = ctx.build_matrix(seqs, k, 4)
        assert m.count_bytes == (4 if k == 7 else width)  # (16-bit rows up to 4096 bins)
        got = m.counts().astype(np.uint64)
        for i in (0, 7, 1500, 2999):
            assert (got[i] == oracle.count_kmers(seqs[i], 4, k)).all()
        H = m.entropy()
        for i in (0, 7, 2999):
            assert abs(H[i] - oracle.to_kfreqs(seqs[i], 4, k)[1]) <= 1e-11 * max(1.0, H[i])
        exp = oracle.nmost(seqs, n, k, 4)
        sel = m.nmost(n)
        assert sel.summary().engine == 1
        _assert_selection(sel, exp)
        monkeypatch.setenv("DVS_NO_PERSIST", "1")
        sel = m.nmost(n)
        assert sel.summary().engine == 0
        _assert_selection(sel, exp)
        monkeypatch.delenv("DVS_NO_PERSIST")
        if k == 6:
            _assert_selection(m.max_divergent(5, 60, "stdev"), oracle.max_divergent(seqs, 5, 60, k, 4, "stdev"))
            from diverseseq_amd import distance

            d = distance.euclidean_distances(seqs[:40], k, 4, ctx=ctx)  # (tiled kernel, either width)
            fr = [oracle.to_kfreqs(s_, 4, k)[0] for s_ in seqs[:40]]
            for i, j in ((1, 0), (7, 3), (39, 38), (20, 7)):
                assert abs(d[i, j] - oracle.euclidean_distance(fr[i], fr[j])) <= 1e-9
                assert d[i, j] == d[j, i]
    # a sequence of more than one tile: 32-bit rows whatever the knob says
    m = ctx.build_matrix(seqs[:20] + [np.zeros(40_000, dtype=np.uint8)], 6, 4)
    assert m.count_bytes == 4
    assert int(m.counts(20, 1)[0][0]) == 40_000 - 5
    # other alphabets take the same path (20 states, k=2: 400 bins)
    prot = [np.random.default_rng(i).integers(0, 21, size=900, dtype=np.uint8) for i in range(300)]
    m = ctx.build_matrix(prot, 2, 20)
    assert m.count_bytes == width
    assert (m.counts(5, 1)[0] == oracle.count_kmers(prot[5], 20, 2)).all()
    _assert_selection(m.nmost(7), oracle.nmost(prot, 7, 2, 20))


# ---------------------------------------------------------------- candidates inside the bands
COARSE_BAND_K6 = 1.25 * 2.0**-24 * (9.0 * 12 + 7.4)  # select_dev.h coarse_band(4096)
FAST_BAND = 4e-7                                       # select_dev.h


def _craft(oset, start, target, k, rng, tol):
    """substitutions of `start` (one to three bases at a time) until the oracle's score of the
    sequence against `oset` sits at threshold + target (records.rs:70-92) within tol.  Equal-length
    sequences give the score a grain of ~1e-8, so tol is a few of those (0.08 of the band), not zero."""
    seq = start.copy()
    thr = oset.total_jsd

    def margin(s):
        f, h = oracle.to_kfreqs(s, 4, k)
        return oset.delta_jsd(f, h) - thr

    cur = margin(seq)
    for _ in range(60_000):
        if abs(cur - target) <= tol:
            return seq, cur
        m = int(rng.integers(1, 4))
        pos = rng.integers(0, seq.size, size=m)
        old = seq[pos].copy()
        seq[pos] = (old + rng.integers(1, 4, size=m)) % 4
        d = margin(seq)
        if abs(d - target) < abs(cur - target):
            cur = d
        else:
            seq[pos] = old
    raise AssertionError(f"no sequence found at margin {target}: stuck at {cur}")


def _band_stream(k, n, length, nprefix, targets, seed):
    """a random prefix, then for every target margin a near-copy of the set's lowest member whose exact
    score is threshold + margin (each followed by a few ordinary rows), the oracle tracking the set"""
    rng = np.random.default_rng(seed)
    seqs = synth_seqs(nprefix, length, seed)
    oset = oracle.nmost(seqs, n, k, 4)
    margins = []
    for t in targets:
        labels = oset.members()[0]
        low = seqs[int(labels[oset.lowest_index])]
        cand, got = _craft(oset, low, t, k, rng, tol=0.08 * (FAST_BAND if abs(t) < 1e-6 else COARSE_BAND_K6))
        f, h = oracle.to_kfreqs(cand, 4, k)
        if oset.increases_jsd(f, h, len(seqs)):
            oset.replace_lowest(f, h, len(seqs))
        seqs.append(cand)
        margins.append(got)
        for s in synth_seqs(3, length, int(rng.integers(1 << 30))):
            f, h = oracle.to_kfreqs(s, 4, k)
            if oset.increases_jsd(f, h, len(seqs)):
                oset.replace_lowest(f, h, len(seqs))
            seqs.append(s)
    return seqs, oset, margins


MULTS = (-1.1, -0.9, -0.5, -0.1, 0.1, 0.5, 0.9, 1.1)


@pytest.fixture(scope="module")
def band_k6():
    targets = [m * COARSE_BAND_K6 for m in MULTS] + [m * FAST_BAND for m in MULTS]
    return _band_stream(6, 10, 5000, 700, targets, 91)


@pytest.fixture(scope="module")
def band_k7():
    return _band_stream(7, 10, 5000, 300, [m * FAST_BAND for m in (-0.5, -0.1, 0.1, 0.5)], 92)  # (16384-bin scores: ~3 s each)


@pytest.mark.parametrize("env", [{}, {"DVS_PERSIST_WG_ROUNDS": "0"}, {"DVS_PERSIST_WG_ROUNDS": "100000"},
                                 {"DVS_NO_PERSIST": "1"}])
def test_candidates_inside_the_f32_bands_k6(ctx, band_k6, env, monkeypatch):
    """reference src/records.rs:86-92 (`jsd > total_jsd + EPSILON`) for candidates whose exact score is
    the threshold +- {0.1, 0.5, 0.9, 1.1} x COARSE_BAND and x FAST_BAND: the f32 tiers may not decide
    them, the f64 tier must, and the selected ids stay the oracle's.  Both row mappings of the
    persistent engine and the multi-launch engine."""
    for k_, v_ in env.items():
        monkeypatch.setenv(k_, v_)
    seqs, oset, margins = band_k6
    exp = oracle.nmost(seqs, 10, 6, 4)
    assert exp.members()[0].tolist() == oset.members()[0].tolist()  # (the tracked set is the stream's)
    nprefix = 700
    m = ctx.build_matrix(seqs, 6, 4)
    sel = m.nmost(10)
    s = _assert_selection(sel, exp)
    assert s.engine == (0 if "DVS_NO_PERSIST" in env else 1)
    assert s.n_arbitrated == 0
    # the same prefix without the crafted rows: what the bands cost on ordinary input
    m0 = ctx.build_matrix(seqs[:nprefix], 6, 4)
    s0 = m0.nmost(10).summary()
    sure_fast = sum(1 for g in margins[len(MULTS):] if abs(g) <= 0.6 * FAST_BAND)
    assert sure_fast >= 4
    assert s.rows_rechecked >= s0.rows_rechecked + sure_fast, (s.rows_rechecked, s0.rows_rechecked)
    if s.engine == 1:
        sure_coarse = sum(1 for g in margins if abs(g) <= 0.6 * COARSE_BAND_K6)
        assert sure_coarse >= 4 + len(MULTS)
        assert s.rows_coarse_passed >= s0.rows_coarse_passed + sure_coarse, (s.rows_coarse_passed, s0.rows_coarse_passed)


@pytest.mark.parametrize("env", [{}, {"DVS_NO_PERSIST": "1"}])
def test_candidates_inside_the_fast_band_k7(ctx, band_k7, env, monkeypatch):
    """the same at k=7 (16384 bins: no COARSE tier, the FAST tier decides or lists every row)"""
    for k_, v_ in env.items():
        monkeypatch.setenv(k_, v_)
    seqs, oset, margins = band_k7
    exp = oracle.nmost(seqs, 10, 7, 4)
    assert exp.members()[0].tolist() == oset.members()[0].tolist()
    m = ctx.build_matrix(seqs, 7, 4)
    s = _assert_selection(m.nmost(10), exp)
    assert s.n_arbitrated == 0
    s0 = ctx.build_matrix(seqs[:300], 7, 4).nmost(10).summary()
    sure = sum(1 for g in margins if abs(g) <= 0.6 * FAST_BAND)
    assert s.rows_rechecked >= s0.rows_rechecked + sure >= 4


# ---------------------------------------------------------------- stated configurations
@pytest.mark.parametrize("composition", [True, False])
def test_config_c3_scaled_1050_genomes_vs_oracle(ctx, composition):
    """C3 at N = 1050 (SURVEY 8d's CI variant): genomes of 2.5-3.5 Mb, k=6, `max` min_size=100,
    max_size=N, stdev -- select_max_divergent (records.rs:390-454) in full against the oracle (~10 s of
    CPU: 3.15 Gbases counted, ~1000 tentative pushes).  composition=True: every genome with its own base
    composition; False: i.i.d. uniform genomes, whose k-mer spectra agree to 1e-3 -- every tentative push
    is a near-tie (thousands of rows through the f64 tier, a handful of decisions through the host
    arbiter, records.rs:86-92,427-451)."""
    seqs, offs = synth_device(1050, 2_500_000, 3_500_000, 20260430, composition=composition)
    total = int(offs[-1])
    m = ctx.build_matrix_device(seqs.data_ptr(), offs, 6, 4)
    sel = m.max_divergent(100, 1050, "stdev")
    host = seqs[:total].cpu().numpy()
    exp = oracle.max_divergent_concat(host, offs, 100, 1050, 6, 4, "stdev")
    s = _assert_selection(sel, exp)
    assert s.size >= 100
    tot = m.totals()
    assert (tot.astype(np.int64) == np.diff(offs.astype(np.int64)) - 5).all()
    for i in (0, 523, 1049):
        c = oracle.count_kmers(host[int(offs[i]):int(offs[i + 1])], 4, 6)
        assert (m.counts(i, 1)[0] == c).all()


def test_config_c3_full_size_properties(ctx):
    """C3 as stated: 10.5k genomes of ~3 Mb (31.5 Gbases resident in HBM), k=6, `max` min_size=100.
    Too big for the oracle in seconds, so size-independent properties: row totals, spot rows against
    the oracle's count_kmers, the same members from a different window partition and from the
    multi-launch engine, and the prefix of the stream reproducing the oracle-checked scaled case's
    kind of answer (members are stream positions in range, no duplicates, delta_jsd finite)."""
    N = 10_500
    seqs, offs = synth_device(N, 2_500_000, 3_500_000, 20260431, composition=True)
    m = ctx.build_matrix_device(seqs.data_ptr(), offs, 6, 4)
    tot = m.totals()
    assert (tot.astype(np.int64) == np.diff(offs.astype(np.int64)) - 5).all()
    for i in (0, 7777, N - 1):
        a, b = int(offs[i]), int(offs[i + 1])
        c = oracle.count_kmers(seqs[a:b].cpu().numpy(), 4, 6)
        assert (m.counts(i, 1)[0] == c).all()
    a = m.max_divergent(100, N, "stdev")
    sa, ma = a.summary(), a.members(False)
    assert sa.size >= 100 and len(set(ma.positions.tolist())) == sa.size
    assert np.isfinite(ma.delta_jsd).all() and int(ma.positions.max()) < N
    b = m.max_divergent(100, N, "stdev", window=777)  # another window partition: the same answer
    mb = b.members(False)
    assert ma.positions.tolist() == mb.positions.tolist()
    np.testing.assert_allclose(ma.delta_jsd, mb.delta_jsd, rtol=1e-9)
    os.environ["DVS_NO_PERSIST"] = "1"
    ctx.refresh_knobs()
    try:
        c = m.max_divergent(100, N, "stdev")  # the other engine: the same answer
    finally:
        del os.environ["DVS_NO_PERSIST"]
        ctx.refresh_knobs()
    assert c.summary().engine == 0 and sa.engine == 1
    mc = c.members(False)
    assert ma.positions.tolist() == mc.positions.tolist()
    np.testing.assert_allclose(ma.delta_jsd, mc.delta_jsd, rtol=1e-9)
    # the members' own rows, taken alone, form a set with the same statistics (get_result is a
    # function of the members only): make_summed_records over them in member order
    rows = [seqs[int(offs[p]):int(offs[p + 1])].cpu().numpy() for p in ma.positions[:sa.size]]
    oset = oracle.SummedRecords.from_seqs(rows, 6, 4)
    assert abs(oset.total_jsd - sa.total_jsd) <= RTOL * oset.total_jsd
    np.testing.assert_allclose(ma.delta_jsd, oset.members()[1], rtol=RTOL, atol=1e-13)


def test_config_c4_full_input_k7_n100_vs_oracle(ctx):
    """C4's input on one GPU: 100k x 5 kb, k=7 (16384 bins, a 6.55 GB count matrix), nmost n=100, in
    full against the oracle (~30 s of CPU)"""
    seqs, offs = synth_device(100_000, 5000, 5000, 20260432)
    m = ctx.build_matrix_device(seqs.data_ptr(), offs, 7, 4)
    sel = m.nmost(100)
    host = seqs[:int(offs[-1])].cpu().numpy()
    exp, acc = oracle.nmost_concat(host, offs, 100, 7, 4)
    s = _assert_selection(sel, exp)
    assert s.engine == 1 and s.n_arbitrated == 0
    assert s.n_accepts == acc


def test_config_c5_100_genomes_sketches_and_matrix_vs_oracle(ctx):
    """C5 at N = 100: mash sketches (k=12, s=3000, plain and canonical) of ~3 Mb genomes bit-exact
    against the oracle's mash_sketch (distance.rs:151-182), and the full N x N distance matrix against
    the oracle's restatement of mash_distance (distance.py:230-291)"""
    from diverseseq_amd import distance

    N = 100
    seqs, offs = synth_device(N, 2_900_000, 3_100_000, 20260433, composition=True)
    host = seqs[:int(offs[-1])].cpu().numpy()
    rows = [host[int(offs[i]):int(offs[i + 1])] for i in range(N)]
    for canonical in (False, True):
        sk, lens = distance.sketch_batch(rows, 12, 3000, 4, canonical, ctx=ctx)
        with ThreadPoolExecutor(8) as ex:  # (the C oracle releases the GIL)
            exp = list(ex.map(lambda r: oracle.mash_sketch(r, 12, 3000, 4, canonical), rows))
        for i in range(N):
            assert int(lens[i]) == exp[i].size
            assert (sk[i, :lens[i]] == exp[i]).all(), f"sketch {i} differs (canonical={canonical})"
        d = distance.distances_from_sketches(sk, lens, 12, 3000, ctx=ctx)
        de = oracle.mash_distances(exp, 12, 3000)
        np.testing.assert_allclose(d, de, rtol=RTOL, atol=0)
        assert (d == d.T).all() and (np.diag(d) == 0).all()


def _random_sketches(rng, n, s, pool):
    """n sorted rows of s distinct hashes out of a pool of `pool` values (pairs share about s^2 / pool)"""
    space = np.sort(rng.choice(2**32 - 1, size=pool, replace=False).astype(np.uint32))
    sk = np.zeros((n, s), dtype=np.uint32)
    for i in range(n):
        sk[i] = np.sort(rng.choice(space, size=s, replace=False))
    return sk


def test_config_c5_pairs_at_1000_sketches_vs_oracle(ctx):
    """C5's distance stage at its stated size -- 1000 sketches of 3000 hashes, k=12, 499 500 pairs -- against the
    oracle's mash_distance (distance.py:230-291) on 5000 sampled pairs and on every pair of the rows that are
    not full (shorter sketches, an empty one: the waves holding them take the kernel's general path, all others
    the exactly-s-steps path), then the strided row subsets of cluster.py:640-644."""
    from diverseseq_amd import distance

    rng = np.random.default_rng(20260434)
    N, S, K = 1000, 3000, 12
    sk = _random_sketches(rng, N, S, 24_000)
    lens = np.full(N, S, dtype=np.uint32)
    short = {5: 2990, 63: 100, 64: 0, 300: 1, 777: 2999, 999: 1500}
    for i, n in short.items():
        lens[i] = n
        sk[i, n:] = 0
    d = distance.distances_from_sketches(sk, lens, K, S, ctx=ctx)
    assert (d == d.T).all() and (np.diag(d) == 0).all()
    pairs = {(int(i), int(j)) for i, j in zip(rng.integers(1, N, 5000), rng.integers(0, N, 5000)) if j < i}
    pairs |= {(max(i, j), min(i, j)) for i in short for j in range(N) if i != j}
    inter = 0
    for i, j in pairs:
        e = oracle.mash_distance(sk[i, :lens[i]], sk[j, :lens[j]], K, S)
        assert d[i, j] == e or abs(d[i, j] - e) <= RTOL * abs(e), (i, j, d[i, j], e)
        inter += 0.0 < e < 1.0
    assert inter > 4000  # (the sample is not all zeros and ones)
    acc = np.zeros((N, N))
    for start in range(3):
        distance.distances_from_sketches(sk, lens, K, S, row_start=start, row_stride=3, symmetric=False, out=acc, ctx=ctx)
    np.testing.assert_array_equal(acc + acc.T, d)


def test_pairs_of_sketches_longer_than_the_staged_row(ctx):
    """sketches of more than 8192 hashes: the shared row is read through L1 instead of LDS"""
    from diverseseq_amd import distance

    rng = np.random.default_rng(20260435)
    N, S = 70, 9000
    sk = _random_sketches(rng, N, S, 40_000)
    lens = np.full(N, S, dtype=np.uint32)
    lens[3], lens[69] = 8000, 8193
    d = distance.distances_from_sketches(sk, lens, 16, S, ctx=ctx)
    for i in range(1, N):
        for j in range(i):
            e = oracle.mash_distance(sk[i, :lens[i]], sk[j, :lens[j]], 16, S)
            assert d[i, j] == e or abs(d[i, j] - e) <= RTOL * abs(e), (i, j, d[i, j], e)


@pytest.mark.parametrize("k,nseq,n,reps", [(7, 12_500, 100, 400), (6, 100_000, 10, 150)])
def test_a_repeated_selection_gives_the_same_answer_every_time(ctx, k, nseq, n, reps):
    """ONE selection repeated over the same sequences: accepts, arbitrations, events, windows and the members
    must never change.  (Sets of 64 members and more read their leave-one-out totals behind a grid barrier:
    partials sent as non-returning atomic adds were overtaken by the barrier arrival in ~0.7 % of the
    repetitions at the C4 share's shape, and a third of those ended with a wrong set -- persist.hip,
    grid_barrier.  The first parameter set is that shape, the second the north star's.)"""
    seqs, offs = synth_device(nseq, 5000, 5000, 20260440 + k)
    seen = {}
    for i in range(reps):
        m = ctx.build_matrix_device(seqs.data_ptr(), offs, k, 4)
        sel = m.nmost(n)
        s, mem = sel.summary(), sel.members(False)
        key = (s.n_accepts, s.n_arbitrated, s.n_events, s.n_windows, repr(s.total_jsd), mem.positions.tobytes())
        seen[key] = seen.get(key, 0) + 1
        sel.close()
        m.close()
    assert len(seen) == 1, sorted((v, k_[:5]) for k_, v in seen.items())


@pytest.mark.parametrize("k", [6, 7])
def test_a_repeated_max_selection_gives_the_same_answer_every_time(ctx, k):
    """the same for `max` over a stream of back-to-back events: the persistent engine's batches (k = 6) and
    the multi-launch kernels' (k = 7) hand their jobs' results over behind one rendezvous / between two launches"""
    seqs, offs = synth_device(900, 20_000, 30_000, 20260450 + k, composition=True)
    seen = {}
    for i in range(120):
        m = ctx.build_matrix_device(seqs.data_ptr(), offs, k, 4)
        sel = m.max_divergent(40, 900, "stdev")
        s, mem = sel.summary(), sel.members(False)
        key = (s.size, s.n_accepts, s.n_arbitrated, s.n_events, repr(s.total_jsd), mem.positions.tobytes())
        seen[key] = seen.get(key, 0) + 1
        sel.close()
        m.close()
    assert len(seen) == 1, sorted((v, k_[:5]) for k_, v in seen.items())


# ---------------------------------------------------------------- head phase (CU split)
def _device_build(ctx, seqs, k):
    import torch

    data, offs = oracle.concat(seqs)
    t = torch.from_numpy(np.concatenate([data, np.zeros(16, np.uint8)])).to("cuda:0")
    torch.cuda.synchronize()
    return ctx.build_matrix_device(t.data_ptr(), offs, k, 4), t


@pytest.mark.parametrize("k,n", [(6, 10), (5, 7), (6, 40)])
@pytest.mark.parametrize("bad_seeds", [(), (0, 3)])
def test_head_phase_of_a_split_build_vs_oracle(ctx, k, n, bad_seeds):
    """A device-resident build of more than 17k short sequences is split (head rows first, the rest on
    the CU-masked stream) and the persistent engine walks the head rows on the head CUs meanwhile: two persistent
    launches, the oracle's answer -- also when some of the first rows have no valid k-mer and are
    skipped as seeds (records.rs:299-306)."""
    seqs = synth_seqs(20_000, 260, seed=4242 + k + n, ragged=True, invalid_frac=0.001)
    for i, r in enumerate(bad_seeds):
        seqs[r] = np.full(40, 4, dtype=np.uint8) if i % 2 else seqs[r][:k - 1].copy()
    exp = oracle.nmost(seqs, n, k, 4)
    ctx.set_timing(True)
    try:
        m, keep = _device_build(ctx, seqs, k)
        assert m.count_bytes == 2
        sel = m.nmost(n)
        s = _assert_selection(sel, exp)
        assert s.engine == 1
        assert s.scan_launches == 2, "head phase + full-grid launch expected"
        sel.close()
        m.close()
    finally:
        ctx.set_timing(False)
    del keep


@pytest.mark.parametrize("env", [{"DVS_NO_HEAD_PHASE": "1"}, {"DVS_PERSIST_NO_SEEDED": "1"}, {"DVS_BUILD_WAIT": "1"}])
def test_head_phase_knobs_do_not_change_the_answer(env, monkeypatch):
    from diverseseq_amd import engine

    for k_, v_ in env.items():
        monkeypatch.setenv(k_, v_)
    c = engine.Context(0)  # (the CU split is a property of the context: a fresh one sees the knobs)
    seqs = synth_seqs(20_000, 260, seed=99, ragged=True, invalid_frac=0.001)
    exp = oracle.nmost(seqs, 9, 6, 4)
    m, keep = _device_build(c, seqs, 6)
    sel = m.nmost(9)
    s = _assert_selection(sel, exp)
    assert s.engine == 1
    sel.close()
    m.close()
    c.close()
    del keep


def test_seeded_start_hands_a_tied_initial_set_back(ctx):
    """The persistent kernel works the initial set out itself (no set-up launches) unless its first
    argmin is too close to call: two identical seed rows tie exactly, the launch ends having changed
    nothing, the set-up kernels and the arbiter take over -- one more persistent launch, the oracle's
    answer.  Without the tie: a single launch."""
    seqs = synth_seqs(3000, 300, seed=808, ragged=True)
    ctx.set_timing(True)
    try:
        for tied in (False, True):
            if tied:
                seqs[1] = seqs[0].copy()
            exp = oracle.nmost(seqs, 6, 5, 4)
            m = ctx.build_matrix(seqs, 5, 4)
            sel = m.nmost(6)
            s = _assert_selection(sel, exp)
            assert s.engine == 1
            assert (s.scan_launches >= 2) == tied, (tied, s.scan_launches, s.n_arbitrated)
            sel.close()
            m.close()
    finally:
        ctx.set_timing(False)


# ---------------------------------------------------------------- the bench's own path at the bench's own shape
@pytest.fixture(scope="module")
def north_star_device():
    """100 000 x 5 000 bp generated in HBM (as bench.py does), 0.1 % invalid symbols, and the oracle's
    selections of the same bytes (n = 10 and n = 100: ~10 s of CPU)."""
    import torch

    seqs, offs = synth_device(100_000, 5_000, 5_000, seed=20260421)
    g = torch.Generator(device="cuda:0")
    g.manual_seed(77)
    bad = torch.randint(0, 100_000 * 5_000, (100_000 * 5_000 // 1000,), device="cuda:0", generator=g)
    seqs[bad] = 4
    torch.cuda.synchronize()
    host = seqs[: 100_000 * 5_000].cpu().numpy()
    exp = {n: oracle.nmost_concat(host, offs, n, 6, 4) for n in (10, 100)}
    return seqs, offs, exp


@pytest.mark.parametrize("env", [{}, {"DVS_NO_HEAD_PHASE": "1"}, {"DVS_PERSIST_NO_SEEDED": "1"},
                                 {"DVS_PERSIST_NO_SMALL": "1"}])
def test_the_timed_path_is_the_tested_path(north_star_device, env, monkeypatch):
    """bench.py's path at bench.py's shape against the oracle (src/records.rs:311-342): device-resident
    sequences -> build_matrix_device (not waited for, split in a head launch and the rest on the CU-masked
    stream) -> nmost with the head phase on the head CUs, the seeded start and the full-grid launch --
    ids, member order, frequency rows bit-exact, the floats within 1e-6, the accept count the oracle's,
    no tie arbitration; the same with each of those stages switched off."""
    from diverseseq_amd import engine

    for k_, v_ in env.items():
        monkeypatch.setenv(k_, v_)
    seqs, offs, exp = north_star_device
    c = engine.Context(0)  # (the knobs are read when the context / selection is made)
    c.set_timing(True)
    try:
        for n in (10, 100):
            oset, oacc = exp[n]
            for rep in range(2):  # the second build takes the offsets from the context's cache, as bench.py's steps do
                m = c.build_matrix_device(seqs.data_ptr(), offs, 6, 4)
                assert m.count_bytes == 2
                sel = m.nmost(n)
                s = _assert_selection(sel, oset)
                assert s.engine == 1 and s.n_arbitrated == 0
                assert s.n_accepts == oacc
                # rows scored: the stream once, plus what was in flight when a window ended (at most a row per
                # wave of the grid) -- the counter feeds bench.py's algorithmic bytes (a stale row counter summed
                # into it by a set-up on a side stream made it 4.4 M too high for n = 100, round 4)
                assert 100_000 - n <= s.rows_scored <= 100_000 + s.n_windows * 2100, (s.rows_scored, s.n_windows)
                if n == 10:
                    assert s.scan_launches == (1 if "DVS_NO_HEAD_PHASE" in env else 2), s.scan_launches
                sel.close()
                m.close()
    finally:
        c.close()


def test_stepwise_selection_without_an_order_array(ctx):
    """dvs_select_step_pack / _apply driven at world 1 on a selection started with order=None and
    SELECT_STEPWISE (the public C API allows it): the set-up kernels must have run -- a seeded 'light'
    start would leave the step kernels on uninitialised state (round-2 advisor finding) -- and the answer
    is the oracle's (src/records.rs:311-342)."""
    import torch

    from diverseseq_amd import _lib
    from diverseseq_amd.parallel import HipStepper, drive_exact

    seqs = synth_seqs(1200, 500, 31, invalid_frac=0.001, ragged=True)
    exp = oracle.nmost(seqs, 9, 5, 4)
    m = ctx.build_matrix(seqs, 5, 4)
    sel = m.select(_lib.MODE_NMOST, 9, window=4096, flags=_lib.SELECT_STEPWISE)
    drive_exact(HipStepper(ctx, sel, m.nbins, torch.device("cuda:0")), 1, torch.device("cuda:0"))
    _assert_selection(sel, exp)
    sel.close()
    m.close()


def test_uniform_length_builds_need_no_offsets_on_the_device(ctx, monkeypatch):
    """Sequences of one length laid end to end are built from (base, stride) with nothing uploaded; the
    rows are those of the offsets path (and of count_kmers, src/record.rs:41-84), also with a leading gap
    in front of the first sequence, a length below k, and a single sequence."""
    rng = np.random.default_rng(2026)
    for nseq, length, k, lead in ((700, 333, 5, 0), (64, 1000, 6, 48), (5, 3, 4, 0), (1, 900, 3, 16)):
        data = rng.integers(0, 4, size=lead + nseq * length, dtype=np.uint8)
        data[rng.integers(0, data.size, size=max(1, data.size // 500))] = 4
        offs = (lead + np.arange(nseq + 1, dtype=np.uint64) * np.uint64(length)).astype(np.uint64)
        m = ctx.build_matrix_concat(data, offs, k, 4)
        got = m.counts()
        m.close()
        monkeypatch.setenv("DVS_NO_UNIFORM_OFFSETS", "1")
        m2 = ctx.build_matrix_concat(data, offs, k, 4)
        ref = m2.counts()
        m2.close()
        monkeypatch.delenv("DVS_NO_UNIFORM_OFFSETS")
        assert (got == ref).all()
        for i in (0, nseq // 2, nseq - 1):
            a = lead + i * length
            assert (got[i] == oracle.count_kmers(data[a:a + length], 4, k)).all()


def test_packed_upload_builds_the_same_matrix(ctx, monkeypatch):
    """A host buffer of four-state sequences crosses PCIe as 2 + 1 bits per base and is expanded on the
    device (csrc/pack.hip); the count matrix is the one the plain one-byte-per-base upload gives and
    count_kmers' (src/record.rs:41-84) -- with invalid symbols of every value, a ragged total length and
    several 4 MiB chunks."""
    rng = np.random.default_rng(808)
    nseq = 9000
    lens = rng.integers(3000, 4500, size=nseq)
    offs = np.zeros(nseq + 1, dtype=np.uint64)
    offs[1:] = np.cumsum(lens)
    total = int(offs[-1])
    assert total > (32 << 20) and total % 32 != 0
    data = rng.integers(0, 4, size=total, dtype=np.uint8)
    bad = rng.integers(0, total, size=total // 700)
    data[bad] = rng.integers(4, 256, size=bad.size, dtype=np.uint8)
    m = ctx.build_matrix_concat(data, offs, 6, 4)
    got = m.counts()
    m.close()
    monkeypatch.setenv("DVS_NO_PACKED_UPLOAD", "1")
    m2 = ctx.build_matrix_concat(data, offs, 6, 4)
    ref = m2.counts()
    m2.close()
    assert (got == ref).all()
    for i in (0, 4321, nseq - 1):
        assert (got[i] == oracle.count_kmers(data[int(offs[i]):int(offs[i + 1])], 4, 6)).all()
