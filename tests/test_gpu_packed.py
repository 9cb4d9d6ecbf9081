"""The packed sequence form (include/dvs_hip.h "packed sequences": 2-bit codes + 1-bit invalid mask, 3 bits per
base in HBM) through the C ABI: the planes themselves against a numpy statement of the layout, then k-mer counts
(count_kmers, src/record.rs:41-84) and sketches (get_kmer_hashes / mash_sketch, src/distance.rs:101-182) read
from the packed words, bit-exact against the oracle -- invalid symbols at word, tile and chunk edges included --
and selections over matrices built that way."""
import numpy as np
import pytest

import oracle
from conftest import pack_reference, synth_seqs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from diverseseq_amd import engine

    return engine.default_context()


def _edge_seqs(seed=3):
    """ragged sequences whose invalid symbols sit on the seams of the packed layout: around multiples of 16
    (a word) and of 32 (a host packer group), first and last base, runs longer than k, lengths around k"""
    rng = np.random.default_rng(seed)
    seqs = synth_seqs(48, 900, seed, invalid_frac=0.01, ragged=True)
    for s in seqs[:24]:
        n = s.size
        for p in (0, 15, 16, 17, 31, 32, 33, 47, 48, n - 1, n - 16, n - 17):
            if 0 <= p < n and rng.random() < 0.6:
                s[p] = rng.integers(4, 256)
    seqs += [np.zeros(0, dtype=np.uint8), np.array([1, 2], dtype=np.uint8), np.full(50, 4, dtype=np.uint8),
             np.array([4] + [0, 1, 2, 3] * 5 + [4], dtype=np.uint8), np.array([0, 1, 2, 3, 0, 1], dtype=np.uint8),
             rng.integers(0, 4, size=17, dtype=np.uint8), rng.integers(0, 4, size=16, dtype=np.uint8),
             rng.integers(0, 4, size=15, dtype=np.uint8), np.full(100, 255, dtype=np.uint8),
             np.concatenate([rng.integers(0, 4, size=40, dtype=np.uint8), np.full(20, 9, np.uint8),
                             rng.integers(0, 4, size=40, dtype=np.uint8)])]
    return seqs


def _counts_of(seqs, k):
    return np.stack([oracle.count_kmers(s, 4, k) for s in seqs]).astype(np.uint64)


def _check_counts(m, seqs, k):
    got = m.counts().astype(np.uint64)
    exp = _counts_of(seqs, k)
    assert (got == exp).all(), f"k-mer counts from packed words differ (k={k})"
    tot = m.totals()
    assert (tot == exp.sum(axis=1)).all()
    H = m.entropy()
    for i, s in enumerate(seqs):
        if tot[i]:
            _, h = oracle.to_kfreqs(s, 4, k)
            assert abs(H[i] - h) <= 1e-11 * max(1.0, abs(h))


def test_planes_from_host_and_from_device(ctx):
    """dvs_pack_sequences: the host packer (threads + chunked copies) and the device kernel write the same words
    as the numpy statement of the layout, at lengths around the 16-base words and across the 1 Mi-base chunks of the
    host packer (whose copies take 1, 2, 4, 8, then 16 chunks at a time)"""
    import torch

    rng = np.random.default_rng(11)
    for n in (1, 15, 16, 17, 33, 4099, (1 << 20) - 1, (1 << 20) + 1, (3 << 20) + 17, (4 << 20) - 1, (9 << 20) + 5, (33 << 20) + 77):
        src = rng.integers(0, 4, size=n, dtype=np.uint8)
        src[rng.integers(0, n, size=n // 50 + 1)] = rng.integers(4, 256, size=n // 50 + 1, dtype=np.uint8)
        src[-1] = 7
        ec, em = pack_reference(src)
        ph = ctx.pack_host(src)
        c, m = ph.planes()
        assert ph.nbases == n and ph.nwords == (n + 15) // 16
        assert (c == ec).all() and (m == em).all(), ("host", n)
        ph.close()
        t = torch.zeros(n + 16, dtype=torch.uint8, device="cuda:0")
        t[:n] = torch.from_numpy(src).to("cuda:0")
        torch.cuda.synchronize()
        pd = ctx.pack_device(t.data_ptr(), n)
        c, m = pd.planes()
        assert (c == ec).all() and (m == em).all(), ("device", n)
        pd.close()


@pytest.mark.parametrize("k", [1, 2, 3, 5, 6, 7, 8])
def test_counts_from_packed_words(ctx, k):
    """every histogram instantiation with a packed form: 16-bit rows (k <= 6), 32-bit rows in LDS (k = 7), rows
    in L2 (k = 8)"""
    from diverseseq_amd import engine

    seqs = _edge_seqs()
    data, offsets = engine.concat(seqs)
    p = ctx.pack_host(data)
    m = ctx.build_matrix_packed(p, offsets, k)
    _check_counts(m, seqs, k)
    m.close()
    p.close()


def test_counts_from_packed_genome_length_tiles(ctx):
    """multi-tile rows (explicit tile lists, global atomics) and the fullest single tile from packed words;
    invalid symbols on tile seams (a tile is 32 768 windows)"""
    from diverseseq_amd import engine

    rng = np.random.default_rng(5)
    seqs = [rng.integers(0, 4, size=n, dtype=np.uint8) for n in (200_000, 32_768 + 5, 32_773, 70_001, 900)]
    seqs[0][rng.integers(0, 200_000, size=200)] = 4
    for p in (32_767, 32_768, 32_769, 32_772, 32_773, 65_535, 65_541):
        seqs[0][p] = 5
        seqs[3][p] = 6
    seqs.append(np.full(32768 + 5, 1, dtype=np.uint8))  # one bin takes every count of a full tile
    data, offsets = engine.concat(seqs)
    pk = ctx.pack_host(data)
    for k in (6, 3, 7):
        m = ctx.build_matrix_packed(pk, offsets, k)
        _check_counts(m, seqs, k)
        m.close()
    pk.close()


def test_host_builds_cross_pcie_packed_and_are_read_packed(ctx, monkeypatch):
    """dvs_matrix_build from a host pointer of >= 32 MB packs on the host and builds from the packed words; the
    same call with DVS_NO_PACKED_UPLOAD=1 copies bytes: identical matrices, both the oracle's on sampled rows"""
    rng = np.random.default_rng(23)
    n, L, k = 9000, 4000, 6
    data = rng.integers(0, 4, size=n * L, dtype=np.uint8)
    data[rng.integers(0, data.size, size=4000)] = 4
    for a in ((1 << 20) - 1, 1 << 20, (3 << 20) - 1, 3 << 20, (4 << 20) - 2, (4 << 20) - 1, 4 << 20, (4 << 20) + 1, (7 << 20) - 1,
              7 << 20, (15 << 20) - 1, 15 << 20, (31 << 20) - 1, 31 << 20):  # the host packer's chunk seams and its copies' (1, 2, 4, 8, 16 chunks)
        data[a] = 9
    offsets = np.arange(n + 1, dtype=np.uint64) * np.uint64(L)
    m1 = ctx.build_matrix_concat(data, offsets, k, 4)
    c1 = m1.counts()
    monkeypatch.setenv("DVS_NO_PACKED_UPLOAD", "1")
    m2 = ctx.build_matrix_concat(data, offsets, k, 4)
    c2 = m2.counts()
    assert (c1 == c2).all() and (m1.totals() == m2.totals()).all() and (m1.entropy() == m2.entropy()).all()
    for r in [0, 1, (1 << 20) // L, (3 << 20) // L, (4 << 20) // L - 1, (4 << 20) // L, (4 << 20) // L + 1, (7 << 20) // L, (15 << 20) // L,
              (31 << 20) // L, n - 1] + rng.integers(0, n, 20).tolist():
        assert (c1[r] == oracle.count_kmers(data[r * L:(r + 1) * L], 4, k)).all(), r
    m1.close()
    m2.close()


@pytest.mark.parametrize("canonical", [False, True])
@pytest.mark.parametrize("k", [4, 12, 16, 17, 31, 32])
def test_sketches_from_packed_words(ctx, k, canonical):
    from diverseseq_amd import distance, engine

    seqs = _edge_seqs(7) + [np.random.default_rng(k).integers(0, 4, size=40_000, dtype=np.uint8)]
    seqs[-1][[8191, 8192, 8193, 16_383, 16_384 + k]] = 4  # a sketch tile is 8192 windows
    data, offsets = engine.concat(seqs)
    p = ctx.pack_host(data)
    sk = distance.Sketches(None, k, 300, 4, canonical, ctx=ctx, packed=p, offsets=offsets)
    got, lens = sk.to_host()
    for i, s in enumerate(seqs):
        exp = oracle.mash_sketch(s, k, 300, 4, canonical)
        assert got[i, : lens[i]].tolist() == exp.tolist(), (i, k, canonical)
    sk.close()
    p.close()


def test_packed_sketch_limits_fail_loudly(ctx):
    from diverseseq_amd import distance, engine

    data, offsets = engine.concat(synth_seqs(3, 200, 1))
    p = ctx.pack_host(data)
    with pytest.raises((RuntimeError, ValueError, NotImplementedError)):
        distance.Sketches(None, 33, 50, 4, False, ctx=ctx, packed=p, offsets=offsets)
    bad = offsets.copy()
    bad[-1] += 40  # offsets beyond the packed batch
    with pytest.raises(ValueError):
        ctx.build_matrix_packed(p, bad, 4)
    p.close()


def test_selection_over_a_matrix_built_from_packed_words(ctx):
    """nmost / max over device-resident PACKED sequences (split build, head phase, persistent engine): the
    oracle's members, order and statistics"""
    import torch

    from gpu_synth import synth_device

    seqs_t, offsets = synth_device(20_000, 1800, 2200, seed=77)
    host = seqs_t.cpu().numpy()
    bad = np.random.default_rng(1).integers(0, int(offsets[-1]), size=3000)
    host[bad] = 4
    seqs_t.copy_(torch.from_numpy(host).to(seqs_t.device))
    torch.cuda.synchronize()
    p = ctx.pack_device(seqs_t.data_ptr(), int(offsets[-1]))
    for n in (10, 40):
        m = ctx.build_matrix_packed(p, offsets, 6)
        sel = m.nmost(n)
        exp, acc = oracle.nmost_concat(host[: int(offsets[-1])], offsets, n, 6, 4)
        lab, delta, _, _ = exp.members()
        mem = sel.members(False)
        s = sel.summary()
        assert mem.positions.tolist() == lab.tolist()
        assert np.allclose(mem.delta_jsd, delta, rtol=1e-6, atol=1e-13)
        assert s.n_accepts == acc and abs(s.total_jsd - exp.total_jsd) <= 1e-6 * abs(exp.total_jsd)
        sel.close()
        m.close()
    p.close()


def test_ingested_batch_packed_in_place(ctx):
    """dvs_seqbatch_pack: a FASTA file ingested on the device, its bases re-stated at 3 bits each and the byte
    form released; counts, sketches and the codes copied back (invalid -> 255) agree with the oracle's parse"""
    from diverseseq_amd import distance

    rng = np.random.default_rng(9)
    recs = []
    for i in range(40):
        n = int(rng.integers(50, 3000))
        s = "".join(rng.choice(list("ACGTN-"), p=[0.24, 0.24, 0.24, 0.24, 0.02, 0.02], size=n))
        recs.append(f">r{i} x\n" + "\n".join(s[j:j + 70] for j in range(0, n, 70)) + "\n")
    raw = "".join(recs).encode()
    names, exp_seqs = oracle.load_fasta(raw)
    b = ctx.encode_fasta(raw)
    assert b.packed is None
    b.pack()
    assert b.dev_ptr == 0 and b.packed is not None and b.packed.nbases == b.total
    codes = b.sequences()
    for got, e in zip(codes, exp_seqs):
        assert (np.where(e > 3, 255, e) == got).all()
    m = b.build_matrix(5, 4)
    _check_counts(m, exp_seqs, 5)
    m.close()
    sk = distance.Sketches(None, 9, 64, 4, True, batch=b)
    got, lens = sk.to_host()
    for i, e in enumerate(exp_seqs):
        assert got[i, : lens[i]].tolist() == oracle.mash_sketch(e, 9, 64, 4, True).tolist()
    sk.close()
    with pytest.raises(ValueError):
        b.build_matrix(2, 17)  # a packed batch has four states
    b.close()


def test_packed_batch_dropped_right_after_the_build(ctx):
    """A build over packed planes is not waited for; the planes go back to the context's block cache when the
    `Packed` goes.  Both guards: the C side (dvs_packed_destroy drains the streams its readers were enqueued on --
    exercised by destroying the handle behind the Python object's back) and the Python side (the matrix keeps the
    `Packed` it was built from), with an allocation of the same size filled right behind the destroy."""
    from gpu_synth import synth_device

    seqs_t, offsets = synth_device(30_000, 1900, 2100, seed=5)
    host = seqs_t.cpu().numpy()
    exp = np.stack([oracle.count_kmers(host[int(offsets[i]): int(offsets[i + 1])], 4, 6) for i in (0, 1023, 1024, 17_000, 29_999)])
    for via_c in (True, False):
        p = ctx.pack_device(seqs_t.data_ptr(), int(offsets[-1]))
        m = ctx.build_matrix_packed(p, offsets, 6)
        if via_c:  # the caller of the C ABI that destroys the batch right after the build call
            ctx._L.dvs_packed_destroy(p._h)
            p._h = None
            m._source = None
        del p
        # the blocks the planes lived in are handed out again at once, for other content
        other = (seqs_t + 1) % 4
        q = ctx.pack_device(other.data_ptr(), int(offsets[-1]))
        got = m.counts()
        for j, i in enumerate((0, 1023, 1024, 17_000, 29_999)):
            assert (got[i] == exp[j]).all(), (via_c, i)
        assert (m.totals() == got.sum(axis=1)).all()
        q.close()
        m.close()
        del other


def test_only_the_dna_alphabet_packs(ctx):
    """dvs_seqbatch_pack refuses a batch encoded with a caller's alphabet table (its symbols >= 4 are states)"""
    import ctypes as C

    from diverseseq_amd import _lib, engine

    raw = np.frombuffer(b">p1\nMKVLAAGIVALLLAAGCSSA\n>p2\nMKKLLPTAAAGLLLLAAQPAMA\n", dtype=np.uint8)
    lut = np.full(256, 255, dtype=np.uint8)
    for i, ch in enumerate(b"ACDEFGHIKLMNPQRSTVWY"):
        lut[ch] = i
    h = C.c_void_p()
    ctx.check(ctx._L.dvs_seqbatch_from_fasta(ctx._h, C.c_void_p(raw.ctypes.data), 0, raw.size, _lib.ptr(lut, C.c_uint8), 0,
                                             C.byref(h)))
    b = engine.SeqBatch(ctx, h, raw)
    before = b.codes()
    with pytest.raises(ValueError, match="DNA / RNA alphabet"):
        b.pack()
    assert b.packed is None and (b.codes() == before).all()  # (nothing was released)
    b.close()
