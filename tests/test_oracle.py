"""Pins the CPU oracle (oracle/dvs_oracle.c) to the reference's own
known-answer values: every exact-equality constant in
the Rust unit tests and the mash_distance vectors captured from the reference's
pure-Python function.  CPU only."""
import math

import numpy as np
import pytest

import oracle
from conftest import str2arr


def test_kmer_to_index(rust_vectors):
    v = rust_vectors["kmer_to_index"]
    for c in v["cases"]:
        assert oracle.kmer_to_index(c["kmer"], v["num_states"], v["max_index"]) == c["index"]


def test_kmer_count(rust_vectors):
    v = rust_vectors["kmer_count"]
    got = oracle.count_kmers(v["seq"], v["num_states"], v["k"])
    assert got.tolist() == v["counts"]


def test_kfreqs(rust_vectors):
    v = rust_vectors["kfreqs"]
    f1, _ = oracle.to_kfreqs(v["seq"], 4, 1)
    assert f1.tolist() == [3.0 / 6.0, 2.0 / 6.0, 1.0 / 6.0, 0.0]
    assert f1.tolist() == v["k1"]
    f2, _ = oracle.to_kfreqs(v["seq"], 4, 2)
    assert f2.tolist() == v["k2"]


def test_k0_and_no_valid_kmers(rust_vectors):
    with pytest.raises(ValueError):
        oracle.to_kfreqs(rust_vectors["k0_panics"]["seq"], 4, 0)
    v = rust_vectors["no_valid_kmers"]
    with pytest.raises(ValueError, match="No valid k-mers"):
        oracle.to_kfreqs(v["seq"], v["num_states"], v["k"])


def test_entropy(rust_vectors):
    v = rust_vectors["entropy"]
    assert oracle.entropy(v["max_entropy"]["freqs"]) == v["max_entropy"]["entropy"]
    for bad in v["panics"]:
        with pytest.raises(ValueError):
            oracle.entropy(bad)


def _summed(v):
    rows = [oracle.to_kfreqs(s, v["num_states"], v["k"]) for s in v["seqs"].values()]
    return oracle.SummedRecords.new(np.array([r[0] for r in rows]), [r[1] for r in rows])


def test_construct_summed_records(rust_vectors):
    v = rust_vectors["summed"]
    s = _summed(v)
    assert s.size == v["size"]
    assert s.total_jsd == v["total_jsd"]
    _, deltas, ents, _ = s.members()
    assert ents.tolist() == v["entropies"]
    assert s.summed_entropies == v["summed_entropies"]
    assert deltas.tolist() == v["delta_jsds"]
    assert s.mean_delta_jsd == v["mean_delta_jsd"]
    assert s.std_delta_jsd == v["std_delta_jsd"]
    assert s.cov_delta_jsd == s.std_delta_jsd / s.mean_delta_jsd
    assert s.mean_jsd == s.total_jsd / s.size


def test_increases_jsd(rust_vectors):
    v = rust_vectors["summed"]
    s = _summed(v)
    f, h = oracle.to_kfreqs(v["better"]["seq"], 4, 1)
    assert s.increases_jsd(f, h, label=99)
    # a member: delta_jsd 0.0, increases_jsd false (records.rs:635-649)
    _, _, ents, freqs = s.members(with_freqs=True)
    assert s.delta_jsd(freqs[0], ents[0], label=0) == 0.0
    assert not s.increases_jsd(freqs[0], ents[0], label=0)
    # replace_lowest / push (records.rs:651-674)
    orig = s.total_jsd
    s.push(f, h, label=99)
    assert s.size == 4 and s.total_jsd != orig
    s2 = _summed(v)
    s2.replace_lowest(f, h, label=99)
    assert s2.size == 3 and 99 in s2.members()[0].tolist()


def test_summed_records_panics():
    with pytest.raises(ValueError, match="records cannot be empty"):
        oracle.SummedRecords.new(np.zeros((0, 4)))
    f, h = oracle.to_kfreqs([0, 0, 0, 2, 2, 2], 4, 1)
    with pytest.raises(ValueError, match="must have > 1 KmerSeq"):
        oracle.SummedRecords.new(f[None, :], [h])


def _sel(rust_vectors, with_invalid=False):
    v = rust_vectors["selector"]
    names = [n for n, _ in v["seqs"]]
    seqs = [s for _, s in v["seqs"]]
    if with_invalid:
        names.append(v["invalid"][0])
        seqs.append(v["invalid"][1])
    return v, names, seqs


def test_checked_most_divergent(rust_vectors):
    """records.rs:694-740.  The Rust test streams ids in FxHashMap order (not
    fixed by the test); with an order consistent with its expected member order
    [seq3, seq4, seq2] every asserted value is reproduced exactly."""
    v, names, seqs = _sel(rust_vectors)
    order_in = v["fxhash_consistent_order"]
    rs = oracle.nmost([seqs[names.index(n)] for n in order_in], 3, 1, 4)
    assert rs.size == 3
    labels, deltas, _, _ = rs.members()
    got_names = [order_in[l] for l in labels]
    assert set(got_names) == set(v["nmost_n3_members"])
    order = v["summed234_order"]
    assert got_names == order  # Vec::remove + push ordering
    expect = oracle.SummedRecords.from_seqs([seqs[names.index(n)] for n in order], 1, 4)
    _, ed, _, _ = expect.members()
    assert deltas.tolist() == ed.tolist()  # exact, records.rs:733-739
    assert rs.total_jsd == expect.total_jsd
    # insertion order also selects {seq2, seq3, seq4}
    rs = oracle.nmost(seqs, 3, 1, 4)
    assert {names[l] for l in rs.members()[0]} == set(v["nmost_n3_members"])


def test_most_divergent_misc(rust_vectors):
    v, names, seqs = _sel(rust_vectors)
    with pytest.raises(ValueError, match="The number of sequences 5 is < n 20"):
        oracle.nmost(seqs, 20, 1, 4)
    sub = [seqs[names.index(n)] for n in v["with_seqids"]]
    rs = oracle.nmost(sub, 3, 1, 4)
    assert rs.size == 3
    # duplicates are ignored (records.rs:742-762): same label twice
    rs = oracle.nmost(seqs + [seqs[1]], 3, 1, 4, labels=[0, 1, 2, 3, 4, 1])
    assert rs.size == 3
    # invalid sequence is skipped (records.rs:779-789)
    _, n2, s2 = _sel(rust_vectors, with_invalid=True)
    rs = oracle.nmost(s2, 3, 1, 4)
    assert rs.size == 3 and 5 not in rs.members()[0].tolist()


@pytest.mark.parametrize("stat", ["stdev", "cov"])
def test_max_divergent(rust_vectors, stat):
    v, names, seqs = _sel(rust_vectors, with_invalid=True)
    sr = oracle.max_divergent(seqs, 3, 4, 1, 4, stat)
    assert 3 <= sr.size <= 4
    sr = oracle.max_divergent(seqs, 3, 10, 1, 4, "stdev")
    assert 3 <= sr.size <= 6
    with pytest.raises(ValueError, match="is < n 30"):
        oracle.max_divergent(seqs, 30, 40, 1, 4, "stdev")


def test_bats(rust_vectors):
    v = rust_vectors["bats"]
    seqs = [s for _, s in v["seqs"]]
    sr = oracle.SummedRecords.from_seqs(seqs, v["k"], v["num_states"])
    assert not math.isnan(sr.total_jsd)
    f, h = oracle.to_kfreqs(seqs[2], 4, 3)
    assert not math.isnan(h)


def test_python_level_total_jsd(rust_vectors):
    """tests/test_records.py:34-42: total_jsd == JSD of the freq vectors (k=1,2),
    checked here against an independent numpy formula."""
    v = rust_vectors["python_level"]["seqs"]
    seqs = [str2arr(v[n]) for n in ("b", "c", "d")]  # unique_seqids: 'a' and 'b' dedupe
    for k in (1, 2):
        rows = np.array([oracle.to_kfreqs(s, 4, k)[0] for s in seqs])
        sr = oracle.SummedRecords.new(rows)

        def H(p):
            p = p[p > 0]
            return float(-(p * np.log2(p)).sum())

        expect = H(rows.mean(axis=0)) - np.mean([H(r) for r in rows])
        np.testing.assert_allclose(sr.total_jsd, expect, rtol=1e-12)


def test_final_merge_matches_direct(brca1):
    """chunk + merge (records.py:225-245; records.rs:363-382): the merge of one
    chunk's own result is that result (all members seed, nothing streams)."""
    names = list(brca1)[:20]
    seqs = [brca1[n] for n in names]
    a = oracle.nmost(seqs[:10], 5, 2, 4)
    b = oracle.nmost(seqs[10:], 5, 2, 4)
    la, _, _, fa = a.members(with_freqs=True)
    lb, _, _, fb = b.members(with_freqs=True)
    merged = oracle.final_nmost(np.vstack([fa, fb]), 5, labels=np.concatenate([la, lb + 10]))
    assert merged.size == 5
    with pytest.raises(ValueError, match="is < n 500"):
        oracle.final_nmost(np.vstack([fa, fb]), 500)
    m2 = oracle.final_max(np.vstack([fa, fb]), 3, 6, "stdev",
                          labels=np.concatenate([la, lb + 10]))
    assert 3 <= m2.size <= 6


def test_reverse_complement(rust_vectors):
    v = rust_vectors["reverse_complement"]
    assert oracle.reverse_complement(v["kmer"]).tolist() == v["expect"]


def test_hash_source_fidelity():
    """independent python restatement of src/distance.rs:21-49 (unpinned by the
    reference's tests; source text is the contract)"""
    def ref_hash(data):
        M = 0xFFFFFFFF
        rotl = lambda x, r: ((x << r) | (x >> (32 - r))) & M
        h = 0x9747B28C ^ len(data)
        for v in data:
            k = (v * 0xCC9E2D51) & M
            k = rotl(k, 15)
            k = (k * 0x1B873593) & M
            h ^= k
            h = rotl(h, 13)
            h = (h * 5 + 0xE6546B64) & M
        h ^= h >> 16
        h = (h * 0x85EBCA6B) & M
        h ^= h >> 13
        h = (h * 0xC2B2AE35) & M
        h ^= h >> 16
        return h

    rng = np.random.default_rng(1)
    for _ in range(50):
        d = rng.integers(0, 4, size=int(rng.integers(0, 24)), dtype=np.uint8)
        assert oracle.murmurhash3_32(d) == ref_hash(d.tolist())
    # canonical picks min(kmer, revcomp) lexicographically; ties -> original
    k = np.array([3, 3, 0, 1], dtype=np.uint8)  # revcomp = [3,2,1,1] < kmer
    rc = oracle.reverse_complement(k)
    assert rc.tolist() == [3, 2, 1, 1]
    assert oracle.hash_kmer(k, True) == oracle.murmurhash3_32(rc)
    assert oracle.hash_kmer(k, False) == oracle.murmurhash3_32(k)


def test_kmer_hashes_skip_and_sketch():
    seq = np.array([0, 1, 2, 4, 3, 2, 1, 0, 0, 1], dtype=np.uint8)
    h = oracle.kmer_hashes(seq, 3, 4)
    valid = [i for i in range(len(seq) - 2) if (seq[i:i + 3] < 4).all()]
    assert len(h) == len(valid)
    assert h.tolist() == [oracle.hash_kmer(seq[i:i + 3]) for i in valid]
    assert oracle.kmer_hashes(seq[:2], 3, 4).size == 0
    sk = oracle.mash_sketch(seq, 3, 4)
    assert sk.tolist() == sorted(set(h.tolist()))[:4]
    assert oracle.mash_sketch(seq, 3, 100).tolist() == sorted(set(h.tolist()))


def test_mash_distance_vectors(mash_vectors):
    for c in mash_vectors["mash_distance"]:
        got = oracle.mash_distance(c["left"], c["right"], c["k"], c["sketch_size"])
        if c["distance"] == "ZeroDivisionError":
            assert math.isnan(got)
        else:
            assert got == c["distance"], c  # same libm log, same op order -> exact
    for c in mash_vectors["euclidean_distance"]:
        np.testing.assert_allclose(oracle.euclidean_distance(c["a"], c["b"]),
                                   c["distance"], rtol=1e-13)


def test_brca1_demo_c1(brca1):
    """config C1: nmost on the bundled demo data, k=4, n=10 (CPU path)"""
    names = list(brca1)
    assert len(names) == 55
    lens = [brca1[n].size for n in names]
    assert min(lens) == 2382 and max(lens) == 2889
    sr = oracle.nmost([brca1[n] for n in names], 10, 4, 4)
    assert sr.size == 10
    assert sr.total_jsd > 0


def test_chunk_and_merge_on_threads_equals_the_sequential_scheme():
    """orc_nmost_chunks_mt (bench.py's all-cores CPU baseline): the reference's -np scheme
    (diverse_seq/records.py:225-245) with threads as workers gives what the same chunks give
    one after the other"""
    from diverseseq_amd.parallel import chunk_bounds

    rng = np.random.default_rng(5)
    nseq, length, k, n = 600, 400, 4, 6
    data = rng.integers(0, 4, size=nseq * length, dtype=np.uint8)
    off = (np.arange(nseq + 1) * length).astype(np.uint64)
    bounds = chunk_bounds(nseq, 5)
    got = oracle.nmost_chunks_threads(data, off, bounds, n, k)
    rows, labs = [], []
    for lo, hi in bounds:
        res, _ = oracle.nmost_concat(data[off[lo]:off[hi]], (off[lo:hi + 1] - off[lo]).astype(np.uint64), n, k, 4,
                                     labels=np.arange(lo, hi, dtype=np.uint32))
        lab, _, _, f = res.members(with_freqs=True)
        rows.append(f)
        labs.append(lab)
    exp = oracle.final_nmost(np.vstack(rows), n, labels=np.concatenate(labs))
    assert got.members()[0].tolist() == exp.members()[0].tolist()
    assert got.total_jsd == exp.total_jsd
