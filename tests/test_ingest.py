"""FASTA ingest: the oracle's restatement of the reference's parse + str2arr
against the reference's own pins, and the device ingest (csrc/ingest.hip) against the oracle."""
import pathlib

import numpy as np
import pytest

import oracle

GOLDEN = pathlib.Path(__file__).parent / "golden"


@pytest.fixture(scope="module")
def ctx():
    from diverseseq_amd import engine

    return engine.default_context()


# ------------------------------------------------------------------ oracle vs the reference's pins
def test_str2arr_pins_of_the_reference():
    """tests/test_util.py:9-16 of the reference: ACGTT -> dna.alphabet.to_indices (T0 C1 A2 G3),
    a non-canonical character gets an index above the canonical states"""
    assert oracle.str2arr("ACGTT").tolist() == [2, 1, 3, 0, 0]
    g = oracle.str2arr("ACGNT")
    assert g[-2] > 3 and g[[0, 1, 2, 4]].tolist() == [2, 1, 3, 0]
    assert oracle.str2arr("").size == 0
    assert oracle.str2arr("AYGTT")[1] > 3
    assert oracle.str2arr("ACGU", "rna").tolist() == [2, 1, 3, 0]


def test_parse_brca1_records():
    """the reference's sample data (diverse_seq/data/brca1.fa: 55 aligned sequences of 3009 columns)"""
    labels, seqs = oracle.load_fasta((GOLDEN / "brca1.fasta").read_bytes())
    assert len(labels) == 55 and len(set(labels)) == 55
    assert {s.size for s in seqs} == {3009}
    assert "Human" in labels and "Chimpanzee" in labels
    joined = oracle.load_fasta((GOLDEN / "brca1.fasta").read_bytes(), join_records=True)[1]
    assert len(joined) == 1 and joined[0].size == 55 * 3009 + 54


# ------------------------------------------------------------------ device vs oracle
def _random_fasta(rng, nrec, crlf=False, junk_front=False, trailing_newline=True):
    out = bytearray()
    if junk_front:
        out += b"; comment line\nACGT\n"
    alphabet = b"ACGTacgtNRY-?X"
    for r in range(nrec):
        out += b">rec%d some description > with a bracket" % r
        out += b"\r\n" if crlf else b"\n"
        length = int(rng.integers(0, 700))
        width = int(rng.integers(1, 90))
        seq = bytes(rng.choice(np.frombuffer(alphabet, dtype=np.uint8), size=length,
                               p=[.22, .22, .22, .22, .02, .02, .02, .02, .01, .01, .005, .005, .005, .005]))
        for i in range(0, length, width):
            out += seq[i:i + width]
            if rng.random() < 0.1:
                out += b" \t"
            out += b"\r\n" if crlf else b"\n"
        if rng.random() < 0.2:
            out += b"\n\n"
    if not trailing_newline:
        while out and out[-1:] in b"\r\n":
            out = out[:-1]
    return bytes(out)


def _check(ctx, raw, join):
    labels, seqs = oracle.load_fasta(raw, join_records=join)
    b = ctx.encode_fasta(raw, join_records=join)
    assert b.labels == labels
    assert b.nseq == len(seqs)
    exp_off = np.concatenate([[0], np.cumsum([s.size for s in seqs])]).astype(np.uint64) if seqs else np.zeros(1, np.uint64)
    assert b.offsets.tolist() == exp_off.tolist()
    got = b.codes()
    exp = np.concatenate(seqs) if seqs else np.zeros(0, np.uint8)
    assert got.size == exp.size
    assert (got == exp).all(), int(np.flatnonzero(got != exp)[0])
    b.close()


@pytest.mark.gpu
def test_device_ingest_matches_oracle_on_brca1(ctx):
    raw = (GOLDEN / "brca1.fasta").read_bytes()
    _check(ctx, raw, False)
    _check(ctx, raw, True)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(6))
def test_device_ingest_matches_oracle_on_ragged_files(ctx, seed):
    """CRLF, blank lines, white space inside lines, lower case, '>' inside a line, text in front of
    the first header, empty records, no trailing newline; sizes around the 16-byte chunk and the
    4 KiB block of the scans"""
    rng = np.random.default_rng(seed)
    raw = _random_fasta(rng, nrec=int(rng.integers(1, 60)), crlf=bool(seed & 1), junk_front=bool(seed & 2),
                        trailing_newline=bool(seed & 4))
    for join in (False, True):
        _check(ctx, raw, join)
        for cut in (0, 1, 15, 16, 17, 4095, 4096, 4097, 8192 + 5):
            if cut <= len(raw):
                _check(ctx, raw[:cut], join)


@pytest.mark.gpu
def test_device_ingest_across_many_scan_blocks(ctx):
    """~1 MB of ragged records: the carries of the three-pass scans cross several 128 KiB blocks"""
    raw = _random_fasta(np.random.default_rng(77), nrec=2500, crlf=True)
    assert len(raw) > 5 * 131072
    _check(ctx, raw, False)
    _check(ctx, raw, True)


@pytest.mark.gpu
def test_matrix_from_device_ingest_equals_matrix_from_host_sequences(ctx):
    """the encoded bases feed the histogram kernel without leaving HBM: same counts as the
    host-encoded sequences give (and as the oracle counts)"""
    raw = (GOLDEN / "brca1.fasta").read_bytes()
    _, seqs = oracle.load_fasta(raw)
    b = ctx.encode_fasta(raw)
    m = b.build_matrix(4, 4)
    counts = m.counts()
    ref = ctx.build_matrix(seqs, 4, 4).counts()
    assert (counts == ref).all()
    for i in (0, 17, 54):
        assert (counts[i] == oracle.count_kmers(seqs[i], 4, 4)).all()
    sel = m.nmost(10)
    exp = oracle.nmost(seqs, 10, 4, 4)
    assert sel.members(with_freqs=False).positions.tolist() == exp.members()[0].tolist()
    m.close()
    b.close()
