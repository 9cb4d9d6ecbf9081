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


GENBANK = b"""LOCUS       SCU49845     5028 bp    DNA             PLN       21-JUN-1999
DEFINITION  Saccharomyces cerevisiae TCP1-beta gene.
FEATURES             Location/Qualifiers
     source          1..5028
                     /organism="Saccharomyces cerevisiae"
ORIGIN
        1 gatcctccat atacaacggt atctccacct caggtttaga tctcaacaac ggaaccattg
       61 ccgacatgag acagttaggt atcgtcgaga gttacaagct aaaacgagca gtagtcagct
      121 ctgcatctga agccgctgaa gttctactaa gggtggataa catcatccgt gcaagaccaa
//
LOCUS       NOSEQ        10 bp    DNA
DEFINITION  a record without a sequence block.
//
LOCUS       SECOND       12 bp    DNA
ORIGIN
        1 acgtnnacgt ry
//
"""


def _random_genbank(rng, nrec):
    out = bytearray()
    for r in range(nrec):
        length = int(rng.integers(0, 2000))
        seq = bytes(rng.choice(np.frombuffer(b"acgtnry", dtype=np.uint8), size=length, p=[.24, .24, .24, .24, .02, .01, .01]))
        out += b"LOCUS       REC%d  %d bp    DNA\nDEFINITION  LOCUS ORIGIN // in the text.\nFEATURES   x\n" % (r, length)
        if r % 4 != 3:
            out += b"ORIGIN\n"
            for i in range(0, length, 60):
                out += b"%9d" % (i + 1)
                for j in range(i, min(i + 60, length), 10):
                    out += b" " + seq[j:j + 10]
                out += b"\n"
        out += b"//\n"
    return bytes(out)


def test_genbank_reframing_matches_the_line_parser():
    """the product's byte-search reframing (engine.genbank_to_fasta) against the oracle's
    line-by-line GenBank reader, on a hand-made file and random ones (CRLF too)"""
    from diverseseq_amd.engine import genbank_to_fasta

    recs = oracle.parse_genbank(GENBANK)
    assert [(lab, len(seq)) for lab, seq in recs] == [("SCU49845", 180), ("NOSEQ", 0), ("SECOND", 12)]
    assert recs[2][1] == "acgtnnacgtry"
    rng = np.random.default_rng(5)
    for raw in [GENBANK, GENBANK.replace(b"\n", b"\r\n"), b"", b"no records here\n"] + \
               [_random_genbank(rng, int(rng.integers(1, 12))) for _ in range(8)]:
        assert oracle.parse_fasta(genbank_to_fasta(raw)) == oracle.parse_genbank(raw)
    labels, seqs = oracle.load_genbank(GENBANK)
    assert labels == ["SCU49845", "NOSEQ", "SECOND"] and seqs[2].tolist()[:4] == [2, 1, 3, 0]
    assert seqs[2][4] > 3 and seqs[1].size == 0


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


@pytest.mark.gpu
def test_device_ingest_of_genbank_files(ctx):
    rng = np.random.default_rng(9)
    for raw in (GENBANK, _random_genbank(rng, 9)):
        labels, seqs = oracle.load_genbank(raw)
        b = ctx.encode_genbank(raw)
        assert b.labels == labels
        got = b.sequences()
        assert len(got) == len(seqs)
        for g, e in zip(got, seqs):
            assert np.array_equal(g, e)
        joined = ctx.encode_genbank(raw, join_records=True)
        exp = oracle.load_genbank(raw, join_records=True)[1]
        assert np.array_equal(joined.sequences()[0], exp[0])
        b.close()
        joined.close()


@pytest.mark.gpu
def test_streamed_upload_gives_the_same_batch(monkeypatch):
    """a file in host memory beyond 96 MiB travels in 32 MiB chunks through pinned staging, every scan
    carrying its state from chunk to chunk: the same codes, offsets and header positions as the one-copy
    path, with records cut by the chunk seams (and the one-copy path is the oracle-checked one above)"""
    from diverseseq_amd import engine

    ctx = engine.default_context()
    rng = np.random.default_rng(12)
    recs = []
    for r in range(37):
        n = int(rng.integers(1_000_000, 6_000_000))
        body = np.frombuffer(b"ACGTNacgt-", dtype=np.uint8)[rng.integers(0, 10, size=n)]
        lines = np.concatenate([body, np.zeros((-n) % 61, dtype=np.uint8)]).reshape(-1, 61)
        lines = np.concatenate([lines, np.full((lines.shape[0], 1), 10, dtype=np.uint8)], axis=1).ravel()
        recs.append(np.frombuffer(b">rec%03d some description\r\n" % r, dtype=np.uint8))
        recs.append(lines[lines != 0])
    raw = np.concatenate(recs)
    assert raw.size > 100 << 20
    a = ctx.encode_fasta(raw)
    monkeypatch.setenv("DVS_INGEST_NO_STREAM", "1")
    b = ctx.encode_fasta(raw)
    assert a.nseq == b.nseq == 37 and a.total == b.total
    assert (a.offsets == b.offsets).all() and (a.header_positions == b.header_positions).all()
    assert (a.codes() == b.codes()).all()
    assert a.labels[5] == "rec005 some description"
    _, seqs = oracle.load_fasta(raw[: int(a.header_positions[2])].tobytes())  # the first two records, in full
    got = a.sequences()
    assert (got[0] == seqs[0]).all() and (got[1] == seqs[1]).all()
    # joined: one sequence, a gap between records
    monkeypatch.delenv("DVS_INGEST_NO_STREAM")
    j = ctx.encode_fasta(raw, join_records=True)
    assert j.nseq == 1 and j.total == a.total + 36


@pytest.mark.gpu
def test_more_records_than_the_arrays_were_sized_for():
    """records of a few bytes each: the record arrays (sized for records of >= 256 bytes) overflow, the
    kernel says so and the last pass is repeated with arrays of the right size"""
    from diverseseq_amd import engine

    ctx = engine.default_context()
    nrec = 1_500_000
    raw = np.tile(np.frombuffer(b">a\nACGT\n", dtype=np.uint8), nrec)
    b = ctx.encode_fasta(raw)
    assert b.nseq == nrec and b.total == 4 * nrec
    assert (b.offsets == np.arange(nrec + 1, dtype=np.uint64) * 4).all()
    assert (b.header_positions == np.arange(nrec, dtype=np.uint64) * 8).all()
    assert (b.codes()[:8] == np.array([2, 1, 3, 0, 2, 1, 3, 0], dtype=np.uint8)).all()
