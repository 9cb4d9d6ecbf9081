"""Wide random sweep of the other kernels against the oracle (a one-off, GPU box only):
    python scripts/micro/stress_other.py SEED NCASES
k-mer counts (k 1..9, num_states 2..6, ragged lengths incl. shorter than k, invalid symbols),
mash sketches + pair distances (k 1..20, sketch sizes 1..3000, canonical on/off, related
sequences so that sketches share hashes), FASTA ingest (random line widths, CRLF, junk)."""
import sys

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import oracle  # noqa: E402

from diverseseq_amd import distance, engine  # noqa: E402


def rand_seqs(rng, nseq, lo, hi, ns, invalid):
    out = []
    for _ in range(nseq):
        length = int(rng.integers(lo, hi + 1))
        s = rng.integers(0, ns, size=length, dtype=np.uint8)
        if invalid and length:
            s[rng.random(length) < invalid] = ns + int(rng.integers(0, 3))
        out.append(s)
    return out


def counts_case(ctx, rng):
    ns = int(rng.integers(2, 7))
    k = int(rng.integers(1, 10 if ns <= 4 else 6))
    seqs = rand_seqs(rng, int(rng.integers(1, 40)), 0, int(rng.integers(1, 3000)), ns, float(rng.choice([0, 0.01, 0.2])))
    counts, totals, ent = ctx.kmer_counts(seqs, k, ns)
    for i, s in enumerate(seqs):
        exp = oracle.count_kmers(s, ns, k)
        if not np.array_equal(counts[i], exp):
            return f"counts k={k} ns={ns} row {i}"
        if int(totals[i]) != int(exp.sum()):
            return f"totals k={k} ns={ns} row {i}"
        if exp.sum():
            h = oracle.to_kfreqs(s, ns, k)[1]
            if abs(ent[i] - h) > 1e-9 * max(1.0, abs(h)):
                return f"entropy k={k} ns={ns} row {i}: {ent[i]} vs {h}"
    return None


def mash_case(ctx, rng):
    k = int(rng.integers(1, 21))
    s = int(rng.choice([1, 2, 7, 50, 400, 1000, 3000]))
    canonical = bool(rng.integers(0, 2))
    nseq = int(rng.integers(2, 14))
    base = rng.integers(0, 4, size=int(rng.integers(0, 30000)), dtype=np.uint8)
    seqs = []
    for _ in range(nseq):
        t = base.copy()
        if t.size:
            m = rng.random(t.size) < float(rng.choice([0.0, 0.01, 0.1]))
            t[m] = rng.integers(0, 4, size=int(m.sum()), dtype=np.uint8)
            if rng.random() < 0.3:
                t[rng.random(t.size) < 0.002] = 4
            if rng.random() < 0.3:
                t = t[: int(rng.integers(0, t.size + 1))]
        seqs.append(t)
    sk, lens = distance.sketch_batch(seqs, k, s, 4, canonical, ctx=ctx)
    exp = [oracle.mash_sketch(t, k, s, 4, canonical) for t in seqs]
    for i in range(nseq):
        if sk[i, : int(lens[i])].tolist() != list(exp[i]):
            return f"sketch k={k} s={s} canonical={canonical} row {i} (len {seqs[i].size})"
    want = oracle.mash_distances(exp, k, s)
    try:
        got = distance.distances_from_sketches(sk, lens, k, s, ctx=ctx)
    except ZeroDivisionError:  # two empty sketches: the reference divides by zero (NaN in the oracle)
        return None if np.isnan(want).any() else "distances raised ZeroDivisionError, the oracle has no NaN"
    if np.isnan(want).any():
        return "the oracle has a NaN (0 / 0), the device did not raise"
    if not np.allclose(got, want, rtol=1e-12, atol=0):
        return f"distances k={k} s={s} canonical={canonical}: max diff {np.abs(got - want).max()}"
    return None


def ingest_case(ctx, rng):
    out = bytearray()
    crlf = bool(rng.integers(0, 2))
    if rng.random() < 0.3:
        out += b"junk before the first header\n"
    alphabet = np.frombuffer(b"ACGTacgtNRYKM-?XU", dtype=np.uint8)
    for r in range(int(rng.integers(0, 30))):
        out += b">r%d text" % r + (b"\r\n" if crlf else b"\n")
        length = int(rng.integers(0, 9000))
        width = int(rng.integers(1, 200))
        seq = bytes(rng.choice(alphabet, size=length))
        for i in range(0, length, width):
            out += seq[i:i + width] + (b"\r\n" if crlf else b"\n")
        if rng.random() < 0.2:
            out += b"\n"
    if out and rng.random() < 0.3:
        out = out[:-1]
    raw = bytes(out)
    for join in (False, True):
        labels, seqs = oracle.load_fasta(raw, join_records=join)
        b = ctx.encode_fasta(raw, join_records=join)
        got = b.sequences()
        if len(got) != len(seqs) or any(not np.array_equal(g, e) for g, e in zip(got, seqs)):
            return f"ingest join={join} ({len(raw)} bytes, crlf={crlf})"
        if not join and b.labels != labels:
            return "ingest labels"
        b.close()
    return None


def main(seed, ncases):
    ctx = engine.default_context()
    rng = np.random.default_rng(seed)
    bad = 0
    kinds = (("counts", counts_case), ("mash", mash_case), ("ingest", ingest_case))
    for case in range(ncases):
        name, fn = kinds[case % 3]
        msg = fn(ctx, rng)
        if msg:
            bad += 1
            print("MISMATCH case", case, name, msg, flush=True)
    print("cases", ncases, "bad", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(int(sys.argv[1]), int(sys.argv[2])))
