"""Wide random sweep of small selections against the oracle (a one-off, GPU box only):
    python scripts/micro/stress_selections.py SEED NCASES [big | head]
Sequence count 12..900, length 20..600, k 1..6, n 2..70, nmost / max stdev / max cov, with
duplicates and invalid symbols.  A case passes when the ids and total_jsd agree, or when both
sides raise the reference's panic with the same message (k = 1 sets do: record.rs:99-104).
`head`: 18k-40k short sequences built from device-resident input, k 3..6, n 2..60: the split build and
the persistent engine's head phase on the CU-masked stream (two persistent launches per selection)."""
import sys

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import oracle  # noqa: E402
from conftest import synth_seqs  # noqa: E402

from diverseseq_amd import engine  # noqa: E402


def attempt(fn):
    try:
        return fn(), None
    except ValueError as e:
        return None, str(e)


def main(seed: int, ncases: int, big: bool = False, head: bool = False) -> int:
    ctx = engine.default_context()
    rng = np.random.default_rng(seed)
    bad = panics = arb = 0
    eng = {0: 0, 1: 0}
    two_launches = 0
    if head:
        import torch

        ctx.set_timing(True)
    for case in range(ncases):
        if head:
            nseq = int(rng.integers(18000, 40000))
            length = int(rng.integers(60, 500))
            k = int(rng.integers(3, 7))
            n = int(rng.integers(2, 61))
        elif big:  # thousands of rows, sets up to 300 members (the barrier path from 128 on), k up to 7
            nseq = int(rng.integers(2000, 30000))
            length = int(rng.integers(200, 3000))
            k = int(rng.integers(4, 8))
            n = int(rng.choice([2, 5, 10, 33, 64, 100, 127, 128, 129, 200, 300]))
            if k == 7:
                n = min(n, 100)  # (the oracle's leave-one-out pass costs n * 4^k log2 per event)
        else:
            nseq = int(rng.integers(12, 900))
            length = int(rng.integers(20, 600))
            k = int(rng.integers(1, 7))
            n = int(rng.integers(2, min(70, nseq - 1)))
        seqs = synth_seqs(nseq, length, seed=int(rng.integers(0, 1 << 30)), ragged=bool(case & 1),
                          invalid_frac=0.01 if case % 3 == 0 else 0.0)
        if case % 5 == 0:
            for _ in range(3):
                seqs[int(rng.integers(0, nseq))] = seqs[int(rng.integers(0, nseq))].copy()
        if head:
            data, offs = oracle.concat(seqs)
            dev_seqs = torch.from_numpy(np.concatenate([data, np.zeros(16, np.uint8)])).to("cuda:0")
            torch.cuda.synchronize()
            m = ctx.build_matrix_device(dev_seqs.data_ptr(), offs, k, 4)
        else:
            m = ctx.build_matrix(seqs, k, 4)
        mode = 0 if head else case % 3
        if big and n > 64:
            mode = 0  # (`max` clones the set per tentative push: minutes in the oracle at this size)
        if big or head:
            print("case", case, nseq, length, k, n, mode, flush=True)
        if mode == 0:
            sel, gerr = attempt(lambda: m.nmost(n))
            exp, oerr = attempt(lambda: oracle.nmost(seqs, n, k, 4))
        else:
            stat = "stdev" if mode == 1 else "cov"
            mx = nseq if (case % 2 and not big) else min(nseq, n + int(rng.integers(0, 40)))
            sel, gerr = attempt(lambda: m.max_divergent(n, mx, stat))
            exp, oerr = attempt(lambda: oracle.max_divergent(seqs, n, mx, k, 4, stat))
        if gerr is not None or oerr is not None:
            if gerr == oerr:
                panics += 1
            else:
                bad += 1
                print("MISMATCH case", case, nseq, length, k, n, mode, "gpu:", gerr, "oracle:", oerr)
            if sel is not None:
                sel.close()
            m.close()
            continue
        got = sel.members(with_freqs=False)
        s = sel.summary()
        elab = exp.members()[0]
        ok = got.positions.tolist() == elab.tolist() and \
            abs(s.total_jsd - exp.total_jsd) <= 1e-6 * max(abs(exp.total_jsd), 1e-300) + 1e-13
        eng[s.engine] += 1
        two_launches += s.scan_launches >= 2
        arb += s.n_arbitrated
        if not ok:
            bad += 1
            print("MISMATCH case", case, nseq, length, k, n, mode, got.positions.tolist()[:8], elab.tolist()[:8])
        sel.close()
        m.close()
    print("cases", ncases, "bad", bad, "same panic on both sides", panics, "engines", eng, "arbitrations", arb,
          *(("with a head phase", two_launches) if head else ()))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(int(sys.argv[1]), int(sys.argv[2]), len(sys.argv) > 3 and sys.argv[3] == "big",
                  len(sys.argv) > 3 and sys.argv[3] == "head"))
