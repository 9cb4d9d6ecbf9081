"""Side benchmark (not the headline metric): MinHash sketching of genome-length sequences and
the N x N mash distance matrix (config C5 shape, scaled to one GPU)."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from diverseseq_amd import distance, engine  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nseq", type=int, default=64)
    ap.add_argument("--length", type=int, default=3_000_000)
    ap.add_argument("-k", type=int, default=12)
    ap.add_argument("-s", type=int, default=3000)
    ap.add_argument("--npair-seqs", type=int, default=1000)
    a = ap.parse_args()
    rng = np.random.default_rng(5)
    ctx = engine.default_context()
    seqs = [rng.integers(0, 4, size=a.length, dtype=np.uint8) for _ in range(a.nseq)]
    distance.sketch_batch(seqs[:2], a.k, a.s, 4, True)  # warm up
    t0 = time.perf_counter()
    sk, lens = distance.sketch_batch(seqs, a.k, a.s, 4, True)
    dt = time.perf_counter() - t0
    print(f"sketch: {a.nseq} x {a.length} bp, k={a.k}, s={a.s}: {dt * 1e3:.1f} ms "
          f"({a.nseq * a.length / dt / 1e9:.2f} Gbase/s incl. H2D of {a.nseq * a.length / 1e6:.0f} MB)")
    n = a.npair_seqs
    skp = np.sort(rng.integers(0, 2**32, size=(n, a.s), dtype=np.uint64).astype(np.uint32), axis=1)
    lp = np.full(n, a.s, dtype=np.uint32)
    distance.distances_from_sketches(skp[:8], lp[:8], a.k, a.s)
    t0 = time.perf_counter()
    d = distance.distances_from_sketches(skp, lp, a.k, a.s)
    dt = time.perf_counter() - t0
    print(f"pairs: {n} sketches -> {n * (n - 1) // 2} distances: {dt * 1e3:.1f} ms; checksum {d.sum():.6f}")


if __name__ == "__main__":
    main()
