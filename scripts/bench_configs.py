#!/usr/bin/env python3
"""The BASELINE.json configurations that fit one MI355X, one JSON line each:
C2 (10k x 2 kb, k=6, nmost n=10 / n=100), C3 scaled (1050 genomes of ~3 Mb, k=6, `max`
min_size=100), C4's per-GPU share and its whole input on one GPU (k=7, n=100), the north-star
shape at n=100, and C5 (1000 x 3 Mb mash sketches, k=12, s=3000, + the N x N distances).
Synthetic uniform DNA generated on the device; sequences resident in HBM when the clock starts
(the headline metric's rule); side numbers, not the bench.py metric."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from diverseseq_amd import _lib, distance, engine  # noqa: E402

dev = torch.device("cuda:0")
ctx = engine.Context(0)
ctx.set_timing(True)  # (HIP events around the engine's launches: `launches`, `engine_ms` below)


def synth(nseq, lo, hi, seed, composition=False):
    """uniform i.i.d. symbols, or (composition=True) every sequence with its own base composition
    (Dirichlet(4,4,4,4): GC content varies the way real microbial genomes' does) -- i.i.d. uniform
    genomes of 3 Mb all have the same k-mer spectrum to within 1e-3 and are a degenerate input"""
    rng = np.random.default_rng(seed)
    lens = rng.integers(lo, hi + 1, size=nseq, dtype=np.int64) if hi > lo else np.full(nseq, lo, np.int64)
    offsets = np.zeros(nseq + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum(lens)
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    total = int(offsets[-1])
    if not composition:
        seqs = torch.randint(0, 4, (total + 16,), dtype=torch.uint8, device=dev, generator=g)
    else:
        seqs = torch.zeros(total + 16, dtype=torch.uint8, device=dev)
        r = torch.randint(0, 256, (total,), dtype=torch.uint8, device=dev, generator=g)
        cuts = np.minimum(255, np.cumsum(rng.dirichlet([4.0] * 4, size=nseq), axis=1)[:, :3] * 256).astype(np.uint8)
        for i in range(nseq):
            a, b = int(offsets[i]), int(offsets[i + 1])
            x = r[a:b]
            seqs[a:b] = (x >= int(cuts[i, 0])).to(torch.uint8) + (x >= int(cuts[i, 1])).to(torch.uint8) + \
                (x >= int(cuts[i, 2])).to(torch.uint8)
        del r
    torch.cuda.synchronize()
    return seqs, offsets


def timed(fn, reps):
    fn()
    ctx.sync()
    t = []
    for _ in range(reps):
        t0 = time.perf_counter()
        out = fn()
        ctx.sync()
        t.append(time.perf_counter() - t0)
    return min(t), out


def select_case(name, nseq, lo, hi, k, mode, reps=3, composition=False, **kw):
    seqs, offsets = synth(nseq, lo, hi, 20260421 + len(name), composition)

    phase = {}

    def hist_only():  # the build alone, waited for (a separate repetition: the timed run does not wait)
        m = ctx.build_matrix_device(seqs.data_ptr(), offsets, k, 4)
        ctx.sync()
        m.close()

    phase["hist_ms"] = round(timed(hist_only, reps)[0] * 1e3, 3)
    # the same sequences resident in HBM in the packed form (3 bits per base, packed once outside the clock)
    packed = ctx.pack_device(seqs.data_ptr(), int(offsets[-1])) if PACKED else None
    if packed is not None:
        ctx.sync()

        def hist_packed():
            m = ctx.build_matrix_packed(packed, offsets, k)
            ctx.sync()
            m.close()

        phase["hist_packed_ms"] = round(timed(hist_packed, reps)[0] * 1e3, 3)
        phase["input_form"] = "packed"

    def run():
        # as a caller runs it: the build does not wait for its kernels, the selection follows at once
        m = (ctx.build_matrix_packed(packed, offsets, k) if packed is not None
             else ctx.build_matrix_device(seqs.data_ptr(), offsets, k, 4))
        sel = m.nmost(kw["n"]) if mode == "nmost" else m.max_divergent(kw["min_size"], nseq, "stdev")
        s = sel.summary()
        out = dict(size=s.size, accepts=s.n_accepts, rows_scored=s.rows_scored, rechecked=s.rows_rechecked, windows=s.n_windows, engine=s.engine,
                   n_arbitrated=s.n_arbitrated, host_arbiter_ms=round(s.arbiter_ms, 3), events=s.n_events,
                   launches=s.scan_launches, engine_ms=round(s.scan_ms, 3), last_launch_ms=round(s.scan_ms_last, 3),
                   last_launch_rows=s.rows_scored_last, total_jsd=s.total_jsd)
        sel.close()
        m.close()
        return out

    dt, out = timed(run, reps)
    rec = dict(config=name, nseq=nseq, length=[lo, hi], k=k, mode=mode, **kw, ms=round(dt * 1e3, 3),
               sequences_per_s=round(nseq / dt), gbases_per_s=round(float(offsets[-1]) / dt / 1e9, 2), **phase, **out)
    print(json.dumps(rec), flush=True)
    if packed is not None:
        packed.close()
    del seqs
    torch.cuda.empty_cache()


def mash_case(name, nseq, lo, hi, k, s, canonical, reps=2, composition=False):
    seqs, offsets = synth(nseq, lo, hi, 777, composition)
    sk = np.zeros((nseq, s), dtype=np.uint32)
    lens = np.zeros(nseq, dtype=np.uint32)

    def sketch():
        ctx.check(ctx._L.dvs_mash_sketch(ctx._h, C.c_void_p(seqs.data_ptr()), 1, _lib.ptr(offsets, C.c_uint64),
                                         nseq, k, s, 4, int(canonical), _lib.ptr(sk, C.c_uint32),
                                         _lib.ptr(lens, C.c_uint32)))

    dt_s, _ = timed(sketch, reps)
    extra = {}
    if PACKED:  # the same sketches from the packed form (3 bits per base in HBM)
        packed = ctx.pack_device(seqs.data_ptr(), int(offsets[-1]))
        ctx.sync()

        def sketch_packed():
            r = distance.Sketches(None, k, s, 4, canonical, ctx=ctx, packed=packed, offsets=offsets)
            return r

        dt_k, r = timed(sketch_packed, reps)
        sk2, lens2 = r.to_host()
        r.close()
        sketch()
        assert np.array_equal(lens, lens2) and np.array_equal(sk, sk2)
        extra = dict(sketch_packed_ms=round(dt_k * 1e3, 2))
        packed.close()
    dt_p, d = timed(lambda: distance.distances_from_sketches(sk, lens, k, s, ctx=ctx), reps)
    # the ctree path: the sketches stay in HBM between the two stages (distance.mash_distances)
    res = distance.Sketches(None, k, s, 4, canonical, ctx=ctx, dev_ptr=seqs.data_ptr(), offsets=offsets)
    dt_r, d_r = timed(lambda: res.distances(), max(reps, 3))
    out = np.zeros((nseq, nseq))
    dt_o, _ = timed(lambda: res.distances(out=out), max(reps, 3))
    res.close()
    assert np.array_equal(d, d_r) and np.array_equal(d, out)
    rec = dict(config=name, nseq=nseq, length=[lo, hi], k=k, sketch_size=s, canonical=canonical,
               sketch_ms=round(dt_s * 1e3, 2), gbases_per_s=round(float(offsets[-1]) / dt_s / 1e9, 2),
               pairs=nseq * (nseq - 1) // 2, pairs_ms=round(dt_r * 1e3, 2),
               pairs_ms_from_host_sketches=round(dt_p * 1e3, 2), pairs_ms_into_a_kept_matrix=round(dt_o * 1e3, 2),
               mean_distance=float(d[np.tril_indices(nseq, -1)].mean()), **extra)
    print(json.dumps(rec), flush=True)
    del seqs
    torch.cuda.empty_cache()


def ingest_case(name, nrec, length, reps=3):
    """FASTA bytes (80-column lines) -> index codes in HBM (csrc/ingest.hip): device-resident file
    (the kernels alone) and host file (upload included)"""
    rng = np.random.default_rng(99)
    width = 80
    rows = length // width
    recs = []
    for r in range(nrec):
        body = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(rows, width), dtype=np.uint8)]
        body = np.concatenate([body, np.full((rows, 1), 10, dtype=np.uint8)], axis=1).ravel()
        recs.append(np.frombuffer(b">genome%05d synthetic\n" % r, dtype=np.uint8))
        recs.append(body)
    raw = np.concatenate(recs)
    t_raw = torch.from_numpy(raw).to(dev)
    torch.cuda.synchronize()

    def on_device():
        b = ctx.encode_fasta(None, dev_ptr=t_raw.data_ptr(), nbytes=raw.size)
        n = (b.nseq, b.total)
        b.close()
        return n

    def from_host():
        b = ctx.encode_fasta(raw)
        n = (b.nseq, b.total)
        b.close()
        return n

    dt_d, n = timed(on_device, reps)
    dt_h, _ = timed(from_host, reps)
    print(json.dumps(dict(config=name, records=n[0], bases=n[1], file_bytes=int(raw.size),
                          device_ms=round(dt_d * 1e3, 3), device_gbytes_per_s=round(raw.size / dt_d / 1e9, 1),
                          host_ms=round(dt_h * 1e3, 3), host_gbytes_per_s=round(raw.size / dt_h / 1e9, 1))), flush=True)


PACKED = False

if __name__ == "__main__":
    which = set(sys.argv[1:])
    if "PACKED" in which:  # every case also from the packed form of its sequences
        which.discard("PACKED")
        PACKED = True
    def want(n):
        return not which or n in which
    if want("C2"):
        select_case("C2 n=10", 10_000, 2000, 2000, 6, "nmost", n=10)
        select_case("C2 n=100", 10_000, 2000, 2000, 6, "nmost", n=100)
    if want("NS"):
        select_case("north-star n=10", 100_000, 5000, 5000, 6, "nmost", n=10)
        select_case("north-star n=100", 100_000, 5000, 5000, 6, "nmost", n=100)
    if want("C4"):
        select_case("C4 per-GPU share (1/8)", 12_500, 5000, 5000, 7, "nmost", n=100)
        select_case("C4 whole input, 1 GPU", 100_000, 5000, 5000, 7, "nmost", n=100)
    if want("C3"):
        select_case("C3 scaled (1050 genomes, own base composition each)", 1050, 2_500_000, 3_500_000, 6, "max",
                    reps=2, composition=True, min_size=100)
        select_case("C3 scaled (1050 genomes, i.i.d. uniform: degenerate)", 1050, 2_500_000, 3_500_000, 6, "max",
                    reps=2, min_size=100)
    if "C3K7" in which:  # (`max` at 4^7 bins: not on the persistent engine)
        select_case("C3 scaled at k=7 (1050 genomes, own base composition each)", 1050, 2_500_000, 3_500_000, 7, "max",
                    reps=2, composition=True, min_size=100)
    if "C3F" in which:  # (only when asked for: 31.5 GB of sequence)
        select_case("C3 as stated (10 500 genomes, own base composition each)", 10_500, 2_500_000, 3_500_000, 6, "max",
                    reps=2, composition=True, min_size=100)
    if want("ING"):
        ingest_case("FASTA ingest, 100 x 3 Mb genomes", 100, 3_000_000)
        ingest_case("FASTA ingest, 100k x 5 kb records", 100_000, 5_040)
    if want("C5"):
        mash_case("C5 mash", 1000, 2_900_000, 3_100_000, 12, 3000, False)
        mash_case("C5 mash canonical", 1000, 2_900_000, 3_100_000, 12, 3000, True)
