"""gloo tests of the N>1 paths' exchange steps (no GPU), world sizes 2, 4 and 8.  Chunk mode: each rank
selects on its chunk with the oracle, the winners are all_gathered exactly as bench.py /
diverseseq_amd.parallel do on RCCL, and the merged result must equal the reference's chunk-and-merge
(`-np G`, chunks in order) computed in one process.  Exact mode: the driver of the row-sharded
selection (parallel.drive_exact: one all_gather per greedy step) with the oracle as each rank's
compute must give the one-process answer of select_nmost_divergent / select_max_divergent."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT, synth_seqs


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_world(target, world, timeout=240):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=timeout) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def _worker(rank, world, port, q):
    sys.path.insert(0, str(ROOT))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist

    import oracle
    from diverseseq_amd import parallel

    dist.init_process_group("gloo", rank=rank, world_size=world)
    seqs = synth_seqs(301, 300, 99, ragged=True)
    lo, hi = parallel.chunk_bounds(len(seqs), world)[rank]
    local = oracle.nmost(seqs[lo:hi], 6, 3, 4)
    lab, _, _, rows = local.members(with_freqs=True)
    rows_all, ids_all = parallel.gather_winners(rows, lab.astype(np.int64) + lo, world,
                                                torch.device("cpu"), cap=6)
    merged = oracle.final_nmost(rows_all, 6, labels=np.arange(len(ids_all), dtype=np.uint32))
    pos, delta, _, _ = merged.members()
    q.put((rank, ids_all[pos].tolist(), delta.tolist(), merged.total_jsd))
    dist.destroy_process_group()


def test_chunk_sizes_match_reference():
    from diverseseq_amd import parallel

    assert parallel.determine_chunk_size(10, 3) == [4, 3, 3]  # reference tests/test_util.py:40-42
    assert parallel.chunk_bounds(10, 3) == [(0, 4), (4, 7), (7, 10)]


@pytest.mark.parametrize("world", [2, 4, 8])
def test_gather_and_merge(world):
    import oracle
    from diverseseq_amd import parallel

    res = _run_world(_worker, world)
    # single-process statement of the same `-np G` semantics (chunks in order)
    seqs = synth_seqs(301, 300, 99, ragged=True)
    rows, ids = [], []
    for lo, hi in parallel.chunk_bounds(len(seqs), world):
        r = oracle.nmost(seqs[lo:hi], 6, 3, 4)
        lab, _, _, f = r.members(with_freqs=True)
        rows.append(f)
        ids.append(lab.astype(np.int64) + lo)
    rows, ids = np.vstack(rows), np.concatenate(ids)
    exp = oracle.final_nmost(rows, 6)
    epos, edelta, _, _ = exp.members()
    for rank, got_ids, got_delta, got_total in res:
        assert got_ids == ids[epos].tolist()
        assert got_delta == edelta.tolist()
        assert got_total == exp.total_jsd


def _mash_worker(rank, world, port, q):
    sys.path.insert(0, str(ROOT))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist

    import oracle
    from diverseseq_amd import parallel

    dist.init_process_group("gloo", rank=rank, world_size=world)
    seqs = synth_seqs(23, 400, 5, ragged=True)
    k, s = 8, 64

    def sketcher(chunk):
        sk = np.zeros((len(chunk), s), dtype=np.uint32)
        lens = np.zeros(len(chunk), dtype=np.uint32)
        for i, x in enumerate(chunk):
            h = oracle.mash_sketch(x, k, s)
            sk[i, : len(h)] = h
            lens[i] = len(h)
        return sk, lens

    def pair_rows(sk, lens, row_start, row_stride):
        n = sk.shape[0]
        out = np.zeros((n, n))
        for i in range(row_start, n, row_stride):  # cluster.py:640-644
            for j in range(i):
                out[i, j] = oracle.mash_distance(sk[i, : lens[i]], sk[j, : lens[j]], k, s)
        return out

    d = parallel.mash_distances_sharded(seqs, k, s, rank, world, torch.device("cpu"),
                                        sketcher=sketcher, pair_rows=pair_rows)
    q.put((rank, d))
    dist.destroy_process_group()


def test_sharded_mash_distances_world2():
    """ctree over ranks: chunked sketching, all_gather, strided triangle rows,
    SUM all-reduce, symmetrise == the one-process matrix"""
    import torch.multiprocessing as mp

    import oracle

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_mash_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=90) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    seqs = synth_seqs(23, 400, 5, ragged=True)
    exp = oracle.mash_distances([oracle.mash_sketch(x, 8, 64) for x in seqs], 8, 64)
    for _, d in res:
        np.testing.assert_array_equal(d, exp)


# ----------------------------------------------------------------------------- exact mode
from bench_cpu_engine import OracleStepper as _OracleStepper  # (shared with bench.py's CPU test hook)


def _exact_cpu_worker(rank, world, port, q):
    sys.path.insert(0, str(ROOT))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist

    from diverseseq_amd import parallel

    dist.init_process_group("gloo", rank=rank, world_size=world)
    seqs = synth_seqs(400, 300, 7, invalid_frac=0.002, ragged=True)
    out = [rank]
    for mode, n_seed, kw in (("nmost", 6, {}), ("max", 4, {"max_size": 12, "stat": "stdev"}),
                             ("max", 4, {"max_size": 400, "stat": "cov"})):
        owned, _ = parallel.shard_order(len(seqs), n_seed, rank, world, block=16)
        st = _OracleStepper(seqs, owned, mode, n_seed, 3, window=24 * world, **kw)
        parallel.drive_exact(st, world, torch.device("cpu"), poll_every=3)
        lab, delta, _, _ = st.set.members()
        out.append((lab.tolist(), delta.tolist(), st.set.total_jsd))
    q.put(tuple(out))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_exact_mode_driver(world):
    """rows dealt block-cyclically over 2 / 4 / 8 ranks, ONE all_gather per greedy step: every rank ends
    with the one-process selection, nmost and max (stdev, cov), bit for bit"""
    import oracle

    res = _run_world(_exact_cpu_worker, world)
    seqs = synth_seqs(400, 300, 7, invalid_frac=0.002, ragged=True)
    exp = [oracle.nmost(seqs, 6, 3, 4), oracle.max_divergent(seqs, 4, 12, 3, 4, "stdev"),
           oracle.max_divergent(seqs, 4, 400, 3, 4, "cov")]
    for r in res:
        for got, e in zip(r[1:], exp):
            lab, delta, _, _ = e.members()
            assert got[0] == lab.tolist()
            assert got[1] == delta.tolist()
            assert got[2] == e.total_jsd
