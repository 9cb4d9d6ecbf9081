"""world_size-2 gloo test of the N>1 path's exchange step (no GPU): each rank
selects on its chunk with the oracle, the winners are all_gathered exactly as
bench.py / diverseseq_amd.parallel do on RCCL, and the merged result must equal
the reference's chunk-and-merge (`-np 2`) semantics computed in one process."""
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


def test_gather_and_merge_world2():
    import torch.multiprocessing as mp

    import oracle
    from diverseseq_amd import parallel

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=90) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process statement of the same `-np 2` semantics
    seqs = synth_seqs(301, 300, 99, ragged=True)
    rows, ids = [], []
    for lo, hi in parallel.chunk_bounds(len(seqs), 2):
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
