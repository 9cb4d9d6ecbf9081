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


class _HistoryStepper:
    """the library's stepwise engine as the driver sees it, without a device: launches are applied in order, the engine
    stops for its arbiter at given launches (the syncing poll resolves that) and is done behind launch `end`; `peek(lag)`
    answers from the status word of the launch `lag` launches back, as dvs_select_step_peek does (launches the last poll
    has accounted for read as running), and asks for a poll when `ring_half` steps have passed since the last one"""

    def __init__(self, end, stops=(), ring_half=10**9, peek=True):
        self.end, self.stops, self.ring_half = end, set(stops), ring_half
        self.launches = self.polls = self.floor = self.since_poll = 0
        self.state = 0  # 0 running, 1 done, 2 stopped for the arbiter
        self.words = {}
        self.progress = 0  # launches that did something
        if not peek:
            self.peek = None

    def pack(self):
        return None

    def apply(self, slot, world):
        self.launches += 1
        self.since_poll += 1
        if self.state == 0:
            self.progress += 1
            if self.progress in self.stops:
                self.state = 2
            elif self.progress >= self.end:
                self.state = 1
        self.words[self.launches] = self.state

    def peek(self, lag):
        must = self.since_poll + lag > self.ring_half
        back = self.launches - lag
        if back <= 0 or back <= self.floor:
            return 0, must
        return self.words[back], must

    def done(self):
        self.polls += 1
        self.since_poll = 0
        self.floor = self.launches
        if self.state == 2:  # the arbiter's verdict: on with the steps
            self.state = 0
        return self.state == 1


@pytest.mark.parametrize("end,stops,ring_half,poll_every", [(88, (), 10**9, 16), (5, (), 10**9, 16), (200, (17, 90, 91), 10**9, 8),
                                                          (120, (3,), 16, 16), (1, (), 10**9, 4)])
def test_exact_driver_looks_without_syncing(end, stops, ring_half, poll_every):
    """parallel.drive_exact over an engine that keeps a status history (dvs_select_step_peek): it runs to the end whatever
    the engine's stops, calls the syncing poll only at the start, for every stop, when the accepted rows' ring asks for it and
    at the end -- not every `poll_every` steps -- and enqueues at most two short batches of no-ops behind the end.  Without a
    history (max_divergent, the multi-launch kernels) it polls every `poll_every` steps as before."""
    import torch

    from diverseseq_amd import parallel

    st = _HistoryStepper(end, stops, ring_half)
    parallel.drive_exact(st, 1, torch.device("cpu"), poll_every=poll_every)
    assert st.state == 1 and st.progress == end
    ring_polls = 0 if ring_half > 10**6 else st.launches // max(1, ring_half - 4) + 1
    assert st.polls <= 2 + len(stops) + ring_polls, (st.polls, st.launches)
    assert st.launches <= end + len(stops) * 8 + max(poll_every, 8) + 8, (st.launches, end)
    old = _HistoryStepper(end, stops, ring_half, peek=False)
    parallel.drive_exact(old, 1, torch.device("cpu"), poll_every=poll_every)
    assert old.state == 1 and old.polls >= end // poll_every
