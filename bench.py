#!/usr/bin/env python3
"""Headline benchmark: sequences scored/sec for the greedy delta-JSD selection
(`dvs nmost`) at k=6 on synthetic DNA, inputs resident in HBM.

One "step" = one full pass of the hot path over one batch: k-mer histogram build
(N x 4^k count matrix) + the windowed delta-JSD scan + every set update, for N
sequences per GPU.

`python bench.py --gpus G` with G > 1 and no launcher around it starts the G ranks
ITSELF: the parent process -- which never touches the GPU -- runs
`python -m torch.distributed.run --nnodes=1 --nproc-per-node G ... bench.py ...`
as a child and exits with its code.  Under a launcher (RANK / WORLD_SIZE in the
environment) it is one rank of the job; WORLD_SIZE != --gpus is an error.

Multi-GPU, both schemes in one line (`--mode both`, the default for G > 1):
  * `value`: chunk + merge -- each rank owns a contiguous shard of the sequences and
    runs the greedy on it, the reference's own `-np G` scheme (diverse_seq/records.py:
    225-245, diverse_seq/util.py:82-102) -- then the G*n winners' frequency rows are
    exchanged with ONE RCCL all_gather and merged with the reference's final_nmost
    (src/records.rs:363-382) on the device.  Work per GPU is fixed: weak scaling.
  * `value_exact`: rows sharded block-cyclically, set state replicated, ONE all_gather
    per greedy step (the `-np 1` answer).

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--nseq", type=int, default=100_000, help="sequences per GPU")
    ap.add_argument("--length", type=int, default=5_000)
    ap.add_argument("-k", type=int, default=6)
    ap.add_argument("-n", type=int, default=10, help="size of the divergent set")
    ap.add_argument("--window", type=int, default=0)
    ap.add_argument("--mode", choices=("chunk", "exact", "both"), default=None,
                    help="multi-GPU scheme: 'chunk' = the reference's -np G (independent greedy per GPU + "
                         "one all_gather + final_nmost); 'exact' = rows sharded block-cyclically, set state "
                         "replicated, ONE all_gather per greedy step: every rank's first event + its candidate row "
                         "(same answer as 1 GPU); 'both' (the default with --gpus > 1) = chunk as `value`, then "
                         "exact as `value_exact` in the same line")
    ap.add_argument("--input", choices=("bytes", "packed"), default="bytes",
                    help="form of the sequences resident in HBM: 'bytes' = one byte per base, as the reference's "
                         "boundary delivers them (src/record.rs:205-209); 'packed' = 2-bit codes + 1-bit invalid "
                         "mask (dvs_pack_sequences, done once outside the timed region)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-side-runs", action="store_true",
                    help="only the timed steps: no streaming passes, no configs[1] side number (profiling runs: "
                         "every launch of the dominant kernel is then a headline launch)")
    ap.add_argument("--no-large-point", action="store_true",
                    help="skip the 1 000 000 x 5 kb side point (8.2 GB of 16-bit rows: the largest single-GPU "
                         "stream, where the scan dominates the event chain)")
    ap.add_argument("--cpu-sample", type=int, default=100_000,
                    help="sequences of the same workload the 1-thread CPU oracle is timed on")
    return ap.parse_args()


def usable_cores() -> int:
    """host cores this process may really use: the affinity mask, capped by the cgroup CPU quota
    (a GPU box hands out a share of its cores, not all of them)"""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p))
        except (OSError, ValueError):
            pass
    return n


def self_launch(a) -> int:
    """--gpus G > 1 without a launcher: start the G ranks as children of THIS process, which has not
    imported torch or touched the GPU (no exec after a GPU call: a fresh child per rank), and hand their
    exit code on.  Rank 0's JSON line reaches stdout through the inherited descriptor."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # (dmabuf IPC: what RCCL needs on this host driver)
    env["DVS_BENCH_SELF_LAUNCHED"] = "1"
    return subprocess.run(cmd, env=env).returncode


class Bench:
    """one rank's state: device, context, communicator"""

    def __init__(self, a):
        self.a = a
        import torch
        import torch.distributed as dist

        self.torch, self.dist = torch, dist
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local = int(os.environ.get("LOCAL_RANK", "0"))
        if self.world != a.gpus:
            raise SystemExit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={self.world}")
        # TEST HOOK (tests/test_bench_launch.py only): a stand-in compute module and the gloo backend, so
        # that the launch / exchange / reporting logic of this file runs end to end on a box without a GPU.
        # Never set outside the CPU test suite; the line says so ("engine_module") when it is.
        self.test_engine = os.environ.get("DVS_BENCH_TEST_ENGINE")
        self.on_gpu = not self.test_engine
        if self.on_gpu:
            torch.cuda.set_device(self.local)
            self.dev = torch.device("cuda", self.local)
        else:
            self.dev = torch.device("cpu")
        self.force_dist = bool(os.environ.get("DVS_BENCH_FORCE_DIST"))  # exercise the RCCL path on 1 GPU
        self.dist_on = self.world > 1 or self.force_dist
        if self.dist_on:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
            if self.on_gpu:
                dist.init_process_group("nccl", device_id=self.dev)
            else:
                dist.init_process_group("gloo")
        if self.on_gpu:
            from diverseseq_amd import engine

            self.engine = engine
            # kernels and collectives ordered on one torch stream: no host waits in between
            self.stream = torch.cuda.Stream() if (self.dist_on or a.mode in ("exact", "both")) else None
            if self.stream is not None:
                torch.cuda.set_stream(self.stream)
                self.ctx = engine.Context(self.local, stream=self.stream.cuda_stream)
            else:
                self.ctx = engine.Context(self.local)
        else:
            import importlib

            self.engine = importlib.import_module(self.test_engine)
            self.ctx = self.engine.Context(self.local)
        self.B = 4 ** a.k

    # ---- plumbing that differs between a GPU and the CPU test hook
    def sync(self):
        self.ctx.sync()
        if self.on_gpu:
            self.torch.cuda.synchronize()

    def randint(self, count, seed):
        torch = self.torch
        g = torch.Generator(device=self.dev)
        g.manual_seed(seed)
        return torch.randint(0, 4, (count,), dtype=torch.uint8, device=self.dev, generator=g)

    def build(self, seqs, offsets, k, packed=None):
        if not self.on_gpu:
            return self.ctx.build_matrix_tensor(seqs, offsets, k, 4)
        if packed is not None:
            return self.ctx.build_matrix_packed(packed, offsets, k)
        return self.ctx.build_matrix_device(seqs.data_ptr(), offsets, k, 4)

    # ---- one scheme, timed
    def run(self, mode: str, collect_side: bool):
        a, torch, dist = self.a, self.torch, self.dist
        rank, world, dev, ctx = self.rank, self.world, self.dev, self.ctx
        exact = mode == "exact"
        if self.on_gpu:
            from diverseseq_amd.parallel import merge_nmost, nmost_exact, shard_order
        else:
            from diverseseq_amd.parallel import merge_nmost, shard_order

            nmost_exact = self.engine.nmost_exact
        order = None
        if exact:
            # global stream of world * nseq positions; the n seeds are replicated, the rest owned
            # block-cyclically: this rank's matrix = [seeds, owned rows]
            npos = a.nseq * world
            owned, order = shard_order(npos, a.n, rank, world)
            nlocal = a.n + owned.size
            seqs = torch.cat([self.randint(a.n * a.length, 20260421),  # the same seed rows on every rank
                              self.randint(owned.size * a.length, 20260422 + rank)])
            offsets = np.arange(nlocal + 1, dtype=np.uint64) * np.uint64(a.length)
        else:
            seqs = self.randint(a.nseq * a.length, 20260421 + rank)
            offsets = np.arange(a.nseq + 1, dtype=np.uint64) * np.uint64(a.length)
        packed = None
        if a.input == "packed" and self.on_gpu:  # (once, outside the timed region: the resident form)
            packed = ctx.pack_device(seqs.data_ptr(), int(seqs.numel()))
        self.sync()

        stats = {"rows_scored": 0, "scan_ms": 0.0, "scan_launches": 0, "n_accepts": 0,
                 "n_windows": 0, "n_arbitrated": 0, "hist_ms": 0.0, "engine": 0, "count_bytes": 4,
                 "scan_ms_last": 0.0, "rows_scored_last": 0, "arbiter_ms": 0.0}
        timing = {}   # exact: per-step all_gather; chunk: the merge's all_gathers (device events / host clock)
        last = {}     # members and statistics of the last timed step (checked against the oracle below)
        merge_buffers = {}

        def step(collect: bool, keep: bool = False):
            t0 = time.perf_counter()
            m = self.build(seqs, offsets, a.k, packed)
            t1 = time.perf_counter()
            if exact:
                sel = nmost_exact(ctx, m, order, a.n, dev, world, window=a.window, timing=timing)
            else:
                sel = m.nmost(a.n, window=a.window)
            if self.dist_on and not exact:
                merged = merge_nmost(ctx, sel, a.n, rank, world, rank * a.nseq, dev,
                                     chunk_starts=[r * a.nseq for r in range(world)], shared_stream=True,
                                     buffers=merge_buffers, timing=timing if collect else None)
                merged.close()
            if collect:
                s = sel.summary()
                for key in ("rows_scored", "scan_ms", "scan_launches", "n_accepts", "n_windows", "n_arbitrated",
                            "scan_ms_last", "rows_scored_last", "arbiter_ms"):
                    stats[key] += getattr(s, key, 0)
                stats["hist_ms"] += (t1 - t0) * 1e3
                stats["engine"] = s.engine
                stats["count_bytes"] = m.count_bytes
                if keep:
                    mem = sel.members(False)
                    last.update(positions=np.asarray(mem.positions).copy(), delta_jsd=np.asarray(mem.delta_jsd).copy(),
                                total_jsd=s.total_jsd, n_accepts=s.n_accepts, size=s.size)
            sel.close()
            m.close()

        def timed_loop(collect: bool) -> float:
            if self.dist_on:
                dist.barrier()
            self.sync()
            t0 = time.perf_counter()
            for i in range(a.steps):
                step(collect, keep=collect and i == a.steps - 1)
            self.sync()
            if self.dist_on:
                dist.barrier()
            dt = time.perf_counter() - t0
            if self.dist_on:
                t = torch.tensor([dt], dtype=torch.float64, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dt = float(t.item())
            return dt

        # The library remembers the last build's offsets (compared by content) and skips their
        # validation and upload when the same batch is built again -- which only a loop like this one
        # does.  The headline is therefore timed with that cache OFF: every step validates and uploads
        # its offsets as a first build would; the cached number is reported beside it.
        os.environ["DVS_NO_OFFSETS_CACHE"] = "1"
        ctx.refresh_knobs()  # (the library reads its switches once per context, and on this call)
        ctx.set_timing(True)  # HIP-event pairs around every scan launch, read after the run
        # The warm-up steps take exactly the timed steps' path -- the summary, and on the last of them the members'
        # read-back the oracle check needs -- so that what that path sets up once (events, pinned blocks) is set up
        # here: a 5-step run showed 1.5 to 2.9 ms per step against 1.45 with it inside the timed region.  What they
        # collected is thrown away.
        for i in range(a.warmup):
            step(True, keep=i == a.warmup - 1)
        for key in stats:
            stats[key] = type(stats[key])(0) if key not in ("engine", "count_bytes") else stats[key]
        last.clear()
        timing.clear()
        elapsed = timed_loop(True)
        del os.environ["DVS_NO_OFFSETS_CACHE"]
        ctx.refresh_knobs()
        step(False)
        elapsed_cached = timed_loop(False)
        if self.on_gpu:
            from diverseseq_amd.parallel import collective_times

            collective_times(timing)  # (chunk mode: device events -> milliseconds)
        res = {"mode": mode, "elapsed": elapsed, "elapsed_cached": elapsed_cached, "stats": stats,
               "timing": timing, "last": last, "seqs": seqs, "offsets": offsets, "packed": packed}
        if collect_side:
            self.side_runs(res)
        return res

    # ---- single-GPU diagnostics outside the timed region
    def side_runs(self, res):
        a, torch, ctx = self.a, self.torch, self.ctx
        seqs, offsets, packed, stats = res["seqs"], res["offsets"], res["packed"], res["stats"]
        # the scan arithmetic alone, one launch over the whole stream
        m = self.build(seqs, offsets, a.k, packed)
        sel = m.nmost(a.n, window=a.window)
        stats["scan_stream_ms"], stats["scan_stream_rows"] = sel.bench_scan(5)
        sel.close()
        m.close()
        # ... and the persistent kernel itself as a pure stream: the same launch with a threshold no
        # row reaches (no events, a handful of long windows), timed by the same HIP events
        knobs = {"DVS_PERSIST_NO_EVENTS": "1", "DVS_PERSIST_WG_ROUNDS": "0",  # (no events: also long windows)
                 "DVS_NO_HEAD_PHASE": "1", "DVS_PERSIST_NO_SEEDED": "1"}  # (ONE launch over the whole stream, set up by the set-up kernels)
        saved = {k_: os.environ.get(k_) for k_ in knobs}
        os.environ.update(knobs)
        ctx.refresh_knobs()
        try:
            best = None
            for _ in range(3):
                m = self.build(seqs, offsets, a.k, packed)
                sel = m.nmost(a.n)
                s_ = sel.summary()
                if s_.engine == 1 and s_.scan_launches == 1 and (best is None or s_.scan_ms < best[0]):
                    best = (s_.scan_ms, s_.rows_scored)
                sel.close()
                m.close()
            if best:
                stats["persist_stream_ms"], stats["persist_stream_rows"] = best
        finally:
            for k_, v_ in saved.items():
                if v_ is None:
                    os.environ.pop(k_, None)
                else:
                    os.environ[k_] = v_
            ctx.refresh_knobs()
        # the histogram on its own (a build that waits for its kernels, nothing else in flight)
        best = None
        os.environ["DVS_BUILD_WAIT"] = "1"
        ctx.refresh_knobs()
        try:
            for _ in range(5):
                self.sync()
                t0 = time.perf_counter()
                m = self.build(seqs, offsets, a.k, packed)
                dt = (time.perf_counter() - t0) * 1e3
                m.close()
                best = dt if best is None or dt < best else best
        finally:
            del os.environ["DVS_BUILD_WAIT"]
            ctx.refresh_knobs()
        stats["hist_alone_ms"] = best
        # also outside the timed region: the same workload through the drop-in module, as a caller of
        # the reference's API sees it (diverse_seq._dvs.nmost_divergent(store, n, k), src/lib.rs:59-73):
        # sequences start on the HOST in an in-memory store, so this includes gathering them, the
        # upload and the copy of the members' rows back -- never the headline
        from diverseseq_amd import _dvs

        host_all = seqs.cpu().numpy()
        store = _dvs.make_zarr_store()
        for i in range(a.nseq):
            store.write(f"s{i:06d}", host_all[i * a.length:(i + 1) * a.length].tobytes())
        ids = [f"s{i:06d}" for i in range(a.nseq)]
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            r_ = _dvs.nmost_divergent(store, a.n, a.k, seqids=ids)
            dt = time.perf_counter() - t0
            best = dt if best is None or dt < best else best
        t0 = time.perf_counter()
        _, data_, offs_, _ = _dvs._gather(store, ids)
        t_gather = time.perf_counter() - t0
        t_build = None
        for _ in range(3):  # (the first call makes the context's pinned staging block)
            t0 = time.perf_counter()
            m_ = ctx.build_matrix_concat(data_, offs_, a.k, 4)
            ctx.sync()
            dt = time.perf_counter() - t0
            t_build = dt if t_build is None or dt < t_build else t_build
            m_.close()
        stats["dvs_module"] = {
            "what": "diverseseq_amd._dvs.nmost_divergent(store, n, k, seqids) on the same sequences held in an "
                    "in-memory store on the host (the reference's call, src/lib.rs:59-73)",
            "ms": best * 1e3, "sequences_per_s": a.nseq / best, "engine": r_.stats["engine"],
            "host_gather_ms": t_gather * 1e3, "upload_and_histogram_ms": t_build * 1e3,
            "upload": "four-state sequences cross PCIe packed (2 + 1 bits per base, packed by host threads chunk by "
                      "chunk beside the copies) and the histogram kernel reads the packed words as they are "
                      "(csrc/pack.hip, kmer_hist.hip)",
        }
        del data_, store, host_all
        # BASELINE.json configs[1] (10k x 2 kb, k=6, nmost n=10) as a
        # side number -- the headline workload above is the shape the north star quotes its target on
        c2 = self.randint(10_000 * 2_000, 20260423)
        c2_off = np.arange(10_001, dtype=np.uint64) * np.uint64(2_000)
        self.sync()
        best = None
        for _ in range(4):
            t0 = time.perf_counter()
            m2 = ctx.build_matrix_device(c2.data_ptr(), c2_off, 6, 4)
            s2 = m2.nmost(10)
            acc2 = s2.summary().n_accepts
            dt = time.perf_counter() - t0
            s2.close()
            m2.close()
            best = dt if best is None or dt < best else best
        stats["c2"] = {"workload": "BASELINE.json configs[1]: 10000 x 2000 bp, k=6, nmost n=10, inputs in HBM",
                       "ms": best * 1e3, "sequences_per_s": 10_000 / best, "accepts": acc2}
        del c2
        # The largest single-GPU point of this shape: 1 000 000 x 5 kb, k=6 -- 8.2 GB of 16-bit rows, ~115
        # accepts -- where the stream, not the chain of greedy events, is most of the selection.
        if not a.no_large_point and (a.n, a.length, a.k) == (10, 5_000, 6):
            try:
                big_n = 1_000_000
                free_b, _ = torch.cuda.mem_get_info()
                if free_b > 24 << 30:
                    big = torch.cat([self.randint(100_000 * a.length, 20260500 + i) for i in range(10)])
                    big_off = np.arange(big_n + 1, dtype=np.uint64) * np.uint64(a.length)
                    self.sync()
                    best = None
                    for _ in range(3):
                        t0 = time.perf_counter()
                        mb = ctx.build_matrix_device(big.data_ptr(), big_off, a.k, 4)
                        sb = mb.nmost(a.n)
                        sm = sb.summary()
                        dt = time.perf_counter() - t0
                        cbb = mb.count_bytes
                        sb.close()
                        mb.close()
                        if best is None or dt < best[0]:
                            best = (dt, sm)
                    dt, sm = best
                    gb = sm.rows_scored * self.B * cbb / 1e9
                    stats["large_point"] = {
                        "workload": f"{big_n} x {a.length} bp, k={a.k}, nmost n={a.n}, inputs in HBM "
                                    f"({big_n * self.B * cbb / 1e9:.1f} GB count matrix)",
                        "ms": dt * 1e3, "sequences_per_s": big_n / dt, "accepts": sm.n_accepts, "windows": sm.n_windows,
                        "scan_ms": sm.scan_ms, "scan_launches": sm.scan_launches, "rows_scored": sm.rows_scored,
                        "achieved": gb / (sm.scan_ms * 1e-3) if sm.scan_ms else None, "unit": "GB/s",
                        "frac": gb / (sm.scan_ms * 1e-3) / 8000.0 if sm.scan_ms else None, "engine": sm.engine,
                    }
                    del big
            except Exception as e:  # (a side number must not take the headline down)
                stats["large_point"] = {"error": repr(e)}


def collective_summary(timing, what):
    cm = timing.get("collective_ms") if timing else None
    if not cm:
        return None
    return {"what": what, "mean_us": 1e3 * sum(cm) / len(cm), "max_us": 1e3 * max(cm), "samples": len(cm)}


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        raise SystemExit(self_launch(a))
    if a.mode is None:
        a.mode = "both" if a.gpus > 1 else "chunk"
    # RCCL prints a version banner on stdout when the process exits; the contract is ONE JSON
    # line on stdout, so everything C-level goes to stderr and the line is written to the real
    # stdout explicitly.
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    b = Bench(a)
    rank, world, B = b.rank, b.world, b.B
    failed = False
    single = world == 1 and not b.force_dist
    first_mode = "exact" if a.mode == "exact" else "chunk"
    res = b.run(first_mode, collect_side=(rank == 0 and single and first_mode == "chunk" and not a.no_side_runs
                                          and b.on_gpu))
    res_exact = b.run("exact", collect_side=False) if a.mode == "both" else None
    exact = first_mode == "exact"
    stats, elapsed, elapsed_cached, last = res["stats"], res["elapsed"], res["elapsed_cached"], res["last"]
    seqs, offsets = res["seqs"], res["offsets"]
    if rank == 0:
        total_seqs = a.nseq * world * a.steps
        scan_s = stats["scan_ms"] / 1e3
        cb = stats["count_bytes"]  # sizeof(count) of the matrix the scan reads: 2 for whole-sequence rows, else 4
        alg_bytes = stats["rows_scored"] * B * cb  # rows scored x 4^k x sizeof(count)
        achieved = alg_bytes / scan_s / 1e9 if scan_s > 0 else 0.0
        peak = 8000.0  # GB/s, MI355X HBM3E spec (MI355X_MICROARCH.md)
        out = {
            "metric": "sequences scored/sec (delta-JSD, k=6)",
            "value": total_seqs / elapsed,
            "unit": "sequences/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64 (results and every set update in f64; reject/accept decisions tiered: f32 and "
                     "f32-log scores decide rows farther from the threshold than their proven error bands, f64 the rest)",
            "data": "synthetic",
            "config": {
                "workload": (f"dvs nmost n={a.n}: k-mer histogram + greedy delta-JSD scan + set updates over "
                             f"{a.nseq} x {a.length} bp synthetic DNA per GPU, k={a.k} ({B} bins), "
                             "inputs resident in HBM (north-star shape)"),
                "nseq_per_gpu": a.nseq, "length": a.length, "k": a.k, "n": a.n,
                "input_form": ("one byte per base (the reference's boundary form, src/record.rs:205-209)"
                               if res["packed"] is None else
                               "2-bit codes + 1-bit invalid mask per base (packed once outside the timed region)"),
                "launched_by": ("bench.py itself (subprocess: python -m torch.distributed.run)"
                                if os.environ.get("DVS_BENCH_SELF_LAUNCHED") else
                                ("an external launcher" if "WORLD_SIZE" in os.environ else "single process")),
                "parallelism": ("single GPU" if world == 1 and not exact else
                                (f"{world} ranks, rows sharded block-cyclically, replicated set state, ONE RCCL all_gather "
                                 f"per greedy step ({world} x {B + 2} f64: every rank's first event + its candidate row; "
                                 "same answer as 1 GPU)") if exact else
                                (f"{world} shards, independent greedy per GPU + one RCCL all_gather "
                                 "+ final_nmost merge (reference -np semantics)")),
                "accepts_per_step": stats["n_accepts"] / a.steps,
                "scan_launches_per_step": stats["scan_launches"] / a.steps,
                "rows_scored_per_step": stats["rows_scored"] / a.steps,
                "hist_host_ms_per_step": stats["hist_ms"] / a.steps,  # host time of the build call (it does not wait for its kernel)
                "scan_ms_per_step": stats["scan_ms"] / a.steps,
                "tie_arbitrations": stats["n_arbitrated"],
                "host_arbiter_ms": stats["arbiter_ms"],
                "offsets_cache": "off in the timed steps (every step validates its offsets as a first build does; "
                                 "sequences of one length laid end to end -- this workload -- are recognised in that "
                                 "pass and built from (base, stride): nothing is uploaded)",
                "value_with_offsets_cache": total_seqs / elapsed_cached,
                "ms_per_step_with_offsets_cache": elapsed_cached / a.steps * 1e3,
            },
            "roofline": {
                "kernel": (f"persist_nmost_kernel<uint{8 * cb}> (windowed delta-JSD scan + in-kernel set updates behind "
                           "grid barriers; " +
                           ("two launches per selection: the head of the stream on 64 masked CUs beside the histogram "
                            "of the rest of the matrix, then the full grid from the state it leaves"
                            if stats["scan_launches"] >= 2 * a.steps else "one launch per selection") + ")")
                          if stats["engine"] == 1
                          else f"scan_kernel<uint{8 * cb}> (one launch per window)",
                "count_bytes": cb,
                "achieved_if_counts_were_u32": achieved * 4 / cb,  # comparable with round 1's lines (uint32 rows)
                "bound": "hbm",
                "achieved": achieved,
                "peak": peak,
                "unit": "GB/s",
                "frac": achieved / peak,
                "traffic": None,
                "bytes_per_launch": alg_bytes / max(1, stats["scan_launches"]),
                "avg_launch_us": stats["scan_ms"] * 1e3 / max(1, stats["scan_launches"]),
            },
        }
        # The average over a selection's launches hides two very different ones when it starts with a head phase
        # (the event-dense first rows on the head CUs beside the histogram, then the full grid): each on its own
        if stats["engine"] == 1 and stats["scan_launches"] == 2 * a.steps and stats["scan_ms_last"] > 0:
            def launch_line(what, ms, rows):
                byts = rows * B * cb
                return {"what": what, "avg_launch_us": ms * 1e3 / a.steps, "rows_per_launch": rows / a.steps,
                        "bytes_per_launch": byts / a.steps, "achieved": byts / (ms * 1e-3) / 1e9 if ms > 0 else 0.0,
                        "unit": "GB/s", "frac": (byts / (ms * 1e-3) / 1e9 / peak) if ms > 0 else 0.0}
            out["roofline"]["per_launch"] = [
                launch_line("head phase: the stream's first rows on the head CUs, beside the histogram of the rest",
                            stats["scan_ms"] - stats["scan_ms_last"], stats["rows_scored"] - stats["rows_scored_last"]),
                launch_line("full grid: the rest of the stream from the state the head phase leaves",
                            stats["scan_ms_last"], stats["rows_scored_last"]),
            ]
        if b.test_engine:
            out["engine_module"] = b.test_engine  # (CPU test hook: not a measurement)
        if world > 1 or b.force_dist:
            if not exact:
                out["config"]["chunk_mode_collective"] = collective_summary(
                    res["timing"], "the merge's two all_gathers (winners' rows + their positions) per step, rank 0")
            else:
                out["config"]["exact_mode_collective"] = collective_summary(
                    res["timing"], "the per-step all_gather (device events around it, first 64 steps of every "
                                   "selection), rank 0")
        if res_exact is not None:
            se = res_exact["stats"]
            out["value_exact"] = total_seqs / res_exact["elapsed"]
            out["config"]["exact_mode"] = {
                "what": (f"the same {world} x {a.nseq} sequences as ONE stream: rows sharded block-cyclically, set state "
                         f"replicated, ONE all_gather per greedy step ({world} x {B + 2} f64); the answer of one GPU / "
                         "`-np 1`"),
                "value": total_seqs / res_exact["elapsed"], "unit": "sequences/s",
                "ms_per_step": res_exact["elapsed"] / a.steps * 1e3,
                "accepts_per_step": se["n_accepts"] / a.steps, "windows_per_step": se["n_windows"] / a.steps,
                "engine": se["engine"],
                "collective": collective_summary(res_exact["timing"], "the per-step all_gather, rank 0, first 64 "
                                                 "steps of every selection"),
            }
        if "scan_stream_ms" in stats:  # the scan arithmetic alone, streaming the whole matrix once
            gbps = stats["scan_stream_rows"] * B * cb / (stats["scan_stream_ms"] * 1e-3) / 1e9
            out["roofline"]["scan_streaming"] = {
                "what": "one scan_kernel launch over all streamed rows, no events (dvs_select_bench_scan)",
                "ms": stats["scan_stream_ms"], "rows": stats["scan_stream_rows"],
                "achieved": gbps, "unit": "GB/s", "frac": gbps / peak,
            }
        if "persist_stream_ms" in stats:
            gbps = stats["persist_stream_rows"] * B * cb / (stats["persist_stream_ms"] * 1e-3) / 1e9
            out["roofline"]["persist_streaming"] = {
                "what": "the persistent kernel with a threshold no row reaches (DVS_PERSIST_NO_EVENTS: no events, "
                        "four long windows): its scan arithmetic (coarse tier) as a pure stream, launch to exit",
                "ms": stats["persist_stream_ms"], "rows": stats["persist_stream_rows"],
                "achieved": gbps, "unit": "GB/s", "frac": gbps / peak,
            }
        if "hist_alone_ms" in stats:
            hb = a.nseq * a.length * (1.0 if res["packed"] is None else 0.375) + a.nseq * B * cb
            out["roofline"]["histogram_alone"] = {
                "what": "kmer_hist_kernel over the whole batch, nothing beside it (host clock around a build that "
                        "waits for its kernels: includes the launch and one stream sync)",
                "ms": stats["hist_alone_ms"], "algorithmic_bytes": hb,
                "achieved": hb / (stats["hist_alone_ms"] * 1e-3) / 1e9, "unit": "GB/s",
                "frac": hb / (stats["hist_alone_ms"] * 1e-3) / 1e9 / peak,
            }
        if "large_point" in stats:
            out["roofline"]["large_stream_point"] = stats["large_point"]
        if "c2" in stats:
            out["config"]["also_configs_1"] = stats["c2"]
        if "dvs_module" in stats:
            out["config"]["through_dvs_module"] = stats["dvs_module"]
        # HBM traffic of the dominant kernel: measured separately with rocprofv3 PMC passes
        # (bench.py cannot run under the profiler and time itself); committed in profiles/.
        # The headline workload MUST find its profile: a missing or drifted file is an error, not a
        # silently absent field; a profile whose algorithmic bytes per launch are more than 15 % away
        # from this run's is flagged stale.
        wl = f"nmost n={a.n}, {a.nseq} x {a.length} bp, k={a.k}"
        pmc_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "pmc_traffic.json")
        headline = (a.n, a.nseq, a.length, a.k) == (10, 100_000, 5_000, 6) and world == 1 and not exact and b.on_gpu
        try:
            pmc = json.load(open(pmc_path))
            if pmc.get("workload") != wl:
                raise KeyError(f"profile is for {pmc.get('workload')!r}, this run is {wl!r}")
            if stats["engine"] == 1:
                pk = pmc["persist_nmost_kernel"]
                out["roofline"]["traffic"] = pk["hbm_bytes_per_launch"]
                out["roofline"]["traffic_source"] = "profiles/pmc_traffic.json (2 x FETCH_SIZE + WRITE_SIZE, KiB -> B)"
                prof_alg = pk.get("algorithmic_bytes_per_launch")
                run_alg = out["roofline"]["bytes_per_launch"]
                if prof_alg:
                    drift = abs(prof_alg - run_alg) / run_alg if run_alg else 1.0
                    out["roofline"]["traffic_profile_algorithmic_bytes_per_launch"] = prof_alg
                    out["roofline"]["traffic_stale"] = bool(drift > 0.15)
                else:
                    out["roofline"]["traffic_stale"] = None  # (a profile from before the field existed)
        except (OSError, KeyError, ValueError) as e:
            if headline:
                raise SystemExit(f"bench.py: profiles/pmc_traffic.json does not serve the headline workload: {e}")
            out["roofline"]["traffic_source"] = f"none: no PMC profile committed for this workload ({wl})"
        if world == 1 and not a.no_cpu_baseline:
            import oracle

            ns = min(a.cpu_sample, a.nseq)
            host = seqs[: ns * a.length].cpu().numpy()
            t0 = time.perf_counter()
            oset, oacc = oracle.nmost_concat(host, offsets[: ns + 1], a.n, a.k, 4)
            dt = time.perf_counter() - t0
            if ns == a.nseq and last:
                # the oracle has just selected from the very sequences the timed steps used: the last
                # timed step's members must be its members (ids and order bit-exact, floats within the
                # north star's 1e-6 relative)
                elab, edelta, _, _ = oset.members()
                ok = (last["size"] == oset.size and last["positions"].tolist() == elab.tolist()
                      and last["n_accepts"] == oacc
                      and bool(np.allclose(last["delta_jsd"], edelta, rtol=1e-6, atol=1e-13))
                      and abs(last["total_jsd"] - oset.total_jsd) <= 1e-6 * abs(oset.total_jsd))
                out["verified_vs_oracle"] = ok
                if not ok:
                    out["verified_detail"] = {"got_ids": last["positions"].tolist(), "oracle_ids": elab.tolist(),
                                              "got_accepts": last["n_accepts"], "oracle_accepts": int(oacc),
                                              "got_total_jsd": last["total_jsd"], "oracle_total_jsd": oset.total_jsd}
            else:
                out["verified_vs_oracle"] = None  # (the CPU sample is not the whole workload)
            out["cpu_baseline"] = {
                "value": ns / dt, "unit": "sequences/s", "cores": 1, "kind": "port",
                "sample": (f"the first {ns} sequences of the same workload, 1 thread, C restatement of the "
                           "Rust path (oracle/dvs_oracle.c), data in RAM"),
            }
            # the reference's own parallel scheme on every host core (records.py:225-245): contiguous
            # chunks, an independent selection per worker, final_nmost over the winners -- the workers
            # are threads inside the C restatement (oracle/dvs_oracle.c orc_nmost_chunks_mt), so no
            # process is forked from one that holds the GPU
            from diverseseq_amd.parallel import chunk_bounds

            ncores = max(1, min(usable_cores(), ns // max(4 * a.n, 1)))
            t0 = time.perf_counter()
            oracle.nmost_chunks_threads(host, offsets[: ns + 1], chunk_bounds(ns, ncores), a.n, a.k, 4)
            dt_all = time.perf_counter() - t0
            out["cpu_baseline"].update({"value_all_cores": ns / dt_all, "cores_all": ncores,
                                        "sample_all_cores": "same sample, the reference's chunk + merge "
                                        "(-np cores), one worker thread per usable host core"})
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
        if out.get("verified_vs_oracle") is False:
            failed = True
    if world > 1 or b.force_dist:
        b.dist.destroy_process_group()
    if failed:
        raise SystemExit("bench.py: the last timed step's selection differs from the oracle's")


if __name__ == "__main__":
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    main()
