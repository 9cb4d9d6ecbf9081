"""One process per GPU: the reference's chunk-and-merge parallelism
(diverse_seq/records.py:225-245 `apply_app`, diverse_seq/util.py:82-102
`determine_chunk_size` / `chunked`) over ranks instead of worker processes.

Each rank runs the greedy selection on its contiguous chunk of the candidate
stream with no communication; the only exchange step is ONE all_gather of every
rank's winning frequency rows (n x 4^k f64 per rank; RCCL over xGMI on GPUs,
gloo in the CPU tests), after which every rank runs the reference's merge
(`final_nmost` / `final_max`, src/records.rs:363-382,456-507) on the G*n rows.
The result equals `final_nmost` / `final_max` over the per-chunk results taken in CHUNK order.
(The reference's `apply_app` collects its workers' results in completion order,
records.py:236-243, and the merge is order-dependent -- the first n rows seed the set -- so the
reference's own `-np G` output is this one whenever its workers finish in chunk order, and one of
the other orderings otherwise; the order here is fixed so that runs are reproducible.)
"""

from __future__ import annotations

import numpy as np


def determine_chunk_size(total_items: int, num_chunks: int) -> list[int]:
    """diverse_seq/util.py:82-90"""
    base, rem = divmod(total_items, num_chunks)
    return [base + 1 if i < rem else base for i in range(num_chunks)]


def chunk_bounds(total_items: int, num_chunks: int) -> list[tuple[int, int]]:
    """diverse_seq/util.py:93-102: contiguous [start, end) per chunk"""
    sizes = determine_chunk_size(total_items, num_chunks)
    ends = np.cumsum(sizes)
    starts = np.concatenate([[0], ends[:-1]])
    return [(int(s), int(e)) for s, e in zip(starts, ends)]


def gather_winners(rows: np.ndarray, ids: np.ndarray, world: int, device, cap: int, timing: dict | None = None):
    """all_gather of every rank's member rows (padded to `cap` rows) and their global ids.
    Returns (rows [sum sizes, B], ids [sum sizes]) in rank order, i.e. the list order
    `apply_app` hands to the merge.  timing: {"collective_ms": [...]} gets the two collectives' duration
    (host clock: this is the host-staged form)."""
    import time

    import torch
    import torch.distributed as dist

    B = rows.shape[1]
    pad = np.zeros((cap, B), dtype=np.float64)
    pad[: rows.shape[0]] = rows
    meta = np.full(cap + 1, -1, dtype=np.int64)
    meta[0] = rows.shape[0]
    meta[1: 1 + rows.shape[0]] = ids
    t_rows = torch.from_numpy(pad).to(device)
    t_meta = torch.from_numpy(meta).to(device)
    all_rows = torch.empty((world * cap, B), dtype=torch.float64, device=device)
    all_meta = torch.empty((world * (cap + 1),), dtype=torch.int64, device=device)
    t0 = time.perf_counter()
    dist.all_gather_into_tensor(all_rows, t_rows)
    dist.all_gather_into_tensor(all_meta, t_meta)
    if timing is not None:
        timing.setdefault("collective_ms", []).append((time.perf_counter() - t0) * 1e3)
    all_rows = all_rows.cpu().numpy().reshape(world, cap, B)
    all_meta = all_meta.cpu().numpy().reshape(world, cap + 1)
    out_rows, out_ids = [], []
    for r in range(world):
        n = int(all_meta[r, 0])
        out_rows.append(all_rows[r, :n])
        out_ids.append(all_meta[r, 1: 1 + n])
    return np.concatenate(out_rows), np.concatenate(out_ids)


def gather_winners_device(ctx, sel, world: int, device, cap: int, *, shared_stream: bool = False,
                          buffers: dict | None = None, timing: dict | None = None):
    """The same exchange with no host round trip: the members are gathered on the device
    straight into the all_gather's send buffer.  Returns device tensors (rows [world*cap, B],
    meta [world*cap, 2] = (chunk-local position, valid)); ranks with fewer than `cap` members
    contribute zero rows flagged invalid, which the merge skips.

    shared_stream: the ctx launches on torch's current stream (Context(stream=...)), so the gather
    kernel, the collectives and whatever consumes their output are ordered by the stream alone and
    nothing here waits on the host.  buffers: a dict the four tensors are kept in between calls.
    timing: {"collective_events": [...]} collects (start, end) device events around the two collectives
    (no host wait here; `collective_times` turns them into milliseconds afterwards)."""
    import torch
    import torch.distributed as dist

    B = sel.matrix.nbins
    key = (world, cap, B, str(device))
    if buffers is not None and buffers.get("key") == key:
        t_rows, t_meta, all_rows, all_meta = buffers["tensors"]
    else:
        t_rows = torch.empty((cap, B), dtype=torch.float64, device=device)
        t_meta = torch.empty((cap, 2), dtype=torch.float64, device=device)
        all_rows = torch.empty((world * cap, B), dtype=torch.float64, device=device)
        all_meta = torch.empty((world * cap, 2), dtype=torch.float64, device=device)
        if buffers is not None:
            buffers["key"], buffers["tensors"] = key, (t_rows, t_meta, all_rows, all_meta)
        if not shared_stream:
            torch.cuda.current_stream().synchronize()  # the buffers exist before the library writes them
    sel.gather_members(t_rows.data_ptr(), t_meta.data_ptr(), cap)
    if not shared_stream:
        ctx.sync()
    ev = None
    if timing is not None and len(timing.setdefault("collective_events", [])) < 256:
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
    dist.all_gather_into_tensor(all_rows, t_rows)
    dist.all_gather_into_tensor(all_meta, t_meta)
    if ev is not None:
        ev[1].record()
        timing["collective_events"].append(ev)
    if not shared_stream:
        torch.cuda.current_stream().synchronize()
    return all_rows, all_meta


def collective_times(timing: dict | None):
    """device events collected by gather_winners_device -> timing["collective_ms"] (waits for them)"""
    if timing and timing.get("collective_events"):
        import torch

        torch.cuda.synchronize()
        timing.setdefault("collective_ms", []).extend(a.elapsed_time(b) for a, b in timing.pop("collective_events"))
    return timing


def _global_ids(all_meta, chunk_starts, cap: int, src_rows=None) -> np.ndarray:
    """global stream position of every row of the merge matrix (-1 for padding); src_rows maps a
    matrix row to the gathered row it was copied from (the matrix keeps the real rows first)"""
    meta = all_meta.cpu().numpy()
    starts = np.repeat(np.asarray(chunk_starts, dtype=np.int64), cap)
    gids = np.where(meta[:, 1] != 0, meta[:, 0].astype(np.int64) + starts, -1)
    return gids if src_rows is None else gids[np.asarray(src_rows, dtype=np.int64)]


class _LazyGlobalIds:
    """global ids of the merge matrix's rows, copied from the device on first use"""

    def __init__(self, all_meta, chunk_starts, cap, matrix=None):
        self._args, self._matrix, self._val = (all_meta, chunk_starts, cap), matrix, None

    def get(self):
        if self._val is None:
            src = self._matrix.source_rows() if self._matrix is not None else None
            self._val = _global_ids(*self._args, src_rows=src)
        return self._val


def merge_nmost(ctx, sel, n: int, rank: int, world: int, chunk_start: int, device,
                chunk_starts=None, *, shared_stream: bool = False, buffers: dict | None = None,
                timing: dict | None = None):
    """exchange the winners and run final_nmost on the device; returns the merged Selection
    (its member positions index the gathered row list; `merged.global_ids` maps them to
    global stream positions).  With `chunk_starts` (every rank's chunk start) on a GPU the
    rows never leave HBM."""
    if chunk_starts is not None and getattr(device, "type", str(device)) == "cuda":
        all_rows, all_meta = gather_winners_device(ctx, sel, world, device, cap=n, shared_stream=shared_stream,
                                                   buffers=buffers, timing=timing)
        m = ctx.matrix_from_device_freqs(all_rows.data_ptr(), world * n, sel.matrix.nbins,
                                         all_meta.data_ptr())
        merged = m.nmost(n)
        merged._lazy_gids = _LazyGlobalIds(all_meta, chunk_starts, n, m)  # (device -> host copies: on demand)
        merged._keep = m
        return merged
    mem = sel.members(with_freqs=True)
    ids = mem.positions.astype(np.int64) + chunk_start
    rows, gids = gather_winners(mem.kfreqs, ids, world, device, cap=n, timing=timing)
    m = ctx.matrix_from_freqs(rows)
    merged = m.nmost(n)  # ids are unique across chunks: identity labels
    merged.global_ids = gids
    merged._keep = m  # the matrix must outlive the selection
    return merged


def merge_max(ctx, sel, min_size: int, max_size: int, stat: str, chunk_start: int, world: int,
              device, cap: int, chunk_starts=None, *, shared_stream: bool = False, buffers: dict | None = None):
    """exchange the winners and run final_max (select_max_divergent_final, src/records.rs:456-507) on
    the device.  `cap`: the largest set any rank may bring (every rank passes the same value; a rank
    with fewer members pads).  With `chunk_starts` on a GPU the winners never leave HBM -- gathered
    into the all_gather's send buffer, wrapped as a frequency matrix (padding rows moved behind the
    real ones) and merged there, as merge_nmost does."""
    if chunk_starts is not None and getattr(device, "type", str(device)) == "cuda":
        all_rows, all_meta = gather_winners_device(ctx, sel, world, device, cap=cap, shared_stream=shared_stream,
                                                   buffers=buffers)
        m = ctx.matrix_from_device_freqs(all_rows.data_ptr(), world * cap, sel.matrix.nbins,
                                         all_meta.data_ptr())
        merged = m.max_divergent(min_size, max_size, stat)
        merged._lazy_gids = _LazyGlobalIds(all_meta, chunk_starts, cap, m)
        merged._keep = m
        return merged
    mem = sel.members(with_freqs=True)
    ids = mem.positions.astype(np.int64) + chunk_start
    rows, gids = gather_winners(mem.kfreqs, ids, world, device, cap=cap)
    m = ctx.matrix_from_freqs(rows)
    merged = m.max_divergent(min_size, max_size, stat,
                             labels=np.arange(rows.shape[0], dtype=np.uint32))
    merged.global_ids = gids
    merged._keep = m
    return merged


# ---------------------------------------------------------------------------------------
# Exact multi-GPU selection (same answer as one GPU / the reference's `-np 1`): rows are
# sharded block-cyclically by global candidate order, the set state is replicated, and each
# greedy step costs ONE small collective (BASELINE north_star): every rank packs
# its own first event of the window -- position, row entropy, the candidate's 4^k frequency
# row -- into its slot of an all_gather (world x (4^k + 2) f64: 8 x 32 KB at k=6); every rank
# then takes the earliest position among the slots and resolves that candidate against its
# replica of the set (identical arithmetic, so the replicas stay bit-identical).
ROW_REMOTE = 0xFFFFFFFF


def shard_order(npos: int, n_seed: int, rank: int, world: int, block: int = 256):
    """Block-cyclic ownership of stream positions >= n_seed; positions < n_seed (the seeds) are
    replicated on every rank.  Returns (owned positions ascending, order[npos] uint32) where
    order[p] is the row in THIS rank's matrix (seeds first, then the owned rows) or ROW_REMOTE."""
    pos = np.arange(n_seed, npos, dtype=np.int64)
    mine = ((pos - n_seed) // block) % world == rank
    owned = pos[mine]
    order = np.full(npos, ROW_REMOTE, dtype=np.uint32)
    order[:n_seed] = np.arange(n_seed, dtype=np.uint32)
    order[owned] = n_seed + np.arange(owned.size, dtype=np.uint32)
    return owned, order


class HipStepper:
    """the library's stepwise entry points (include/dvs_hip.h) behind the driver below"""

    def __init__(self, ctx, sel, nbins: int, device):
        import torch

        self.ctx, self.sel, self.nbins = ctx, sel, nbins
        self.slot = torch.zeros(nbins + 2, dtype=torch.float64, device=device)
        self._no_peek = False

    def pack(self):
        import ctypes as C

        self.ctx.check(self.ctx._L.dvs_select_step_pack(self.ctx._h, self.sel._h, C.c_void_p(self.slot.data_ptr())))
        return self.slot

    def apply(self, all_slots, world: int):
        import ctypes as C

        self.ctx.check(self.ctx._L.dvs_select_step_apply(self.ctx._h, self.sel._h, C.c_void_p(all_slots.data_ptr()),
                                                         world))

    def peek(self, lag: int):
        """(status, must_poll) behind the apply launch `lag` launches back, without a sync (dvs_select_step_peek);
        None when the selection keeps no status history -- the driver then polls"""
        import ctypes as C

        from . import _lib

        if self._no_peek:
            return None
        status, must = C.c_uint32(), C.c_int()
        rc = self.ctx._L.dvs_select_step_peek(self.ctx._h, self.sel._h, lag, C.byref(status), C.byref(must))
        if rc == _lib.ERR_UNSUPPORTED:
            self._no_peek = True
            return None
        self.ctx.check(rc)
        return status.value, bool(must.value)

    def done(self) -> bool:
        import ctypes as C

        status, cursor = C.c_uint32(), C.c_uint64()
        self.ctx.check(self.ctx._L.dvs_select_step_poll(self.ctx._h, self.sel._h, C.byref(status), C.byref(cursor)))
        if status.value not in (0, 1):
            raise RuntimeError(f"selection engine in state {status.value}")
        return status.value == 1


def drive_exact(stepper, world: int, device, *, poll_every: int = 16, timing: dict | None = None):
    """The greedy loop of the exact mode: `poll_every` steps are enqueued between two looks at the
    engine's status (a finished selection turns the remaining steps into no-ops).  One all_gather per
    step.  `stepper` is HipStepper on GPUs; the CPU tests drive the oracle through the same loop.
    timing: {"collective_ms": [...]} gets the duration of the first steps' all_gather (device events)."""
    import torch
    import torch.distributed as dist

    all_slots = None
    on_gpu = getattr(device, "type", str(device)) == "cuda"

    flat_ok = [True]

    def gather(dst, src):
        if flat_ok[0]:
            try:
                dist.all_gather_into_tensor(dst, src)
                return
            except RuntimeError:  # (a backend without the flat form for this device: per-rank views)
                flat_ok[0] = False
        dist.all_gather(list(dst.view(world, -1).unbind(0)), src)
    events = []
    peek = getattr(stepper, "peek", None)
    first = True
    batch = poll_every
    while True:
        # a look at the status every `poll_every` steps.  Where the engine keeps a status history the look is at the word
        # of the launch `poll_every` steps back -- no sync, the queue stays full, and every rank reads the same word at the
        # same step count (the number of all_gathers must not depend on timing) -- and the syncing poll (which also
        # arbitrates ties and drains the accepted rows' ring) runs only when that word, or the ring, asks for it.
        look = None if first or peek is None else peek(batch)
        first = False
        if look is None or look[0] != 0 or look[1]:
            if stepper.done():
                break
        if look is not None:
            batch = min(poll_every, 4)  # (a look costs nothing now, and the steps enqueued behind the end are no-ops that still launch)
        for _ in range(batch):
            slot = stepper.pack()
            if world > 1:
                if all_slots is None:
                    all_slots = torch.empty(world * slot.numel(), dtype=slot.dtype, device=slot.device)  # (flat: gloo wants it so)
                if timing is not None and on_gpu and len(events) < 64:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    gather(all_slots, slot)
                    e1.record()
                    events.append((e0, e1))
                else:
                    gather(all_slots, slot)
                stepper.apply(all_slots, world)
            else:
                stepper.apply(slot, 1)
    if timing is not None and events:
        torch.cuda.synchronize()
        timing.setdefault("collective_ms", []).extend(a.elapsed_time(b) for a, b in events)


def _select_exact(ctx, matrix, order, mode, n_seed, device, world, *, max_size=0, stat=0, window=0,
                  poll_every=16, timing=None):
    from . import _lib

    sel = matrix.select(mode, n_seed, max_size=max_size, stat=stat, order=order, window=window or 4096 * world,
                        flags=_lib.SELECT_STEPWISE)
    drive_exact(HipStepper(ctx, sel, matrix.nbins, device), world, device, poll_every=poll_every, timing=timing)
    return sel


def nmost_exact(ctx, matrix, order: np.ndarray, n: int, device, world: int, *, window: int = 0,
                poll_every: int = 16, timing: dict | None = None):
    """Greedy nmost over a row-sharded stream (select_nmost_divergent, src/records.rs:311-342).
    `ctx` must have been created on the torch stream that is current here
    (Context(device, stream=torch.cuda.current_stream().cuda_stream)), so the library's kernels and
    the collective are ordered on one stream with no host syncs in between.  Returns the Selection;
    its member positions are global stream positions."""
    from . import _lib

    return _select_exact(ctx, matrix, order, _lib.MODE_NMOST, n, device, world, window=window,
                         poll_every=poll_every, timing=timing)


def max_exact(ctx, matrix, order: np.ndarray, min_size: int, max_size: int, stat: str, device, world: int, *,
              window: int = 0, poll_every: int = 16, timing: dict | None = None):
    """The same for select_max_divergent (src/records.rs:390-454): the set grows while the standard
    deviation (or coefficient of variation) of the members' delta_jsd rises; every rank takes the
    commit-or-rollback decision from the same numbers."""
    from . import _lib

    st = _lib.STAT_STDEV if stat == "stdev" else _lib.STAT_COV
    return _select_exact(ctx, matrix, order, _lib.MODE_MAX, min_size, device, world, max_size=max_size, stat=st,
                         window=window, poll_every=poll_every, timing=timing)


# ---------------------------------------------------------------------------------------
# ctree distances over ranks (diverse_seq/cluster.py:607-644 `dvs_par_ctree`):
# each rank sketches its contiguous share of the sequences, ONE all_gather hands every rank
# all N sketches (N x s u32; 12 MB at N=1000, s=3000), rank g fills rows g, g+G, ... of the
# lower triangle (the reference's stride, which balances the triangle), and a SUM all-reduce
# of the N x N f64 assembles it; symmetrise as cluster.py:294 does (D + D^T - diag).
def mash_distances_sharded(seqs, k: int, sketch_size: int, rank: int, world: int, device, *,
                           num_states: int = 4, mash_canonical: bool = False,
                           sketcher=None, pair_rows=None, collectives=None) -> np.ndarray:
    """`seqs`: the full list (every rank sees the store, as the reference's workers do); only
    this rank's chunk is sketched here.  `sketcher(chunk) -> (uint32 [m, stride], uint32 [m])`
    and `pair_rows(sk, lens, row_start, row_stride) -> N x N lower-triangle rows` default to the
    HIP kernels (dvs_mash_sketch / dvs_mash_distances); the CPU tests inject the oracle.  On a GPU with the
    defaults everything stays in HBM (`_mash_distances_sharded_device`); `collectives` = (all_gather_into_tensor,
    all_reduce) replaces torch.distributed's there (the GPU test of two ranks sharing one card runs over gloo,
    which has no all_gather of device tensors)."""
    import torch
    import torch.distributed as dist

    from . import distance

    n = len(seqs)
    if sketcher is None and pair_rows is None and torch.device(device).type == "cuda":
        return _mash_distances_sharded_device(seqs, k, sketch_size, rank, world, device, num_states, mash_canonical,
                                              collectives)
    if sketcher is None:
        def sketcher(chunk):
            return distance.sketch_batch(chunk, k, sketch_size, num_states, mash_canonical)
    if pair_rows is None:
        def pair_rows(sk, lens, row_start, row_stride):
            return distance.distances_from_sketches(sk, lens, k, sketch_size, row_start=row_start,
                                                    row_stride=row_stride, symmetric=False)
    longest = max((len(s) for s in seqs), default=0) - k + 1
    stride = max(1, min(int(sketch_size), max(0, longest)))  # same on every rank
    bounds = chunk_bounds(n, world)
    lo, hi = bounds[rank]
    cap = max(1, max(e - s for s, e in bounds))
    sk_pad = np.zeros((cap, stride), dtype=np.uint32)
    len_pad = np.zeros(cap, dtype=np.int32)
    if hi > lo:
        sk, lens = sketcher(seqs[lo:hi])
        sk_pad[: hi - lo, : sk.shape[1]] = sk[:, :stride]
        len_pad[: hi - lo] = lens
    t_sk = torch.from_numpy(sk_pad.view(np.int32)).to(device)
    t_len = torch.from_numpy(len_pad).to(device)
    all_sk = torch.empty((world * cap, stride), dtype=torch.int32, device=device)
    all_len = torch.empty((world * cap,), dtype=torch.int32, device=device)
    dist.all_gather_into_tensor(all_sk, t_sk)
    dist.all_gather_into_tensor(all_len, t_len)
    all_sk = all_sk.cpu().numpy().view(np.uint32).reshape(world, cap, stride)
    all_len = all_len.cpu().numpy().astype(np.uint32).reshape(world, cap)
    sk_all = np.concatenate([all_sk[r, : e - s] for r, (s, e) in enumerate(bounds)])
    len_all = np.concatenate([all_len[r, : e - s] for r, (s, e) in enumerate(bounds)])
    part = pair_rows(sk_all, len_all, rank, world)  # rows rank, rank + world, ...
    t_d = torch.from_numpy(np.ascontiguousarray(part)).to(device)
    dist.all_reduce(t_d, op=dist.ReduceOp.SUM)  # every entry is written by exactly one rank
    lower = t_d.cpu().numpy()
    return lower + lower.T - np.diag(np.diag(lower))


def _mash_distances_sharded_device(seqs, k, sketch_size, rank, world, device, num_states, mash_canonical,
                                   collectives=None) -> np.ndarray:
    """The same on the HIP kernels with nothing but the result crossing PCIe: this rank's sketches are copied
    device-to-device into the all_gather's send buffer, the gathered N sketches are read in place by the pair kernel
    (rows rank, rank + world, ... of the lower triangle into a device matrix), the SUM all_reduce assembles it and the
    symmetrisation (cluster.py:294) runs on the device; the N x N matrix is copied out once, at the end."""
    import torch
    import torch.distributed as dist

    from . import distance, engine

    n = len(seqs)
    dev = torch.device(device)
    ctx = engine.default_context()
    gather, reduce_ = collectives if collectives else (dist.all_gather_into_tensor, dist.all_reduce)
    longest = max((len(s) for s in seqs), default=0) - k + 1
    stride = max(1, min(int(sketch_size), max(0, longest)))  # same on every rank
    bounds = chunk_bounds(n, world)
    lo, hi = bounds[rank]
    cap = max(1, max(e - s for s, e in bounds))
    t_sk = torch.zeros((cap, stride), dtype=torch.int32, device=dev)
    t_len = torch.zeros((cap,), dtype=torch.int32, device=dev)
    own = None
    if hi > lo and sketch_size:
        own = distance.Sketches(seqs[lo:hi], k, sketch_size, num_states, mash_canonical, ctx=ctx)
        torch.cuda.current_stream(dev).synchronize()  # (the send buffers' zero fill, on torch's stream)
        if own.stride:
            own.copy_to_device(t_sk.data_ptr(), stride, t_len.data_ptr())
        ctx.sync()  # (the library's stream -> torch's stream: the collective reads the buffers)
    all_sk = torch.empty((world * cap, stride), dtype=torch.int32, device=dev)
    all_len = torch.empty((world * cap,), dtype=torch.int32, device=dev)
    gather(all_sk, t_sk)
    gather(all_len, t_len)
    if all(e - s == cap for s, e in bounds):  # (equal chunks: the gathered rows ARE the N sketches, in order)
        sk_all, len_all = all_sk, all_len
    else:  # the real rows of every rank's padded block, gathered on the device
        idx = torch.tensor([r * cap + i for r, (s, e) in enumerate(bounds) for i in range(e - s)], dtype=torch.long, device=dev)
        sk_all, len_all = all_sk.index_select(0, idx).contiguous(), all_len.index_select(0, idx).contiguous()
    t_d = torch.zeros((n, n), dtype=torch.float64, device=dev)
    t_flag = torch.zeros((1,), dtype=torch.int32, device=dev)
    torch.cuda.current_stream(dev).synchronize()  # (torch's stream -> the library's)
    if n >= 2:
        gathered = distance.Sketches.from_device(ctx, sk_all.data_ptr(), len_all.data_ptr(), n, stride, k, sketch_size,
                                                 keep=(sk_all, len_all))
        try:
            gathered.distances_device(t_d.data_ptr(), t_flag.data_ptr(), row_start=rank, row_stride=world, symmetric=False)
            ctx.sync()
        finally:
            gathered.close()
    if own is not None:
        own.close()
    reduce_(t_d, op=dist.ReduceOp.SUM)  # every entry is written by exactly one rank
    reduce_(t_flag, op=dist.ReduceOp.MAX)
    if int(t_flag.item()):
        raise ZeroDivisionError("division by zero")  # two empty sketches (distance.py:283)
    full = t_d + t_d.T - torch.diag(torch.diagonal(t_d))
    return full.cpu().numpy()
