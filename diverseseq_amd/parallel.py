"""One process per GPU: the reference's chunk-and-merge parallelism
(diverse_seq/records.py:225-245 `apply_app`, diverse_seq/util.py:82-102
`determine_chunk_size` / `chunked`) over ranks instead of worker processes.

Each rank runs the greedy selection on its contiguous chunk of the candidate
stream with no communication; the only exchange step is ONE all_gather of every
rank's winning frequency rows (n x 4^k f64 per rank; RCCL over xGMI on GPUs,
gloo in the CPU tests), after which every rank runs the reference's merge
(`final_nmost` / `final_max`, src/records.rs:363-382,456-507) on the G*n rows.
The result equals the reference run with `-np G`.
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


def gather_winners(rows: np.ndarray, ids: np.ndarray, world: int, device, cap: int):
    """all_gather of every rank's member rows (padded to `cap` rows) and their global ids.
    Returns (rows [sum sizes, B], ids [sum sizes]) in rank order, i.e. the list order
    `apply_app` hands to the merge."""
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
    dist.all_gather_into_tensor(all_rows, t_rows)
    dist.all_gather_into_tensor(all_meta, t_meta)
    all_rows = all_rows.cpu().numpy().reshape(world, cap, B)
    all_meta = all_meta.cpu().numpy().reshape(world, cap + 1)
    out_rows, out_ids = [], []
    for r in range(world):
        n = int(all_meta[r, 0])
        out_rows.append(all_rows[r, :n])
        out_ids.append(all_meta[r, 1: 1 + n])
    return np.concatenate(out_rows), np.concatenate(out_ids)


def merge_nmost(ctx, sel, n: int, rank: int, world: int, chunk_start: int, device):
    """exchange the winners and run final_nmost on the device; returns the merged Selection
    (its member positions index the gathered row list; see `gather_winners`)."""
    mem = sel.members(with_freqs=True)
    ids = mem.positions.astype(np.int64) + chunk_start
    rows, gids = gather_winners(mem.kfreqs, ids, world, device, cap=n)
    m = ctx.matrix_from_freqs(rows)
    merged = m.nmost(n, labels=np.arange(rows.shape[0], dtype=np.uint32))
    merged.global_ids = gids
    merged._keep = m  # the matrix must outlive the selection
    return merged


def merge_max(ctx, sel, min_size: int, max_size: int, stat: str, chunk_start: int, world: int,
              device, cap: int):
    mem = sel.members(with_freqs=True)
    ids = mem.positions.astype(np.int64) + chunk_start
    rows, gids = gather_winners(mem.kfreqs, ids, world, device, cap=cap)
    m = ctx.matrix_from_freqs(rows)
    merged = m.max_divergent(min_size, max_size, stat,
                             labels=np.arange(rows.shape[0], dtype=np.uint32))
    merged.global_ids = gids
    merged._keep = m
    return merged
