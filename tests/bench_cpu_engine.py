"""Stand-in compute for bench.py on a box without a GPU (DVS_BENCH_TEST_ENGINE=bench_cpu_engine, set by
tests/test_bench_launch.py only): the objects bench.py drives -- context, matrix, selection, the exact
mode's stepper -- stated with the CPU oracle, so that the launch, exchange and reporting logic of bench.py
can be run end to end under gloo.  Test infrastructure: never imported by the product."""
import types

import numpy as np

import oracle
from diverseseq_amd import parallel


class OracleStepper:
    """the per-rank compute of the exact mode stated with the oracle: scan this rank's rows of the
    window against the replicated set, pack the first local event, apply the gathered winner"""

    def __init__(self, seqs, owned, mode, n_seed, k, window, max_size=0, stat="stdev", row_of=None):
        self.o, self.seqs, self.k, self.window = oracle, seqs, k, window
        self.mode, self.stat = mode, stat
        self.owned = set(int(p) for p in owned)
        self.row_of = row_of  # stream position -> index into seqs (None: identity)
        self.npos = len(seqs) if row_of is None else len(row_of)
        self.max_size = min(max_size, self.npos)
        seeds = [self._seq(p) for p in range(n_seed)]
        self.set = oracle.SummedRecords.from_seqs(seeds, k, 4, labels=np.arange(n_seed, dtype=np.uint32))
        self.cursor, self.B = n_seed, 4 ** k
        self.n_accepts = self.n_windows = 0

    def _seq(self, p):
        return self.seqs[p if self.row_of is None else int(self.row_of[p])]

    def pack(self):
        import torch

        slot = np.zeros(self.B + 2)
        slot[0] = -1.0
        for p in range(self.cursor, min(self.cursor + self.window, self.npos)):
            if p not in self.owned or self._seq(p).size < self.k:
                continue
            f, h = self.o.to_kfreqs(self._seq(p), 4, self.k)
            if self.set.increases_jsd(f, h, p):
                slot[0], slot[1], slot[2:] = p, h, f
                break
        return torch.from_numpy(slot)

    def _stat(self, s):
        return s.std_delta_jsd if self.stat == "stdev" else s.cov_delta_jsd

    def apply(self, all_slots, world):
        if self.done():
            return
        self.n_windows += 1
        a = all_slots.numpy().reshape(world, -1)
        live = [r for r in range(world) if a[r, 0] >= 0]
        if not live:
            self.cursor = min(self.cursor + self.window, self.npos)
            return
        r = min(live, key=lambda i: a[i, 0])
        p, h, f = int(a[r, 0]), float(a[r, 1]), a[r, 2:].copy()
        if self.mode == "nmost" or self.set.size >= self.max_size:
            self.set.replace_lowest(f, h, p)
            self.n_accepts += 1
        else:  # records.rs:427-451: clone + push, kept iff the statistic rose
            lab, _, ent, fr = self.set.members(with_freqs=True)
            grown = self.o.SummedRecords.new(np.vstack([fr, f[None]]), np.append(ent, h),
                                             np.append(lab, p).astype(np.uint32))
            if self._stat(grown) > self._stat(self.set):
                self.set = grown
                self.n_accepts += 1
        self.cursor = p + 1

    def done(self):
        return self.cursor >= self.npos


class _Members:
    def __init__(self, lab, delta, rows):
        self.positions, self.delta_jsd, self.kfreqs = lab, delta, rows


class Selection:
    def __init__(self, matrix, srec, n_accepts=0, n_windows=0):
        self.matrix, self._s, self._acc, self._win = matrix, srec, n_accepts, n_windows

    def summary(self):
        return types.SimpleNamespace(rows_scored=self.matrix.nrows, scan_ms=0.0, scan_launches=0, n_accepts=self._acc,
                                     n_windows=self._win, n_arbitrated=0, engine=0, total_jsd=self._s.total_jsd,
                                     size=self._s.size)

    def members(self, with_freqs=False):
        lab, delta, _, rows = self._s.members(with_freqs=with_freqs)
        return _Members(lab, delta, rows)

    def close(self):
        pass


class Matrix:
    count_bytes = 4

    def __init__(self, seqs=None, k=0, freqs=None):
        self.seqs, self.k, self.freqs = seqs, k, freqs
        self.nrows = len(seqs) if seqs is not None else freqs.shape[0]
        self.nbins = 4 ** k if seqs is not None else freqs.shape[1]

    def nmost(self, n, window=0):
        if self.freqs is not None:
            return Selection(self, oracle.final_nmost(self.freqs, n))
        data, offsets = oracle.concat(self.seqs)
        srec, acc = oracle.nmost_concat(data, offsets, n, self.k, 4)
        return Selection(self, srec, acc)

    def close(self):
        pass


class Context:
    def __init__(self, device=0):
        pass

    def sync(self):
        pass

    def refresh_knobs(self):
        pass

    def set_timing(self, on):
        pass

    def build_matrix_tensor(self, seqs, offsets, k, num_states):
        host = seqs.numpy()
        off = np.asarray(offsets, dtype=np.int64)
        return Matrix([host[off[i]:off[i + 1]] for i in range(off.size - 1)], k)

    def matrix_from_freqs(self, rows):
        return Matrix(freqs=np.ascontiguousarray(rows, dtype=np.float64))


def nmost_exact(ctx, m, order, n, dev, world, *, window=0, timing=None):
    """bench.py's exact mode: this rank's matrix rows are [seeds, owned rows]; order[p] = row or ROW_REMOTE"""
    owned = np.nonzero((order != parallel.ROW_REMOTE) & (np.arange(order.size) >= n))[0]
    st = OracleStepper(m.seqs, owned, "nmost", n, m.k, window or 64 * world, row_of=np.where(
        order == parallel.ROW_REMOTE, 0, order))
    parallel.drive_exact(st, world, dev, poll_every=4, timing=timing)
    return Selection(m, st.set, st.n_accepts, st.n_windows)
