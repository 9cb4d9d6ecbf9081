"""Host side of the distance path: mirrors diverse_seq/distance.py's functions
(mash_sketches :178-227, mash_distances :119-175, mash_distance :230-291,
euclidean_distances :294-332) and the strided chunks of
diverse_seq/cluster.py:607-644, with the arithmetic in libdvs_hip.so."""

from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib, engine

_U32_MAX = 0xFFFFFFFF


def sketch_batch(seqs, k: int, sketch_size: int, num_states: int = 4,
                 mash_canonical: bool = False, ctx: engine.Context | None = None):
    """bottom-`sketch_size` sketches of a batch -> (uint32 [n, stride], lens uint32 [n]);
    stride = min(sketch_size, longest possible sketch)"""
    ctx = ctx or engine.default_context()
    data, offsets = engine.concat(seqs)
    n = len(seqs)
    lens_in = np.diff(offsets.astype(np.int64)) if n else np.zeros(0, dtype=np.int64)
    longest = int(max(0, (lens_in.max() if n else 0) - k + 1))
    if sketch_size < 0 or sketch_size > _U32_MAX:
        raise OverflowError("sketch_size out of range for u32")  # pyo3 usize/u32 extraction
    stride = max(1, min(int(sketch_size), longest))
    sk = np.zeros((n, stride), dtype=np.uint32)
    lens = np.zeros(n, dtype=np.uint32)
    if n and sketch_size:
        ctx.check(ctx._L.dvs_mash_sketch(ctx._h, data.ctypes.data_as(C.c_void_p), 0,
                                         _lib.ptr(offsets, C.c_uint64), n, k, stride, num_states,
                                         int(bool(mash_canonical)), _lib.ptr(sk, C.c_uint32),
                                         _lib.ptr(lens, C.c_uint32)))
    return sk, lens


def mash_sketches(seqs, k: int, sketch_size: int, num_states: int = 4,
                  mash_canonical: bool = False) -> list[list[int]]:
    sk, lens = sketch_batch(seqs, k, sketch_size, num_states, mash_canonical)
    return [sk[i, : int(lens[i])].tolist() for i in range(len(seqs))]


def distances_from_sketches(sk: np.ndarray, lens: np.ndarray, k: int, sketch_size: int, *,
                            row_start: int = 0, row_stride: int = 1, symmetric: bool = True,
                            out: np.ndarray | None = None,
                            ctx: engine.Context | None = None) -> np.ndarray:
    """lower-triangle mash distances for rows row_start, row_start+row_stride, ...
    (compute_mash_chunk_distances, diverse_seq/cluster.py:640-644)"""
    ctx = ctx or engine.default_context()
    sk = np.ascontiguousarray(sk, dtype=np.uint32)
    lens = np.ascontiguousarray(lens, dtype=np.uint32)
    n = sk.shape[0]
    if out is not None and (not isinstance(out, np.ndarray) or out.shape != (n, n) or out.dtype != np.float64
                            or not out.flags["C_CONTIGUOUS"]):
        raise ValueError(f"out must be a C-contiguous float64 array of shape ({n}, {n})")
    dist = np.zeros((n, n), dtype=np.float64) if out is None else out
    ctx.check(ctx._L.dvs_mash_distances(ctx._h, _lib.ptr(sk, C.c_uint32), sk.shape[1],
                                        _lib.ptr(lens, C.c_uint32), n, k,
                                        min(int(sketch_size), _U32_MAX), row_start, row_stride,
                                        int(symmetric), _lib.ptr(dist, C.c_double)))
    return dist


def mash_distance(left_sketch, right_sketch, k: int, sketch_size: int) -> float:
    """diverse_seq/distance.py:230-291 for one pair"""
    l = np.asarray(left_sketch, dtype=np.uint32)
    r = np.asarray(right_sketch, dtype=np.uint32)
    stride = max(1, l.size, r.size)
    sk = np.zeros((2, stride), dtype=np.uint32)
    sk[1, : l.size] = l  # pair (i=1, j=0): left is row i
    sk[0, : r.size] = r
    d = distances_from_sketches(sk, np.array([r.size, l.size], dtype=np.uint32), k, sketch_size)
    return float(d[1, 0])


class Sketches:
    """bottom-s sketches of a batch resident in HBM (dvs_sketches): the hand-over between the two
    stages of ctree (diverse_seq/cluster.py:241-297) without a trip through the host"""

    def __init__(self, seqs, k: int, sketch_size: int, num_states: int = 4, mash_canonical: bool = False,
                 ctx: engine.Context | None = None, dev_ptr: int | None = None, offsets=None,
                 packed: "engine.Packed | None" = None, batch: "engine.SeqBatch | None" = None):
        """seqs: host sequences; or dev_ptr + offsets: bytes already in HBM; or packed + offsets: a packed batch
        (engine.Packed); or batch: an ingested engine.SeqBatch (packed or not)"""
        self.ctx = ctx or (batch.ctx if batch is not None else packed.ctx if packed is not None else engine.default_context())
        if batch is not None:
            offsets = batch.offsets
        elif packed is not None or dev_ptr is not None:
            offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        else:
            data, offsets = engine.concat(seqs)
        self.n = offsets.size - 1
        lens_in = np.diff(offsets.astype(np.int64)) if self.n else np.zeros(0, dtype=np.int64)
        longest = int(max(0, (lens_in.max() if self.n else 0) - k + 1))
        if sketch_size < 0 or sketch_size > _U32_MAX:
            raise OverflowError("sketch_size out of range for u32")
        self.k, self.sketch_size = k, int(sketch_size)
        self.stride = max(1, min(int(sketch_size), longest)) if sketch_size else 0
        h = C.c_void_p()
        L, flag = self.ctx._L, int(bool(mash_canonical))
        if batch is not None:
            self.ctx.check(L.dvs_sketches_build_from_seqbatch(self.ctx._h, batch._h, k, self.stride, num_states, flag,
                                                             C.byref(h)))
        elif packed is not None:
            if num_states != 4:
                raise ValueError("packed sequences have four states")
            self.ctx.check(L.dvs_sketches_build_packed(self.ctx._h, packed._h, _lib.ptr(offsets, C.c_uint64), self.n, k,
                                                      self.stride, flag, C.byref(h)))
        else:
            if dev_ptr is None:
                src, on_dev = data.ctypes.data_as(C.c_void_p), 0
            else:  # sequences already in HBM
                src, on_dev = C.c_void_p(dev_ptr), 1
            self.ctx.check(L.dvs_sketches_build(self.ctx._h, src, on_dev, _lib.ptr(offsets, C.c_uint64), self.n, k,
                                               self.stride, num_states, flag, C.byref(h)))
        self._h = h
        self._source = batch if batch is not None else packed  # (its planes must outlive the sketch kernels)

    def close(self):
        if getattr(self, "_h", None):
            self.ctx._L.dvs_sketches_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def to_host(self):
        """(uint32 [n, stride], lens uint32 [n])"""
        sk = np.zeros((self.n, max(self.stride, 1)), dtype=np.uint32)
        lens = np.zeros(self.n, dtype=np.uint32)
        self.ctx.check(self.ctx._L.dvs_sketches_get(self.ctx._h, self._h, _lib.ptr(sk, C.c_uint32), _lib.ptr(lens, C.c_uint32)))
        return sk, lens

    @classmethod
    def from_device(cls, ctx: engine.Context, sk_ptr: int, lens_ptr: int, n: int, stride: int, k: int,
                    sketch_size: int, keep=None) -> "Sketches":
        """sketches that already sit in HBM (uint32 [n, stride] ascending, uint32 [n] lengths) -- e.g. an
        all_gather's output -- wrapped without a copy; `keep`: whatever owns the two buffers, kept alive beside the
        handle"""
        self = cls.__new__(cls)
        self.ctx, self.n, self.k, self.sketch_size, self.stride = ctx, int(n), k, int(sketch_size), int(stride)
        h = C.c_void_p()
        ctx.check(ctx._L.dvs_sketches_from_device(ctx._h, C.c_void_p(sk_ptr), C.c_void_p(lens_ptr), self.n, self.stride,
                                                  C.byref(h)))
        self._h, self._source = h, keep
        return self

    def copy_to_device(self, dst_ptr: int, dst_stride: int, dst_lens_ptr: int):
        """this batch's sketches into a caller's device buffer whose rows are dst_stride words apart (a collective's
        send buffer); enqueued on the context's stream"""
        self.ctx.check(self.ctx._L.dvs_sketches_copy_to_device(self.ctx._h, self._h, C.c_void_p(dst_ptr), int(dst_stride),
                                                               C.c_void_p(dst_lens_ptr)))

    def distances_device(self, dist_ptr: int, zerodiv_ptr: int, *, row_start: int = 0, row_stride: int = 1,
                         symmetric: bool = True):
        """the visited cells into a device matrix (float64 [n, n]) of the caller's; enqueued, not waited for;
        the uint32 at zerodiv_ptr is set where `distances` would raise ZeroDivisionError"""
        self.ctx.check(self.ctx._L.dvs_sketches_distances_device(self.ctx._h, self._h, self.k, min(self.sketch_size, _U32_MAX),
                                                                 row_start, row_stride, int(symmetric), C.c_void_p(dist_ptr),
                                                                 C.c_void_p(zerodiv_ptr)))

    def distances(self, *, row_start: int = 0, row_stride: int = 1, symmetric: bool = True,
                  out: np.ndarray | None = None) -> np.ndarray:
        if out is not None and (not isinstance(out, np.ndarray) or out.shape != (self.n, self.n)
                                or out.dtype != np.float64 or not out.flags["C_CONTIGUOUS"]):
            raise ValueError(f"out must be a C-contiguous float64 array of shape ({self.n}, {self.n})")
        dist = np.zeros((self.n, self.n), dtype=np.float64) if out is None else out
        self.ctx.check(self.ctx._L.dvs_sketches_distances(self.ctx._h, self._h, self.k, min(self.sketch_size, _U32_MAX),
                                                          row_start, row_stride, int(symmetric), _lib.ptr(dist, C.c_double)))
        return dist


def mash_distances(seqs, k: int, sketch_size: int, num_states: int = 4,
                   mash_canonical: bool = False, ctx: engine.Context | None = None) -> np.ndarray:
    """diverse_seq/distance.py:119-175: sketches, then the symmetric N x N matrix; the sketches stay in
    HBM between the two stages"""
    sk = Sketches(seqs, k, sketch_size, num_states, mash_canonical, ctx=ctx)
    try:
        return sk.distances()
    finally:
        sk.close()


def euclidean_distances(seqs, k: int, num_states: int = 4,
                        ctx: engine.Context | None = None) -> np.ndarray:
    """diverse_seq/distance.py:294-332: ||kfreqs_i - kfreqs_j||_2"""
    ctx = ctx or engine.default_context()
    m = ctx.build_matrix(seqs, k, num_states)
    try:
        dist = np.zeros((m.nrows, m.nrows), dtype=np.float64)
        ctx.check(ctx._L.dvs_euclidean_distances(ctx._h, m._h, _lib.ptr(dist, C.c_double)))
        return dist
    finally:
        m.close()
