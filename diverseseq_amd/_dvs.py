"""Drop-in for the reference's PyO3 extension module ``diverse_seq._dvs``.

Same names, keyword arguments, return attributes and exception types as the
functions/classes registered in the reference's ``src/lib.rs:175-189``; the
arithmetic runs in libdvs_hip.so (HIP kernels for gfx950) through ctypes.

    reference                                     here
    ---------                                     ----
    make_zarr_store(path=None, mode="r")          in-memory store, or an on-disk .dvseqsz directory (zarr_store.py)
    get_seqids_from_store(path)                   ids of an on-disk .dvseqsz store
    nmost_divergent / final_nmost                 dvs_select_run MODE_NMOST
    max_divergent / final_max                     dvs_select_run MODE_MAX
    get_delta_jsd_calculator                      dvs_select_run MODE_SET + dvs_select_delta_jsd
    mash_sketch                                   dvs_mash_sketch
    SummedRecordsResult, LazySeq, ZarrStoreWrapper python classes with the same attributes
"""

from __future__ import annotations

import operator


import numpy as np

from . import _lib, engine

__all__ = [
    "LazySeq", "SummedRecordsResult", "ZarrStoreWrapper", "final_max", "final_nmost",
    "get_delta_jsd_calculator", "get_seqids_from_store", "make_zarr_store", "mash_sketch",
    "max_divergent", "nmost_divergent",
]


# --------------------------------------------------------------------- classes
class SummedRecordsResult:
    """src/records_py.rs:7-88: read-only result of a selection, picklable"""

    __slots__ = ("total_jsd", "records", "mean_delta_jsd", "std_delta_jsd", "cov_delta_jsd",
                 "size", "k", "num_states", "stats")

    def __init__(self):
        self.total_jsd = 0.0
        self.records: list[tuple[str, list[float], float]] = []
        self.mean_delta_jsd = 0.0
        self.std_delta_jsd = 0.0
        self.cov_delta_jsd = 0.0
        self.size = 0
        self.k = 0
        self.num_states = 0
        self.stats = None  # engine statistics (not part of the reference's object)

    @property
    def record_names(self) -> list[str]:
        return [r[0] for r in self.records]

    def __getstate__(self) -> dict:
        return {n: getattr(self, n) for n in ("total_jsd", "records", "mean_delta_jsd",
                                              "std_delta_jsd", "cov_delta_jsd", "size", "k",
                                              "num_states")}

    def __setstate__(self, state: dict) -> None:
        for n in ("total_jsd", "records", "mean_delta_jsd", "std_delta_jsd", "cov_delta_jsd",
                  "size", "k", "num_states"):
            setattr(self, n, state[n])  # KeyError on a missing field, as the reference
        self.stats = None

    def __repr__(self):
        return f"SummedRecordsResult(size={self.size}, total_jsd={self.total_jsd})"


class ZarrStoreWrapper:
    """The python face of src/zarr_py.rs:9-247 over an in-memory store (path None) or an on-disk
    ``.dvseqsz`` directory (diverseseq_amd/zarr_store.py: the reference's Zarr v3 + zstd layout).

    Sequences are deduplicated by content: ``unique_seqids`` holds one id per
    distinct sequence, the last one written (src/zarr_io.rs:376-384;
    reference tests/test_zarr_store.py:48-61).  The reference returns ids in
    FxHashMap iteration order; here it is insertion order (parity is defined on
    "same ordered seqids in -> same ids out", SURVEY.md hard part 6).
    """

    def __init__(self, path: str | None = None, mode: str = "r"):
        # in-memory store: every sequence appended to ONE byte arena (ids -> spans), so a selection
        # over the store's ids in insertion order uploads the arena as it stands, with no gather
        self._arena = bytearray()
        self._index: dict[str, int] = {}   # seqid -> entry
        self._starts: list[int] = []
        self._lens: list[int] = []
        self._keys: list[bytes] = []       # content digest per entry (the reference names content by its xxh3)
        self._meta: dict[str, dict] = {}
        self._spans = None                 # (starts, lens, offsets, laid end to end?) as arrays, rebuilt after writes
        self._ids: list[str] = []          # the ids in insertion order (a list: `ids == self._ids` is one C-level pass)
        self._disk = None
        self.source = ""
        if path is not None:
            from .zarr_store import DvseqszDir

            self._disk = DvseqszDir(str(path), mode)  # FileNotFoundError / RuntimeError as zarr_py.rs:41-57
            self.source = str(path)

    def __repr__(self):
        src = self.source if self.source else "'in memory'"
        return f"ZarrStoreWrapper(source={src}, num members={len(self)})"

    def __contains__(self, key: str) -> bool:
        return key in (self._disk.seqid_to_hash if self._disk else self._index)

    def __len__(self) -> int:
        return len(self._disk.seqid_to_hash if self._disk else self._index)

    # pickling by path (src/zarr_py.rs:90-133); an in-memory store refuses
    def __getstate__(self):
        if self._disk is None:
            raise TypeError("Cannot pickle in-memory store")
        if self._disk.mode != "r":
            self._disk.save_metadata()
        return {"path": self.source}

    def __setstate__(self, state):
        self.__init__(state["path"], "r")

    def __getnewargs__(self):
        return (self.source,)

    def close(self):
        """what Drop does (src/zarr_io.rs:404-422): persist the id map of a writable store"""
        if self._disk is not None and self._disk.mode != "r":
            self._disk.save_metadata()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def write(self, seqid: str, seq, metadata: dict | None = None) -> None:
        data = bytes(seq)
        meta = {str(k): str(v) for k, v in metadata.items()} if metadata else {"source": "unknown"}
        if len(data) == 0:
            raise ValueError(f"Failed to create add {seqid}")
        if self._disk is not None:
            try:
                self._disk.add(seqid, data, meta)
            except (OSError, ValueError, RuntimeError) as e:
                raise ValueError(f"Failed to create add {seqid}") from e
            return
        if seqid in self._index:
            return  # idempotent per seqid (src/zarr_io.rs:217-219)
        import xxhash

        self._index[seqid] = len(self._starts)
        self._ids.append(seqid)
        self._starts.append(len(self._arena))
        self._lens.append(len(data))
        self._keys.append(xxhash.xxh3_128_digest(data))
        self._arena += data  # (BufferError while a selection holds a view of the arena: not re-entrant)
        self._meta[seqid] = meta
        self._spans = None

    def write_log(self, unique_id: str, data: str) -> None:
        pass

    def write_citations(self, data) -> None:
        pass

    def read(self, seqid: str) -> bytes:
        try:
            if self._disk:
                return self._disk.read(seqid)
            i = self._index[seqid]
            a = self._starts[i]
            return bytes(self._arena[a:a + self._lens[i]])
        except (KeyError, OSError, RuntimeError, ValueError):
            raise RuntimeError(f"Failed to create add {seqid}") from None

    def read_metadata(self, seqid: str) -> dict:
        try:
            return self._disk.read_metadata(seqid) if self._disk else dict(self._meta[seqid])
        except (KeyError, OSError, ValueError) as e:
            raise RuntimeError(f"Failed to read metadata for {seqid}: {e}") from None

    def _content_keys(self):
        """(seqid, content key) in insertion order: the content hash on disk, a digest in memory"""
        if self._disk:
            return self._disk.seqid_to_hash.items()
        return zip(self._index, self._keys)

    def _concat(self, ids, own: bool | None = None):
        """in-memory store: (uint8 data, uint64 offsets[n+1]) of the sequences of `ids`, in that order.
        Ids that follow one another in the arena (the usual call: the store's own id list) come back
        as a VIEW of the arena; anything else is gathered."""
        if self._spans is None:
            st, ln = np.array(self._starts, dtype=np.int64), np.array(self._lens, dtype=np.int64)
            off = np.zeros(len(ln) + 1, dtype=np.uint64)
            np.cumsum(ln, out=off[1:], dtype=np.uint64)
            self._spans = (st, ln, off, bool(np.array_equal(st[1:], st[:-1] + ln[:-1])))
        index = self._index
        if own is None:
            own = ids is self._ids or ids == self._ids
        if own:  # the store's own id list: everything is already there
            starts, lens, offsets, contiguous = self._spans
            if len(ids) and contiguous:
                arena = np.frombuffer(self._arena, dtype=np.uint8)
                return arena[starts[0]:starts[0] + int(offsets[-1])], offsets
        else:
            try:
                idx = np.fromiter((index[sid] for sid in ids), dtype=np.int64, count=len(ids))
            except KeyError as e:  # read_uint8_array(..).unwrap() panics (src/record.rs:206)
                raise ValueError(f"sequence {e.args[0]!r} not in store") from None
            starts, lens = self._spans[0][idx], self._spans[1][idx]
        offsets = np.zeros(len(ids) + 1, dtype=np.uint64)
        np.cumsum(lens, out=offsets[1:], dtype=np.uint64)
        if not len(ids):
            return np.zeros(16, dtype=np.uint8), offsets
        arena = np.frombuffer(self._arena, dtype=np.uint8)
        if np.array_equal(starts[1:], starts[:-1] + lens[:-1]):
            return arena[starts[0]:starts[0] + int(offsets[-1])], offsets
        mv = memoryview(self._arena)
        return np.frombuffer(b"".join([mv[a:a + n] for a, n in zip(starts.tolist(), lens.tolist())]),
                             dtype=np.uint8), offsets

    def num_unique(self) -> int:
        return len({key for _, key in self._content_keys()})

    @property
    def unique_seqids(self) -> list[str]:
        last: dict = {}
        for sid, key in self._content_keys():
            last[key] = sid
        return list(last.values())

    def get_seqids(self) -> list[str]:
        return [sid for sid, _ in self._content_keys()]

    def get_lazyseq(self, seqid: str, num_states: int) -> "LazySeq":
        return LazySeq(seqid, self, num_states)

    def get_lazyseqs(self, num_states: int) -> list["LazySeq"]:
        return [self.get_lazyseq(s, num_states) for s in self.get_seqids()]


class LazySeq:
    """src/record.rs:212-269"""


    def __init__(self, seqid: str, storage: ZarrStoreWrapper, num_states: int):
        self.seqid, self._storage, self.num_states = seqid, storage, num_states

    def __repr__(self):
        return f"LazySeq(seqid={self.seqid}, num_states={self.num_states}, storage={self._storage!r})"

    def get_seq(self) -> bytes:
        return self._storage.read(self.seqid)

    def get_kcounts(self, k: int) -> list[int]:
        counts, _, _ = engine.default_context().kmer_counts([self.get_seq()], k, self.num_states)
        return counts[0].tolist()

    def get_kfreqs(self, k: int) -> list[float]:
        counts, totals, _ = engine.default_context().kmer_counts([self.get_seq()], k, self.num_states)
        # record.rs:256-261: no zero check -> NaN for an all-invalid sequence
        with np.errstate(invalid="ignore", divide="ignore"):
            return (counts[0].astype(np.float64) / np.float64(totals[0])).tolist()


# ------------------------------------------------------------------- functions
def make_zarr_store(path: str | None = None, mode: str = "r") -> ZarrStoreWrapper:
    return ZarrStoreWrapper(path, mode)


def get_seqids_from_store(path: str) -> list[str]:
    """src/lib.rs:29-34: every seqid of the store at `path` (opened read-only)"""
    return ZarrStoreWrapper(path, "r").get_seqids()


def _labels(ids) -> np.ndarray:
    """integer identity labels (same id -> same label, first-occurrence order)"""
    label_of = dict.fromkeys(ids)
    if len(label_of) == len(ids):
        return np.arange(len(ids), dtype=np.uint32)
    for i, sid in enumerate(label_of):
        label_of[sid] = i
    return np.fromiter((label_of[sid] for sid in ids), dtype=np.uint32, count=len(ids))


def _gather(store: ZarrStoreWrapper, seqids):
    """the sequences of `seqids` as one stream (uint8 data, uint64 offsets[n+1]) + identity labels"""
    # (a list is used as it is -- nothing here changes it)
    ids = list(store.unique_seqids) if seqids is None else seqids if isinstance(seqids, list) else list(seqids)
    if store._disk is None:
        # (the store's own id list -- one C-level list comparison -- needs no look-ups, and its ids are
        # distinct by construction: the labels are the positions)
        own = ids == store._ids
        data, offsets = store._concat(ids, own)
        if own:
            return ids, data, offsets, np.arange(len(ids), dtype=np.uint32)
    else:
        for sid in ids:
            if sid not in store:
                raise ValueError(f"sequence {sid!r} not in store")
        try:
            data, offsets = store._disk.read_many(ids)  # decoded in place, on a thread pool
        except (KeyError, OSError, ValueError):
            raise RuntimeError("Failed to read the store's arrays") from None
        if not data.size:
            data = np.zeros(16, dtype=np.uint8)
    return ids, data, offsets, _labels(ids)


def _result(sel: engine.Selection, ids, k: int, num_states: int) -> SummedRecordsResult:
    s = sel.summary()
    mem = sel.members(with_freqs=True)
    r = SummedRecordsResult()
    r.total_jsd = s.total_jsd
    r.mean_delta_jsd = s.mean_delta_jsd
    r.std_delta_jsd = s.std_delta_jsd
    r.cov_delta_jsd = s.cov_delta_jsd
    r.size = int(s.size)
    r.k, r.num_states = k, num_states
    r.records = [(ids[int(p)], mem.kfreqs[i].tolist(), float(mem.delta_jsd[i]))
                 for i, p in enumerate(mem.positions)]
    r.stats = {n: getattr(s, n) for n in ("rows_scored", "rows_rechecked", "n_windows", "n_events",
                                          "n_accepts", "n_arbitrated", "scan_ms", "scan_launches", "engine")}
    return r


def _check_k(k: int):
    if k == 0:
        raise ValueError("k cannot be 0")  # src/record.rs:126


def _gather_and_build(store: ZarrStoreWrapper, seqids, n_min: int, k: int, num_states: int):
    """ids, count matrix and labels of a selection's stream.  `data` may be a VIEW of the in-memory
    store's arena: it is let go on every way out of here -- also when the checks or the build raise, since
    a traceback would otherwise keep the frame's view alive and the store's next write fail (BufferError)."""
    ids, data, offsets, labels = _gather(store, seqids)
    try:
        if len(ids) < n_min:
            raise ValueError(f"The number of sequences {len(ids)} is < n {n_min}")
        _check_k(k)
        return ids, engine.default_context().build_matrix_concat(data, offsets, k, num_states), labels
    finally:
        data = None


def nmost_divergent(store: ZarrStoreWrapper, n: int, k: int, num_states: int = 4,
                    seqids=None) -> SummedRecordsResult:
    """src/lib.rs:59-73 -> select_nmost_divergent (src/records.rs:311-342)"""
    ids, m, labels = _gather_and_build(store, seqids, n, k, num_states)
    try:
        sel = m.nmost(n, labels=labels)
        try:
            return _result(sel, ids, k, num_states)
        finally:
            sel.close()
    finally:
        m.close()


def max_divergent(store: ZarrStoreWrapper, min_size: int, max_size: int, k: int,
                  num_states: int = 4, seqids=None, stat: str = "stdev") -> SummedRecordsResult:
    """src/lib.rs:105-137 -> select_max_divergent (src/records.rs:390-454)"""
    ids, m, labels = _gather_and_build(store, seqids, min_size, k, num_states)
    try:
        sel = m.max_divergent(min_size, max_size, stat, labels=labels)
        try:
            return _result(sel, ids, k, num_states)
        finally:
            sel.close()
    finally:
        m.close()


def _merge_inputs(records):
    """get_kmerseqs_and_init_summed_records (src/records.rs:344-360): concatenate the member
    rows of every result in list order; NB k / num_states come back swapped (:353)."""
    ids, rows, labels, label_of = [], [], [], {}
    for sr in records:
        for sid, kfreqs, _ in sr.records:
            ids.append(sid)
            rows.append(kfreqs)
            labels.append(label_of.setdefault(sid, len(label_of)))
    first = records[0] if records else None
    k, ns = (first.num_states, first.k) if first is not None else (0, 0)
    return ids, rows, np.asarray(labels, dtype=np.uint32), k, ns


def final_nmost(records: list[SummedRecordsResult], n: int) -> SummedRecordsResult:
    """src/lib.rs:95-103 -> select_nmost_divergent_final (src/records.rs:363-382)"""
    ids, rows, labels, k, ns = _merge_inputs(records)
    if len(ids) < n:
        raise ValueError(f"The number of sequences {len(ids)} is < n {n}")
    if not ids:
        raise ValueError("records cannot be empty")
    ctx = engine.default_context()
    m = ctx.matrix_from_freqs(np.asarray(rows, dtype=np.float64))
    try:
        sel = m.nmost(n, labels=labels)
        try:
            return _result(sel, ids, k, ns)
        finally:
            sel.close()
    finally:
        m.close()


def final_max(records: list[SummedRecordsResult], min_size: int, max_size: int,
              stat: str = "stdev") -> SummedRecordsResult:
    """src/lib.rs:139-160 -> select_max_divergent_final (src/records.rs:456-507)"""
    ids, rows, labels, k, ns = _merge_inputs(records)
    if len(ids) < min_size:
        raise ValueError(f"The number of sequences {len(ids)} is < n {min_size}")
    if not ids:
        raise ValueError("records cannot be empty")
    ctx = engine.default_context()
    m = ctx.matrix_from_freqs(np.asarray(rows, dtype=np.float64))
    try:
        sel = m.max_divergent(min_size, max_size, stat, labels=labels)
        try:
            return _result(sel, ids, k, ns)
        finally:
            sel.close()
    finally:
        m.close()


class SummedRecordsWrapper:
    """src/records_py.rs:90-125: stateful delta-JSD calculator"""


    def __init__(self, records, k: int, num_states: int = 4):
        _check_k(k)
        self._k, self._num_states = k, num_states
        self._ctx = engine.default_context()
        self._ids = [sid for sid, _ in records]
        label_of: dict[str, int] = {}
        labels = np.asarray([label_of.setdefault(sid, len(label_of)) for sid in self._ids],
                            dtype=np.uint32)
        self._label_of = label_of
        self._matrix = self._ctx.build_matrix([seq for _, seq in records], k, num_states)
        self._sel = self._matrix.as_set(labels=labels)  # make_summed_records, records.rs:509-524

    def delta_jsd(self, seqid: str, seq) -> float:
        q = self._ctx.build_matrix([bytes(seq)], self._k, self._num_states)
        try:
            if int(q.totals()[0]) == 0:
                raise ValueError(f"delta_jsd('{seqid}') failed: No valid k-mers for '{seqid}'")
            lab = self._label_of.get(seqid, 0xFFFFFFFF)
            return float(self._sel.delta_jsd(q, [lab])[0])
        finally:
            q.close()

    def get_result(self) -> SummedRecordsResult:
        return _result(self._sel, self._ids, self._k, self._num_states)


def get_delta_jsd_calculator(seqids_seqs, k: int, num_states: int = 4) -> SummedRecordsWrapper:
    """src/lib.rs:162-171"""
    return SummedRecordsWrapper(list(seqids_seqs), k, num_states)


def mash_sketch(seq_array, k: int, sketch_size: int, num_states: int = 4,
                mash_canonical: bool = False) -> list[int]:
    """src/distance.rs:136-182"""
    from .distance import sketch_batch

    sk, lens = sketch_batch([bytes(seq_array)], k, sketch_size, num_states, mash_canonical)
    return sk[0, : int(lens[0])].tolist()
