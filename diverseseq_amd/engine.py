"""Thin object layer over the C ABI: device context, count matrix, selection.

Host-side plumbing only; every number comes out of libdvs_hip.so.
"""

from __future__ import annotations

import ctypes as C
import weakref
from dataclasses import dataclass

import numpy as np

from . import _lib


def concat(seqs) -> tuple[np.ndarray, np.ndarray]:
    """list of byte-like / uint8 arrays -> (concatenated uint8, uint64 offsets[n+1])"""
    if seqs and all(type(s) is bytes for s in seqs):  # the store's own representation: one C-level join
        offsets = np.zeros(len(seqs) + 1, dtype=np.uint64)
        offsets[1:] = np.cumsum(np.fromiter(map(len, seqs), dtype=np.int64, count=len(seqs)), dtype=np.uint64)
        total = int(offsets[-1])
        data = np.frombuffer(b"".join(seqs), dtype=np.uint8) if total else np.zeros(16, dtype=np.uint8)
        return data, offsets
    arrs = []
    for s in seqs:
        if isinstance(s, (bytes, bytearray, memoryview)):
            arrs.append(np.frombuffer(s, dtype=np.uint8))
        else:
            arrs.append(np.ascontiguousarray(s, dtype=np.uint8).reshape(-1))
    offsets = np.zeros(len(arrs) + 1, dtype=np.uint64)
    if arrs:
        offsets[1:] = np.cumsum([a.size for a in arrs], dtype=np.uint64)
    total = int(offsets[-1])
    data = np.concatenate(arrs) if total else np.zeros(16, dtype=np.uint8)
    return np.ascontiguousarray(data), offsets


_GB_DROP = bytes(range(48, 58)) + b" \t\r\n"


def genbank_to_fasta(raw: bytes) -> bytes:
    """LOCUS name + the ORIGIN..// block of every record of a GenBank flat file, re-framed as FASTA
    records (position numbers and blanks removed); the format the reference reaches through
    cogent3's get_format_parser(path, "genbank") (diverse_seq/io.py:94-96).  A record without an
    ORIGIN block becomes an empty record."""
    out, pos = [], 0
    while True:
        loc = 0 if raw.startswith(b"LOCUS", pos) and pos == 0 else raw.find(b"\nLOCUS", pos)
        if loc < 0:
            break
        if raw[loc:loc + 1] == b"\n":
            loc += 1
        eol = raw.find(b"\n", loc)
        eol = len(raw) if eol < 0 else eol
        toks = raw[loc:eol].split()
        name = toks[1] if len(toks) > 1 else b""
        end = raw.find(b"\n//", eol)
        end = len(raw) if end < 0 else end
        org = raw.find(b"\nORIGIN", eol, end)
        seq = b""
        if org >= 0:
            first = raw.find(b"\n", org + 1, end)
            if first >= 0:
                seq = raw[first + 1:end].translate(None, _GB_DROP)
        out.append(b">" + name + b"\n" + seq + b"\n")
        pos = end + 1
        if pos >= len(raw):
            break
    return b"".join(out)


_live_contexts = weakref.WeakSet()


def refresh_all_knobs():
    """every live Context re-reads the DVS_* environment switches (tests / A-B runs that flip one)"""
    for c in list(_live_contexts):
        if c._h:
            c.refresh_knobs()


class Context:
    """one per process per GPU (dvs_ctx)"""

    def __init__(self, device: int = -1, stream: int | None = None):
        self._L = _lib.load()
        h = C.c_void_p()
        rc = self._L.dvs_ctx_create(device, C.c_void_p(stream) if stream else None, C.byref(h))
        if rc:
            _lib.raise_for(rc, None)
        self._h = h
        _live_contexts.add(self)

    def close(self):
        if getattr(self, "_h", None):
            self._L.dvs_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, rc: int):
        _lib.raise_for(rc, self._h)

    def sync(self):
        self.check(self._L.dvs_ctx_sync(self._h))

    def refresh_knobs(self):
        """re-read the DVS_* environment switches (they are otherwise read once, when the context is made)"""
        self.check(self._L.dvs_ctx_refresh_knobs(self._h))

    def set_timing(self, on: bool):
        self.check(self._L.dvs_ctx_set_timing(self._h, int(on)))

    def device_info(self) -> dict:
        name = C.create_string_buffer(256)
        ncu, mem = C.c_int(), C.c_uint64()
        self.check(self._L.dvs_ctx_device_info(self._h, name, 256, C.byref(ncu), C.byref(mem)))
        return {"name": name.value.decode(), "n_cu": ncu.value, "hbm_bytes": mem.value}

    # ---- matrices -----------------------------------------------------------
    def build_matrix(self, seqs, k: int, num_states: int = 4) -> "CountMatrix":
        """k-mer count matrix of host sequences (list of uint8 arrays / bytes)"""
        data, offsets = concat(seqs)
        return self.build_matrix_concat(data, offsets, k, num_states)

    def build_matrix_concat(self, data: np.ndarray, offsets: np.ndarray, k: int,
                            num_states: int = 4) -> "CountMatrix":
        h = C.c_void_p()
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        self.check(self._L.dvs_matrix_build(self._h, data.ctypes.data_as(C.c_void_p), 0,
                                            _lib.ptr(offsets, C.c_uint64), offsets.size - 1, k,
                                            num_states, C.byref(h)))
        return CountMatrix(self, h, k, num_states)

    def build_matrix_device(self, dev_ptr: int, offsets: np.ndarray, k: int,
                            num_states: int = 4) -> "CountMatrix":
        """sequences already resident in HBM (e.g. a torch uint8 tensor's data_ptr()).  The call
        does not wait for its kernels: keep the buffer alive and unmodified until the matrix is
        first used from the host side (a selection, counts(), ctx.sync())"""
        h = C.c_void_p()
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        self.check(self._L.dvs_matrix_build(self._h, C.c_void_p(dev_ptr), 1,
                                            _lib.ptr(offsets, C.c_uint64), offsets.size - 1, k,
                                            num_states, C.byref(h)))
        return CountMatrix(self, h, k, num_states)

    # ---- packed sequences (3 bits per base in HBM: csrc/pack.hip) ------------------
    def pack_device(self, dev_ptr: int, nbases: int) -> "Packed":
        """four-state sequences resident in HBM one byte per base (16-byte aligned) -> the packed form; the
        kernel is not waited for: keep the byte buffer until the next ctx.sync()"""
        h = C.c_void_p()
        self.check(self._L.dvs_pack_sequences(self._h, C.c_void_p(dev_ptr), 1, int(nbases), C.byref(h)))
        return Packed(self, h)

    def pack_host(self, data: np.ndarray) -> "Packed":
        """host sequences (concatenated uint8) -> the packed form in HBM; 3/8 of the bytes cross PCIe"""
        data = np.ascontiguousarray(data, dtype=np.uint8).reshape(-1)
        h = C.c_void_p()
        self.check(self._L.dvs_pack_sequences(self._h, C.c_void_p(data.ctypes.data) if data.size else None, 0,
                                              data.size, C.byref(h)))
        return Packed(self, h)

    def build_matrix_packed(self, packed: "Packed", offsets: np.ndarray, k: int) -> "CountMatrix":
        """k-mer count matrix of a packed batch (the histogram kernel reads the packed words as they are)"""
        h = C.c_void_p()
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        self.check(self._L.dvs_matrix_build_packed(self._h, packed._h, _lib.ptr(offsets, C.c_uint64), offsets.size - 1,
                                                   k, C.byref(h)))
        m = CountMatrix(self, h, k, 4)
        m._source = packed  # (the build is not waited for: the planes must outlive its kernels)
        return m

    # ---- ingest ---------------------------------------------------------------
    def encode_fasta(self, raw, join_records: bool = False, moltype: str = "dna",
                     dev_ptr: int | None = None, nbytes: int | None = None) -> "SeqBatch":
        """FASTA file bytes -> index-coded sequences in HBM (csrc/ingest.hip), replacing the host
        parse + str2arr of diverse_seq/io.py:75-104 / util.py:32-45.  `raw`: bytes / uint8 array of
        the file (or None with dev_ptr + nbytes for bytes already on the device, in which case
        record names are not extracted).  join_records: one sequence per file, records joined by
        a gap symbol (io.py:100); else one sequence per record."""
        lut = np.zeros(256, dtype=np.uint8)
        self._L.dvs_default_alphabet_lut(int(moltype.lower() == "rna"), _lib.ptr(lut, C.c_uint8))
        h = C.c_void_p()
        if dev_ptr is not None:
            self.check(self._L.dvs_seqbatch_from_fasta(self._h, C.c_void_p(dev_ptr), 1, int(nbytes),
                                                       _lib.ptr(lut, C.c_uint8), int(join_records), C.byref(h)))
            host = None
        else:
            host = np.frombuffer(raw, dtype=np.uint8) if not isinstance(raw, np.ndarray) else np.ascontiguousarray(raw, dtype=np.uint8)
            self.check(self._L.dvs_seqbatch_from_fasta(self._h, C.c_void_p(host.ctypes.data) if host.size else None, 0,
                                                       host.size, _lib.ptr(lut, C.c_uint8), int(join_records),
                                                       C.byref(h)))
        return SeqBatch(self, h, host)

    def encode_genbank(self, raw: bytes, join_records: bool = False, moltype: str = "dna") -> "SeqBatch":
        """GenBank flat file -> index-coded sequences in HBM.  The host only re-frames the file
        (`genbank_to_fasta`: byte searches and one `translate` per record, no per-base Python);
        coding, line joining and the per-record offsets are the device ingest's, as for FASTA."""
        return self.encode_fasta(genbank_to_fasta(bytes(raw)), join_records=join_records, moltype=moltype)

    def matrix_from_freqs(self, freqs: np.ndarray) -> "CountMatrix":
        f = np.ascontiguousarray(freqs, dtype=np.float64)
        if f.ndim != 2:
            raise ValueError("freqs must be 2-D")
        h = C.c_void_p()
        self.check(self._L.dvs_matrix_from_freqs(self._h, _lib.ptr(f, C.c_double), f.shape[0],
                                                 f.shape[1], C.byref(h)))
        return CountMatrix(self, h, 0, 0)

    def matrix_from_device_freqs(self, dev_ptr: int, nrows: int, nbins: int,
                                 meta_ptr: int | None = None) -> "CountMatrix":
        """frequency rows already in HBM (f64 [nrows, nbins]); meta (f64 [nrows, 2]) marks
        padding rows with meta[r, 1] == 0"""
        h = C.c_void_p()
        self.check(self._L.dvs_matrix_from_device_freqs(self._h, C.c_void_p(dev_ptr),
                                                        C.c_void_p(meta_ptr) if meta_ptr else None,
                                                        nrows, nbins, C.byref(h)))
        return CountMatrix(self, h, 0, 0)

    def kmer_counts(self, seqs, k: int, num_states: int = 4):
        """-> (counts uint32 [n, ns^k], totals uint32 [n], entropy f64 [n])"""
        m = self.build_matrix(seqs, k, num_states)
        try:
            return m.counts(), m.totals(), m.entropy()
        finally:
            m.close()


class Packed:
    """four-state sequences in HBM at 3 bits per base (dvs_packed): code words (uint32 per 16 bases, the first
    base in the top bit pair) and mask words (uint16 per 16 bases, bit 15 - i = base i invalid)"""

    def __init__(self, ctx: Context, handle, owned: bool = True):
        self.ctx, self._h, self._owned = ctx, handle, owned
        nb, nw = C.c_uint64(), C.c_uint64()
        ctx._L.dvs_packed_info(handle, C.byref(nb), C.byref(nw))
        self.nbases, self.nwords = nb.value, nw.value

    @property
    def dev_codes(self) -> int:
        return int(self.ctx._L.dvs_packed_dev_codes(self._h) or 0)

    @property
    def dev_mask(self) -> int:
        return int(self.ctx._L.dvs_packed_dev_mask(self._h) or 0)

    def planes(self) -> tuple[np.ndarray, np.ndarray]:
        """(codes uint32 [nwords], mask uint16 [nwords]) copied to the host (tests)"""
        codes = np.zeros(max(1, self.nwords), dtype=np.uint32)
        mask = np.zeros(max(1, self.nwords), dtype=np.uint16)
        self.ctx.check(self.ctx._L.dvs_packed_get(self.ctx._h, self._h, _lib.ptr(codes, C.c_uint32),
                                                  _lib.ptr(mask, C.c_uint16)))
        return codes[: self.nwords], mask[: self.nwords]

    def close(self):
        if getattr(self, "_h", None) and self._owned:
            self.ctx._L.dvs_packed_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class SeqBatch:
    """index-coded sequences resident in HBM, as produced by Context.encode_fasta"""

    def __init__(self, ctx: Context, handle, raw_host):
        self.ctx, self._h = ctx, handle
        nseq, total, nrec = C.c_uint32(), C.c_uint64(), C.c_uint32()
        ctx._L.dvs_seqbatch_info(handle, C.byref(nseq), C.byref(total), C.byref(nrec))
        self.nseq, self.total, self.nrecords = nseq.value, total.value, nrec.value
        self.offsets = np.zeros(self.nseq + 1, dtype=np.uint64)
        ctx._L.dvs_seqbatch_offsets(handle, _lib.ptr(self.offsets, C.c_uint64))
        self.header_positions = np.zeros(self.nrecords, dtype=np.uint64)
        if self.nrecords:
            ctx._L.dvs_seqbatch_header_positions(handle, _lib.ptr(self.header_positions, C.c_uint64))
        self._raw_host = raw_host
        self._labels = None

    @property
    def labels(self):
        """record labels: the header line without '>' (host slices of the file on first use; no
        pass over the bases); None when the file was handed over as a device pointer"""
        raw_host = self._raw_host
        if self._labels is None and raw_host is not None:
            self.labels_ = []
            for p in self.header_positions:  # a window per header: the file itself is never copied
                p, width = int(p), 256
                while True:
                    line = raw_host[p + 1: p + 1 + width]
                    nl = np.flatnonzero(line == 10)
                    if nl.size or p + 1 + width >= raw_host.size:
                        break
                    width *= 16
                end = int(nl[0]) if nl.size else line.size
                self.labels_.append(line[:end].tobytes().decode("utf8", "replace").strip())
            self._labels = self.labels_
        return self._labels

    @property
    def dev_ptr(self) -> int:
        return int(self.ctx._L.dvs_seqbatch_dev_codes(self._h) or 0)

    def codes(self) -> np.ndarray:
        """the encoded symbols, copied to the host (tests, store writers)"""
        out = np.zeros(self.total, dtype=np.uint8)
        self.ctx.check(self.ctx._L.dvs_seqbatch_get_codes(self.ctx._h, self._h, _lib.ptr(out, C.c_uint8)))
        return out

    def sequences(self) -> list:
        c = self.codes()
        return [c[int(a): int(b)] for a, b in zip(self.offsets[:-1], self.offsets[1:])]

    def pack(self) -> "SeqBatch":
        """re-state the batch's bases at 3 bits each and release the byte form (four-state use only: every
        symbol >= 4 becomes "invalid"); build_matrix / sketches then read the packed words"""
        self.ctx.check(self.ctx._L.dvs_seqbatch_pack(self.ctx._h, self._h))
        return self

    @property
    def packed(self) -> "Packed | None":
        h = self.ctx._L.dvs_seqbatch_packed(self._h)
        if not h:
            return None
        view = Packed(self.ctx, C.c_void_p(h), owned=False)
        view._batch = self  # (a view of this batch's planes: the batch lives as long as the view)
        return view

    def build_matrix(self, k: int, num_states: int = 4) -> "CountMatrix":
        """k-mer count matrix straight from the encoded bases in HBM (no host round trip)"""
        h = C.c_void_p()
        self.ctx.check(self.ctx._L.dvs_matrix_build_from_seqbatch(self.ctx._h, self._h, k, num_states, C.byref(h)))
        m = CountMatrix(self.ctx, h, k, num_states)
        m._source = self  # (the build is not waited for: the batch must outlive its kernels)
        return m

    def close(self):
        if getattr(self, "_h", None):
            self.ctx._L.dvs_seqbatch_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class CountMatrix:
    """N x num_states^k matrix resident in HBM (dvs_matrix)"""

    def __init__(self, ctx: Context, handle, k: int, num_states: int):
        self.ctx, self._h, self.k, self.num_states = ctx, handle, k, num_states
        self._source = None  # the Packed / SeqBatch the matrix was built from, kept alive beside it
        L = ctx._L
        self.nrows = int(L.dvs_matrix_nrows(handle))
        self.nbins = int(L.dvs_matrix_nbins(handle))

    def close(self):
        if getattr(self, "_h", None):
            self.ctx._L.dvs_matrix_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def counts(self, row0: int = 0, nrows: int | None = None) -> np.ndarray:
        nrows = self.nrows - row0 if nrows is None else nrows
        out = np.zeros((nrows, self.nbins), dtype=np.uint32)
        self.ctx.check(self.ctx._L.dvs_matrix_get_counts(self.ctx._h, self._h, row0, nrows,
                                                         _lib.ptr(out, C.c_uint32)))
        return out

    def totals(self) -> np.ndarray:
        out = np.zeros(self.nrows, dtype=np.uint32)
        self.ctx.check(self.ctx._L.dvs_matrix_get_totals(self.ctx._h, self._h,
                                                         _lib.ptr(out, C.c_uint32)))
        return out

    def entropy(self) -> np.ndarray:
        out = np.zeros(self.nrows, dtype=np.float64)
        self.ctx.check(self.ctx._L.dvs_matrix_get_entropy(self.ctx._h, self._h,
                                                          _lib.ptr(out, C.c_double)))
        return out

    def source_rows(self) -> np.ndarray:
        """(matrix_from_device_freqs with flags) the input row every matrix row was copied from"""
        out = np.zeros(self.nrows, dtype=np.uint32)
        self.ctx.check(self.ctx._L.dvs_matrix_get_source_rows(self.ctx._h, self._h, _lib.ptr(out, C.c_uint32)))
        return out

    def dev_counts(self) -> int:
        return int(self.ctx._L.dvs_matrix_dev_counts(self._h) or 0)

    @property
    def count_bytes(self) -> int:
        """width of a count on the device: 4, or 2 when every sequence was a single tile (0: frequency rows)"""
        return int(self.ctx._L.dvs_matrix_count_bytes(self._h))

    # ---- selection -----------------------------------------------------------
    def select(self, mode: int, n_seed: int, *, max_size: int = 0, stat: int = _lib.STAT_STDEV,
               order=None, labels=None, npos: int | None = None, window: int = 0,
               flags: int = 0) -> "Selection":
        order_a = None if order is None else np.ascontiguousarray(order, dtype=np.uint32)
        labels_a = None if labels is None else np.ascontiguousarray(labels, dtype=np.uint32)
        if npos is None:
            npos = order_a.size if order_a is not None else self.nrows
        p = _lib.SelectParams(mode, n_seed, max_size, stat, window, flags)
        h = C.c_void_p()
        self.ctx.check(self.ctx._L.dvs_select_run(self.ctx._h, self._h,
                                                  _lib.ptr(order_a, C.c_uint32),
                                                  _lib.ptr(labels_a, C.c_uint32), npos,
                                                  C.byref(p), C.byref(h)))
        return Selection(self, h)

    def nmost(self, n: int, **kw) -> "Selection":
        return self.select(_lib.MODE_NMOST, n, **kw)

    def max_divergent(self, min_size: int, max_size: int, stat: str = "stdev", **kw) -> "Selection":
        st = _lib.STAT_STDEV if stat == "stdev" else _lib.STAT_COV  # src/lib.rs:116-120
        return self.select(_lib.MODE_MAX, min_size, max_size=max_size, stat=st, **kw)

    def as_set(self, **kw) -> "Selection":
        return self.select(_lib.MODE_SET, 0, **kw)


@dataclass
class Members:
    positions: np.ndarray
    labels: np.ndarray
    delta_jsd: np.ndarray
    entropy: np.ndarray
    kfreqs: np.ndarray | None


class Selection:
    """a SummedRecords set living on the device (dvs_select)"""

    def __init__(self, matrix: CountMatrix, handle):
        self.matrix, self.ctx, self._h = matrix, matrix.ctx, handle
        self._gids = None
        self._lazy_gids = None

    @property
    def global_ids(self):
        """(merged selections, parallel.merge_*) global stream position of every gathered row"""
        if self._gids is None and self._lazy_gids is not None:
            self._gids = self._lazy_gids.get()
        return self._gids

    @global_ids.setter
    def global_ids(self, value):
        self._gids = value

    def close(self):
        if getattr(self, "_h", None):
            self.ctx._L.dvs_select_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def summary(self) -> _lib.SelectSummary:
        s = _lib.SelectSummary()
        self.ctx.check(self.ctx._L.dvs_select_get_summary(self.ctx._h, self._h, C.byref(s)))
        return s

    def members(self, with_freqs: bool = True) -> Members:
        n = self.summary().size
        pos = np.zeros(n, dtype=np.uint64)
        lab = np.zeros(n, dtype=np.uint32)
        dl = np.zeros(n, dtype=np.float64)
        en = np.zeros(n, dtype=np.float64)
        fr = np.zeros((n, self.matrix.nbins), dtype=np.float64) if with_freqs else None
        self.ctx.check(self.ctx._L.dvs_select_get_members(
            self.ctx._h, self._h, _lib.ptr(pos, C.c_uint64), _lib.ptr(lab, C.c_uint32),
            _lib.ptr(dl, C.c_double), _lib.ptr(en, C.c_double), _lib.ptr(fr, C.c_double)))
        return Members(pos, lab, dl, en, fr)

    def gather_members(self, rows_ptr: int, meta_ptr: int, cap_rows: int):
        """members in set order into device buffers rows[cap, nbins], meta[cap, 2] (position, valid);
        enqueued on the context's stream"""
        self.ctx.check(self.ctx._L.dvs_select_gather_members(self.ctx._h, self._h, C.c_void_p(rows_ptr),
                                                             C.c_void_p(meta_ptr), cap_rows))

    def bench_scan(self, repeats: int = 5) -> tuple[float, int]:
        """(ms per launch, rows) of one scan launch over the whole stream, no events"""
        ms, rows = C.c_double(), C.c_uint64()
        self.ctx.check(self.ctx._L.dvs_select_bench_scan(self.ctx._h, self._h, repeats, C.byref(ms),
                                                         C.byref(rows)))
        return ms.value, rows.value

    def delta_jsd(self, queries: CountMatrix, qlabels=None) -> np.ndarray:
        out = np.zeros(queries.nrows, dtype=np.float64)
        ql = None if qlabels is None else np.ascontiguousarray(qlabels, dtype=np.uint32)
        self.ctx.check(self.ctx._L.dvs_select_delta_jsd(self.ctx._h, self._h, queries._h,
                                                        _lib.ptr(ql, C.c_uint32),
                                                        _lib.ptr(out, C.c_double)))
        return out


_default_ctx: Context | None = None


def default_context() -> Context:
    """process-wide context on the current device (created on first use)"""
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context()
    return _default_ctx
