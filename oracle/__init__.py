"""ctypes/numpy face of the CPU ORACLE (oracle/dvs_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, ``__graft_entry__.smoke()`` and
the ``cpu_baseline`` leg of ``bench.py``.  Nothing under ``diverseseq_amd/``
may import this package (tests/test_boundary.py enforces it).

Each helper mirrors one reference entry point; the arithmetic is in the C file,
which cites the reference file:line per function.
"""

from __future__ import annotations

import ctypes as C
import pathlib
import subprocess

import numpy as np

_HERE = pathlib.Path(__file__).resolve().parent
_SO = _HERE / "libdvs_oracle.so"

OK, ERR_PANIC, ERR_NOKMERS, ERR_ALLOC = 0, 1, 2, 3


class OraclePanic(ValueError):
    """the reference would panic here (surfaced to python as ValueError, lib.rs:36-57)"""


def build(force: bool = False) -> pathlib.Path:
    src = _HERE / "dvs_oracle.c"
    if force or not _SO.exists() or _SO.stat().st_mtime < src.stat().st_mtime:
        subprocess.run(["make", "-C", str(_HERE), "-B", "libdvs_oracle.so"], check=True,
                       capture_output=True)
    return _SO


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(str(_SO))
        u8p, u32p, u64p, f64p = (C.POINTER(C.c_uint8), C.POINTER(C.c_uint32),
                                 C.POINTER(C.c_uint64), C.POINTER(C.c_double))
        vp, sz = C.c_void_p, C.c_size_t
        L.orc_last_error.restype = C.c_char_p
        L.orc_kmer_to_index.restype = C.c_uint64
        L.orc_kmer_to_index.argtypes = [u8p, C.c_uint, C.c_uint, C.c_uint64]
        L.orc_count_kmers.argtypes = [u8p, sz, C.c_uint, C.c_uint, u64p]
        L.orc_entropy.argtypes = [f64p, sz, f64p]
        L.orc_to_kfreqs.argtypes = [u8p, sz, C.c_uint, C.c_uint, u64p, f64p, sz, f64p]
        L.orc_set_new.argtypes = [f64p, f64p, u32p, sz, sz, C.POINTER(vp)]
        L.orc_set_free.argtypes = [vp]
        L.orc_set_delta_jsd.argtypes = [vp, f64p, C.c_double, C.c_uint32, f64p]
        L.orc_set_increases_jsd_pub.argtypes = [vp, f64p, C.c_double, C.c_uint32,
                                                C.POINTER(C.c_int)]
        L.orc_set_push_pub.argtypes = [vp, f64p, C.c_double, C.c_uint32]
        L.orc_set_replace_lowest_pub.argtypes = [vp, f64p, C.c_double, C.c_uint32]
        for name in ("orc_set_size", "orc_set_nbins"):
            getattr(L, name).restype = sz
            getattr(L, name).argtypes = [vp]
        for name in ("orc_set_total_jsd", "orc_set_summed_entropies",
                     "orc_set_mean_delta_jsd", "orc_set_std_delta_jsd",
                     "orc_set_cov_delta_jsd"):
            getattr(L, name).restype = C.c_double
            getattr(L, name).argtypes = [vp]
        L.orc_set_lowest_index.restype = C.c_uint32
        L.orc_set_lowest_index.argtypes = [vp]
        L.orc_set_summed_kfreqs.argtypes = [vp, f64p]
        L.orc_set_members.argtypes = [vp, u32p, f64p, f64p, f64p]
        L.orc_nmost.argtypes = [u8p, u64p, u32p, sz, sz, C.c_uint, C.c_uint,
                                C.POINTER(vp), u64p]
        L.orc_max.argtypes = [u8p, u64p, u32p, sz, sz, sz, C.c_int, C.c_uint, C.c_uint,
                              C.POINTER(vp)]
        L.orc_final_nmost.argtypes = [f64p, u32p, sz, sz, sz, C.POINTER(vp)]
        L.orc_nmost_chunks_mt.argtypes = [u8p, u64p, u64p, sz, sz, C.c_uint, C.c_uint, f64p, u32p, u64p]
        L.orc_final_max.argtypes = [f64p, u32p, sz, sz, sz, sz, C.c_int, C.POINTER(vp)]
        L.orc_make_summed_records.argtypes = [u8p, u64p, u32p, sz, C.c_uint, C.c_uint,
                                              C.POINTER(vp)]
        L.orc_reverse_complement.argtypes = [u8p, sz, u8p]
        L.orc_murmurhash3_32.restype = C.c_uint32
        L.orc_murmurhash3_32.argtypes = [u8p, sz, C.c_uint32]
        L.orc_hash_kmer.restype = C.c_uint32
        L.orc_hash_kmer.argtypes = [u8p, sz, C.c_int]
        L.orc_kmer_hashes.restype = sz
        L.orc_kmer_hashes.argtypes = [u8p, sz, sz, C.c_uint, C.c_int, u32p]
        L.orc_mash_sketch.restype = sz
        L.orc_mash_sketch.argtypes = [u8p, sz, sz, sz, C.c_uint, C.c_int, u32p]
        L.orc_mash_distance.restype = C.c_double
        L.orc_mash_distance.argtypes = [u32p, sz, u32p, sz, C.c_uint, sz]
        L.orc_euclidean_distance.restype = C.c_double
        L.orc_euclidean_distance.argtypes = [f64p, f64p, sz]
        _lib = L
    return _lib


def _p(a: np.ndarray, ctype):
    return a.ctypes.data_as(C.POINTER(ctype))


def _u8(seq) -> np.ndarray:
    if isinstance(seq, (bytes, bytearray, memoryview)):
        return np.frombuffer(bytes(seq), dtype=np.uint8)
    return np.ascontiguousarray(seq, dtype=np.uint8)


def _check(rc: int):
    if rc == OK:
        return
    msg = lib().orc_last_error().decode()
    if rc in (ERR_PANIC, ERR_NOKMERS):
        raise OraclePanic(msg)
    raise MemoryError(msg)


def concat(seqs) -> tuple[np.ndarray, np.ndarray]:
    """list of uint8 sequences -> (concatenated bytes, uint64 offsets[n+1])"""
    arrs = [_u8(s) for s in seqs]
    offsets = np.zeros(len(arrs) + 1, dtype=np.uint64)
    if arrs:
        offsets[1:] = np.cumsum([a.size for a in arrs], dtype=np.uint64)
    data = np.concatenate(arrs) if arrs else np.zeros(0, dtype=np.uint8)
    if data.size == 0:
        data = np.zeros(1, dtype=np.uint8)  # keep a valid pointer
    return np.ascontiguousarray(data), offsets


# ------------------------------------------------------------------ counting
def kmer_to_index(kmer, num_states: int, max_index: int) -> int:
    a = _u8(kmer)
    return int(lib().orc_kmer_to_index(_p(a, C.c_uint8), a.size, num_states, max_index))


def count_kmers(seq, num_states: int, k: int) -> np.ndarray:
    """src/record.rs:124-131 to_kcounts"""
    a = _u8(seq)
    if k == 0:
        raise OraclePanic("k cannot be 0")
    out = np.zeros(num_states**k, dtype=np.uint64)
    buf = a if a.size else np.zeros(1, dtype=np.uint8)
    _check(lib().orc_count_kmers(_p(buf, C.c_uint8), a.size, num_states, k, _p(out, C.c_uint64)))
    return out


def entropy(kfreqs) -> float:
    f = np.ascontiguousarray(kfreqs, dtype=np.float64)
    out = C.c_double()
    _check(lib().orc_entropy(_p(f, C.c_double) if f.size else None, f.size, C.byref(out)))
    return out.value


def to_kfreqs(seq, num_states: int, k: int) -> tuple[np.ndarray, float]:
    """src/record.rs:133-141 to_kmerseq -> (kfreqs, entropy)"""
    a = _u8(seq)
    if k == 0:
        raise OraclePanic("k cannot be 0")
    B = num_states**k
    scratch = np.zeros(B, dtype=np.uint64)
    f = np.zeros(B, dtype=np.float64)
    h = C.c_double()
    buf = a if a.size else np.zeros(1, dtype=np.uint8)
    _check(lib().orc_to_kfreqs(_p(buf, C.c_uint8), a.size, num_states, k,
                               _p(scratch, C.c_uint64), _p(f, C.c_double), B, C.byref(h)))
    return f, h.value


# --------------------------------------------------------------- set algebra
class SummedRecords:
    """src/records.rs:10-216 SummedRecords, identity by integer label"""

    def __init__(self, handle):
        self._h = C.c_void_p(handle)

    @classmethod
    def new(cls, freqs, entropies=None, labels=None) -> "SummedRecords":
        f = np.ascontiguousarray(freqs, dtype=np.float64)
        m, B = f.shape if f.ndim == 2 else (0, 0)
        if entropies is None:
            entropies = [entropy(row) for row in f]
        h = np.ascontiguousarray(entropies, dtype=np.float64)
        lab = np.arange(m, dtype=np.uint32) if labels is None else np.ascontiguousarray(
            labels, dtype=np.uint32)
        out = C.c_void_p()
        _check(lib().orc_set_new(_p(f, C.c_double) if m else None,
                                 _p(h, C.c_double) if m else None,
                                 _p(lab, C.c_uint32) if m else None, m, B, C.byref(out)))
        return cls(out.value)

    @classmethod
    def from_seqs(cls, seqs, k: int, num_states: int = 4, labels=None) -> "SummedRecords":
        """src/records.rs:509-524 make_summed_records"""
        data, offs = concat(seqs)
        lab = None if labels is None else np.ascontiguousarray(labels, dtype=np.uint32)
        out = C.c_void_p()
        _check(lib().orc_make_summed_records(_p(data, C.c_uint8), _p(offs, C.c_uint64),
                                             _p(lab, C.c_uint32) if lab is not None else None,
                                             len(seqs), k, num_states, C.byref(out)))
        return cls(out.value)

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.orc_set_free(self._h)
            self._h = None

    size = property(lambda s: int(lib().orc_set_size(s._h)))
    nbins = property(lambda s: int(lib().orc_set_nbins(s._h)))
    total_jsd = property(lambda s: lib().orc_set_total_jsd(s._h))
    summed_entropies = property(lambda s: lib().orc_set_summed_entropies(s._h))
    lowest_index = property(lambda s: int(lib().orc_set_lowest_index(s._h)))
    mean_delta_jsd = property(lambda s: lib().orc_set_mean_delta_jsd(s._h))
    std_delta_jsd = property(lambda s: lib().orc_set_std_delta_jsd(s._h))
    cov_delta_jsd = property(lambda s: lib().orc_set_cov_delta_jsd(s._h))

    @property
    def mean_jsd(self):
        return self.total_jsd / self.size

    @property
    def summed_kfreqs(self) -> np.ndarray:
        out = np.zeros(self.nbins, dtype=np.float64)
        lib().orc_set_summed_kfreqs(self._h, _p(out, C.c_double))
        return out

    def members(self, with_freqs: bool = False):
        n, B = self.size, self.nbins
        labels = np.zeros(n, dtype=np.uint32)
        deltas = np.zeros(n, dtype=np.float64)
        ents = np.zeros(n, dtype=np.float64)
        freqs = np.zeros((n, B), dtype=np.float64) if with_freqs else None
        lib().orc_set_members(self._h, _p(labels, C.c_uint32), _p(deltas, C.c_double),
                              _p(ents, C.c_double),
                              _p(freqs, C.c_double) if with_freqs else None)
        return labels, deltas, ents, freqs

    def delta_jsd(self, kfreqs, h: float | None = None, label: int = 0xFFFFFFFF) -> float:
        f = np.ascontiguousarray(kfreqs, dtype=np.float64)
        h = entropy(f) if h is None else h
        out = C.c_double()
        _check(lib().orc_set_delta_jsd(self._h, _p(f, C.c_double), h, label, C.byref(out)))
        return out.value

    def increases_jsd(self, kfreqs, h: float | None = None, label: int = 0xFFFFFFFF) -> bool:
        f = np.ascontiguousarray(kfreqs, dtype=np.float64)
        h = entropy(f) if h is None else h
        out = C.c_int()
        _check(lib().orc_set_increases_jsd_pub(self._h, _p(f, C.c_double), h, label,
                                               C.byref(out)))
        return bool(out.value)

    def push(self, kfreqs, h: float | None = None, label: int = 0xFFFFFFFF):
        f = np.ascontiguousarray(kfreqs, dtype=np.float64)
        h = entropy(f) if h is None else h
        _check(lib().orc_set_push_pub(self._h, _p(f, C.c_double), h, label))

    def replace_lowest(self, kfreqs, h: float | None = None, label: int = 0xFFFFFFFF):
        f = np.ascontiguousarray(kfreqs, dtype=np.float64)
        h = entropy(f) if h is None else h
        _check(lib().orc_set_replace_lowest_pub(self._h, _p(f, C.c_double), h, label))

    def result(self, with_freqs: bool = True) -> dict:
        """src/records.rs:205-216 get_result"""
        labels, deltas, ents, freqs = self.members(with_freqs)
        return {
            "labels": labels, "delta_jsd": deltas, "entropies": ents, "kfreqs": freqs,
            "total_jsd": self.total_jsd, "mean_delta_jsd": self.mean_delta_jsd,
            "std_delta_jsd": self.std_delta_jsd, "cov_delta_jsd": self.cov_delta_jsd,
            "size": self.size,
        }


# ------------------------------------------------------------------ selectors
def _lab(labels):
    if labels is None:
        return None, None
    a = np.ascontiguousarray(labels, dtype=np.uint32)
    return a, _p(a, C.c_uint32)


def nmost_concat(data, offsets, n: int, k: int, num_states: int = 4, labels=None):
    """select_nmost_divergent over pre-concatenated sequences -> (SummedRecords, n_accepts)"""
    keep, lp = _lab(labels)
    out, acc = C.c_void_p(), C.c_uint64(0)
    _check(lib().orc_nmost(_p(data, C.c_uint8), _p(offsets, C.c_uint64), lp,
                           offsets.size - 1, n, k, num_states, C.byref(out), C.byref(acc)))
    return SummedRecords(out.value), acc.value


def nmost(seqs, n: int, k: int, num_states: int = 4, labels=None) -> SummedRecords:
    """src/records.rs:311-342 select_nmost_divergent over the sequences in list order"""
    data, offs = concat(seqs)
    return nmost_concat(data, offs, n, k, num_states, labels)[0]


def max_divergent_concat(data, offsets, min_size: int, max_size: int, k: int, num_states: int = 4,
                         stat: str = "stdev", labels=None) -> SummedRecords:
    """select_max_divergent over pre-concatenated sequences (genome-sized inputs: no second copy)"""
    keep, lp = _lab(labels)
    out = C.c_void_p()
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    _check(lib().orc_max(_p(data, C.c_uint8), _p(offsets, C.c_uint64), lp, offsets.size - 1,
                         min_size, max_size, int(stat == "stdev"), k, num_states,
                         C.byref(out)))
    return SummedRecords(out.value)


def max_divergent(seqs, min_size: int, max_size: int, k: int, num_states: int = 4,
                  stat: str = "stdev", labels=None) -> SummedRecords:
    """src/records.rs:390-454 select_max_divergent (stat != 'stdev' means cov, lib.rs:116-120)"""
    data, offs = concat(seqs)
    return max_divergent_concat(data, offs, min_size, max_size, k, num_states, stat, labels)


def nmost_chunks_threads(data, offsets, bounds, n: int, k: int, num_states: int = 4):
    """the reference's `-np len(bounds)` scheme (diverse_seq/records.py:225-245) with one THREAD per
    chunk: an independent select_nmost per contiguous chunk, then final_nmost over the winners in
    chunk order.  bounds: [(start, end)] per chunk.  Returns the merged SummedRecords."""
    nch = len(bounds)
    B = num_states ** k
    edges = np.ascontiguousarray([b[0] for b in bounds] + [bounds[-1][1]], dtype=np.uint64)
    rows = np.zeros((nch * n, B), dtype=np.float64)
    labels = np.zeros(nch * n, dtype=np.uint32)
    sizes = np.zeros(nch, dtype=np.uint64)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    _check(lib().orc_nmost_chunks_mt(_p(data, C.c_uint8), _p(offsets, C.c_uint64), _p(edges, C.c_uint64), nch,
                                     n, k, num_states, _p(rows, C.c_double), _p(labels, C.c_uint32),
                                     _p(sizes, C.c_uint64)))
    keep = np.concatenate([np.arange(c * n, c * n + int(sizes[c])) for c in range(nch)])
    return final_nmost(rows[keep], n, labels=labels[keep])


def final_nmost(freq_rows, n: int, labels=None) -> SummedRecords:
    """src/records.rs:363-382 select_nmost_divergent_final over concatenated member rows"""
    f = np.ascontiguousarray(freq_rows, dtype=np.float64)
    keep, lp = _lab(labels)
    out = C.c_void_p()
    _check(lib().orc_final_nmost(_p(f, C.c_double), lp, f.shape[0], f.shape[1], n,
                                 C.byref(out)))
    return SummedRecords(out.value)


def final_max(freq_rows, min_size: int, max_size: int, stat: str = "stdev",
              labels=None) -> SummedRecords:
    """src/records.rs:456-507 select_max_divergent_final"""
    f = np.ascontiguousarray(freq_rows, dtype=np.float64)
    keep, lp = _lab(labels)
    out = C.c_void_p()
    _check(lib().orc_final_max(_p(f, C.c_double), lp, f.shape[0], f.shape[1], min_size,
                               max_size, int(stat == "stdev"), C.byref(out)))
    return SummedRecords(out.value)


# ----------------------------------------------------------------------- mash
def reverse_complement(kmer) -> np.ndarray:
    a = _u8(kmer)
    out = np.zeros_like(a)
    lib().orc_reverse_complement(_p(a, C.c_uint8), a.size, _p(out, C.c_uint8))
    return out


def murmurhash3_32(data, seed: int = 0) -> int:
    a = _u8(data)
    buf = a if a.size else np.zeros(1, dtype=np.uint8)
    return int(lib().orc_murmurhash3_32(_p(buf, C.c_uint8), a.size, seed))


def hash_kmer(kmer, canonical: bool = False) -> int:
    a = _u8(kmer)
    return int(lib().orc_hash_kmer(_p(a, C.c_uint8), a.size, int(canonical)))


def kmer_hashes(seq, k: int, num_states: int = 4, canonical: bool = False) -> np.ndarray:
    a = _u8(seq)
    if a.size < k:
        return np.zeros(0, dtype=np.uint32)
    out = np.zeros(a.size - k + 1, dtype=np.uint32)
    n = lib().orc_kmer_hashes(_p(a, C.c_uint8), a.size, k, num_states, int(canonical),
                              _p(out, C.c_uint32))
    return out[:n].copy()


def mash_sketch(seq, k: int, sketch_size: int, num_states: int = 4,
                canonical: bool = False) -> np.ndarray:
    """src/distance.rs:136-182 mash_sketch"""
    a = _u8(seq)
    out = np.zeros(max(sketch_size, 1), dtype=np.uint32)
    buf = a if a.size else np.zeros(1, dtype=np.uint8)
    n = lib().orc_mash_sketch(_p(buf, C.c_uint8), a.size, k, sketch_size, num_states,
                              int(canonical), _p(out, C.c_uint32))
    return out[:n].copy()


def mash_distance(left, right, k: int, sketch_size: int) -> float:
    """diverse_seq/distance.py:230-291 (NaN where python raises ZeroDivisionError)"""
    l = np.ascontiguousarray(left, dtype=np.uint32)
    r = np.ascontiguousarray(right, dtype=np.uint32)
    lb = l if l.size else np.zeros(1, dtype=np.uint32)
    rb = r if r.size else np.zeros(1, dtype=np.uint32)
    return lib().orc_mash_distance(_p(lb, C.c_uint32), l.size, _p(rb, C.c_uint32), r.size,
                                   k, sketch_size)


def mash_distances(sketches, k: int, sketch_size: int) -> np.ndarray:
    """diverse_seq/distance.py:165-173 N x N fill (lower triangle mirrored)"""
    n = len(sketches)
    d = np.zeros((n, n), dtype=np.float64)
    for i in range(1, n):
        for j in range(i):
            d[i, j] = d[j, i] = mash_distance(sketches[i], sketches[j], k, sketch_size)
    return d


def euclidean_distance(a, b) -> float:
    x = np.ascontiguousarray(a, dtype=np.float64)
    y = np.ascontiguousarray(b, dtype=np.float64)
    return lib().orc_euclidean_distance(_p(x, C.c_double), _p(y, C.c_double), x.size)


# ----------------------------------------------------------------------- ingest
DNA_ORDER = "TCAG-NRYWSKMBDHV?"  # cogent3 get_moltype("dna").most_degen_alphabet()


def str2arr(text: str, moltype: str = "dna") -> np.ndarray:
    """diverse_seq/util.py:32-45 str2arr: alphabet.to_indices over the most degenerate alphabet
    (canonical states first: T0 C1 A2 G3, then '-', the ambiguity codes, '?'); lower case folded,
    any other character 255.  Pinned by tests/test_util.py:9-16 of the reference (ACGTT ->
    the canonical indices, N > 3)."""
    order = DNA_ORDER.replace("T", "U") if moltype.lower() == "rna" else DNA_ORDER
    lut = {c: i for i, c in enumerate(order)}
    lut.update({c.lower(): i for c, i in list(lut.items()) if c.isalpha()})
    lut["U" if moltype.lower() != "rna" else "T"] = 0
    lut["u" if moltype.lower() != "rna" else "t"] = 0
    return np.array([lut.get(c, 255) for c in text], dtype=np.uint8)


def parse_fasta(raw: bytes):
    """records of a FASTA file as (label, sequence text): a line starting with '>' opens a record,
    every other line's characters minus white space belong to the open record (what cogent3's
    MinimalFastaParser yields for diverse_seq/io.py:95-96); text in front of the first header is
    dropped"""
    records, cur = [], None
    for line in raw.split(b"\n"):
        if line.startswith(b">"):
            cur = [line[1:].decode("utf8", "replace").strip(), []]
            records.append(cur)
        elif cur is not None:
            cur[1].append(bytes(ch for ch in line if ch not in b" \t\r\x00").decode("latin1"))
    return [(lab, "".join(parts)) for lab, parts in records]


def load_fasta(raw: bytes, join_records: bool = False, moltype: str = "dna"):
    """diverse_seq/io.py:92-104 dvs_load_seqs.main (join_records: the file's sequences joined by
    "-" into one) or one coded sequence per record; returns (labels, [uint8 arrays])"""
    recs = parse_fasta(raw)
    if join_records:
        return [lab for lab, _ in recs], ([str2arr("-".join(s for _, s in recs), moltype)] if recs else [])
    return [lab for lab, _ in recs], [str2arr(s, moltype) for _, s in recs]


def parse_genbank(raw: bytes):
    """records of a GenBank flat file as (locus name, sequence text): the name is the second
    token of the LOCUS line, the sequence is every letter of the lines between ORIGIN and // (the
    position numbers and blanks dropped) -- what a GenBank parser hands diverse_seq/io.py:94-96
    through cogent3's get_format_parser(path, "genbank"); a record without ORIGIN has no sequence"""
    records, name, parts, in_seq = [], None, [], False
    for line in raw.split(b"\n"):
        line = line.rstrip(b"\r")
        if line.startswith(b"LOCUS"):
            toks = line.split()
            name, parts, in_seq = (toks[1].decode("utf8", "replace") if len(toks) > 1 else ""), [], False
        elif line.startswith(b"ORIGIN"):
            in_seq = True
        elif line.startswith(b"//"):
            if name is not None:
                records.append((name, "".join(parts)))
            name, parts, in_seq = None, [], False
        elif in_seq:
            parts.append("".join(chr(ch) for ch in line if chr(ch).isalpha() or chr(ch) in "-?"))
    return records


def load_genbank(raw: bytes, join_records: bool = False, moltype: str = "dna"):
    """as load_fasta, for GenBank files (letters are upper-cased first: the alphabet is upper case)"""
    recs = [(lab, seq.upper()) for lab, seq in parse_genbank(raw)]
    if join_records:
        return [lab for lab, _ in recs], ([str2arr("-".join(s for _, s in recs), moltype)] if recs else [])
    return [lab for lab, _ in recs], [str2arr(s, moltype) for _, s in recs]
