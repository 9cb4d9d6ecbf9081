"""The reference's app surface over the HIP path: `dvs_nmost`, `dvs_max`, `dvs_delta_jsd`
(diverse_seq/records.py:254-429) and `dvs_ctree` / `dvs_par_ctree` (diverse_seq/cluster.py:98-188,
399-495), the five names its pyproject registers under the `cogent3.app` entry-point group
(pyproject.toml:89-94; `pyproject.toml` here registers the same five).

Constructor arguments, defaults, seeding (`numpy.random.default_rng(seed).shuffle` of the unique
ids) and error messages are the reference's.  cogent3 is OPTIONAL: when it is importable the classes
are wrapped with its `define_app` and take / return its sequence collections; without it (this
image) they are plain callables over `{name: sequence}` mappings -- str (IUPAC letters), bytes or
uint8 arrays of alphabet indices -- and return the selected mapping, (name, delta) pair or Newick
string.  Everything numeric goes through `diverseseq_amd._dvs`, i.e. the C ABI.
"""

from __future__ import annotations

import numpy as np

from . import _dvs as dvs
from . import cluster as _cluster

try:  # pragma: no cover - cogent3 is not in this image
    from cogent3.app.composable import define_app as _define_app
    from cogent3 import get_moltype as _get_moltype
    HAVE_COGENT3 = True
except Exception:  # noqa: BLE001
    HAVE_COGENT3 = False

    def _define_app(*a, **kw):
        def wrap(cls):
            cls.__call__ = lambda self, *args, **kwargs: self.main(*args, **kwargs)
            return cls
        return wrap if not (len(a) == 1 and isinstance(a[0], type)) else wrap(a[0])

__all__ = ["dvs_nmost", "dvs_max", "dvs_delta_jsd", "dvs_ctree", "dvs_par_ctree"]

# len(get_moltype(m).alphabet) of the reference (records.py:299, 415-416)
_NUM_STATES = {"dna": 4, "rna": 4, "protein": 20, "text": 26, "bytes": 256}
_ALPHABET = {"dna": "TCAG", "rna": "UCAG", "protein": "ACDEFGHIKLMNPQRSTVWY"}


def _num_states(moltype: str) -> int:
    if HAVE_COGENT3:  # pragma: no cover
        return len(_get_moltype(moltype).alphabet)
    try:
        return _NUM_STATES[moltype.lower()]
    except KeyError:
        raise ValueError(f"unknown moltype {moltype!r}") from None


def _encode(seq, moltype: str) -> bytes:
    """sequence -> alphabet indices, one byte per symbol (diverse_seq/util.py:32-45 str2arr); gaps and
    ambiguity codes become indices >= num_states and invalidate the k-mers that contain them"""
    if isinstance(seq, (bytes, bytearray, memoryview)):
        return bytes(seq)
    if isinstance(seq, np.ndarray):
        return np.ascontiguousarray(seq, dtype=np.uint8).tobytes()
    text = str(seq).replace("-", "").replace("?", "")  # degap (records.py: seqs.degap())
    canon = _ALPHABET.get(moltype.lower())
    if canon is None:
        raise ValueError(f"cannot encode text for moltype {moltype!r} without cogent3")
    lut = np.full(256, len(canon), dtype=np.uint8)
    for i, ch in enumerate(canon):
        lut[ord(ch)] = lut[ord(ch.lower())] = i
    if moltype.lower() == "dna":
        lut[ord("U")] = lut[ord("u")] = 0
    return lut[np.frombuffer(text.encode("ascii", "replace"), dtype=np.uint8)].tobytes()


def _as_mapping(seqs, moltype: str):
    """(names, {name: index bytes}, taker) for a cogent3 collection or a plain mapping"""
    if HAVE_COGENT3 and hasattr(seqs, "take_seqs"):  # pragma: no cover
        degapped = seqs.degap()
        data = {s.name: np.array(s).tobytes() for s in degapped.seqs}
        return list(data), data, seqs.take_seqs
    data = {str(n): _encode(s, moltype) for n, s in dict(seqs).items()}

    def take(names):  # (the store's names are str(n): compare on those, whatever the mapping's keys are)
        sel = {str(n) for n in names}
        return {n: seqs[n] for n in seqs if str(n) in sel}

    return list(data), data, take


def _populate_inmem_zstore(data: dict):
    """diverse_seq/util.py:176-184"""
    zstore = dvs.make_zarr_store()
    for name, arr in data.items():
        zstore.write(name, arr)
    return zstore


@_define_app
class dvs_max:
    """select the maximally divergent seqs from a sequence collection (records.py:254-321)"""

    def __init__(self, min_size: int = 5, max_size: int = 30, stat: str = "stdev", moltype: str = "dna",
                 include: list[str] | str | None = None, k: int = 6, seed: int | None = None) -> None:
        self._k = k
        self._moltype = moltype
        self._num_states = _num_states(moltype)
        self._min_size = min_size
        self._max_size = max_size
        self._stat = stat
        self._rng = np.random.default_rng(seed)
        self._include = [include] if isinstance(include, str) else include

    def main(self, seqs):
        _, data, take = _as_mapping(seqs, self._moltype)
        zstore = _populate_inmem_zstore(data)
        seqids = list(zstore.unique_seqids)
        self._rng.shuffle(seqids)
        result = dvs.max_divergent(zstore, min_size=self._min_size, max_size=self._max_size, k=self._k,
                                   num_states=self._num_states, seqids=seqids, stat=self._stat)
        return take(set(result.record_names) | set(self._include or []))


@_define_app
class dvs_nmost:
    """select the n-most diverse seqs from a sequence collection (records.py:324-373)"""

    def __init__(self, n: int = 10, moltype: str = "dna", include: list[str] | str | None = None, k: int = 6,
                 seed: int | None = None) -> None:
        self._k = k
        self._n = n
        self._moltype = moltype
        self._rng = np.random.default_rng(seed)
        self._include = [include] if isinstance(include, str) else include

    def main(self, seqs):
        _, data, take = _as_mapping(seqs, self._moltype)
        zstore = _populate_inmem_zstore(data)
        seqids = list(zstore.unique_seqids)
        self._rng.shuffle(seqids)
        result = dvs.nmost_divergent(zstore, n=self._n, k=self._k, seqids=seqids)  # (num_states left at 4, :371)
        return take(set(result.record_names) | set(self._include or []))


@_define_app
class dvs_delta_jsd:
    """delta JSD of a sequence against a fixed reference set (records.py:376-429)"""

    def __init__(self, seqs, moltype: str = "dna", k: int = 6) -> None:
        _, data, _ = _as_mapping(seqs, moltype)
        zero_len = ", ".join(n for n, s in data.items() if len(s) == 0)
        if zero_len:
            raise ValueError(f"cannot compute delta_jsd with zero-length sequences: {zero_len}")
        self.moltype = moltype
        self._sr = dvs.get_delta_jsd_calculator(list(data.items()), k, _num_states(moltype))

    def main(self, seq):
        if HAVE_COGENT3 and hasattr(seq, "moltype"):  # pragma: no cover
            if seq.moltype.name != self.moltype:
                seq = seq.to_moltype(self.moltype)
            seq = seq.degap()
            name, arr = seq.name, np.array(seq).tobytes()
        else:
            name, raw = seq  # (name, sequence)
            arr = _encode(raw, self.moltype)
        if len(arr) == 0:
            return name, float("nan")
        return name, self._sr.delta_jsd(name, arr)


class _ClusterTreeBase:
    """argument checks of ClusterTreeBase.__init__ (cluster.py:36-95)"""

    def __init__(self, *, k: int = 12, sketch_size: int | None = 3_000, moltype: str = "dna",
                 distance_mode: str = "mash", mash_canonical_kmers: bool | None = None,
                 show_progress: bool = False) -> None:
        if mash_canonical_kmers is None:
            mash_canonical_kmers = False
        if distance_mode not in ("mash", "euclidean"):
            raise ValueError(f"Unexpected distance {distance_mode!r}.")
        if moltype not in ("dna", "rna") and mash_canonical_kmers:
            raise ValueError("Canonical kmers only supported for dna/rna sequences.")
        if distance_mode == "mash" and sketch_size is None:
            raise ValueError("Expected sketch size for mash distance measure.")
        if distance_mode != "mash":  # (the sketch size means nothing to the euclidean mode: cli.py:546-560)
            sketch_size = None
        self._moltype = moltype
        self._k = k
        self._num_states = _num_states(moltype)
        self._sketch_size = sketch_size
        self._distance_mode = distance_mode
        self._mash_canonical = mash_canonical_kmers
        self._progress = show_progress

    def main(self, seqs):
        names, data, _ = _as_mapping(seqs, self._moltype)
        arrays = {n: np.frombuffer(data[n], dtype=np.uint8) for n in names}
        newick = _cluster.ctree(arrays, k=self._k, sketch_size=self._sketch_size, distance_mode=self._distance_mode,
                                mash_canonical_kmers=self._mash_canonical, num_states=self._num_states)
        if HAVE_COGENT3:  # pragma: no cover
            from cogent3 import make_tree

            return make_tree(newick, underscore_unmunge=True)
        return newick


@_define_app
class dvs_ctree(_ClusterTreeBase):
    """Create a cluster tree from kmer distances (cluster.py:98-188)."""

    def __init__(self, *, k: int = 12, sketch_size: int | None = 3_000, moltype: str = "dna",
                 distance_mode: str = "mash", mash_canonical_kmers: bool | None = None,
                 show_progress: bool = False) -> None:
        super().__init__(k=k, sketch_size=sketch_size, moltype=moltype, distance_mode=distance_mode,
                         mash_canonical_kmers=mash_canonical_kmers, show_progress=show_progress)


@_define_app
class dvs_par_ctree(_ClusterTreeBase):
    """The same tree with the reference's worker-process knobs accepted (cluster.py:399-495).  The
    reference spreads sketches and strided distance rows over `max_workers` processes; here one GPU does
    both stages (several GPUs: diverseseq_amd.parallel.mash_distances_sharded), so `max_workers` and
    `parallel` only keep the signature."""

    def __init__(self, *, k: int = 12, sketch_size: int | None = 3_000, moltype: str = "dna",
                 distance_mode: str = "mash", mash_canonical_kmers: bool | None = None,
                 show_progress: bool = False, max_workers: int | None = None, parallel: bool = True) -> None:
        super().__init__(k=k, sketch_size=sketch_size, moltype=moltype, distance_mode=distance_mode,
                         mash_canonical_kmers=mash_canonical_kmers, show_progress=show_progress)
        self._max_workers = max_workers
        self._parallel = parallel
