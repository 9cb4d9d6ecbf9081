"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports
every symbol include/dvs_hip.h declares, the product never touches oracle/, and
without a GPU the product fails loudly instead of falling back."""
import ast
import ctypes
import pathlib
import pickle
import re

import numpy as np
import pytest

from conftest import ROOT

from conftest import pack_reference
from diverseseq_amd import _dvs, _lib, engine


def _header_functions():
    text = (ROOT / "include" / "dvs_hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return set(re.findall(r"\b(dvs_[a-z_0-9]+)\s*\(", text))


def test_library_exports_every_declared_symbol():
    declared = _header_functions()
    assert declared, "no declarations parsed"
    assert declared == set(_lib.EXPORTS)
    lib = ctypes.CDLL(str(_lib.LIB_PATH))
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in dvs_hip.h but not exported"
    assert _lib.load().dvs_abi_version() == 3


def test_product_never_imports_oracle():
    for py in (ROOT / "diverseseq_amd").rglob("*.py"):
        tree = ast.parse(py.read_text())
        for node in ast.walk(tree):
            names = []
            if isinstance(node, ast.Import):
                names = [a.name for a in node.names]
            elif isinstance(node, ast.ImportFrom):
                names = [node.module or ""]
            assert not any(n.split(".")[0] == "oracle" for n in names), f"{py} imports oracle"
    for src in (ROOT / "diverseseq_amd" / "csrc").glob("*"):
        if src.is_file():
            assert "oracle/" not in src.read_text(errors="ignore").replace("oracle/_ref", "")


def _has_gpu():
    lib = _lib.load()
    h = ctypes.c_void_p()
    rc = lib.dvs_ctx_create(-1, None, ctypes.byref(h))
    if rc == 0:
        lib.dvs_ctx_destroy(h)
    return rc == 0


def test_no_cpu_fallback_without_gpu():
    if _has_gpu():
        pytest.skip("a GPU is visible")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        engine.Context()
    store = _dvs.make_zarr_store()
    for i, s in enumerate(([0, 1, 2, 3], [0, 0, 1, 1], [2, 2, 3, 3])):
        store.write(f"s{i}", bytes(s))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _dvs.nmost_divergent(store, n=2, k=1)


def test_store_semantics():
    """reference tests/test_zarr_store.py:48-61 and src/zarr_py.rs"""
    st = _dvs.make_zarr_store()
    dup = bytes([0, 1, 2, 3])
    for sid in ("s1", "s2", "s3", "s4"):
        st.write(sid, dup)
    assert len(st) == 4 and st.num_unique() == 1
    assert st.unique_seqids == ["s4"]  # the last id written wins
    assert st.get_seqids() == ["s1", "s2", "s3", "s4"]
    assert "s2" in st and "zz" not in st
    assert st.read("s1") == dup
    with pytest.raises(ValueError):
        st.write("e", b"")
    with pytest.raises(RuntimeError):
        st.read("missing")
    with pytest.raises(TypeError):
        pickle.dumps(st)
    lz = st.get_lazyseq("s1", 4)
    assert lz.seqid == "s1" and lz.num_states == 4 and lz.get_seq() == dup
    assert [l.seqid for l in st.get_lazyseqs(4)] == st.get_seqids()
    with pytest.raises(FileNotFoundError):  # on-disk stores: tests/test_dvseqsz_store.py
        _dvs.make_zarr_store("/tmp/no-such-store.dvseqsz", mode="r")


def test_in_memory_store_hands_over_one_stream():
    """the stream a selection uploads: a view of the store's arena for the store's own id order,
    gathered bytes for any other order; repeated ids share a label; an unknown id is the reference's
    panic (src/record.rs:206) as ValueError"""
    rng = np.random.default_rng(11)
    seqs = [rng.integers(0, 5, int(rng.integers(20, 60)), dtype=np.uint8).tobytes() for _ in range(200)]
    seqs[10] = seqs[3]
    st = _dvs.make_zarr_store()
    for i, s in enumerate(seqs):
        st.write(f"s{i}", s)
    st.write("s5", b"\x01")  # a known id is not rewritten (src/zarr_io.rs:217-219)
    assert st.read("s5") == seqs[5]
    all_ids = [f"s{i}" for i in range(200)]
    ids, data, offs, labels = _dvs._gather(st, all_ids)
    assert data.tobytes() == b"".join(seqs) and labels.tolist() == list(range(200))
    assert np.shares_memory(data, np.frombuffer(st._arena, dtype=np.uint8))
    del data
    perm = rng.permutation(200).tolist()
    ids, data, offs, labels = _dvs._gather(st, [f"s{i}" for i in perm])
    for j, i in enumerate(perm):
        assert data[int(offs[j]):int(offs[j + 1])].tobytes() == seqs[i]
    del data
    ids, data, offs, labels = _dvs._gather(st, None)  # unique content, the last id written named
    assert ids[3] == "s10" and "s3" not in ids and len(ids) == 199
    assert data.tobytes() == b"".join(s for i, s in enumerate(seqs) if i != 10)
    del data
    ids, data, offs, labels = _dvs._gather(st, ["s1", "s7", "s1"])
    assert labels.tolist() == [0, 1, 0] and data.tobytes() == seqs[1] + seqs[7] + seqs[1]
    del data
    with pytest.raises(ValueError, match="not in store"):
        _dvs._gather(st, ["s1", "nope"])
    st.write("late", b"\x00\x01")  # no view outstanding: the arena grows again
    assert st.read("late") == b"\x00\x01" and len(st) == 201


def test_result_pickles():
    """reference tests/test_records.py:88-97; src/records_py.rs:49-87"""
    r = _dvs.SummedRecordsResult()
    assert r.size == 0 and r.records == [] and r.record_names == []
    r.records = [("a", [0.5, 0.5], 0.1)]
    r.size, r.k, r.num_states, r.total_jsd = 1, 2, 4, 0.25
    r2 = pickle.loads(pickle.dumps(r))
    assert r2.record_names == ["a"] and r2.size == 1 and r2.k == 2 and r2.total_jsd == 0.25
    with pytest.raises(KeyError):
        _dvs.SummedRecordsResult().__setstate__({"total_jsd": 1.0})


def test_argument_errors_before_any_device_work():
    st = _dvs.make_zarr_store()
    for i, s in enumerate(([0, 1, 2, 3], [0, 0, 1, 1], [2, 2, 3, 3])):
        st.write(f"s{i}", bytes(s))
    # records.rs:323-325 / 404-410: checked before anything is computed
    with pytest.raises(ValueError, match="The number of sequences 3 is < n 30"):
        _dvs.nmost_divergent(st, n=30, k=1)
    with pytest.raises(ValueError, match="The number of sequences 3 is < n 30"):
        _dvs.max_divergent(st, min_size=30, max_size=2, k=1)
    with pytest.raises(ValueError, match="is < n 5"):
        _dvs.final_nmost([], n=5)


def test_concat():
    data, offs = engine.concat([b"\x00\x01", np.array([2, 3, 0], dtype=np.uint8), b""])
    assert data.tolist() == [0, 1, 2, 3, 0] and offs.tolist() == [0, 2, 5, 5]
    data, offs = engine.concat([])
    assert offs.tolist() == [0]


def test_host_packer_of_the_packed_upload():
    """csrc/pack_host.cpp (AVX-512, AVX2 or scalar, whichever the CPU runs -- every form it runs is checked): 2
    bits per base + 1 invalid bit per base in the kernels' own word layout (uint32 of codes per 16 bases with the
    FIRST base in the top bit pair, uint16 of mask with the first base in bit 15), positions behind the end of a
    ragged tail marked invalid -- against a numpy restatement, at every length around the 16-, 32- and 64-base
    groups"""
    lib = ctypes.CDLL(str(_lib.LIB_PATH))
    f = lib.dvs_pack_bases_level
    f.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    f.restype = None
    lib.dvs_pack_level.restype = ctypes.c_int
    top = lib.dvs_pack_level()
    assert 0 <= top <= 2
    rng = np.random.default_rng(3)
    for n in (0, 1, 15, 16, 17, 31, 32, 33, 47, 48, 49, 63, 64, 65, 127, 128, 129, 1000, 4099, 1 << 16):
        src = rng.integers(0, 4, size=n, dtype=np.uint8)
        if n:
            src[rng.integers(0, n, size=n // 10 + 1)] = rng.integers(4, 256, size=n // 10 + 1, dtype=np.uint8)
        nw = (n + 15) // 16
        ec, em = pack_reference(src)
        for level in range(top + 1):
            codes = np.full(nw + 4, 0xDEADBEEF, np.uint32)
            mask = np.full(nw + 4, 0xBEEF, np.uint16)
            f(src.ctypes.data, n, codes.ctypes.data, mask.ctypes.data, level)
            assert (codes[:nw] == ec).all() and (mask[:nw] == em).all(), (n, level)
            # nothing written behind the last word
            assert (codes[nw:] == 0xDEADBEEF).all() and (mask[nw:] == 0xBEEF).all(), (n, level)
