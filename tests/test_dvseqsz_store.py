"""On-disk .dvseqsz stores (diverseseq_amd/zarr_store.py): the behaviours
the reference's tests/test_zarr_store.py asserts, and the on-disk layout as src/zarr_io.rs writes
it (names, JSON members, zstd frame, postcard bytes).  No GPU involved: the store is host I/O.

Parity note: the reference ships no store fixture and cannot run here, so the layout checks pin
this module to the reference's SOURCE, not to bytes the reference produced."""
import json
import pickle
import struct

import numpy as np
import pytest

from diverseseq_amd import _dvs as dvs
from diverseseq_amd import zarr_store


def test_invalid_path():
    with pytest.raises(FileNotFoundError):  # reference tests/test_zarr_store.py:8-10
        dvs.make_zarr_store("nonexistent_path.zarr", mode="r")
    with pytest.raises(FileNotFoundError):
        dvs.get_seqids_from_store("nonexistent_path.zarr")


@pytest.fixture(params=["disk", "memory"])
def zstore(request, tmp_path):
    return dvs.make_zarr_store(str(tmp_path / "s.dvseqsz"), mode="w") if request.param == "disk" \
        else dvs.make_zarr_store()


def test_add_seq_and_dedup(zstore):
    """reference tests/test_zarr_store.py:38-61"""
    assert not len(zstore)
    seq = np.array([0, 3, 2, 0], dtype=np.uint8)
    zstore.write("s3", seq.tobytes())
    assert "s3" in zstore and "zz" not in zstore
    assert zstore.read("s3") == seq.tobytes()
    zstore.write("s3", seq.tobytes())
    assert zstore.num_unique() == 1 and len(zstore) == 1
    zstore.write("s4", seq.tobytes())
    assert zstore.num_unique() == 1 and len(zstore) == 2
    assert zstore.unique_seqids == ["s4"]
    with pytest.raises(ValueError):
        zstore.write("empty", b"")
    with pytest.raises(RuntimeError):
        zstore.read("nope")


def test_metadata_roundtrip(zstore):
    """reference tests/test_zarr_store.py:64-79"""
    zstore.write("Human", bytes([2, 1, 3, 0, 5]), {"source": "brca1:Human"})
    zstore.write("plain", bytes([1, 1, 1]))
    assert zstore.read_metadata("Human")["source"] == "brca1:Human"
    assert zstore.read_metadata("plain") == {"source": "unknown"}  # zarr_py.rs:143-150


def test_persistence_pickle_and_reopen(tmp_path):
    """reference tests/test_zarr_store.py:82-110: a store pickles as its path; the id map survives"""
    path = str(tmp_path / "p.dvseqsz")
    st = dvs.make_zarr_store(path, mode="w")
    rng = np.random.default_rng(3)
    seqs = {f"s{i}": rng.integers(0, 6, size=int(rng.integers(1, 5000)), dtype=np.uint8).tobytes() for i in range(20)}
    seqs["zeros"] = bytes(1000)  # equal to the fill value: the chunk is not stored
    for k, v in seqs.items():
        st.write(k, v, {"source": k})
    clone = pickle.loads(pickle.dumps(st))
    assert clone.source == path and len(clone) == len(seqs)
    del st
    again = dvs.make_zarr_store(path, mode="r")
    assert again.get_seqids() == list(seqs) == dvs.get_seqids_from_store(path)
    for k, v in seqs.items():
        assert again.read(k) == v == clone.read(k)
        assert again.read_metadata(k) == {"source": k}
    with pytest.raises(TypeError):
        pickle.dumps(dvs.make_zarr_store())


def test_layout_on_disk_follows_zarr_io_rs(tmp_path):
    path = tmp_path / "layout.dvseqsz"
    st = dvs.make_zarr_store(str(path), mode="w")
    data = bytes([0, 1, 2, 3] * 600)
    st.write("seqA", data, {"source": "x"})
    st.write("seqB", data)  # same content: no second array
    st.close()
    hexd = zarr_store.xxh3_hex(data)
    assert len(hexd) == 16 and int(hexd, 16) >= 0
    assert json.loads((path / "seqdata" / "zarr.json").read_text())["node_type"] == "group"
    arrays = [p.name for p in (path / "seqdata").iterdir() if p.is_dir()]
    assert arrays == [hexd]
    meta = json.loads((path / "seqdata" / hexd / "zarr.json").read_text())
    assert meta["zarr_format"] == 3 and meta["node_type"] == "array"
    assert meta["shape"] == [2400] and meta["data_type"] == "uint8" and meta["fill_value"] == 0
    assert meta["chunk_grid"]["configuration"]["chunk_shape"] == [2400]  # one chunk (zarr_io.rs:241)
    assert meta["codecs"] == [{"name": "bytes"}, {"name": "zstd", "configuration": {"level": 3, "checksum": True}}]
    assert bytes(meta["attributes"]["metadata"]) == b"\x01\x06source\x01x"  # postcard {"source": "x"}
    chunk = (path / "seqdata" / hexd / "c" / "0").read_bytes()
    assert struct.unpack("<I", chunk[:4])[0] == 0xFD2FB528  # zstd frame magic
    assert chunk[4] & 0x04  # Content_Checksum_flag of the frame header descriptor
    assert zarr_store._Zstd.decompress(chunk, 2400) == data
    # side file: postcard of Vec<(String, [u8; 16])>
    side = (path / ".seqid_to_hash.bin").read_bytes()
    assert side == b"\x02" + b"\x04seqA" + hexd.encode() + b"\x04seqB" + hexd.encode()
    assert zarr_store.decode_side_file(side) == {"seqA": hexd, "seqB": hexd}


def test_postcard_varints():
    long_id = "x" * 300
    buf = zarr_store.encode_side_file({long_id: "0123456789abcdef"})
    assert buf[:3] == b"\x01\xac\x02"  # 1 entry; 300 = 0b10_0101100 -> ac 02
    assert zarr_store.decode_side_file(buf) == {long_id: "0123456789abcdef"}
    m = {"a": "b" * 200, "k": ""}
    assert zarr_store.decode_str_map(zarr_store.encode_str_map(m)) == m


def test_read_many_decodes_in_place(tmp_path):
    """what a selection over an on-disk store uploads: every array decoded straight into one buffer
    (thread pool, zstd in place) equals the per-id reads joined; a chunk equal to the fill value has
    no file (zarrs default) and a repeated id is decoded again"""
    rng = np.random.default_rng(3)
    st = dvs.make_zarr_store(str(tmp_path / "m.dvseqsz"), mode="w")
    seqs = {f"s{i}": rng.integers(0, 5, int(rng.integers(1, 4000)), dtype=np.uint8).tobytes() for i in range(60)}
    seqs["zeros"] = bytes(100)
    seqs["same"] = seqs["s7"]
    for sid, s in seqs.items():
        st.write(sid, s)
    ids = list(seqs) + ["s3", "zeros"]
    for workers in (1, 8):
        data, offs = st._disk.read_many(ids, workers=workers)
        assert offs.dtype == np.uint64 and offs[0] == 0 and int(offs[-1]) == data.size
        assert data.tobytes() == b"".join(seqs[sid] for sid in ids)
        assert [int(b - a) for a, b in zip(offs[:-1], offs[1:])] == [len(seqs[sid]) for sid in ids]
    data, offs = st._disk.read_many([])
    assert data.size == 0 and offs.tolist() == [0]
    with pytest.raises(KeyError):
        st._disk.read_many(["s1", "absent"])
    # the module-level gather takes this path for a store on disk
    got_ids, data, offs, labels = dvs._gather(st, ["s2", "same", "s2"])
    assert data.tobytes() == seqs["s2"] + seqs["same"] + seqs["s2"] and labels.tolist() == [0, 1, 0]
    with pytest.raises(ValueError, match="not in store"):
        dvs._gather(st, ["s2", "absent"])
    # a damaged chunk is an error, not silence
    hexd = st._disk.seqid_to_hash["s5"]
    chunk = tmp_path / "m.dvseqsz" / "seqdata" / hexd / "c" / "0"
    chunk.write_bytes(chunk.read_bytes()[:-6])
    with pytest.raises(RuntimeError):
        st._disk.read_many(["s5"])
