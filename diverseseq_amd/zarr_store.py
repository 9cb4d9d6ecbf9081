"""On-disk ``.dvseqsz`` stores: the format ``dvs prep`` writes.

Restates the storage layout of the reference's ``src/zarr_io.rs`` so that stores written by the
real ``dvs prep`` can be opened here and vice versa (host-side I/O, not part of the arithmetic
path; ``_dvs._gather`` reads sequences through it exactly as the reference's selectors read
through ``ZarrStore::read_uint8_array``, src/record.rs:205-209):

* a Zarr v3 hierarchy on the file system with one group ``seqdata`` (zarr_io.rs:82-95) and, per
  DISTINCT sequence, one 1-D uint8 array named by the 16-hex-digit xxh3-64 of its bytes
  (:222-224), a single chunk as long as the sequence (:239-247), codecs ``bytes`` then
  ``zstd`` level 3 with content checksum (:238), fill value 0; optional per-array attribute
  ``metadata`` = the postcard bytes of a ``{str: str}`` map as a JSON array of ints (:248-256);
* a side file ``.seqid_to_hash.bin`` = postcard of ``struct { seqid_to_hash: Vec<(String,
  [u8; 16])> }`` (:57-60, :139-155), replaced atomically (tmp + rename, :157-187).

Third-party pieces the reference takes from crates and this module from what the image has:
zstd through ``libzstd.so.1`` (ctypes), xxh3-64 through the ``xxhash`` module, postcard's wire
format (LEB128 varint lengths, strings as length + UTF-8, fixed arrays as their bytes) by hand.
Parity is UNPINNED: the reference ships no store fixture and cannot be run here, so the tests
pin this module to the layout as read from the source (names, JSON members, zstd frame flags,
postcard bytes of a worked example) and to its own round trips.
"""
from __future__ import annotations

import ctypes as C
import json
import os
import pathlib

ROOT = "seqdata"
SIDE_FILE = ".seqid_to_hash.bin"


# ------------------------------------------------------------------------- zstd
class _Zstd:
    _lib = None

    @classmethod
    def lib(cls):
        if cls._lib is None:
            L = C.CDLL("libzstd.so.1")
            L.ZSTD_compressBound.argtypes = [C.c_size_t]
            L.ZSTD_compressBound.restype = C.c_size_t
            L.ZSTD_createCCtx.restype = C.c_void_p
            L.ZSTD_freeCCtx.argtypes = [C.c_void_p]
            L.ZSTD_CCtx_setParameter.argtypes = [C.c_void_p, C.c_int, C.c_int]
            L.ZSTD_CCtx_setParameter.restype = C.c_size_t
            L.ZSTD_compress2.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
            L.ZSTD_compress2.restype = C.c_size_t
            L.ZSTD_decompress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
            L.ZSTD_decompress.restype = C.c_size_t
            L.ZSTD_isError.argtypes = [C.c_size_t]
            L.ZSTD_isError.restype = C.c_uint
            L.ZSTD_getErrorName.argtypes = [C.c_size_t]
            L.ZSTD_getErrorName.restype = C.c_char_p
            cls._lib = L
        return cls._lib

    @classmethod
    def compress(cls, data: bytes, level: int = 3, checksum: bool = True) -> bytes:
        L = cls.lib()
        cap = L.ZSTD_compressBound(len(data))
        dst = C.create_string_buffer(cap)
        cctx = L.ZSTD_createCCtx()
        try:
            L.ZSTD_CCtx_setParameter(cctx, 100, level)         # ZSTD_c_compressionLevel
            L.ZSTD_CCtx_setParameter(cctx, 201, int(checksum))  # ZSTD_c_checksumFlag
            n = L.ZSTD_compress2(cctx, dst, cap, data, len(data))
        finally:
            L.ZSTD_freeCCtx(cctx)
        if L.ZSTD_isError(n):
            raise RuntimeError("zstd: " + L.ZSTD_getErrorName(n).decode())
        return dst.raw[:n]

    @classmethod
    def decompress(cls, data: bytes, size: int) -> bytes:
        L = cls.lib()
        dst = C.create_string_buffer(max(size, 1))
        n = L.ZSTD_decompress(dst, size, data, len(data))
        if L.ZSTD_isError(n):
            raise RuntimeError("zstd: " + L.ZSTD_getErrorName(n).decode())
        if n != size:
            raise RuntimeError(f"zstd: chunk holds {n} bytes, the array has {size}")
        return dst.raw[:size]


def xxh3_hex(data: bytes) -> str:
    """zarr_io.rs:222-223: format!("{:016x}", xxh3_64(data))"""
    import xxhash

    return format(xxhash.xxh3_64_intdigest(data), "016x")


# --------------------------------------------------------------------- postcard
def _varint(n: int) -> bytes:
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        if n:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _read_varint(buf: bytes, pos: int) -> tuple[int, int]:
    shift = val = 0
    while True:
        b = buf[pos]
        pos += 1
        val |= (b & 0x7F) << shift
        if not b & 0x80:
            return val, pos
        shift += 7


def _pc_str(s: str) -> bytes:
    b = s.encode("utf8")
    return _varint(len(b)) + b


def _pc_read_str(buf: bytes, pos: int) -> tuple[str, int]:
    n, pos = _read_varint(buf, pos)
    return buf[pos: pos + n].decode("utf8"), pos + n


def encode_side_file(seqid_to_hash: dict[str, str]) -> bytes:
    """postcard of Metadata { seqid_to_hash: Vec<(String, [u8; 16])> } (zarr_io.rs:57-60)"""
    out = bytearray(_varint(len(seqid_to_hash)))
    for seqid, hexd in seqid_to_hash.items():
        h = hexd.encode("ascii")
        if len(h) != 16:
            raise ValueError("hash digests are 16 hex characters")
        out += _pc_str(seqid) + h
    return bytes(out)


def decode_side_file(buf: bytes) -> dict[str, str]:
    n, pos = _read_varint(buf, 0)
    out: dict[str, str] = {}
    for _ in range(n):
        seqid, pos = _pc_read_str(buf, pos)
        out[seqid] = buf[pos: pos + 16].decode("ascii")
        pos += 16
    return out


def encode_str_map(m: dict[str, str]) -> bytes:
    """postcard of FxHashMap<String, String> (zarr_io.rs:249)"""
    out = bytearray(_varint(len(m)))
    for k, v in m.items():
        out += _pc_str(str(k)) + _pc_str(str(v))
    return bytes(out)


def decode_str_map(buf: bytes) -> dict[str, str]:
    n, pos = _read_varint(buf, 0)
    out = {}
    for _ in range(n):
        k, pos = _pc_read_str(buf, pos)
        v, pos = _pc_read_str(buf, pos)
        out[k] = v
    return out


# ------------------------------------------------------------------------ store
class DvseqszDir:
    """file-system side of ZarrStore (src/zarr_io.rs:65-422)"""

    def __init__(self, path: str, mode: str = "r"):
        self.path = pathlib.Path(path)
        self.mode = mode
        if mode == "r" and not self.path.exists():
            raise FileNotFoundError(f'Path does not exist: "{path}"')  # zarr_py.rs:41-48
        try:
            if mode != "r":
                self.path.mkdir(parents=True, exist_ok=True)
            self.seqid_to_hash: dict[str, str] = {}
            side = self.path / SIDE_FILE
            if side.exists():
                try:
                    self.seqid_to_hash = decode_side_file(side.read_bytes())
                except (IndexError, UnicodeDecodeError, ValueError):
                    self.seqid_to_hash = {}  # zarr_io.rs:110-118: an unreadable file is an empty map
            group = self.path / ROOT / "zarr.json"
            if not group.exists():  # zarr_io.rs:82-95 (the reference rewrites it on every open)
                group.parent.mkdir(parents=True, exist_ok=True)
                group.write_text(json.dumps({"zarr_format": 3, "node_type": "group"}))
        except OSError as e:
            raise RuntimeError(f'Failed to create ZarrStore: "{path}"') from e  # zarr_py.rs:52-57

    # -- arrays
    def _array_dir(self, hexd: str) -> pathlib.Path:
        return self.path / ROOT / hexd

    def add(self, seqid: str, data: bytes, metadata: dict | None) -> None:
        """add_uint8_array, zarr_io.rs:211-282"""
        if seqid in self.seqid_to_hash:
            return
        hexd = xxh3_hex(data)
        known = hexd in self.seqid_to_hash.values()
        self.seqid_to_hash[seqid] = hexd
        if known:
            return
        n = len(data)
        if n == 0:
            del self.seqid_to_hash[seqid]
            raise ValueError("a zero-length array has no valid chunk shape")
        meta = {
            "zarr_format": 3, "node_type": "array", "shape": [n], "data_type": "uint8",
            "chunk_grid": {"name": "regular", "configuration": {"chunk_shape": [n]}},
            "chunk_key_encoding": {"name": "default", "configuration": {"separator": "/"}},
            "fill_value": 0,
            "codecs": [{"name": "bytes"}, {"name": "zstd", "configuration": {"level": 3, "checksum": True}}],
        }
        if metadata is not None:
            meta["attributes"] = {"metadata": list(encode_str_map(metadata))}
        d = self._array_dir(hexd)
        (d / "c").mkdir(parents=True, exist_ok=True)
        if data.count(0) != n:  # a chunk equal to the fill value is not stored (zarrs default)
            (d / "c" / "0").write_bytes(_Zstd.compress(data, 3, True))
        (d / "zarr.json").write_text(json.dumps(meta))

    def _meta(self, seqid: str) -> tuple[dict, pathlib.Path]:
        if seqid not in self.seqid_to_hash:
            raise KeyError(f"member '{seqid}' not found")  # zarr_io.rs:192-194
        d = self._array_dir(self.seqid_to_hash[seqid])
        return json.loads((d / "zarr.json").read_text()), d

    def read(self, seqid: str) -> bytes:
        """read_uint8_array, zarr_io.rs:309-313"""
        meta, d = self._meta(seqid)
        if meta.get("data_type") != "uint8" or len(meta.get("shape", [])) != 1:
            raise RuntimeError(f"array of '{seqid}' is not 1-D uint8")
        n = int(meta["shape"][0])
        sep = meta.get("chunk_key_encoding", {}).get("configuration", {}).get("separator", "/")
        kind = meta.get("chunk_key_encoding", {}).get("name", "default")
        key = ("c" + sep + "0") if kind == "default" else "0"
        chunk = d / key.replace("/", os.sep)
        if not chunk.exists():
            return bytes([int(meta.get("fill_value", 0))]) * n
        buf = chunk.read_bytes()
        for codec in reversed(meta.get("codecs", [])):
            name = codec.get("name")
            if name == "zstd":
                buf = _Zstd.decompress(buf, n)
            elif name == "bytes":
                pass  # one byte per element: no byte order
            else:
                raise RuntimeError(f"codec '{name}' is not one the reference writes")
        if len(buf) != n:
            raise RuntimeError(f"chunk of '{seqid}' holds {len(buf)} bytes, expected {n}")
        return buf

    def _plan(self, seqid: str):
        """(length, chunk path or None, fill value, needs zstd) of the array behind `seqid`"""
        meta, d = self._meta(seqid)
        if meta.get("data_type") != "uint8" or len(meta.get("shape", [])) != 1:
            raise RuntimeError(f"array of '{seqid}' is not 1-D uint8")
        n = int(meta["shape"][0])
        sep = meta.get("chunk_key_encoding", {}).get("configuration", {}).get("separator", "/")
        kind = meta.get("chunk_key_encoding", {}).get("name", "default")
        key = ("c" + sep + "0") if kind == "default" else "0"
        chunk = d / key.replace("/", os.sep)
        zstd = False
        for codec in meta.get("codecs", []):
            name = codec.get("name")
            if name == "zstd":
                zstd = True
            elif name != "bytes":
                raise RuntimeError(f"codec '{name}' is not one the reference writes")
        return n, (chunk if chunk.exists() else None), int(meta.get("fill_value", 0)), zstd

    def read_many(self, seqids, workers: int | None = None):
        """the arrays of `seqids` decoded straight into ONE buffer: (uint8 data, uint64 offsets[n+1]).
        What a selection over an on-disk store uploads; the per-array work (zarr.json, the chunk
        file, zstd) runs on a thread pool -- file reads and libzstd release the GIL -- and every
        chunk is decompressed in place, so no per-sequence bytes object is made or joined."""
        import numpy as np
        from concurrent.futures import ThreadPoolExecutor

        ids = list(seqids)
        workers = workers or min(32, len(os.sched_getaffinity(0)) * 2)
        L = _Zstd.lib()
        with ThreadPoolExecutor(max_workers=max(1, workers)) as pool:
            plans = list(pool.map(self._plan, ids, chunksize=64))
            offsets = np.zeros(len(ids) + 1, dtype=np.uint64)
            if ids:
                np.cumsum([p[0] for p in plans], out=offsets[1:], dtype=np.uint64)
            total = int(offsets[-1])
            data = np.empty(max(total, 16), dtype=np.uint8)
            base = data.ctypes.data

            def fill(i):
                n, chunk, fillv, zstd = plans[i]
                a = int(offsets[i])
                if chunk is None:
                    data[a:a + n] = fillv
                    return
                buf = chunk.read_bytes()
                if zstd:
                    got = L.ZSTD_decompress(C.c_void_p(base + a), n, buf, len(buf))
                    if L.ZSTD_isError(got):
                        raise RuntimeError("zstd: " + L.ZSTD_getErrorName(got).decode())
                    if got != n:
                        raise RuntimeError(f"zstd: chunk of '{ids[i]}' holds {got} bytes, the array has {n}")
                else:
                    if len(buf) != n:
                        raise RuntimeError(f"chunk of '{ids[i]}' holds {len(buf)} bytes, expected {n}")
                    data[a:a + n] = np.frombuffer(buf, dtype=np.uint8)

            list(pool.map(fill, range(len(ids)), chunksize=16))
        return data[:total] if total else data[:0], offsets

    def read_metadata(self, seqid: str) -> dict:
        """zarr_io.rs:315-337"""
        meta, _ = self._meta(seqid)
        raw = meta.get("attributes", {}).get("metadata")
        if raw is None:
            return {}
        try:
            return decode_str_map(bytes(raw))
        except (IndexError, ValueError, TypeError, UnicodeDecodeError):
            return {}

    def save_metadata(self) -> None:
        """zarr_io.rs:121-190: tmp file, fsync, rename"""
        tmp = self.path / (SIDE_FILE + ".tmp")
        with open(tmp, "wb") as f:
            f.write(encode_side_file(self.seqid_to_hash))
            f.flush()
            os.fsync(f.fileno())
        os.replace(tmp, self.path / SIDE_FILE)
