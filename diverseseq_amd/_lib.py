"""ctypes binding of libdvs_hip.so (include/dvs_hip.h) -- the only compute path.

There is no CPU fallback: if the HIP library is missing or no GPU is visible the
calls raise.  Nothing here imports ``oracle`` (the CPU restatement is test
infrastructure only).
"""

from __future__ import annotations

import ctypes as C
import os
import pathlib
import threading

import numpy as np

_HERE = pathlib.Path(__file__).resolve().parent
# DVS_HIP_LIB: another build of the same library (csrc/Makefile's host-sanitizer `asan` target)
LIB_PATH = pathlib.Path(os.environ["DVS_HIP_LIB"]).resolve() if os.environ.get("DVS_HIP_LIB") else _HERE / "libdvs_hip.so"

OK, ERR_VALUE, ERR_RUNTIME, ERR_NOMEM, ERR_UNSUPPORTED, ERR_ZERODIV = range(6)
MODE_NMOST, MODE_MAX, MODE_SET = 0, 1, 2
STAT_STDEV, STAT_COV = 0, 1
SELECT_NO_ARBITER = 1
SELECT_STEPWISE = 2
ROW_REMOTE = 0xFFFFFFFF

# every symbol include/dvs_hip.h declares (tests/test_boundary.py checks the .so exports them)
EXPORTS = (
    "dvs_abi_version", "dvs_ctx_create", "dvs_ctx_destroy", "dvs_last_error", "dvs_ctx_sync",
    "dvs_ctx_trim", "dvs_ctx_device_info", "dvs_ctx_set_timing", "dvs_ctx_refresh_knobs", "dvs_matrix_build", "dvs_matrix_from_freqs", "dvs_matrix_from_device_freqs", "dvs_matrix_get_source_rows", "dvs_matrix_count_bytes",
    "dvs_matrix_destroy", "dvs_matrix_nrows", "dvs_matrix_nbins", "dvs_matrix_dev_counts",
    "dvs_matrix_dev_totals", "dvs_matrix_dev_entropy", "dvs_matrix_get_counts",
    "dvs_matrix_get_totals", "dvs_matrix_get_entropy", "dvs_kmer_counts", "dvs_select_run",
    "dvs_select_destroy", "dvs_select_get_summary", "dvs_select_get_members",
    "dvs_select_gather_members",
    "dvs_select_delta_jsd", "dvs_select_step_pack",
    "dvs_select_step_apply", "dvs_select_step_poll", "dvs_select_step_peek", "dvs_select_bench_scan", "dvs_selftest_fast_log2", "dvs_selftest_log2_acc", "dvs_selftest_log2_f32", "dvs_selftest_exact_div", "dvs_selftest_handover", "dvs_mash_sketch", "dvs_mash_distances", "dvs_euclidean_distances", "dvs_sketches_build", "dvs_sketches_destroy", "dvs_sketches_get", "dvs_sketches_dev", "dvs_sketches_dev_lens", "dvs_sketches_distances",
    "dvs_sketches_from_device", "dvs_sketches_copy_to_device", "dvs_sketches_distances_device",
    "dvs_default_alphabet_lut", "dvs_seqbatch_from_fasta", "dvs_seqbatch_destroy", "dvs_seqbatch_info",
    "dvs_seqbatch_offsets", "dvs_seqbatch_header_positions", "dvs_seqbatch_dev_codes", "dvs_seqbatch_get_codes",
    "dvs_matrix_build_from_seqbatch",
    "dvs_pack_sequences", "dvs_packed_destroy", "dvs_packed_info", "dvs_packed_dev_codes", "dvs_packed_dev_mask",
    "dvs_packed_get", "dvs_matrix_build_packed", "dvs_sketches_build_packed", "dvs_seqbatch_pack",
    "dvs_seqbatch_packed", "dvs_sketches_build_from_seqbatch",
)


class SelectParams(C.Structure):
    _fields_ = [("mode", C.c_uint32), ("n_seed", C.c_uint32), ("max_size", C.c_uint32),
                ("stat", C.c_uint32), ("window", C.c_uint32), ("flags", C.c_uint32)]


class SelectSummary(C.Structure):
    _fields_ = [("size", C.c_uint32), ("lowest_index", C.c_uint32),
                ("total_jsd", C.c_double), ("mean_delta_jsd", C.c_double),
                ("std_delta_jsd", C.c_double), ("cov_delta_jsd", C.c_double),
                ("summed_entropies", C.c_double),
                ("rows_scored", C.c_uint64), ("rows_rechecked", C.c_uint64),
                ("n_windows", C.c_uint32), ("n_events", C.c_uint32),
                ("n_accepts", C.c_uint32), ("n_arbitrated", C.c_uint32),
                ("scan_ms", C.c_double), ("scan_launches", C.c_uint64),
                ("engine", C.c_uint32), ("rows_coarse_passed", C.c_uint32),
                ("scan_ms_last", C.c_double), ("rows_scored_last", C.c_uint64), ("arbiter_ms", C.c_double)]


class DvsLibraryMissing(RuntimeError):
    pass


_lib = None
_lock = threading.Lock()


def _preload_torch_hip_runtime() -> None:
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own
    libamdhip64.so.7 / libhsa-runtime64 (same SONAMEs as /opt/rocm's); if this
    library pulled in the system copy first, a later `import torch` would find no
    GPU, and device pointers could not be shared between the two.  So when torch
    is installed, its copy is loaded (RTLD_GLOBAL) before libdvs_hip.so, whose
    NEEDED libamdhip64.so.7 then resolves to it.  torch itself is not imported."""
    import importlib.util
    import sys

    if "torch" in sys.modules:
        return  # already loaded its runtime
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    libdir = pathlib.Path(list(spec.submodule_search_locations)[0]) / "lib"
    for name in ("libhsa-runtime64.so", "libamd_comgr.so", "libamdhip64.so"):
        p = libdir / name
        if p.exists():
            try:
                C.CDLL(str(p), mode=C.RTLD_GLOBAL)
            except OSError:
                return


def load() -> C.CDLL:
    """dlopen libdvs_hip.so and declare the ABI; raises if it has not been built"""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not LIB_PATH.exists():
            raise DvsLibraryMissing(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` (hipcc --offload-arch=gfx950). There is no CPU fallback.")
        _preload_torch_hip_runtime()
        L = C.CDLL(str(LIB_PATH))
        vp, u8p, u32p, u64p, f64p = (C.c_void_p, C.POINTER(C.c_uint8), C.POINTER(C.c_uint32),
                                     C.POINTER(C.c_uint64), C.POINTER(C.c_double))
        L.dvs_abi_version.restype = C.c_int
        L.dvs_ctx_create.argtypes = [C.c_int, vp, C.POINTER(vp)]
        L.dvs_ctx_destroy.argtypes = [vp]
        L.dvs_ctx_destroy.restype = None
        L.dvs_last_error.argtypes = [vp]
        L.dvs_last_error.restype = C.c_char_p
        L.dvs_ctx_sync.argtypes = [vp]
        L.dvs_ctx_trim.argtypes = [vp]
        L.dvs_ctx_set_timing.argtypes = [vp, C.c_int]
        L.dvs_ctx_refresh_knobs.argtypes = [vp]
        L.dvs_ctx_device_info.argtypes = [vp, C.c_char_p, C.c_size_t, C.POINTER(C.c_int), u64p]
        L.dvs_matrix_build.argtypes = [vp, vp, C.c_int, u64p, C.c_uint32, C.c_uint32, C.c_uint32,
                                       C.POINTER(vp)]
        L.dvs_matrix_from_freqs.argtypes = [vp, f64p, C.c_uint32, C.c_uint64, C.POINTER(vp)]
        L.dvs_matrix_from_device_freqs.argtypes = [vp, vp, vp, C.c_uint32, C.c_uint64, C.POINTER(vp)]
        L.dvs_select_gather_members.argtypes = [vp, vp, vp, vp, C.c_uint32]
        L.dvs_matrix_destroy.argtypes = [vp]
        L.dvs_matrix_destroy.restype = None
        L.dvs_matrix_nrows.argtypes = [vp]
        L.dvs_matrix_nrows.restype = C.c_uint32
        L.dvs_matrix_nbins.argtypes = [vp]
        L.dvs_matrix_nbins.restype = C.c_uint64
        for n in ("dvs_matrix_dev_counts", "dvs_matrix_dev_totals", "dvs_matrix_dev_entropy"):
            getattr(L, n).argtypes = [vp]
            getattr(L, n).restype = vp
        L.dvs_matrix_get_counts.argtypes = [vp, vp, C.c_uint32, C.c_uint32, u32p]
        L.dvs_matrix_get_totals.argtypes = [vp, vp, u32p]
        L.dvs_matrix_get_entropy.argtypes = [vp, vp, f64p]
        L.dvs_matrix_get_source_rows.argtypes = [vp, vp, u32p]
        L.dvs_matrix_count_bytes.argtypes = [vp]
        L.dvs_matrix_count_bytes.restype = C.c_uint32
        L.dvs_kmer_counts.argtypes = [vp, u8p, u64p, C.c_uint32, C.c_uint32, C.c_uint32, u32p, u32p,
                                      f64p]
        L.dvs_select_run.argtypes = [vp, vp, u32p, u32p, C.c_uint64, C.POINTER(SelectParams),
                                     C.POINTER(vp)]
        L.dvs_select_destroy.argtypes = [vp]
        L.dvs_select_destroy.restype = None
        L.dvs_select_get_summary.argtypes = [vp, vp, C.POINTER(SelectSummary)]
        L.dvs_select_get_members.argtypes = [vp, vp, u64p, u32p, f64p, f64p, f64p]
        L.dvs_select_delta_jsd.argtypes = [vp, vp, vp, u32p, f64p]
        L.dvs_select_step_pack.argtypes = [vp, vp, vp]
        L.dvs_select_step_apply.argtypes = [vp, vp, vp, C.c_uint32]
        L.dvs_select_step_poll.argtypes = [vp, vp, u32p, u64p]
        L.dvs_select_step_peek.argtypes = [vp, vp, C.c_uint32, u32p, C.POINTER(C.c_int)]
        L.dvs_select_bench_scan.argtypes = [vp, vp, C.c_int, f64p, u64p]
        L.dvs_selftest_fast_log2.argtypes = [vp, f64p]
        L.dvs_selftest_log2_acc.argtypes = [vp, f64p]
        L.dvs_selftest_log2_f32.argtypes = [vp, f64p]
        L.dvs_selftest_exact_div.argtypes = [vp, C.POINTER(C.c_uint64)]
        L.dvs_selftest_handover.argtypes = [vp, C.c_uint32, C.POINTER(C.c_uint64)]
        L.dvs_mash_sketch.argtypes = [vp, vp, C.c_int, u64p, C.c_uint32, C.c_uint32, C.c_uint32,
                                      C.c_uint32, C.c_int, u32p, u32p]
        L.dvs_mash_distances.argtypes = [vp, u32p, C.c_uint32, u32p, C.c_uint32, C.c_uint32,
                                         C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, f64p]
        L.dvs_euclidean_distances.argtypes = [vp, vp, f64p]
        L.dvs_sketches_build.argtypes = [vp, vp, C.c_int, u64p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                         C.c_int, C.POINTER(vp)]
        L.dvs_sketches_destroy.argtypes = [vp]
        L.dvs_sketches_destroy.restype = None
        L.dvs_sketches_get.argtypes = [vp, vp, u32p, u32p]
        for n in ("dvs_sketches_dev", "dvs_sketches_dev_lens"):
            getattr(L, n).argtypes = [vp]
            getattr(L, n).restype = vp
        L.dvs_sketches_distances.argtypes = [vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, f64p]
        L.dvs_sketches_from_device.argtypes = [vp, vp, vp, C.c_uint32, C.c_uint32, C.POINTER(vp)]
        L.dvs_sketches_copy_to_device.argtypes = [vp, vp, vp, C.c_uint32, vp]
        L.dvs_sketches_distances_device.argtypes = [vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, vp, vp]
        L.dvs_default_alphabet_lut.argtypes = [C.c_int, u8p]
        L.dvs_default_alphabet_lut.restype = None
        L.dvs_seqbatch_from_fasta.argtypes = [vp, vp, C.c_int, C.c_uint64, u8p, C.c_int, C.POINTER(vp)]
        L.dvs_seqbatch_destroy.argtypes = [vp]
        L.dvs_seqbatch_destroy.restype = None
        L.dvs_seqbatch_info.argtypes = [vp, u32p, u64p, u32p]
        L.dvs_seqbatch_offsets.argtypes = [vp, u64p]
        L.dvs_seqbatch_header_positions.argtypes = [vp, u64p]
        L.dvs_seqbatch_dev_codes.argtypes = [vp]
        L.dvs_seqbatch_dev_codes.restype = vp
        L.dvs_seqbatch_get_codes.argtypes = [vp, vp, u8p]
        L.dvs_matrix_build_from_seqbatch.argtypes = [vp, vp, C.c_uint32, C.c_uint32, C.POINTER(vp)]
        L.dvs_pack_sequences.argtypes = [vp, vp, C.c_int, C.c_uint64, C.POINTER(vp)]
        L.dvs_packed_destroy.argtypes = [vp]
        L.dvs_packed_destroy.restype = None
        L.dvs_packed_info.argtypes = [vp, u64p, u64p]
        for n in ("dvs_packed_dev_codes", "dvs_packed_dev_mask", "dvs_seqbatch_packed"):
            getattr(L, n).argtypes = [vp]
            getattr(L, n).restype = vp
        L.dvs_packed_get.argtypes = [vp, vp, u32p, C.POINTER(C.c_uint16)]
        L.dvs_matrix_build_packed.argtypes = [vp, vp, u64p, C.c_uint32, C.c_uint32, C.POINTER(vp)]
        L.dvs_sketches_build_packed.argtypes = [vp, vp, u64p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int,
                                                C.POINTER(vp)]
        L.dvs_seqbatch_pack.argtypes = [vp, vp]
        L.dvs_sketches_build_from_seqbatch.argtypes = [vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int,
                                                       C.POINTER(vp)]
        if L.dvs_abi_version() != 3:
            raise RuntimeError("libdvs_hip.so ABI version mismatch")
        _lib = L
        return _lib


def ptr(a: np.ndarray | None, ctype):
    return None if a is None else a.ctypes.data_as(C.POINTER(ctype))


def raise_for(rc: int, ctx) -> None:
    """map a DVS_ERR_* code to the exception the reference's binding raises"""
    if rc == OK:
        return
    msg = (load().dvs_last_error(ctx) or b"").decode(errors="replace")
    if rc == ERR_VALUE:
        raise ValueError(msg)  # panic -> ValueError, src/lib.rs:36-57
    if rc == ERR_NOMEM:
        raise MemoryError(msg)
    if rc == ERR_UNSUPPORTED:
        raise NotImplementedError(msg)
    if rc == ERR_ZERODIV:
        raise ZeroDivisionError(msg)
    raise RuntimeError(msg)
