"""ctypes binding of libkmerseek_amd.so (include/kmerseek_amd.h).

There is deliberately no fallback: if the HIP library is missing or fails to load, every entry
point raises.  The CPU checker used by the tests lives outside this package and is never imported from here.
"""
from __future__ import annotations

import ctypes as C
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
# KMERSEEK_AMD_LIB selects another build of the same library (kernel-tuning variants); never a fallback
SO_PATH = os.environ.get("KMERSEEK_AMD_LIB") or os.path.join(HERE, "libkmerseek_amd.so")

KS_OK = 0
KS_ERR_INVALID_MOLTYPE = 1
KS_ERR_INVALID_KSIZE = 2
KS_ERR_INVALID_RESIDUE = 3
KS_ERR_INVALID_ARG = 4
KS_ERR_OOM = 5
KS_ERR_HIP = 6
KS_ERR_NO_DEVICE = 7
KS_ERR_CAPACITY = 8
KS_ERR_INVALID_SCALED = 9

KS_PROTEIN, KS_DAYHOFF, KS_HP = 0, 1, 2
KS_SEED_DEFAULT = 42


class ks_params(C.Structure):
    _fields_ = [("ksize", C.c_uint32), ("scaled", C.c_uint32), ("moltype", C.c_uint32),
                ("flags", C.c_uint32), ("seed", C.c_uint64)]


class ks_residue_error(C.Structure):
    _fields_ = [("seq_index", C.c_uint32), ("position", C.c_uint32), ("residue", C.c_uint8)]


class ks_kernel_time(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("launches", C.c_uint64), ("total_ms", C.c_double)]


_vp = C.c_void_p
_u8p = C.POINTER(C.c_uint8)
_u32p = C.POINTER(C.c_uint32)
_u64p = C.POINTER(C.c_uint64)
_pp = C.POINTER(C.c_void_p)
_parp = C.POINTER(ks_params)

# name -> (restype, argtypes); mirrors include/kmerseek_amd.h one to one
SIGNATURES = {
    "ks_abi_version": (C.c_uint32, []),
    "ks_status_string": (C.c_char_p, [C.c_int]),
    "ks_moltype_from_string": (C.c_int, [C.c_char_p, _u32p]),
    "ks_max_hash": (C.c_uint64, [C.c_uint32]),
    "ks_ctx_create": (C.c_int, [C.c_int, _vp, _pp]),
    "ks_ctx_destroy": (None, [_vp]),
    "ks_last_error": (C.c_char_p, [_vp]),
    "ks_ctx_stream": (_vp, [_vp]),
    "ks_ctx_synchronize": (C.c_int, [_vp]),
    "ks_ctx_sketch_stats": (C.c_int, [_vp, C.POINTER(C.c_uint64 * 4)]),
    "ks_ctx_search_stats": (C.c_int, [_vp, C.POINTER(C.c_uint64 * 2)]),
    "ks_ctx_fused_stats": (C.c_int, [_vp, C.POINTER(C.c_uint64 * 2)]),
    "ks_ctx_reload_debug_env": (C.c_int, [_vp]),
    "ks_debug_guard_selftest": (C.c_int, [C.c_char_p]),
    "ks_host_alloc": (C.c_int, [_vp, C.c_uint64, _pp]),
    "ks_host_free": (C.c_int, [_vp, _vp]),
    "ks_dev_malloc": (C.c_int, [_vp, C.c_uint64, _pp]),
    "ks_dev_free": (C.c_int, [_vp, _vp]),
    "ks_dev_upload": (C.c_int, [_vp, _vp, _vp, C.c_uint64]),
    "ks_dev_download": (C.c_int, [_vp, _vp, _vp, C.c_uint64]),
    "ks_ctx_pool_stats": (C.c_int, [_vp, _u64p, _u64p, _u64p, _u64p]),
    "ks_validate_and_resolve": (C.c_int, [C.c_char_p, C.c_uint64, C.c_int, C.c_uint64, C.c_char_p, _u64p,
                                          C.POINTER(ks_residue_error)]),
    "ks_sketch_batch": (C.c_int, [_vp, _vp, _vp, C.c_uint32, _parp, _pp]),
    "ks_sketch_batch_device": (C.c_int, [_vp, _vp, _vp, C.c_uint32, C.c_uint64, C.c_uint32, _parp, _pp]),
    "ks_sketch_queries_device": (C.c_int, [_vp, _vp, _vp, _vp, C.c_uint32, C.c_uint64, C.c_uint32, _pp]),
    "ks_sketch_search_device": (C.c_int, [_vp, _vp, _vp, _vp, C.c_uint32, C.c_uint64, C.c_uint32, _pp, _pp]),
    "ks_sketch_search": (C.c_int, [_vp, _vp, _vp, _vp, C.c_uint32, _pp, _pp]),
    "ks_sketches_has_postings": (C.c_int, [_vp]),
    "ks_sketches_n_seqs": (C.c_uint32, [_vp]),
    "ks_sketches_n_hashes": (C.c_uint64, [_vp]),
    "ks_sketches_n_windows": (C.c_uint64, [_vp]),
    "ks_sketches_params": (None, [_vp, _parp]),
    "ks_sketches_device_offsets": (_vp, [_vp]),
    "ks_sketches_device_hashes": (_vp, [_vp]),
    "ks_sketches_device_abunds": (_vp, [_vp]),
    "ks_sketches_copy_to_host": (C.c_int, [_vp, _vp, _vp, _vp, _vp]),
    "ks_sketches_from_host": (C.c_int, [_vp, _vp, _vp, _vp, C.c_uint32, _parp, _pp]),
    "ks_sketches_union": (C.c_int, [_vp, _vp, _pp]),
    "ks_sketches_free": (None, [_vp]),
    "ks_kmer_positions": (C.c_int, [_vp, _vp, _vp, C.c_uint32, _parp, _pp]),
    "ks_kmer_positions_device": (C.c_int, [_vp, _vp, _vp, C.c_uint32, C.c_uint64, _parp, _pp]),
    "ks_kmerpos_count": (C.c_uint64, [_vp]),
    "ks_kmerpos_copy_to_host": (C.c_int, [_vp, _vp, _vp, _vp, _vp]),
    "ks_kmerpos_free": (None, [_vp]),
    "ks_index_build": (C.c_int, [_vp, _vp, _pp]),
    "ks_index_n_targets": (C.c_uint32, [_vp]),
    "ks_index_n_postings": (C.c_uint64, [_vp]),
    "ks_index_free": (None, [_vp]),
    "ks_search": (C.c_int, [_vp, _vp, _vp, _pp]),
    "ks_hits_count": (C.c_uint64, [_vp]),
    "ks_hits_n_pair_instances": (C.c_uint64, [_vp]),
    "ks_hits_partition_path": (C.c_int, [_vp]),
    "ks_hits_bucket_posting_bytes": (C.c_int, [_vp]),
    "ks_hits_copy_to_host": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "ks_hits_device_qid": (_vp, [_vp]),
    "ks_hits_device_tid": (_vp, [_vp]),
    "ks_hits_device_intersect": (_vp, [_vp]),
    "ks_hits_device_n_weighted": (_vp, [_vp]),
    "ks_hits_copy_to_device": (C.c_int, [_vp, _vp, C.c_uint32, C.c_uint32, _vp, _vp, _vp, _vp]),
    "ks_hits_pack64_to_device": (C.c_int, [_vp, _vp, C.c_uint32, C.c_uint32, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp, C.c_uint32]),
    "ks_hits_unpack64_device": (C.c_int, [_vp, _vp, C.c_uint64, C.c_int, C.c_int, _vp, _vp, _vp, _vp]),
    "ks_hits_merge_by_qid_device": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.POINTER(C.c_uint64), C.c_uint32, C.c_uint32, _vp, _vp, _vp, _vp]),
    "ks_hits_free": (None, [_vp]),
    "ks_timing_enable": (C.c_int, [_vp, C.c_int]),
    "ks_timing_reset": (C.c_int, [_vp]),
    "ks_timing_get": (C.c_int, [_vp, C.POINTER(ks_kernel_time), C.c_uint32, _u32p]),
    "ks_bench_gather_rates": (C.c_int, [_vp, C.POINTER(C.c_double * 2)]),
    "ks_bench_device_rates": (C.c_int, [_vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
}

_lib = None


class KmerseekLibraryError(RuntimeError):
    pass


def load():
    """dlopen the HIP library; raises KmerseekLibraryError (never falls back) if it is unavailable."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise KmerseekLibraryError(
            f"{SO_PATH} not found: build it with `python -m kmerseek_amd.build` (hipcc, gfx950). "
            "There is no CPU fallback.")
    # One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64, and whichever copy is mapped first serves
    # both.  If this library pulled in /opt/rocm's copy BEFORE torch was imported, a later `import torch` would bring a
    # second runtime and ks_ctx_create would then see "no HIP device" (measured: bench.py, round 2).  So when torch is
    # installed and the process may use it (device memory, streams, torch.distributed), let it load first.
    if "torch" not in sys.modules and os.environ.get("KS_NO_TORCH_PRELOAD") != "1":
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    try:
        L = C.CDLL(SO_PATH)
    except OSError as e:  # missing libamdhip64 etc.
        raise KmerseekLibraryError(f"cannot load {SO_PATH}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(L, name)
        except AttributeError as e:
            raise KmerseekLibraryError(f"{SO_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L
