"""Object wrappers over the C ABI (include/kmerseek_amd.h): Context, Sketches, Index, Hits.

Host arrays are numpy; device-resident inputs are passed as raw pointers (e.g. ``tensor.data_ptr()``),
so nothing here depends on torch.  All compute happens in the HIP library.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from ._lib import ks_params

MOLTYPES = {"protein": _lib.KS_PROTEIN, "raw": _lib.KS_PROTEIN, "dayhoff": _lib.KS_DAYHOFF, "hp": _lib.KS_HP}
MOLTYPE_NAMES = {_lib.KS_PROTEIN: "protein", _lib.KS_DAYHOFF: "dayhoff", _lib.KS_HP: "hp"}
SEED = _lib.KS_SEED_DEFAULT  # src/rust/signature.rs:12


class KmerseekError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(message)
        self.status = status


class InvalidAminoAcid(KmerseekError):
    """IndexError::InvalidAminoAcid(char, position) — src/rust/errors.rs:14-15."""

    def __init__(self, char: str, position: int, seq_index: int = 0):
        super().__init__(_lib.KS_ERR_INVALID_RESIDUE, f"Invalid amino acid '{char}' found at position {position}")
        self.char, self.position, self.seq_index = char, position, seq_index


def moltype_id(moltype: str) -> int:
    L = _lib.load()
    out = C.c_uint32(0)
    st = L.ks_moltype_from_string(moltype.encode(), C.byref(out))
    if st != _lib.KS_OK:
        # message of src/rust/encoding.rs:22-25
        raise KmerseekError(st, f"Invalid moltype: {moltype}, only 'protein', 'hp', or 'dayhoff' are supported")
    return out.value


def make_params(ksize: int, scaled: int, moltype: str, seed: int = SEED) -> ks_params:
    return ks_params(ksize=int(ksize), scaled=int(scaled), moltype=moltype_id(moltype), flags=0, seed=int(seed))


def max_hash(scaled: int) -> int:
    return int(_lib.load().ks_max_hash(int(scaled)))


def validate_and_resolve(seq: bytes, upper: bool = False, rng_seed: int = 0) -> bytes:
    """Host pre-step (src/rust/aminoacid.rs:74-105); raises InvalidAminoAcid."""
    L = _lib.load()
    out = C.create_string_buffer(len(seq) + 1)
    out_len = C.c_uint64(0)
    err = _lib.ks_residue_error()
    st = L.ks_validate_and_resolve(seq, len(seq), 1 if upper else 0, rng_seed, out, C.byref(out_len), C.byref(err))
    if st == _lib.KS_ERR_INVALID_RESIDUE:
        raise InvalidAminoAcid(chr(err.residue), err.position)
    if st != _lib.KS_OK:
        raise KmerseekError(st, L.ks_status_string(st).decode())
    return out.raw[:out_len.value]


def pack(seqs: Sequence[bytes]) -> Tuple[np.ndarray, np.ndarray]:
    """Concatenate records into (residues u8, offsets u64[n+1]) — the batch layout of the C ABI."""
    offs = np.zeros(len(seqs) + 1, dtype=np.uint64)
    if len(seqs):
        offs[1:] = np.cumsum([len(s) for s in seqs], dtype=np.uint64)
    res = np.frombuffer(b"".join(seqs), dtype=np.uint8).copy() if len(seqs) else np.zeros(0, np.uint8)
    return res, offs


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class _FollowDebugEnv:
    """Library proxy of a diagnostic context: re-reads the KS_DEBUG_* variables before every call (the library itself reads
    them only when a context is created).  The tests use it to force the rarely taken paths on one context."""

    def __init__(self, L, ctx):
        self._L, self._ctx = L, ctx

    def __getattr__(self, name):
        f = getattr(self._L, name)
        if not name.startswith("ks_") or name in ("ks_ctx_create", "ks_ctx_destroy", "ks_ctx_reload_debug_env", "ks_last_error",
                                                    "ks_status_string"):
            return f

        def call(*a):
            if self._ctx._h:
                self._L.ks_ctx_reload_debug_env(self._ctx._h)
            return f(*a)
        return call


class Context:
    """One HIP device + stream + workspace (ks_ctx).  Not thread-safe: one per host thread.

    stream: None — the context creates a non-blocking stream of its own; a raw ``hipStream_t`` value (e.g.
    ``torch.cuda.current_stream().cuda_stream``) — the context launches there, so its work is ordered with the caller's.
    0 is what torch reports for the device's default (null) stream: it is passed on as ``hipStreamLegacy``, and
    ``Context.stream`` reports it as 0 again, so ``ctx.stream == torch.cuda.current_stream().cuda_stream`` holds."""

    _HIP_STREAM_LEGACY = 1  # hip_runtime_api.h: #define hipStreamLegacy ((hipStream_t)1)

    def __init__(self, device: int = 0, stream: Optional[int] = None, follow_debug_env: bool = False):
        self._L = _lib.load()
        self._h = None
        if follow_debug_env:
            self._L = _FollowDebugEnv(self._L, self)
        h = C.c_void_p()
        sp = None if stream is None else C.c_void_p(int(stream) if int(stream) != 0 else self._HIP_STREAM_LEGACY)
        st = self._L.ks_ctx_create(int(device), sp, C.byref(h))
        if st != _lib.KS_OK:
            raise KmerseekError(st, f"ks_ctx_create(device={device}) failed: {self._L.ks_status_string(st).decode()}"
                                    " — the HIP path has no CPU fallback")
        self._h = h
        self.device = device
        self._pinned, self._close_pending = 0, False

    def close(self):
        """Destroy the context (its pool, stream, pinned blocks).  While views of library-owned device arrays are alive
        (`_Owned.pin`: the world-size-1 hit exchange hands out torch views) the destruction waits for the last of them —
        a view must never outlive the memory it points into."""
        if getattr(self, "_pinned", 0) > 0:
            self._close_pending = True
            return
        if getattr(self, "_h", None):
            self._L.ks_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, st: int):
        if st != _lib.KS_OK:
            msg = self._L.ks_last_error(self._h).decode() or self._L.ks_status_string(st).decode()
            raise KmerseekError(st, msg)

    @property
    def stream(self) -> int:
        v = int(self._L.ks_ctx_stream(self._h) or 0)
        return 0 if v == self._HIP_STREAM_LEGACY else v

    def synchronize(self):
        self._check(self._L.ks_ctx_synchronize(self._h))

    def to_device(self, a: np.ndarray) -> "DeviceBuffer":
        """Copy a host array into a plain device buffer (for the *_device entry points without torch)."""
        a = np.ascontiguousarray(a)
        p = C.c_void_p()
        self._check(self._L.ks_dev_malloc(self._h, a.nbytes, C.byref(p)))
        buf = DeviceBuffer(self, p, a.nbytes)
        self._check(self._L.ks_dev_upload(self._h, p, _ptr(a), a.nbytes))
        return buf

    def pinned_empty(self, n: int, dtype) -> np.ndarray:
        """Uninitialised numpy array of n items in pinned host memory (ks_host_alloc): copies between such an array and the
        device run as one DMA at link rate, where a pageable array is staged through pinned buffers by host threads.
        The memory is released when the array (and every view of it) is garbage-collected."""
        dt = np.dtype(dtype)
        nbytes = max(int(n) * dt.itemsize, 1)
        p = C.c_void_p()
        self._check(self._L.ks_host_alloc(self._h, nbytes, C.byref(p)))
        buf = (C.c_char * nbytes).from_address(p.value)
        buf._owner = _PinnedBlock(self, p)  # the ctypes view is the array's base: it keeps the block alive
        return np.frombuffer(buf, dtype=dt, count=int(n))

    def pool_stats(self) -> Dict[str, int]:
        v = [C.c_uint64(0) for _ in range(4)]
        self._check(self._L.ks_ctx_pool_stats(self._h, *[C.byref(x) for x in v]))
        return dict(zip(("blocks", "bytes_held", "bytes_in_use", "mallocs"), (int(x.value) for x in v)))

    def sketch_stats(self) -> Dict[str, int]:
        """Repeats this context needed so far (see ks_ctx_sketch_stats): results never depend on them, run time does."""
        v = (C.c_uint64 * 4)()
        self._check(self._L.ks_ctx_sketch_stats(self._h, C.byref(v)))
        return {"ticket_fallbacks": int(v[0]), "uses_ticket": int(v[1]), "compact_fallbacks": int(v[2]), "cap_fallbacks": int(v[3])}

    def unpack_hits64_device(self, d_packed: int, n: int, qbits: int, tbits: int, d_qid: int, d_tid: int, d_isect: int, d_nw: int):
        """Transport words -> columns, all device pointers (ks_hits_unpack64_device): the receiving side of the exchange."""
        self._check(self._L.ks_hits_unpack64_device(self._h, C.c_void_p(d_packed), n, qbits, tbits, C.c_void_p(d_qid), C.c_void_p(d_tid),
                                                    C.c_void_p(d_isect), C.c_void_p(d_nw)))

    def merge_hits_by_qid_device(self, d_qid: int, d_tid: int, d_isect: int, d_nw: int, block_rows: Sequence[int], n_queries: int,
                                 o_qid: int, o_tid: int, o_isect: int, o_nw: int):
        """Rank blocks of a gathered index-sharded hit list -> one list ordered by (qid, tid), all device pointers
        (ks_hits_merge_by_qid_device: a counting merge instead of a sort)."""
        rows = (C.c_uint64 * len(block_rows))(*[int(x) for x in block_rows])
        self._check(self._L.ks_hits_merge_by_qid_device(self._h, C.c_void_p(d_qid), C.c_void_p(d_tid), C.c_void_p(d_isect), C.c_void_p(d_nw),
                                                        rows, len(block_rows), int(n_queries), C.c_void_p(o_qid), C.c_void_p(o_tid),
                                                        C.c_void_p(o_isect), C.c_void_p(o_nw)))

    def reload_debug_env(self):
        """Re-read the KS_DEBUG_* variables (diagnostics: the library reads them only when a context is created)."""
        self._check(self._L.ks_ctx_reload_debug_env(self._h))

    def search_stats(self) -> Dict[str, int]:
        """Repeats ks_search needed so far on this context (see ks_ctx_search_stats)."""
        v = (C.c_uint64 * 2)()
        self._check(self._L.ks_ctx_search_stats(self._h, C.byref(v)))
        return {"join_retries": int(v[0]), "rows_ticket_fallbacks": int(v[1])}

    # ---- sketch ----
    def sketch_batch(self, residues: np.ndarray, offsets: np.ndarray, ksize: int, scaled: int, moltype: str,
                     seed: int = SEED) -> "Sketches":
        residues = np.ascontiguousarray(residues, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        p = make_params(ksize, scaled, moltype, seed)
        out = C.c_void_p()
        self._check(self._L.ks_sketch_batch(self._h, _ptr(residues), _ptr(offsets), len(offsets) - 1, C.byref(p),
                                            C.byref(out)))
        return Sketches(self, out)

    def sketch_batch_device(self, d_residues: int, d_offsets: int, n_seqs: int, n_residues: int, ksize: int,
                            scaled: int, moltype: str, max_seq_len: int = 0, seed: int = SEED) -> "Sketches":
        p = make_params(ksize, scaled, moltype, seed)
        out = C.c_void_p()
        self._check(self._L.ks_sketch_batch_device(self._h, C.c_void_p(d_residues), C.c_void_p(d_offsets), n_seqs,
                                                   n_residues, max_seq_len, C.byref(p), C.byref(out)))
        return Sketches(self, out)

    def sketch_queries_device(self, index: "Index", d_residues: int, d_offsets: int, n_seqs: int, n_residues: int,
                              max_seq_len: int = 0) -> "Sketches":
        """Sketch a query batch for an immediate search against `index` (also emits pre-partitioned postings)."""
        out = C.c_void_p()
        self._check(self._L.ks_sketch_queries_device(self._h, index._h, C.c_void_p(d_residues), C.c_void_p(d_offsets),
                                                     n_seqs, n_residues, max_seq_len, C.byref(out)))
        return Sketches(self, out)

    def sketch_search_device(self, index: "Index", d_residues: int, d_offsets: int, n_seqs: int, n_residues: int,
                             max_seq_len: int = 0, want_sketches: bool = True):
        """ks_sketch_search_device: sketch a query batch and search it against `index` in one call (two host waits instead
        of three).  Returns (Sketches or None, Hits); same results as sketch_queries_device + search."""
        sk, hits = C.c_void_p(), C.c_void_p()
        self._check(self._L.ks_sketch_search_device(self._h, index._h, C.c_void_p(d_residues), C.c_void_p(d_offsets), n_seqs,
                                                    n_residues, max_seq_len, C.byref(sk) if want_sketches else None, C.byref(hits)))
        return (Sketches(self, sk) if want_sketches else None), Hits(self, hits)

    def sketch_search(self, index: "Index", residues: np.ndarray, offsets: np.ndarray, want_sketches: bool = True):
        """ks_sketch_search: the same from host arrays (upload, sketch, search).  Returns (Sketches or None, Hits)."""
        residues = np.ascontiguousarray(residues, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        sk, hits = C.c_void_p(), C.c_void_p()
        self._check(self._L.ks_sketch_search(self._h, index._h, _ptr(residues), _ptr(offsets), len(offsets) - 1,
                                             C.byref(sk) if want_sketches else None, C.byref(hits)))
        return (Sketches(self, sk) if want_sketches else None), Hits(self, hits)

    def fused_stats(self) -> Dict[str, int]:
        """ks_sketch_search_device calls on this context: with the sketch read-back deferred / repeated the plain way."""
        v = (C.c_uint64 * 2)()
        self._check(self._L.ks_ctx_fused_stats(self._h, C.byref(v)))
        return {"deferred": int(v[0]), "redos": int(v[1])}

    def sketches_from_host(self, offsets: np.ndarray, hashes: np.ndarray, abunds: np.ndarray, ksize: int,
                           scaled: int, moltype: str, seed: int = SEED) -> "Sketches":
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        hashes = np.ascontiguousarray(hashes, dtype=np.uint64)
        abunds = np.ascontiguousarray(abunds, dtype=np.uint32)
        p = make_params(ksize, scaled, moltype, seed)
        out = C.c_void_p()
        self._check(self._L.ks_sketches_from_host(self._h, _ptr(offsets), _ptr(hashes), _ptr(abunds),
                                                  len(offsets) - 1, C.byref(p), C.byref(out)))
        return Sketches(self, out)

    def kmer_positions(self, residues: np.ndarray, offsets: np.ndarray, ksize: int, scaled: int, moltype: str,
                       seed: int = SEED) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        """(seq u32, start u32, hash u64) of every kept window, ordered by (seq, start)."""
        residues = np.ascontiguousarray(residues, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        p = make_params(ksize, scaled, moltype, seed)
        out = C.c_void_p()
        self._check(self._L.ks_kmer_positions(self._h, _ptr(residues), _ptr(offsets), len(offsets) - 1, C.byref(p),
                                              C.byref(out)))
        try:
            n = int(self._L.ks_kmerpos_count(out))
            seq = np.zeros(n, np.uint32); start = np.zeros(n, np.uint32); h = np.zeros(n, np.uint64)
            self._check(self._L.ks_kmerpos_copy_to_host(self._h, out, _ptr(seq), _ptr(start), _ptr(h)))
        finally:
            self._L.ks_kmerpos_free(out)
        return seq, start, h

    def kmer_positions_device(self, d_residues: int, d_offsets: int, n_seqs: int, n_residues: int, ksize: int, scaled: int,
                              moltype: str, seed: int = SEED, fetch: bool = True):
        """Same from device-resident buffers; fetch=False returns only the number of kept windows (table stays on the GPU
        and is freed) — what bench / profiling runs use."""
        p = make_params(ksize, scaled, moltype, seed)
        out = C.c_void_p()
        self._check(self._L.ks_kmer_positions_device(self._h, C.c_void_p(d_residues), C.c_void_p(d_offsets), n_seqs, n_residues,
                                                     C.byref(p), C.byref(out)))
        try:
            n = int(self._L.ks_kmerpos_count(out))
            if not fetch:
                return n
            seq = np.zeros(n, np.uint32); start = np.zeros(n, np.uint32); h = np.zeros(n, np.uint64)
            self._check(self._L.ks_kmerpos_copy_to_host(self._h, out, _ptr(seq), _ptr(start), _ptr(h)))
        finally:
            self._L.ks_kmerpos_free(out)
        return seq, start, h

    # ---- index / search ----
    def index_build(self, targets: "Sketches") -> "Index":
        out = C.c_void_p()
        self._check(self._L.ks_index_build(self._h, targets._h, C.byref(out)))
        return Index(self, out)

    def search(self, index: "Index", queries: "Sketches") -> "Hits":
        out = C.c_void_p()
        self._check(self._L.ks_search(self._h, index._h, queries._h, C.byref(out)))
        return Hits(self, out)

    # ---- measurement ----
    def timing_enable(self, on=True):
        """on: False/0 off, True/1 every launch, 2 only the kernels that carry the bytes (low overhead)."""
        self._check(self._L.ks_timing_enable(self._h, int(on)))

    def timing_reset(self):
        self._check(self._L.ks_timing_reset(self._h))

    def device_rates(self) -> Dict[str, float]:
        """u64-multiply rate, device copy rate and nominal memory bandwidth of this context's GPU (bench.py prints them)."""
        v = [C.c_double(0) for _ in range(3)]
        self._check(self._L.ks_bench_device_rates(self._h, *[C.byref(x) for x in v]))
        return {"u64_gmul_per_s": v[0].value, "copy_gb_per_s": v[1].value, "nominal_gb_per_s": v[2].value}

    def gather_rates(self) -> Dict[str, float]:
        """Random 8-byte gathers per second from a 134 MB and from a 2 MB table (the ceiling of a table-lookup hash for hp)."""
        v = (C.c_double * 2)()
        self._check(self._L.ks_bench_gather_rates(self._h, C.byref(v)))
        return {"gathers_per_s_134mb": v[0], "gathers_per_s_2mb": v[1]}

    def timing(self) -> Dict[str, Tuple[int, float]]:
        """{kernel name: (launches, total ms)} from HIP events on the context's stream."""
        n = C.c_uint32(0)
        rows = (_lib.ks_kernel_time * 64)()
        self._check(self._L.ks_timing_get(self._h, rows, 64, C.byref(n)))
        return {rows[i].name.decode(): (int(rows[i].launches), float(rows[i].total_ms)) for i in range(min(n.value, 64))}


class _PinnedBlock:
    def __init__(self, ctx: "Context", ptr):
        self._ctx, self._p = ctx, ptr

    def __del__(self):
        try:
            if self._p is not None and self._ctx._h:
                self._ctx._L.ks_host_free(self._ctx._h, self._p)
        except Exception:
            pass
        self._p = None


class DeviceBuffer:
    def __init__(self, ctx: Context, ptr, nbytes: int):
        self._ctx, self._p, self.nbytes = ctx, ptr, nbytes

    @property
    def ptr(self) -> int:
        return int(self._p.value or 0)

    def free(self):
        if self._p is not None and self._ctx._h:
            self._ctx._L.ks_dev_free(self._ctx._h, self._p)
        self._p = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class _Owned:
    _free = None

    def __init__(self, ctx: Context, handle):
        self._ctx, self._h = ctx, handle
        self._pins, self._free_pending = 0, False

    def free(self):
        """Release the device object now — or, while views of its device arrays are alive (pin / unpin: the world-size-1
        hit exchange hands out torch views instead of copies), as soon as the last of them is gone."""
        if getattr(self, "_pins", 0) > 0:
            self._free_pending = True
            return
        if getattr(self, "_h", None) and self._ctx._h:
            getattr(self._ctx._L, self._free)(self._h)
        self._h = None

    def pin(self):
        self._pins += 1
        self._ctx._pinned += 1

    def unpin(self):
        self._pins -= 1
        self._ctx._pinned -= 1
        if self._pins == 0 and self._free_pending:
            self._free_pending = False
            self.free()
        if self._ctx._pinned == 0 and self._ctx._close_pending:
            self._ctx._close_pending = False
            self._ctx.close()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Sketches(_Owned):
    """Device-resident CSR of per-sequence sketches (ks_sketches)."""
    _free = "ks_sketches_free"

    @property
    def n_seqs(self) -> int:
        return int(self._ctx._L.ks_sketches_n_seqs(self._h))

    @property
    def n_hashes(self) -> int:
        return int(self._ctx._L.ks_sketches_n_hashes(self._h))

    @property
    def n_windows(self) -> int:
        return int(self._ctx._L.ks_sketches_n_windows(self._h))

    @property
    def has_postings(self) -> bool:
        return bool(self._ctx._L.ks_sketches_has_postings(self._h))

    @property
    def posting_bytes(self) -> int:
        """Bytes per partitioned query posting: 12, 10 (big fingerprint indexes at scaled = 1) or 0 (none attached)."""
        return (0, 12, 10)[int(self._ctx._L.ks_sketches_has_postings(self._h))]

    def device_ptrs(self) -> Tuple[int, int, int]:
        """Raw device pointers of the CSR (offsets u64[n_seqs + 1], hashes u64[n_hashes], abunds u32[n_hashes])."""
        L = self._ctx._L
        return tuple(int(f(self._h) or 0) for f in (L.ks_sketches_device_offsets, L.ks_sketches_device_hashes,
                                                    L.ks_sketches_device_abunds))

    def union(self) -> "Sketches":
        """Combined sketch: sorted unique hashes of all sequences with summed abundances (one sequence)."""
        out = C.c_void_p()
        self._ctx._check(self._ctx._L.ks_sketches_union(self._ctx._h, self._h, C.byref(out)))
        return Sketches(self._ctx, out)

    def to_host(self, pinned: bool = False) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        """(offsets u64[n+1], hashes u64, abunds u32) on the host.  pinned=True returns arrays in pinned memory
        (Context.pinned_empty): one DMA at link rate instead of staged copies — what large sketch sets should use."""
        n, m = self.n_seqs, self.n_hashes
        if pinned:
            offs = self._ctx.pinned_empty(n + 1, np.uint64); hashes = self._ctx.pinned_empty(m, np.uint64)
            abunds = self._ctx.pinned_empty(m, np.uint32)
        else:
            offs = np.empty(n + 1, np.uint64); hashes = np.empty(m, np.uint64); abunds = np.empty(m, np.uint32)
        self._ctx._check(self._ctx._L.ks_sketches_copy_to_host(self._ctx._h, self._h, _ptr(offs), _ptr(hashes),
                                                               _ptr(abunds)))
        return offs, hashes, abunds


class Index(_Owned):
    _free = "ks_index_free"

    @property
    def n_targets(self) -> int:
        return int(self._ctx._L.ks_index_n_targets(self._h))

    @property
    def n_postings(self) -> int:
        return int(self._ctx._L.ks_index_n_postings(self._h))


class Hits(_Owned):
    _free = "ks_hits_free"

    @property
    def count(self) -> int:
        return int(self._ctx._L.ks_hits_count(self._h))

    @property
    def n_pair_instances(self) -> int:
        return int(self._ctx._L.ks_hits_n_pair_instances(self._h))

    @property
    def partition_path(self) -> int:
        """0 sketch regions == buckets, 1 regions + bucket scatter, 2 regions + dense pass, 3 dense from the CSR."""
        return int(self._ctx._L.ks_hits_partition_path(self._h))

    @property
    def bucket_posting_bytes(self) -> int:
        """Bytes per query posting inside the join buckets (12 / 10 / 9; 0 = no bucket scatter ran)."""
        return int(self._ctx._L.ks_hits_bucket_posting_bytes(self._h))

    def device_ptrs(self) -> Tuple[int, int, int, int]:
        """Raw device pointers of the COO columns (qid u32, tid u32, intersect u32, n_weighted u64), `count` entries each."""
        L = self._ctx._L
        return tuple(int(f(self._h) or 0) for f in (L.ks_hits_device_qid, L.ks_hits_device_tid, L.ks_hits_device_intersect,
                                                    L.ks_hits_device_n_weighted))

    def copy_to_device(self, d_qid: int, d_tid: int, d_isect: int, d_nw: int, qid_base: int = 0, tid_base: int = 0):
        """D2D copy of the columns into caller-owned device buffers (ids shifted to global numbering); asynchronous on the
        context's stream.  This is how a shard's hits enter the send block of an all-gather (kmerseek_amd/dist.py)."""
        self._ctx._check(self._ctx._L.ks_hits_copy_to_device(self._ctx._h, self._h, qid_base, tid_base, C.c_void_p(d_qid),
                                                             C.c_void_p(d_tid), C.c_void_p(d_isect), C.c_void_p(d_nw)))

    def pack64_to_device(self, d_packed: int, d_esc_row: int, d_esc_isect: int, d_esc_nw: int, d_n_esc: int, esc_cap: int,
                         qbits: int, tbits: int, qid_base: int = 0, tid_base: int = 0):
        """Rows as 64-bit transport words (ks_hits_pack64_to_device) into caller-owned device buffers: 8 instead of 20
        bytes per row for the all-gather of a multi-GPU search; rows with wide values go to the escape list."""
        self._ctx._check(self._ctx._L.ks_hits_pack64_to_device(self._ctx._h, self._h, qid_base, tid_base, qbits, tbits,
                                                               C.c_void_p(d_packed), C.c_void_p(d_esc_row), C.c_void_p(d_esc_isect),
                                                               C.c_void_p(d_esc_nw), C.c_void_p(d_n_esc), esc_cap))

    def to_host(self) -> Tuple[np.ndarray, np.ndarray, np.ndarray, np.ndarray]:
        n = self.count
        qid = np.zeros(n, np.uint32); tid = np.zeros(n, np.uint32)
        isect = np.zeros(n, np.uint32); nw = np.zeros(n, np.uint64)
        self._ctx._check(self._ctx._L.ks_hits_copy_to_host(self._ctx._h, self._h, _ptr(qid), _ptr(tid), _ptr(isect),
                                                           _ptr(nw)))
        return qid, tid, isect, nw
