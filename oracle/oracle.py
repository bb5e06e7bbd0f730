"""ctypes/numpy front-end of the CPU oracle (oracle/ks_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of bench.py — never by anything under ``kmerseek_amd/``.
Parity status: PINNED against the reference's fixtures (tests/test_oracle_golden.py).
"""
from __future__ import annotations

import ctypes as C
import hashlib
import math
import os
import subprocess
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libks_oracle.so")

MOLTYPES = {"protein": 0, "raw": 0, "dayhoff": 1, "hp": 2}
SEED = 42  # src/rust/signature.rs:12


def build(force: bool = False) -> str:
    """Compile the C restatement with gcc (seconds)."""
    src = os.path.join(_HERE, "ks_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libks_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        u8p, u32p, u64p = C.POINTER(C.c_uint8), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)
        L.kso_hash_murmur.restype = C.c_uint64
        L.kso_hash_murmur.argtypes = [C.c_char_p, C.c_size_t, C.c_uint64]
        L.kso_max_hash.restype = C.c_uint64
        L.kso_max_hash.argtypes = [C.c_uint32]
        L.kso_encode_residue.restype = C.c_uint8
        L.kso_encode_residue.argtypes = [C.c_uint8, C.c_int]
        L.kso_sketch_protein.restype = C.c_size_t
        L.kso_sketch_protein.argtypes = [C.c_char_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_int,
                                         C.c_uint64, u64p, u64p]
        L.kso_sketch_batch.restype = C.c_uint64
        L.kso_sketch_batch.argtypes = [u8p, u64p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int,
                                       C.c_uint64, u64p, u64p, u32p, C.c_int]
        L.kso_kmer_positions.restype = C.c_size_t
        L.kso_kmer_positions.argtypes = [C.c_char_p, C.c_size_t, C.c_uint32, C.c_int, C.c_uint64,
                                         u64p, C.c_size_t, C.c_int, u32p, u64p]
        L.kso_kmer_positions_batch.restype = C.c_uint64
        L.kso_kmer_positions_batch.argtypes = [u8p, u64p, C.c_uint32, C.c_uint32, C.c_int, C.c_uint64, u64p, u64p,
                                               C.c_int, C.c_int]
        L.kso_validate_and_resolve.restype = C.c_int
        L.kso_validate_and_resolve.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t,
                                               C.c_char_p, C.POINTER(C.c_size_t),
                                               C.POINTER(C.c_uint8), C.POINTER(C.c_size_t)]
        L.kso_manysearch.restype = C.c_uint64
        L.kso_manysearch.argtypes = [u64p, u64p, C.c_uint32, C.c_uint32, u64p, u64p, u32p,
                                     C.c_uint32, u32p, u32p, u32p, u64p, C.c_uint64, C.c_int]
        _lib = L
    return _lib


def _p(a: np.ndarray, ty):
    return a.ctypes.data_as(C.POINTER(ty))


def moltype_id(moltype: str) -> int:
    if moltype not in MOLTYPES:
        # text of src/rust/encoding.rs:22-25
        raise ValueError(f"Invalid moltype: {moltype}, only 'protein', 'hp', or 'dayhoff' are supported")
    return MOLTYPES[moltype]


def hash_murmur(data: bytes, seed: int = SEED) -> int:
    return int(lib().kso_hash_murmur(data, len(data), seed))


def max_hash(scaled: int) -> int:
    return int(lib().kso_max_hash(scaled))


def encode(seq: bytes, moltype: str) -> bytes:
    m = moltype_id(moltype)
    L = lib()
    return bytes(L.kso_encode_residue(b, m) for b in seq)


def sketch_protein(seq: bytes, ksize: int, scaled: int, moltype: str,
                   seed: int = SEED) -> Tuple[np.ndarray, np.ndarray]:
    n = max(len(seq) - ksize + 1, 0) + 1
    mins = np.zeros(n, dtype=np.uint64)
    abunds = np.zeros(n, dtype=np.uint64)
    c = lib().kso_sketch_protein(seq, len(seq), ksize, scaled, moltype_id(moltype), seed,
                                 _p(mins, C.c_uint64), _p(abunds, C.c_uint64))
    return mins[:c].copy(), abunds[:c].copy()


def pack(seqs: Sequence[bytes]) -> Tuple[np.ndarray, np.ndarray]:
    """Concatenate sequences into (residues u8, offsets u64[n+1])."""
    offs = np.zeros(len(seqs) + 1, dtype=np.uint64)
    if len(seqs):
        offs[1:] = np.cumsum([len(s) for s in seqs], dtype=np.uint64)
    res = np.frombuffer(b"".join(seqs), dtype=np.uint8).copy() if len(seqs) else np.zeros(0, np.uint8)
    return res, offs


def sketch_batch(residues: np.ndarray, offsets: np.ndarray, ksize: int, scaled: int, moltype: str,
                 seed: int = SEED, n_threads: int = 1) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """CSR sketches: (offsets u64[n+1], mins u64, abunds u32)."""
    residues = np.ascontiguousarray(residues, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    n = len(offsets) - 1
    cap = int(offsets[-1]) + 1 if n > 0 else 1
    out_off = np.zeros(n + 1, dtype=np.uint64)
    mins = np.zeros(cap, dtype=np.uint64)
    abunds = np.zeros(cap, dtype=np.uint32)
    if residues.size == 0:
        residues = np.zeros(1, np.uint8)
    tot = lib().kso_sketch_batch(_p(residues, C.c_uint8), _p(offsets, C.c_uint64), n, ksize, scaled,
                                 moltype_id(moltype), seed, _p(out_off, C.c_uint64),
                                 _p(mins, C.c_uint64), _p(abunds, C.c_uint32), n_threads)
    return out_off, mins[:tot].copy(), abunds[:tot].copy()


def kmer_positions(seq: bytes, ksize: int, moltype: str, mins: np.ndarray, seed: int = SEED,
                   faithful: bool = False) -> Tuple[np.ndarray, np.ndarray]:
    n = max(len(seq) - ksize + 1, 0) + 1
    starts = np.zeros(n, dtype=np.uint32)
    hashes = np.zeros(n, dtype=np.uint64)
    mins = np.ascontiguousarray(mins, dtype=np.uint64)
    mp = _p(mins if mins.size else np.zeros(1, np.uint64), C.c_uint64)
    c = lib().kso_kmer_positions(seq, len(seq), ksize, moltype_id(moltype), seed, mp, mins.size,
                                 1 if faithful else 0, _p(starts, C.c_uint32), _p(hashes, C.c_uint64))
    return starts[:c].copy(), hashes[:c].copy()


def kmer_positions_batch_count(residues: np.ndarray, offsets: np.ndarray, ksize: int, moltype: str, sk_offsets: np.ndarray,
                               sk_mins: np.ndarray, seed: int = SEED, faithful: bool = True, n_threads: int = 1) -> int:
    """process_kmers (src/rust/index.rs:749-786) over a whole batch, rows counted and dropped: bench.py's timing leg for
    the second pass create_protein_signature makes.  faithful = the reference's linear `contains` scan."""
    residues = np.ascontiguousarray(residues, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    sk_offsets = np.ascontiguousarray(sk_offsets, dtype=np.uint64)
    sk_mins = np.ascontiguousarray(sk_mins, dtype=np.uint64)
    if residues.size == 0:
        residues = np.zeros(1, np.uint8)
    if sk_mins.size == 0:
        sk_mins = np.zeros(1, np.uint64)
    return int(lib().kso_kmer_positions_batch(_p(residues, C.c_uint8), _p(offsets, C.c_uint64), len(offsets) - 1, ksize,
                                              moltype_id(moltype), seed, _p(sk_offsets, C.c_uint64),
                                              _p(sk_mins, C.c_uint64), 1 if faithful else 0, n_threads))


def kmer_infos(seq: bytes, ksize: int, moltype: str, mins: np.ndarray) -> Dict[int, dict]:
    """The reference's kmer_infos map (src/rust/kmer.rs:6-12): hash -> {encoded, originals->positions}."""
    starts, hashes = kmer_positions(seq, ksize, moltype, mins)
    out: Dict[int, dict] = {}
    for s, h in zip(starts.tolist(), hashes.tolist()):
        orig = seq[s:s + ksize].decode()
        info = out.setdefault(h, {"ksize": ksize, "hashval": h,
                                  "encoded_kmer": encode(seq[s:s + ksize], moltype).decode(),
                                  "original_kmer_to_position": {}})
        info["original_kmer_to_position"].setdefault(orig, []).append(s)
    return out


class InvalidAminoAcid(ValueError):
    def __init__(self, char: str, pos: int):
        # display format of src/rust/errors.rs:14
        super().__init__(f"Invalid amino acid '{char}' found at position {pos}")
        self.char, self.pos = char, pos


def validate_and_resolve(seq: bytes, choices: bytes = b"") -> bytes:
    out = C.create_string_buffer(len(seq) + 1)
    out_len, bad_pos, bad_char = C.c_size_t(0), C.c_size_t(0), C.c_uint8(0)
    rc = lib().kso_validate_and_resolve(seq, len(seq), choices, len(choices), out, C.byref(out_len),
                                        C.byref(bad_char), C.byref(bad_pos))
    if rc != 0:
        raise InvalidAminoAcid(chr(bad_char.value), bad_pos.value)
    return out.raw[:out_len.value]


def pseudo_md5(mins: np.ndarray) -> str:
    """Rust index key: format!("{:x}", wrapping sum) — src/rust/signature.rs:277-279."""
    s = int(np.sum(np.asarray(mins, dtype=np.uint64), dtype=np.uint64)) if len(mins) else 0
    return format(s, "x")


def sourmash_md5(mins: np.ndarray, protein_ksize: int) -> str:
    """Signature md5sum: MD5(ascii(3*k) || ascii(min) ...) — pinned by the md5 of all 75 golden sigs."""
    m = hashlib.md5()
    m.update(str(protein_ksize * 3).encode())
    for h in np.asarray(mins, dtype=np.uint64).tolist():
        m.update(str(h).encode())
    return m.hexdigest()


def manysearch(q_off, q_mins, t_off, t_mins, t_abund, q_begin: int = 0, q_end: Optional[int] = None,
               n_threads: int = 1):
    """COO hits (qid, tid, intersect, n_weighted) with intersect > 0, sorted by (qid, tid)."""
    q_off = np.ascontiguousarray(q_off, np.uint64); q_mins = np.ascontiguousarray(q_mins, np.uint64)
    t_off = np.ascontiguousarray(t_off, np.uint64); t_mins = np.ascontiguousarray(t_mins, np.uint64)
    t_abund = np.ascontiguousarray(t_abund, np.uint32)
    if q_end is None:
        q_end = len(q_off) - 1
    n_t = len(t_off) - 1
    z64 = np.zeros(1, np.uint64); z32 = np.zeros(1, np.uint32)
    qm = q_mins if q_mins.size else z64
    tm = t_mins if t_mins.size else z64
    ta = t_abund if t_abund.size else z32
    args = [_p(q_off, C.c_uint64), _p(qm, C.c_uint64), q_begin, q_end, _p(t_off, C.c_uint64),
            _p(tm, C.c_uint64), _p(ta, C.c_uint32), n_t]
    n = lib().kso_manysearch(*args, None, None, None, None, 0, n_threads)
    qid = np.zeros(max(n, 1), np.uint32); tid = np.zeros(max(n, 1), np.uint32)
    isect = np.zeros(max(n, 1), np.uint32); nw = np.zeros(max(n, 1), np.uint64)
    lib().kso_manysearch(*args, _p(qid, C.c_uint32), _p(tid, C.c_uint32), _p(isect, C.c_uint32),
                         _p(nw, C.c_uint64), n, n_threads)
    return qid[:n], tid[:n], isect[:n], nw[:n]


MANYSEARCH_COLUMNS = [
    "query_name", "query_md5", "match_name", "containment", "intersect_hashes", "ksize", "scaled",
    "moltype", "match_md5", "jaccard", "max_containment", "average_abund", "median_abund",
    "std_abund", "query_containment_ani", "match_containment_ani", "average_containment_ani",
    "max_containment_ani", "n_weighted_found", "total_weighted_hashes",
    "containment_target_in_query", "f_weighted_target_in_query",
]


def manysearch_row(q_name: str, q_mins: np.ndarray, t_name: str, t_mins: np.ndarray,
                   t_abund: np.ndarray, ksize: int, scaled: int, moltype: str) -> Optional[dict]:
    """One branchwater manysearch CSV row (22 columns of tests/test_search.py:33-39), or None if no overlap."""
    common, _, t_idx = np.intersect1d(q_mins, t_mins, assume_unique=True, return_indices=True)
    overlap = len(common)
    if overlap == 0:
        return None
    ab = np.sort(t_abund[t_idx].astype(np.float64))
    nq, nt = len(q_mins), len(t_mins)
    cont_q = overlap / nq
    cont_t = overlap / nt
    k3 = ksize * 3
    q_ani = cont_q ** (1.0 / k3)
    t_ani = cont_t ** (1.0 / k3)
    n_w = int(t_abund[t_idx].astype(np.uint64).sum())
    tot_w = int(t_abund.astype(np.uint64).sum())
    mean = float(ab.mean())
    n = len(ab)
    median = float(ab[n // 2]) if n % 2 else float((ab[n // 2 - 1] + ab[n // 2]) / 2.0)
    std = math.sqrt(float(((ab - mean) ** 2).sum()) / n)
    return {
        "query_name": q_name, "query_md5": sourmash_md5(q_mins, ksize), "match_name": t_name,
        "containment": cont_q, "intersect_hashes": overlap, "ksize": k3, "scaled": scaled,
        "moltype": moltype, "match_md5": sourmash_md5(t_mins, ksize),
        "jaccard": overlap / (nq + nt - overlap), "max_containment": max(cont_q, cont_t),
        "average_abund": mean, "median_abund": median, "std_abund": std,
        "query_containment_ani": q_ani, "match_containment_ani": t_ani,
        "average_containment_ani": (q_ani + t_ani) / 2.0, "max_containment_ani": max(q_ani, t_ani),
        "n_weighted_found": n_w, "total_weighted_hashes": tot_w,
        "containment_target_in_query": cont_t, "f_weighted_target_in_query": n_w / tot_w,
    }


def read_fasta(path: str) -> List[Tuple[str, bytes]]:
    """Minimal FASTA reader (plain or .gz) for fixtures: [(full header, sequence bytes)]."""
    import gzip
    op = gzip.open if path.endswith(".gz") else open
    recs: List[Tuple[str, List[bytes]]] = []
    with op(path, "rb") as f:
        for line in f.read().splitlines():
            if line.startswith(b">"):
                recs.append((line[1:].decode(), []))
            elif recs and line.strip():
                recs[-1][1].append(line.strip())
    return [(n, b"".join(p)) for n, p in recs]
