"""Python surface of the reference's Rust crate, bound to the C++ host mirror through include/kmerseek_host_c.h.

Mirrors the PyO3 module of src/rust/lib.rs:28-103 — ``sum_as_string``, ``PyProteinEncoding`` (Raw / Dayhoff / HP with
classmethods raw() / dayhoff() / hp()), ``PyProteomeIndex(ksize, scaled, moltype, db_path)`` whose failures surface as
``RuntimeError(str(e))`` — and adds ``ProteomeIndex`` / ``ProteomeIndexBuilder`` wrappers with the Rust method names
(src/rust/index.rs:104-1017, 2975-3061) so the reference's index tests translate line by line.
All computation happens in the HIP library; there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import enum
import json
from typing import Dict, List, Optional, Sequence, Tuple

from . import _lib

_ERR_CAP = 1024

HOST_SIGNATURES = {
    "ksh_index_new": (C.c_int, [C.c_char_p, C.c_uint32, C.c_uint32, C.c_char_p, C.c_int, C.c_int, C.c_int,
                                C.POINTER(C.c_void_p), C.c_char_p, C.c_size_t]),
    "ksh_index_build": (C.c_int, [C.c_char_p, C.c_int, C.c_uint32, C.c_int, C.c_uint32, C.c_char_p, C.c_int, C.c_int,
                                  C.c_int, C.POINTER(C.c_void_p), C.c_char_p, C.c_size_t]),
    "ksh_index_free": (None, [C.c_void_p]),
    "ksh_index_create_signature": (C.c_int, [C.c_void_p, C.c_char_p, C.c_char_p, C.c_int, C.POINTER(C.c_void_p),
                                             C.c_char_p, C.c_size_t]),
    "ksh_index_add_records": (C.c_int, [C.c_void_p, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.c_uint32, C.c_int,
                                        C.c_char_p, C.c_size_t]),
    "ksh_index_process_fasta": (C.c_int, [C.c_void_p, C.c_char_p, C.c_uint32, C.c_uint64, C.c_char_p, C.c_size_t]),
    "ksh_index_search": (C.c_int, [C.c_void_p, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.c_uint32, C.c_int,
                                   C.POINTER(C.c_void_p), C.c_char_p, C.c_size_t]),
    "ksh_index_search_fasta": (C.c_int, [C.c_void_p, C.c_char_p, C.c_uint64, C.POINTER(C.c_void_p), C.c_char_p, C.c_size_t]),
    "ksh_sourmash_md5": (C.c_int, [C.POINTER(C.c_uint64), C.c_uint64, C.c_uint32, C.c_char_p]),
    "ksh_index_signature_count": (C.c_uint64, [C.c_void_p]),
    "ksh_index_combined_minhash_size": (C.c_uint64, [C.c_void_p]),
    "ksh_index_ksize": (C.c_uint32, [C.c_void_p]),
    "ksh_index_scaled": (C.c_uint32, [C.c_void_p]),
    "ksh_index_store_raw_sequences": (C.c_int, [C.c_void_p]),
    "ksh_index_moltype": (C.c_int, [C.c_void_p, C.c_char_p, C.c_size_t]),
    "ksh_index_path": (C.c_int, [C.c_void_p, C.c_char_p, C.c_size_t]),
    "ksh_index_generate_filename": (C.c_int, [C.c_void_p, C.c_char_p, C.c_char_p, C.c_size_t]),
    "ksh_index_dump_json": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]),
    "ksh_index_is_equivalent_to": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.c_char_p, C.c_size_t]),
    "ksh_index_save_state": (C.c_int, [C.c_void_p, C.c_char_p, C.c_size_t]),
    "ksh_index_load": (C.c_int, [C.c_char_p, C.c_int, C.POINTER(C.c_void_p), C.c_char_p, C.c_size_t]),
    "ksh_string_free": (None, [C.c_void_p]),
    "ksh_sketch_fasta": (C.c_int, [C.c_char_p, C.c_uint32, C.c_uint32, C.c_char_p, C.c_int, C.c_int, C.c_uint64, C.c_int,
                                   C.POINTER(C.c_void_p), C.c_char_p, C.c_size_t]),
    "ksh_fs_n_records": (C.c_uint64, [C.c_void_p]),
    "ksh_fs_n_hashes": (C.c_uint64, [C.c_void_p]),
    "ksh_fs_offsets": (C.POINTER(C.c_uint64), [C.c_void_p]),
    "ksh_fs_hashes": (C.POINTER(C.c_uint64), [C.c_void_p]),
    "ksh_fs_abunds": (C.POINTER(C.c_uint32), [C.c_void_p]),
    "ksh_fs_names": (C.c_void_p, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "ksh_fs_stats": (None, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                            C.POINTER(C.c_double)]),
    "ksh_fs_free": (None, [C.c_void_p]),
    "ksh_input_decompress": (C.c_int, [C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.c_char_p, C.c_uint32,
                                       C.c_char_p, C.c_uint32]),
    "ksh_input_free": (None, [C.c_void_p]),
}

_bound = None


def _host():
    global _bound
    if _bound is None:
        L = _lib.load()
        for name, (res, args) in HOST_SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _bound = L
    return _bound


class IndexError_(RuntimeError):
    """kmerseek::IndexError (src/rust/errors.rs); str() is the reference's Display text."""

    KINDS = ["Database", "InvalidMoltype", "InvalidAminoAcid", "InvalidKsize", "NoSavedState", "Io", "Utf8",
             "FastaParsing", "BuilderError", "SourmashError", "ParseError", "ValidationError", "Gpu"]

    def __init__(self, code: int, message: str):
        super().__init__(message)
        self.code = code
        self.kind = self.KINDS[code - 1] if 1 <= code <= len(self.KINDS) else "Other"


def _call(fn, *args):
    err = C.create_string_buffer(_ERR_CAP)
    rc = fn(*args, err, _ERR_CAP)
    if rc != 0:
        raise IndexError_(rc, err.value.decode(errors="replace"))


def _take_string(L, p: C.c_void_p) -> str:
    try:
        return C.string_at(p).decode()
    finally:
        L.ksh_string_free(p)


def decompress(path) -> tuple:
    """-> (bytes, format) of a plain / gzip / zstd file through the ingest's input layer (ks_input.cpp): the
    auto-detection needletail gives the reference (src/rust/index.rs:907-961).  Raises IndexError_ (ParseError) on a
    truncated or corrupt archive.  Needs no GPU."""
    L = _host()
    data, n = C.c_void_p(), C.c_uint64()
    fmt, err = C.create_string_buffer(16), C.create_string_buffer(_ERR_CAP)
    rc = L.ksh_input_decompress(str(path).encode(), C.byref(data), C.byref(n), fmt, 16, err, _ERR_CAP)
    if rc != 0:
        raise IndexError_(rc, err.value.decode(errors="replace"))
    try:
        return C.string_at(data, n.value), fmt.value.decode()
    finally:
        L.ksh_input_free(data)


def sourmash_md5(mins, protein_ksize: int) -> str:
    """md5sum of a sketch as sourmash writes it (native: the search rows' query_md5 / match_md5 come from the same routine)."""
    import numpy as np
    a = np.ascontiguousarray(mins, dtype=np.uint64)
    out = C.create_string_buffer(33)
    if _host().ksh_sourmash_md5(a.ctypes.data_as(C.POINTER(C.c_uint64)), len(a), int(protein_ksize), out) != 0:
        raise IndexError_(1000, "ksh_sourmash_md5 failed")
    return out.value.decode()


def sum_as_string(a: int, b: int) -> str:
    """src/rust/lib.rs:69-72"""
    return str(a + b)


class PyProteinEncoding(enum.Enum):
    """src/rust/lib.rs:29-65"""
    Raw = "protein"
    Dayhoff = "dayhoff"
    HP = "hp"

    @classmethod
    def raw(cls):
        return cls.Raw

    @classmethod
    def dayhoff(cls):
        return cls.Dayhoff

    @classmethod
    def hp(cls):
        return cls.HP

    def __str__(self):
        return self.value


class ProteomeIndex:
    """kmerseek::ProteomeIndex — the C++ mirror of src/rust/index.rs:104-1017, method for method."""

    def __init__(self, path: str, ksize: int, scaled: int, moltype: str, store_raw_sequences: bool = False,
                 device: int = 0, _handle=None):
        self._L = _host()
        if _handle is not None:
            self._h = _handle
            return
        h = C.c_void_p()
        _call(self._L.ksh_index_new, str(path).encode(), ksize, scaled, str(moltype).encode(),
              1 if store_raw_sequences else 0, device, 0, C.byref(h))
        self._h = h

    # -- constructors ------------------------------------------------------------------------------
    @classmethod
    def new(cls, path, ksize, scaled, moltype, store_raw_sequences=False, device=0):
        return cls(path, ksize, scaled, moltype, store_raw_sequences, device)

    @classmethod
    def new_with_auto_filename(cls, base_path, ksize, scaled, moltype, store_raw_sequences=False, device=0):
        L = _host()
        h = C.c_void_p()
        _call(L.ksh_index_new, str(base_path).encode(), ksize, scaled, str(moltype).encode(),
              1 if store_raw_sequences else 0, device, 1, C.byref(h))
        return cls(None, 0, 0, "", _handle=h)

    @classmethod
    def builder(cls) -> "ProteomeIndexBuilder":
        return ProteomeIndexBuilder()

    @classmethod
    def load(cls, path, device=0):
        L = _host()
        h = C.c_void_p()
        _call(L.ksh_index_load, str(path).encode(), device, C.byref(h))
        return cls(None, 0, 0, "", _handle=h)

    def close(self):
        if getattr(self, "_h", None):
            self._L.ksh_index_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- the path -----------------------------------------------------------------------------------
    def create_protein_signature(self, sequence: str, name: str, store: bool = False) -> dict:
        out = C.c_void_p()
        _call(self._L.ksh_index_create_signature, self._h, sequence.encode(), name.encode(), 1 if store else 0,
              C.byref(out))
        sig = json.loads(_take_string(self._L, out))
        sig["kmer_infos"] = {int(k): v for k, v in sig["kmer_infos"].items()}
        return sig

    def add_records(self, records: Sequence[Tuple[str, str]], upper: bool = False) -> None:
        """create_protein_signatures + store_signatures_batch for (sequence, name) records."""
        n = len(records)
        seqs = (C.c_char_p * n)(*[r[0].encode() for r in records])
        names = (C.c_char_p * n)(*[r[1].encode() for r in records])
        _call(self._L.ksh_index_add_records, self._h, seqs, names, n, 1 if upper else 0)

    def process_fasta(self, fasta_path, progress_interval: int = 0, batch_size: int = 1000) -> None:
        _call(self._L.ksh_index_process_fasta, self._h, str(fasta_path).encode(), progress_interval, batch_size)

    def save_state(self) -> None:
        _call(self._L.ksh_index_save_state, self._h)

    # -- search (added: SURVEY 8(b); the reference's crate has none — rows of branchwater manysearch,
    #    src/python/kmerseek/search.py:125-141, for the signatures this index holds) -------------------------------------
    def search(self, records: Sequence[Tuple[str, str]], upper: bool = False) -> List[dict]:
        """(sequence, name) query records against the stored signatures: one dict per (query, match) pair that shares a
        hash, keys = SEARCH_COLUMNS.  Queries are validated / resolved like create_protein_signature's input."""
        n = len(records)
        seqs = (C.c_char_p * n)(*[r[0].encode() for r in records])
        names = (C.c_char_p * n)(*[r[1].encode() for r in records])
        out = C.c_void_p()
        _call(self._L.ksh_index_search, self._h, seqs, names, n, 1 if upper else 0, C.byref(out))
        return json.loads(_take_string(self._L, out))

    def search_fasta(self, fasta_path, batch_size: int = 100000) -> List[dict]:
        """Every record of a FASTA file (plain / gzip / zstd / bzip2 / xz) as queries (upper-cased, as the FASTA path does)."""
        out = C.c_void_p()
        _call(self._L.ksh_index_search_fasta, self._h, str(fasta_path).encode(), batch_size, C.byref(out))
        return json.loads(_take_string(self._L, out))

    # -- getters ------------------------------------------------------------------------------------
    def signature_count(self) -> int:
        return int(self._L.ksh_index_signature_count(self._h))

    def combined_minhash_size(self) -> int:
        return int(self._L.ksh_index_combined_minhash_size(self._h))

    def ksize(self) -> int:
        return int(self._L.ksh_index_ksize(self._h))

    def scaled(self) -> int:
        return int(self._L.ksh_index_scaled(self._h))

    def store_raw_sequences(self) -> bool:
        return bool(self._L.ksh_index_store_raw_sequences(self._h))

    def _str(self, fn, *args) -> str:
        buf = C.create_string_buffer(4096)
        fn(self._h, *args, buf, 4096)
        return buf.value.decode()

    def moltype(self) -> str:
        return self._str(self._L.ksh_index_moltype)

    def path(self) -> str:
        return self._str(self._L.ksh_index_path)

    def generate_filename(self, base_name: str) -> str:
        return self._str(self._L.ksh_index_generate_filename, base_name.encode())

    def dump(self, with_kmers: bool = True) -> dict:
        out = C.c_void_p()
        if self._L.ksh_index_dump_json(self._h, 1 if with_kmers else 0, C.byref(out)) != 0:
            raise IndexError_(1000, "dump failed")
        d = json.loads(_take_string(self._L, out))
        if with_kmers:
            for s in d["signatures"].values():
                s["kmer_infos"] = {int(k): v for k, v in s["kmer_infos"].items()}
        return d

    def get_signatures(self) -> Dict[str, dict]:
        return self.dump(True)["signatures"]

    def is_equivalent_to(self, other: "ProteomeIndex") -> bool:
        eq = C.c_int(0)
        _call(self._L.ksh_index_is_equivalent_to, self._h, other._h, C.byref(eq))
        return bool(eq.value)


class ProteomeIndexBuilder:
    """src/rust/index.rs:2975-3061"""

    def __init__(self):
        self._path = self._moltype = None
        self._ksize = self._scaled = None
        self._raw = False
        self._device = 0

    def path(self, p):
        self._path = str(p); return self

    def ksize(self, k):
        self._ksize = int(k); return self

    def scaled(self, s):
        self._scaled = int(s); return self

    def moltype(self, m):
        self._moltype = str(m); return self

    def store_raw_sequences(self, b):
        self._raw = bool(b); return self

    def device(self, d):
        self._device = int(d); return self

    def _build(self, auto: int) -> ProteomeIndex:
        L = _host()
        h = C.c_void_p()
        _call(L.ksh_index_build, self._path.encode() if self._path is not None else None,
              1 if self._ksize is not None else 0, self._ksize or 0, 1 if self._scaled is not None else 0,
              self._scaled or 0, self._moltype.encode() if self._moltype is not None else None,
              1 if self._raw else 0, auto, self._device, C.byref(h))
        return ProteomeIndex(None, 0, 0, "", _handle=h)

    def build(self) -> ProteomeIndex:
        return self._build(0)

    def build_with_auto_filename(self) -> ProteomeIndex:
        return self._build(1)


# the 22 columns of a search row, in the reference's CSV order (tests/test_search.py:33-39)
SEARCH_COLUMNS = [
    "query_name", "query_md5", "match_name", "containment", "intersect_hashes", "ksize", "scaled", "moltype",
    "match_md5", "jaccard", "max_containment", "average_abund", "median_abund", "std_abund",
    "query_containment_ani", "match_containment_ani", "average_containment_ani", "max_containment_ani",
    "n_weighted_found", "total_weighted_hashes", "containment_target_in_query", "f_weighted_target_in_query",
]


def write_search_csv(rows: List[dict], path) -> int:
    """Rows of ProteomeIndex.search as the reference's manysearch CSV (same header, same column order)."""
    import csv
    with open(path, "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=SEARCH_COLUMNS, lineterminator="\n")
        w.writeheader()
        w.writerows(rows)
    return len(rows)


class PyProteomeIndex:
    """src/rust/lib.rs:74-92: ``PyProteomeIndex(ksize, scaled, moltype, db_path)``, any failure -> RuntimeError(str(e)).
    The reference's class stops at the constructor (lib.rs:76-79: the index it holds is "Not currently used"); the
    ``sketch_*`` / ``search_*`` methods are the ones SURVEY 8(b) asks to add: sketch records into the index
    (ProteomeIndex::process_fasta / create_protein_signature + store_signatures, index.rs:907-961, 719-830) and search
    records against it (rows of src/python/kmerseek/search.py:125-141)."""

    def __init__(self, ksize: int, scaled: int, moltype, db_path: str):
        self.index = ProteomeIndex(db_path, ksize, scaled, str(moltype), False)

    # -- sketch_*: records -> signatures stored in the index ------------------------------------------------------
    def sketch_fasta(self, fasta_path, batch_size: int = 100000) -> int:
        """process_fasta (index.rs:907-961): every record sketched and stored; returns the number of stored signatures."""
        self.index.process_fasta(fasta_path, 0, batch_size)
        return self.index.signature_count()

    def sketch_sequences(self, records: Sequence[Tuple[str, str]]) -> int:
        """(sequence, name) records sketched (one GPU batch) and stored; returns the number of stored signatures."""
        self.index.add_records(list(records), upper=False)
        return self.index.signature_count()

    def signature_count(self) -> int:
        return self.index.signature_count()

    # -- search_*: records -> manysearch rows against the stored signatures --------------------------------------------
    def search_fasta(self, fasta_path, output=None, batch_size: int = 100000) -> List[dict]:
        """Every record of a FASTA file searched against the index; with `output` the rows are also written as the
        reference's 22-column CSV."""
        rows = self.index.search_fasta(fasta_path, batch_size)
        if output is not None:
            write_search_csv(rows, output)
        return rows

    def search_sequences(self, records: Sequence[Tuple[str, str]], output=None) -> List[dict]:
        rows = self.index.search(list(records), upper=False)
        if output is not None:
            write_search_csv(rows, output)
        return rows


def sketch_fasta(path, ksize: int, scaled: int, moltype: str, validate: bool = False, device: int = 0,
                 batch_residues: int = 0, pipeline: bool = True):
    """Pipelined FASTA ingest (kmerseek_amd/csrc/ks_ingest.cpp): (names, offsets u64[n+1], hashes u64[], abunds u32[], stats)
    for every record of a plain / gzip FASTA file.  validate=False hashes the raw record bytes as manysketch does
    (src/python/kmerseek/sketch.py:28-40); validate=True is the Rust index path (upper-case + validate_and_resolve)."""
    import numpy as np
    L = _host()
    h = C.c_void_p()
    _call(L.ksh_sketch_fasta, str(path).encode(), ksize, scaled, str(moltype).encode(), 1 if validate else 0, device,
          int(batch_residues), 1 if pipeline else 0, C.byref(h))
    try:
        n = int(L.ksh_fs_n_records(h))
        nh = int(L.ksh_fs_n_hashes(h))
        offsets = np.ctypeslib.as_array(L.ksh_fs_offsets(h), shape=(n + 1,)).copy()
        hashes = np.ctypeslib.as_array(L.ksh_fs_hashes(h), shape=(nh,)).copy() if nh else np.zeros(0, np.uint64)
        abunds = np.ctypeslib.as_array(L.ksh_fs_abunds(h), shape=(nh,)).copy() if nh else np.zeros(0, np.uint32)
        ln = C.c_uint64(0)
        p = L.ksh_fs_names(h, C.byref(ln))
        blob = C.string_at(p, ln.value).decode() if n else ""
        names = blob.split("\n") if n else []
        nr, nw, nb = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        sec = (C.c_double * 6)()
        L.ksh_fs_stats(h, C.byref(nr), C.byref(nw), C.byref(nb), sec)
        stats = {"residues": nr.value, "windows": nw.value, "batches": nb.value, "wall_s": sec[0], "read_s": sec[1],
                 "pack_s": sec[2], "h2d_s": sec[3], "device_s": sec[4], "collect_s": sec[5]}
    finally:
        L.ksh_fs_free(h)
    return names, offsets, hashes, abunds, stats
