"""Wire formats around the hot path, so the reference's Python callers can consume GPU results unchanged.

* ``sketch()``        — drop-in for src/python/kmerseek/sketch.py:28-40 (branchwater ``do_manysketch(singleton=True)``):
                        writes ``<fasta>.manysketch.csv`` and a sourmash-compatible ``<fasta>.{moltype}.k{k}.scaled{s}.sig.zip``
                        (gzipped JSON signatures + SOURMASH-MANIFEST.csv).
* ``do_manysearch()`` — drop-in for src/python/kmerseek/search.py:125-141 (branchwater ``do_manysearch`` with threshold 0,
                        abundance on, output_all off): writes the 22-column CSV pinned by tests/test_search.py:33-39.
Hashing, sorting, joining and counting all run in the HIP library; this module only formats.  The ratio columns are
f64 arithmetic on the integer results (formulas: SURVEY.md §8(a) row a10).
"""
from __future__ import annotations

import csv
import gzip
import hashlib
import io
import json
import math
import os
import zipfile
from typing import List, Optional, Sequence, Tuple

import numpy as np

from .engine import Context, max_hash

MANYSEARCH_COLUMNS = [
    "query_name", "query_md5", "match_name", "containment", "intersect_hashes", "ksize", "scaled", "moltype",
    "match_md5", "jaccard", "max_containment", "average_abund", "median_abund", "std_abund",
    "query_containment_ani", "match_containment_ani", "average_containment_ani", "max_containment_ani",
    "n_weighted_found", "total_weighted_hashes", "containment_target_in_query", "f_weighted_target_in_query",
]


def read_fasta(path: str) -> List[Tuple[str, bytes]]:
    """[(full header, sequence bytes)] from a plain or gzipped FASTA (manysketch hashes records as given)."""
    op = gzip.open if str(path).endswith(".gz") else open
    recs: List[Tuple[str, List[bytes]]] = []
    with op(path, "rb") as f:
        for line in f.read().splitlines():
            if line.startswith(b">"):
                recs.append((line[1:].decode(), []))
            elif recs and line.strip():
                recs[-1][1].append(line.strip())
    return [(n, b"".join(p)) for n, p in recs]


def sourmash_md5(mins: np.ndarray, protein_ksize: int) -> str:
    """md5sum field of a sourmash signature: MD5(ascii(3k) || ascii(min) ...)."""
    m = hashlib.md5()
    m.update(str(protein_ksize * 3).encode())
    m.update("".join(map(str, np.asarray(mins, dtype=np.uint64).tolist())).encode())
    return m.hexdigest()


# ---------------------------------------------------------------------------------------------------------
# .sig.zip
# ---------------------------------------------------------------------------------------------------------
def write_sig_zip(path: str, names: Sequence[str], offsets: np.ndarray, mins: np.ndarray, abunds: np.ndarray,
                  ksize: int, scaled: int, moltype: str, filename: str) -> None:
    """One gzipped JSON signature per record under signatures/<md5>.sig.gz + SOURMASH-MANIFEST.csv (stored, not deflated)."""
    mh = max_hash(scaled)
    rows = []
    with zipfile.ZipFile(path, "w", compression=zipfile.ZIP_STORED) as z:
        for i, name in enumerate(names):
            m = mins[int(offsets[i]):int(offsets[i + 1])]
            a = abunds[int(offsets[i]):int(offsets[i + 1])]
            md5 = sourmash_md5(m, ksize)
            doc = [{"class": "sourmash_signature", "email": "", "hash_function": "0.murmur64", "filename": filename,
                    "name": name, "license": "CC0",
                    "signatures": [{"num": 0, "ksize": 3 * ksize, "seed": 42, "max_hash": mh, "mins": m.tolist(),
                                    "md5sum": md5, "abundances": a.tolist(), "molecule": moltype}],
                    "version": 0.4}]
            loc = f"signatures/{md5}.sig.gz"
            buf = io.BytesIO()
            with gzip.GzipFile(fileobj=buf, mode="wb", mtime=0) as g:
                g.write(json.dumps(doc, separators=(",", ":")).encode())
            z.writestr(loc, buf.getvalue())
            rows.append([loc, md5, md5[:8], ksize, moltype, 0, scaled, len(m), 1, name, filename])
        out = io.StringIO()
        out.write("# SOURMASH-MANIFEST-VERSION: 1.0\n")
        w = csv.writer(out, lineterminator="\n")
        w.writerow(["internal_location", "md5", "md5short", "ksize", "moltype", "num", "scaled", "n_hashes",
                    "with_abundance", "name", "filename"])
        w.writerows(rows)
        z.writestr("SOURMASH-MANIFEST.csv", out.getvalue())


def read_sig_zip(path: str):
    """-> (names, offsets u64[n+1], mins u64, abunds u32, ksize (protein), scaled, moltype), in manifest order."""
    z = zipfile.ZipFile(path)
    text = "\n".join(l for l in z.read("SOURMASH-MANIFEST.csv").decode().splitlines() if not l.startswith("#"))
    names, mins, abunds, offs = [], [], [], [0]
    ksize = scaled = None
    moltype = None
    for row in csv.DictReader(io.StringIO(text)):
        doc = json.loads(gzip.decompress(z.read(row["internal_location"])))
        sig = doc[0]["signatures"][0]
        names.append(doc[0]["name"])
        mins.extend(sig["mins"])
        abunds.extend(sig.get("abundances") or [1] * len(sig["mins"]))
        offs.append(len(mins))
        have = (sig["ksize"] // 3, int(row["scaled"]), sig["molecule"])
        if ksize is not None and have != (ksize, scaled, moltype):  # every row, not only the last one
            raise ValueError(f"{path}: sketch {doc[0]['name']!r} was made with {have}, earlier ones with "
                             f"{(ksize, scaled, moltype)}: one search takes one set of parameters")
        ksize, scaled, moltype = have
        mh = max_hash(scaled)
        if any(h == 0 or h > mh for h in sig["mins"]) or (sig.get("max_hash") not in (None, 0, mh)):
            raise ValueError(f"{path}: sketch {doc[0]['name']!r} holds hashes outside (0, max_hash(scaled={scaled})]")
    return (names, np.array(offs, np.uint64), np.array(mins, np.uint64), np.array(abunds, np.uint32), ksize, scaled,
            moltype)


# ---------------------------------------------------------------------------------------------------------
# sketch()  — src/python/kmerseek/sketch.py
# ---------------------------------------------------------------------------------------------------------
def make_sketch_kws(moltype: str, ksize: int, scaled: int):
    return dict(ksize=ksize, moltype=moltype, scaled=scaled)


def _make_manysketch_csv(fasta: str) -> str:
    path = f"{fasta}.manysketch.csv"
    with open(path, "w") as f:
        f.write("name,genome_filename,protein_filename\n")
        f.write(f"{os.path.basename(fasta)},,{fasta}\n")
    return path


def _make_sigfile(fasta: str, moltype: str, ksize: int, scaled: int) -> str:
    return f"{fasta}.{moltype}.k{ksize}.scaled{scaled}.sig.zip"


def sketch(fasta: str, moltype: str, ksize: int, scaled: int, ctx: Optional[Context] = None) -> str:
    """sketch() of src/python/kmerseek/sketch.py:28-40.  The records go through the native pipelined ingest
    (csrc/ks_ingest.cpp: parse, pinned staging, H2D and the sketch kernels overlap); `ctx` only picks the device."""
    from . import host
    sigfile = _make_sigfile(fasta, moltype, ksize, scaled)
    _make_manysketch_csv(fasta)
    names, o, m, a, _ = host.sketch_fasta(fasta, ksize, scaled, moltype, validate=False,
                                          device=ctx.device if ctx is not None and hasattr(ctx, "device") else 0)
    write_sig_zip(sigfile, names, o, m, a, ksize, scaled, moltype, os.path.abspath(fasta))
    return sigfile


# ---------------------------------------------------------------------------------------------------------
# do_manysearch()  — src/python/kmerseek/search.py:125-141
# ---------------------------------------------------------------------------------------------------------
def manysearch_rows(q_names, q_off, q_mins, t_names, t_off, t_mins, t_abund, hits, ksize: int, scaled: int,
                    moltype: str) -> List[dict]:
    """The 22 CSV columns for every COO hit (qid, tid, intersect, n_weighted)."""
    qid, tid, isect, nw = hits
    k3 = 3 * ksize
    t_tot = np.add.reduceat(t_abund.astype(np.uint64), t_off[:-1].astype(np.int64)) if len(t_abund) else np.zeros(len(t_off) - 1, np.uint64)
    t_tot = np.where((t_off[1:] - t_off[:-1]) > 0, t_tot, 0)
    md5_q, md5_t = {}, {}
    rows = []
    for q, t, i, w in zip(qid.tolist(), tid.tolist(), isect.tolist(), nw.tolist()):
        qm = q_mins[int(q_off[q]):int(q_off[q + 1])]
        tm = t_mins[int(t_off[t]):int(t_off[t + 1])]
        ta = t_abund[int(t_off[t]):int(t_off[t + 1])]
        # abundance statistics need the per-hash target abundances of the shared hashes (not just their sum)
        shared = ta[np.isin(tm, qm, assume_unique=True)].astype(np.float64)
        assert len(shared) == i
        shared.sort()
        n = len(shared)
        mean = float(shared.mean())
        median = float(shared[n // 2]) if n % 2 else float((shared[n // 2 - 1] + shared[n // 2]) / 2.0)
        std = math.sqrt(float(((shared - mean) ** 2).sum()) / n)
        nq, nt = len(qm), len(tm)
        cq, ct = i / nq, i / nt
        q_ani, t_ani = cq ** (1.0 / k3), ct ** (1.0 / k3)
        if q not in md5_q:
            md5_q[q] = sourmash_md5(qm, ksize)
        if t not in md5_t:
            md5_t[t] = sourmash_md5(tm, ksize)
        tot_w = int(t_tot[t])
        rows.append({
            "query_name": q_names[q], "query_md5": md5_q[q], "match_name": t_names[t], "containment": cq,
            "intersect_hashes": i, "ksize": k3, "scaled": scaled, "moltype": moltype, "match_md5": md5_t[t],
            "jaccard": i / (nq + nt - i), "max_containment": max(cq, ct), "average_abund": mean,
            "median_abund": median, "std_abund": std, "query_containment_ani": q_ani, "match_containment_ani": t_ani,
            "average_containment_ani": (q_ani + t_ani) / 2.0, "max_containment_ani": max(q_ani, t_ani),
            "n_weighted_found": w, "total_weighted_hashes": tot_w, "containment_target_in_query": ct,
            "f_weighted_target_in_query": w / tot_w,
        })
    return rows


def do_manysearch(query_sig: str, target_sig: str, output: str, ksize: int, scaled: int, moltype: str,
                  ctx: Optional[Context] = None) -> int:
    """Search every sketch of query_sig (.sig.zip) against every sketch of target_sig; CSV rows for pairs that share
    at least one hash.  Returns the number of rows written."""
    own = ctx is None
    ctx = ctx or Context(0)
    try:
        qn, qo, qm, qa, qk, qs, qmol = read_sig_zip(query_sig)
        tn, to, tm, ta, tk, ts, tmol = read_sig_zip(target_sig)
        for have in ((qk, qs, qmol), (tk, ts, tmol)):
            if have != (ksize, scaled, moltype):
                raise ValueError(f"sketch parameters {have} do not match the requested {(ksize, scaled, moltype)}")
        Q = ctx.sketches_from_host(qo, qm, qa, ksize, scaled, moltype)
        T = ctx.sketches_from_host(to, tm, ta, ksize, scaled, moltype)
        hits = ctx.search(ctx.index_build(T), Q).to_host()
        rows = manysearch_rows(qn, qo, qm, tn, to, tm, ta, hits, ksize, scaled, moltype)
        with open(output, "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=MANYSEARCH_COLUMNS, lineterminator="\n")
            w.writeheader()
            w.writerows(rows)
        return len(rows)
    finally:
        if own:
            ctx.close()


# ---------------------------------------------------------------------------------------------------------
# k-mer tables and alignment stitching  — src/python/kmerseek/sig2kmer.py:186-219, search.py:37-121,195-276
# ---------------------------------------------------------------------------------------------------------
_DAYHOFF = {**{c: "a" for c in "C"}, **{c: "b" for c in "AGPST"}, **{c: "c" for c in "DENQ"}, **{c: "d" for c in "HKR"},
            **{c: "e" for c in "ILMV"}, **{c: "f" for c in "FWY"}}
_HP = {**{c: "h" for c in "AFGILMPVWY"}, **{c: "p" for c in "NCSTDERHKQ"}}


def encode_kmer(kmer: str, moltype: str) -> str:
    """The `encoded` column: residue-wise re-encode (sig2kmer.py:40-58 / src/rust/encoding.rs:67-105)."""
    if moltype in ("protein", "raw"):
        return kmer
    table = _DAYHOFF if moltype == "dayhoff" else _HP
    return "".join(table.get(c, "X") for c in kmer)


def extract_kmers(ctx: Context, records: Sequence[Tuple[str, bytes]], ksize: int, scaled: int, moltype: str,
                  sequence_file: str = "") -> List[dict]:
    """Rows of the reference's k-mer table (sequence_file, sequence_name, kmer, hashval, encoded, start): one per window
    whose hash is in the sequence's sketch.  Positions and hashes come from the GPU (ks_kmer_positions)."""
    from .engine import pack
    res, offs = pack([s for _, s in records])
    seq_i, start, hashes = ctx.kmer_positions(res, offs, ksize, scaled, moltype)
    rows = []
    for s, st, h in zip(seq_i.tolist(), start.tolist(), hashes.tolist()):
        name, seq = records[s]
        kmer = seq[st:st + ksize].decode().upper()
        rows.append({"sequence_file": sequence_file, "sequence_name": name, "kmer": kmer, "hashval": h,
                     "encoded": encode_kmer(kmer, moltype), "start": st})
    return rows


def write_kmers_parquet(rows: List[dict], path: str) -> None:
    import pyarrow as pa
    import pyarrow.parquet as pq
    cols = ["sequence_file", "sequence_name", "kmer", "hashval", "encoded", "start"]
    tab = pa.table({
        "sequence_file": pa.array([r["sequence_file"] for r in rows], pa.string()),
        "sequence_name": pa.array([r["sequence_name"] for r in rows], pa.string()),
        "kmer": pa.array([r["kmer"] for r in rows], pa.string()),
        "hashval": pa.array([r["hashval"] for r in rows], pa.uint64()),
        "encoded": pa.array([r["encoded"] for r in rows], pa.string()),
        "start": pa.array([r["start"] for r in rows], pa.uint32()),
    })
    assert tab.column_names == cols
    pq.write_table(tab, path)


def single_stitch_together_kmers(kmers: Sequence[str], i_kmers: Sequence[int]) -> str:
    """search.py:37-60, quirks included (a zero step re-appends the whole k-mer: kmer[-0:])."""
    stitched = ""
    prev = 0
    for i, (pos, kmer) in enumerate(zip(i_kmers, kmers)):
        if i == 0:
            stitched = kmer
        else:
            step = pos - prev
            stitched += kmer[-step:]
        prev = pos
    return stitched


def stitch_hits(query_kmers: List[dict], target_kmers: List[dict], hit_pairs: Sequence[Tuple[str, str]]) -> List[dict]:
    """search.py:195-240: join query and target k-mers on (encoded, hashval), keep the pairs that are search hits,
    group by match_name, stitch overlapping k-mers into one aligned region per match; rows sorted by (query_start, query_end)."""
    by_key = {}
    for r in target_kmers:
        by_key.setdefault((r["encoded"], r["hashval"]), []).append(r)
    hitset = set(hit_pairs)
    groups = {}
    for q in query_kmers:
        for t in by_key.get((q["encoded"], q["hashval"]), ()):
            if (q["sequence_name"], t["sequence_name"]) in hitset:
                groups.setdefault(t["sequence_name"], []).append(
                    {"query_name": q["sequence_name"], "match_name": t["sequence_name"], "kmer_query": q["kmer"],
                     "kmer_match": t["kmer"], "encoded": q["encoded"], "start_query": q["start"], "start_match": t["start"]})
    out = []
    for match_name, rows in groups.items():
        rows.sort(key=lambda r: r["start_query"])
        # search.py:79-81: the query k-mers are stitched with the MATCH positions (as the reference does)
        query = single_stitch_together_kmers([r["kmer_query"] for r in rows], [r["start_match"] for r in rows])
        alpha = single_stitch_together_kmers([r["encoded"] for r in rows], [r["start_query"] for r in rows])
        match = single_stitch_together_kmers([r["kmer_match"] for r in rows], [r["start_match"] for r in rows])
        assert len(query) == len(alpha) == len(match)
        length = len(query)
        ms = min(r["start_match"] for r in rows)
        qs = min(r["start_query"] for r in rows)
        qn = rows[0]["query_name"]
        out.append({"match_name": match_name, "query_name": qn, "query_start": qs, "query_end": qs + length,
                    "query": query, "match_start": ms, "match_end": ms + length, "match": match, "encoded": alpha,
                    "length": length,
                    "to_print": f"\n---\nQuery Name: {qn}\nMatch Name: {match_name}\nquery: {query} ({qs}-{qs + length})\n"
                                f"alpha: {alpha}\nmatch: {match} ({ms}-{ms + length})"})
    out.sort(key=lambda r: (r["query_start"], r["query_end"]))
    return out


def search_extract_kmers(query_fasta: str, target_fasta: str, ksize: int, scaled: int, moltype: str,
                         ctx: Optional[Context] = None) -> List[dict]:
    """`kmerseek search --extract-kmers QUERY TARGET` end to end: sketch both, search, k-mer tables, stitch."""
    own = ctx is None
    ctx = ctx or Context(0)
    try:
        from .engine import pack
        q_recs, t_recs = read_fasta(query_fasta), read_fasta(target_fasta)
        Q = ctx.sketch_batch(*pack([s for _, s in q_recs]), ksize, scaled, moltype)
        T = ctx.sketch_batch(*pack([s for _, s in t_recs]), ksize, scaled, moltype)
        qid, tid, _, _ = ctx.search(ctx.index_build(T), Q).to_host()
        pairs = [(q_recs[q][0], t_recs[t][0]) for q, t in zip(qid.tolist(), tid.tolist())]
        qk = extract_kmers(ctx, q_recs, ksize, scaled, moltype, query_fasta)
        tk = extract_kmers(ctx, t_recs, ksize, scaled, moltype, target_fasta)
        return stitch_hits(qk, tk, pairs)
    finally:
        if own:
            ctx.close()
