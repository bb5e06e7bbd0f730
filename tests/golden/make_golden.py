#!/usr/bin/env python3
"""Extract golden vectors (DATA only) from the reference's own test fixtures.

Run once in the build container, where /root/reference is mounted read-only:

    python tests/golden/make_golden.py

Writes small JSON fixtures + the tiny FASTA inputs under tests/golden/.  Nothing here imports
or executes reference code: the reference's Python needs sourmash / branchwater / polars which
are absent from this image (plain absence), and its Rust cannot be built.
What is read:
  * tests/testdata/**/*.sig.zip            -> 75 golden per-protein sketches (mins, abundances, md5)
  * tests/testdata/**/*.kmers.pq           -> k-mer / encoded / hashval / start tables (pyarrow)
  * src/rust/index.rs test tables          -> literal hash KATs and key/count KATs (numbers + strings)
  * tests/test_search.py expected CSV text -> the 5 expected manysearch rows and 5 stitched rows
  * tests/testdata/fasta/*.fasta(.gz)      -> inputs (data files the reference's tests hold)
The output is inputs + expected outputs only; no reference source text is kept.
"""
import csv
import gzip
import hashlib
import io
import json
import os
import re
import shutil
import zipfile

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
TD = os.path.join(REF, "tests", "testdata")
BCL2 = "bcl2_first25_uniprotkb_accession_O43236_OR_accession_2025_02_06.fasta.gz"


def dump(name, obj, gz=False):
    path = os.path.join(OUT, name)
    data = json.dumps(obj, indent=None if gz else 1, sort_keys=True).encode()
    if gz:
        with gzip.GzipFile(path, "wb", mtime=0) as f:
            f.write(data)
    else:
        with open(path, "wb") as f:
            f.write(data + b"\n")
    print(f"wrote {name}: {os.path.getsize(path)} bytes")


def sigzip_to_records(path):
    z = zipfile.ZipFile(path)
    manifest = list(csv.DictReader(io.StringIO(
        "\n".join(l for l in z.read("SOURMASH-MANIFEST.csv").decode().splitlines() if not l.startswith("#")))))
    recs = []
    for row in manifest:
        doc = json.loads(gzip.decompress(z.read(row["internal_location"])))
        assert len(doc) == 1 and len(doc[0]["signatures"]) == 1
        sig = doc[0]["signatures"][0]
        recs.append({
            "name": doc[0]["name"], "md5sum": sig["md5sum"], "ksize": sig["ksize"], "seed": sig["seed"],
            "num": sig["num"], "max_hash": sig["max_hash"], "molecule": sig["molecule"],
            "mins": sig["mins"], "abundances": sig["abundances"],
            "hash_function": doc[0]["hash_function"],
            "manifest_n_hashes": int(row["n_hashes"]), "manifest_scaled": int(row["scaled"]),
            "manifest_ksize": int(row["ksize"]), "manifest_moltype": row["moltype"],
        })
    return recs


def golden_sketches():
    files = {
        "hp.k15.scaled5": os.path.join(TD, "index", BCL2 + ".hp.k15.scaled5.sig.zip"),
        "hp.k16.scaled5": os.path.join(TD, "index", BCL2 + ".hp.k16.scaled5.sig.zip"),
        "hp.k24.scaled5": os.path.join(TD, "fasta", BCL2 + ".hp.k24.scaled5.sig.TRUE.zip"),
    }
    out = {}
    for key, path in files.items():
        recs = sigzip_to_records(path)
        assert len(recs) == 25, (key, len(recs))
        out[key] = {"source": os.path.relpath(path, REF), "signatures": recs}
    dump("sketches.json.gz", out, gz=True)


def golden_kmer_tables():
    import pyarrow.parquet as pq
    files = {
        "hp.k24.scaled5": os.path.join(TD, "fasta", BCL2 + ".hp.k24.scaled5.sig.TRUE.zip.kmers.pq"),
        "hp.k16.scaled5": os.path.join(TD, "index", BCL2 + ".hp.k16.scaled5.sig.zip.kmers.pq"),
        "hp.k15.scaled5": os.path.join(TD, "index", BCL2 + ".hp.k15.scaled5.sig.zip.kmers.pq"),
    }
    out = {}
    for key, path in files.items():
        t = pq.read_table(path).to_pydict()
        cols = sorted(t.keys())
        rows = sorted(zip(t["sequence_name"], t["start"], t["kmer"], t["encoded"], t["hashval"]))
        canon = "\n".join(f"{n}\t{s}\t{k}\t{e}\t{h}" for n, s, k, e, h in rows).encode()
        out[key] = {
            "source": os.path.relpath(path, REF), "columns": cols, "n_rows": len(rows),
            "sha256_sorted_tsv": hashlib.sha256(canon).hexdigest(),
            # full table, grouped per sequence: [start, kmer, encoded, hashval]
            "rows": {},
        }
        for n, s, k, e, h in rows:
            out[key]["rows"].setdefault(n, []).append([s, k, e, h])
    dump("kmer_tables.json.gz", out, gz=True)


def golden_hash_kats():
    src = open(os.path.join(REF, "src", "rust", "index.rs")).read()
    lines = src.splitlines()

    def block(lo, hi):
        return "\n".join(lines[lo - 1:hi])

    kats = {"sequence": "PLANTANDANIMALGENQMES", "ksize": 5, "scaled": 1, "seed": 42,
            "source": "src/rust/index.rs:1084-1103,1187-1205,1309-1326 (literal test tables)"}
    # protein: (hash, ("KMER", [pos])),
    prot = re.findall(r'\((\d+), \("([A-Z]+)", \[(\d+)\]\)\)', block(1084, 1103))
    kats["protein"] = [{"hash": int(h), "encoded": k, "originals": {k: [int(p)]}} for h, k, p in prot]
    # dayhoff: (hash, ("enc", "ORIG", [pos])),
    day = re.findall(r'\((\d+), \("([a-fX]+)", "([A-Z]+)", \[(\d+)\]\)\)', block(1187, 1205))
    kats["dayhoff"] = [{"hash": int(h), "encoded": e, "originals": {o: [int(p)]}} for h, e, o, p in day]
    # hp: (hash, ("enc", vec!["A", "B"], vec![p, q])),
    hp = re.findall(r'\((\d+), \("([hpX]+)", vec!\[([^\]]+)\], vec!\[([^\]]+)\]\)\)', block(1309, 1326))
    kats["hp"] = []
    for h, e, origs, poss in hp:
        o = re.findall(r'"([A-Z]+)"', origs)
        p = [int(x) for x in poss.split(",")]
        kats["hp"].append({"hash": int(h), "encoded": e, "originals": {a: [b] for a, b in zip(o, p)}})
    assert (len(kats["protein"]), len(kats["dayhoff"]), len(kats["hp"])) == (17, 17, 14), \
        (len(kats["protein"]), len(kats["dayhoff"]), len(kats["hp"]))
    # ambiguity-resolution KATs (index.rs:2119-2158, 2205-2243): hash must be present with this encoding
    kats["ambiguity"] = [
        {"moltype": "dayhoff", "sequence": "PLANTANDANIMALGENBMES", "hash": 6161374941338912337, "encoded": "ccecb", "n_kmers": 17},
        {"moltype": "dayhoff", "sequence": "PLANTANDANIMALGENZMES", "hash": 6161374941338912337, "encoded": "ccecb", "n_kmers": 17},
        {"moltype": "dayhoff", "sequence": "PLANTANDANIMALGENJMES", "hash": 9182605311834199497, "encoded": "ceecb", "n_kmers": 17},
        {"moltype": "hp", "sequence": "PLANTANDANIMALGENBMES", "hash": 13058023948041027181, "encoded": "pphpp", "n_kmers": 14},
        {"moltype": "hp", "sequence": "PLANTANDANIMALGENZMES", "hash": 13058023948041027181, "encoded": "pphpp", "n_kmers": 14},
        {"moltype": "hp", "sequence": "PLANTANDANIMALGENJMES", "hash": 10495165127682499337, "encoded": "phhpp", "n_kmers": 14},
    ]
    for a in kats["ambiguity"]:
        assert str(a["hash"]) in src and f'"{a["encoded"]}"' in src
    # encode KATs (src/rust/encoding.rs:195,209)
    enc = open(os.path.join(REF, "src", "rust", "encoding.rs")).read()
    assert "eeeecbbeeec" in enc and "hhhhphhhhhp" in enc
    kats["encode"] = {"sequence": "LIVINGALIVE", "dayhoff": "eeeecbbeeec", "hp": "hhhhphhhhhp", "protein": "LIVINGALIVE"}
    dump("hash_kats.json", kats)


def golden_index_kats():
    """Key / count KATs asserted by the process_fasta tests of index.rs (values cross-checked to be
    present in the source text at generation time)."""
    src = open(os.path.join(REF, "src", "rust", "index.rs")).read()
    small = [("test_protein1", "PLANTANDANIMALGENQMES"), ("test_protein2", "LIVINGALIVE")]
    kats = {
        "source": "src/rust/index.rs:1548-1968, 1975-2076, 2390-2450, 2847-2934",
        "small_fasta": {
            "records": small,
            "cases": [
                {"moltype": "protein", "ksize": 5, "scaled": 1, "combined": 24,
                 "keys": {"f7661cd829e75c0d": 7, "7641839ad508ab8": 17}},
                {"moltype": "dayhoff", "ksize": 5, "scaled": 1, "combined": 24,
                 "keys": {"a963d06839b6d6a9": 7, "84d7545d531dcf51": 17}},
                {"moltype": "hp", "ksize": 5, "scaled": 1, "combined": 16,
                 "keys": {"24ca8d939672666b": 6, "668d7173d661287b": 14}},
            ],
        },
        "bcl2_first25": {
            "fasta": BCL2, "n_signatures": 25,
            "cases": [
                {"moltype": "protein", "ksize": 5, "scaled": 1, "combined": 9049,
                 "keys": {"4d565dee9c8de9db": 474, "4da1f84ad8be618e": 235}},
                {"moltype": "dayhoff", "ksize": 5, "scaled": 1, "combined": 2730,
                 "keys": {"fc27dcd533217385": 433, "3206706fa14185e7": 204}},
                {"moltype": "hp", "ksize": 12, "scaled": 1, "combined": 3549,
                 "keys": {"38ffedf9d3ec7cec": 452, "204716e4d80eb350": 220}},
                {"moltype": "hp", "ksize": 16, "scaled": 5, "combined": 1603, "keys": {}},
            ],
        },
        "single": [
            {"moltype": "protein", "ksize": 5, "scaled": 1, "sequence": "ACDEFGHIKLMNPQRSTVWY",
             "key": "b95f0777d5439d56", "n_kmers": 16},
            {"moltype": "protein", "ksize": 5, "scaled": 1, "sequence": "PLANTANDANIMALGENQMES",
             "key": "7641839ad508ab8", "n_kmers": 17},
        ],
        "invalid": [
            {"sequence": "PLANTANDANIMALGEN1MES", "message": "Invalid amino acid '1'"},
            {"sequence": "PLANTANDANIMALGEN$MES", "message": "Invalid amino acid '$'"},
            {"sequence": "PLANTANDANIMALGEN@MES", "message": "Invalid amino acid '@'"},
        ],
        "mixed_case": {
            "records": [("test_protein_mixed1", "mAaGgCcTt"),
                        ("test_protein_mixed2", "mAaGgCcTtNnRrSsVvWwYyHhKkDdEeFfPpQqIiLl")],
            "moltype": "protein", "ksize": 3, "scaled": 1, "short_kmers": 7, "min_long_kmers": 10,
        },
        "filenames": [
            [16, 5, "hp", "test.fasta.hp.k16.scaled5.kmerseek.rocksdb"],
            [10, 1, "protein", "test.fasta.protein.k10.scaled1.kmerseek.rocksdb"],
            [8, 100, "dayhoff", "test.fasta.dayhoff.k8.scaled100.kmerseek.rocksdb"],
        ],
    }
    # every literal above must appear in the reference test source
    for grp in (kats["small_fasta"]["cases"], kats["bcl2_first25"]["cases"]):
        for c in grp:
            for key, n in c["keys"].items():
                assert f'"{key}"' in src, key
                assert f"== {n}" in src, n
            assert str(c["combined"]) in src
    for s in kats["single"]:
        assert f'"{s["key"]}"' in src
    for f in kats["filenames"]:
        assert f[3] in src
    dump("index_kats.json", kats)


def golden_search():
    txt = open(os.path.join(REF, "tests", "test_search.py")).read()
    blocks = re.findall(r'StringIO\(\s*"""(.*?)"""', txt, flags=re.S)
    assert len(blocks) == 2
    manysearch = list(csv.DictReader(io.StringIO(blocks[0])))
    stitched = list(csv.DictReader(io.StringIO(blocks[1])))
    assert len(manysearch) == 5 and len(manysearch[0]) == 22
    assert len(stitched) == 5
    out = {
        "source": "tests/test_search.py:33-39 and :104-111 (expected CSV text)",
        "query_fasta": "ced9.fasta", "target_fasta": BCL2,
        "moltype": "hp", "ksize": 16, "scaled": 5,
        "manysearch_columns": list(manysearch[0].keys()),
        "manysearch_rows": manysearch,
        "stitched_columns": list(stitched[0].keys()),
        "stitched_rows": stitched,
    }
    # older 16-column schema file kept by the reference (prob_overlap/tf_idf columns are multisearch-only)
    old = list(csv.DictReader(open(os.path.join(TD, "index", "ced9-bcl2-first25.hp.k16.manysearch.csv"))))
    out["older_schema_rows"] = [{k: r[k] for k in ("query_name", "query_md5", "match_name", "match_md5",
                                                   "containment", "max_containment", "jaccard",
                                                   "intersect_hashes", "ksize", "scaled", "moltype")}
                                for r in old]
    dump("search_expected.json", out)


def copy_inputs():
    # (the two larger UniProt downloads are the inputs of the reference's CLI benchmarks, scripts/benchmark_cli.sh:13-15,
    # benches/benchmark_cli.rs: real length distribution, real low-complexity regions, 'X' residues)
    for rel in ("fasta/ced9.fasta", "fasta/" + BCL2, "fasta/test_compression.fasta", "fasta/test_compression.fasta.zst",
                "fasta/uniprotkb_BCL2_AND_model_organism_9606_2025_02_06.fasta.gz",
                "fasta/uniprotkb_protein_name_Uncharacterized_2025_04_15.fasta.gz"):
        dst = os.path.join(OUT, os.path.basename(rel))
        shutil.copyfile(os.path.join(TD, rel), dst)
        os.chmod(dst, 0o644)
        print("copied", os.path.basename(rel))


if __name__ == "__main__":
    copy_inputs()
    golden_sketches()
    golden_kmer_tables()
    golden_hash_kats()
    golden_index_kats()
    golden_search()
