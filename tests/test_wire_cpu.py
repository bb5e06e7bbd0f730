"""Host-side formatting logic that needs no GPU: .sig.zip round trip, manysearch row arithmetic and the stitching
restatement, fed with oracle data (test infrastructure) and checked against the reference's fixtures."""
import csv
import math
import os

import numpy as np

from kmerseek_amd import wire
from oracle import oracle

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
BCL2 = "bcl2_first25_uniprotkb_accession_O43236_OR_accession_2025_02_06.fasta.gz"


def test_sig_zip_round_trip_and_md5(tmp_path, golden_sketches, bcl2_records):
    res, offs = oracle.pack([s for _, s in bcl2_records])
    o, m, a = oracle.sketch_batch(res, offs, 16, 5, "hp")
    names = [n for n, _ in bcl2_records]
    path = tmp_path / "x.sig.zip"
    wire.write_sig_zip(str(path), names, o, m, a, 16, 5, "hp", "/some/file.fasta.gz")
    n2, o2, m2, a2, k, sc, mol = wire.read_sig_zip(str(path))
    assert n2 == names and (k, sc, mol) == (16, 5, "hp")
    assert np.array_equal(o2, o) and np.array_equal(m2, m) and np.array_equal(a2, a)
    gold = {s["name"]: s["md5sum"] for s in golden_sketches["hp.k16.scaled5"]["signatures"]}
    for i, n in enumerate(names):
        assert wire.sourmash_md5(m[int(o[i]):int(o[i + 1])], 16) == gold[n]
    # the reference's own .sig.zip layout is readable too (same manifest columns)
    import zipfile
    z = zipfile.ZipFile(path)
    head = z.read("SOURMASH-MANIFEST.csv").decode().splitlines()
    assert head[0] == "# SOURMASH-MANIFEST-VERSION: 1.0"
    assert head[1] == "internal_location,md5,md5short,ksize,moltype,num,scaled,n_hashes,with_abundance,name,filename"


def test_manysearch_rows_match_expected_csv(search_expected, ced9_records, bcl2_records):
    k, sc, mol = 16, 5, "hp"
    qo, qm, _ = oracle.sketch_batch(*oracle.pack([s for _, s in ced9_records]), k, sc, mol)
    to, tm, ta = oracle.sketch_batch(*oracle.pack([s for _, s in bcl2_records]), k, sc, mol)
    hits = oracle.manysearch(qo, qm, to, tm, ta)
    rows = wire.manysearch_rows([n for n, _ in ced9_records], qo, qm, [n for n, _ in bcl2_records], to, tm, ta, hits, k, sc, mol)
    exp = sorted(search_expected["manysearch_rows"], key=lambda r: r["match_name"])
    rows.sort(key=lambda r: r["match_name"])
    assert [list(r.keys()) for r in rows][0] == wire.MANYSEARCH_COLUMNS == search_expected["manysearch_columns"]
    for g, w in zip(rows, exp):
        for col in wire.MANYSEARCH_COLUMNS:
            if isinstance(g[col], str):
                assert g[col] == w[col]
            else:
                assert math.isclose(float(g[col]), float(w[col]), rel_tol=1e-12, abs_tol=1e-15), col


def test_stitching_restatement(search_expected, ced9_records, bcl2_records):
    k, sc, mol = 16, 5, "hp"

    def table(records):
        rows = []
        for name, seq in records:
            mins, _ = oracle.sketch_protein(seq, k, sc, mol)
            st, hh = oracle.kmer_positions(seq, k, mol, mins)
            for s, h in zip(st.tolist(), hh.tolist()):
                kmer = seq[s:s + k].decode()
                rows.append({"sequence_file": "", "sequence_name": name, "kmer": kmer, "hashval": h,
                             "encoded": wire.encode_kmer(kmer, mol), "start": s})
        return rows

    qk, tk = table(ced9_records), table(bcl2_records)
    assert all(r["encoded"] == oracle.encode(r["kmer"].encode(), mol).decode() for r in qk)
    pairs = [(r["query_name"], r["match_name"]) for r in search_expected["manysearch_rows"]]
    got = sorted(wire.stitch_hits(qk, tk, pairs), key=lambda r: r["match_name"])
    exp = sorted(search_expected["stitched_rows"], key=lambda r: r["match_name"])
    assert len(got) == 5
    for g, w in zip(got, exp):
        for col in search_expected["stitched_columns"]:
            assert str(g[col]) == str(w[col]), col
    # the stitcher's quirks are kept: a repeated position re-appends the whole k-mer
    assert wire.single_stitch_together_kmers(["ABC", "BCD", "BCD"], [0, 1, 1]) == "ABCDBCD"
    assert wire.single_stitch_together_kmers(["ABC", "CDE"], [0, 2]) == "ABCDE"


def test_native_sourmash_md5_equals_hashlib_and_goldens(golden_sketches):
    """The md5sum columns of ProteomeIndex.search rows come from an in-file MD5 (ks_host.cpp, RFC 1321): equal to hashlib on
    every block-boundary length and to the md5sum fields of the reference's golden signatures (tests/testdata/**/*.sig.zip)."""
    from kmerseek_amd import host
    rng = np.random.default_rng(9)
    for n in list(range(0, 12)) + [55, 56, 57, 63, 64, 65, 119, 120, 121, 1000, 4097]:
        mins = np.sort(rng.integers(1, 2**63, size=n, dtype=np.uint64))
        for k in (5, 16, 24, 128):
            assert host.sourmash_md5(mins, k) == wire.sourmash_md5(mins, k)
    n_checked = 0
    for key, ksize in (("hp.k15.scaled5", 15), ("hp.k16.scaled5", 16), ("hp.k24.scaled5", 24)):
        for sig in golden_sketches[key]["signatures"]:
            assert host.sourmash_md5(np.array(sig["mins"], np.uint64), ksize) == sig["md5sum"]
            n_checked += 1
    assert n_checked == 75
