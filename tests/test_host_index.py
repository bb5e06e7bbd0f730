"""The reference's own index / entity / search tests, translated to the C++ host mirror (kmerseek_amd/host.py over
include/kmerseek_host_c.h) and the wire-format layer.  Each test names the reference test it follows.  GPU only:
every signature comes out of the HIP library."""
import csv
import gzip
import math
import os
import shutil

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import kmerseek_amd as ks
from kmerseek_amd import host, wire
from oracle import oracle

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
BCL2 = "bcl2_first25_uniprotkb_accession_O43236_OR_accession_2025_02_06.fasta.gz"
TEST_PROTEIN = "PLANTANDANIMALGENQMES"
TEST_FASTA_CONTENT = ">test_protein1\nPLANTANDANIMALGENQMES\n>test_protein2\nLIVINGALIVE"


def new_index(tmp_path, k, scaled, mol, raw=False, name="test.db"):
    return host.ProteomeIndex(str(tmp_path / name), k, scaled, mol, raw)


# src/rust/index.rs:1041-1135, 1138-1250, 1253-1380 (test_process_kmers_moltype_*)
@pytest.mark.parametrize("mol", ["protein", "dayhoff", "hp"])
def test_process_kmers_moltype(tmp_path, hash_kats, mol):
    ix = new_index(tmp_path, 5, 1, mol)
    sig = ix.create_protein_signature(TEST_PROTEIN, "test_protein")
    rows = hash_kats[mol]
    assert len(sig["kmer_infos"]) == len(rows) == {"protein": 17, "dayhoff": 17, "hp": 14}[mol]
    assert sig["mins"] == sorted(r["hash"] for r in rows)
    for r in rows:
        info = sig["kmer_infos"][r["hash"]]
        assert info["encoded_kmer"] == r["encoded"] and info["ksize"] == 5
        assert info["original_kmer_to_position"] == r["originals"]
    assert sig["minhash_ksize"] == 15 and sig["name"] == "test_protein"


# src/rust/index.rs:1548-1730 (test_process_fasta_moltype_*)
def test_process_fasta_small(tmp_path, index_kats):
    fasta = tmp_path / "test.fasta"
    fasta.write_text(TEST_FASTA_CONTENT)
    for case in index_kats["small_fasta"]["cases"]:
        ix = new_index(tmp_path, case["ksize"], case["scaled"], case["moltype"], name=f"{case['moltype']}.db")
        ix.process_fasta(fasta, 0, 1000)
        sigs = ix.get_signatures()
        assert len(sigs) == 2 == ix.signature_count()
        assert {k: len(v["kmer_infos"]) for k, v in sigs.items()} == case["keys"]
        assert ix.combined_minhash_size() == case["combined"]


# src/rust/index.rs:1791-1970 (test_process_fasta_gz_moltype_*), :2390-2450 (manual vs auto)
def test_process_fasta_gz_bcl2(tmp_path, index_kats):
    fasta = os.path.join(GOLDEN, BCL2)
    for case in index_kats["bcl2_first25"]["cases"]:
        ix = new_index(tmp_path, case["ksize"], case["scaled"], case["moltype"], name=f"bcl2.{case['moltype']}.{case['ksize']}.db")
        ix.process_fasta(fasta, 0, 1000)
        assert ix.signature_count() == 25
        sigs = ix.get_signatures()
        for key, n in case["keys"].items():
            assert len(sigs[key]["kmer_infos"]) == n
        assert ix.combined_minhash_size() == case["combined"]
    # small batches give the same index as one big batch
    a = new_index(tmp_path, 16, 5, "hp", name="a.db"); a.process_fasta(fasta, 0, 7)
    b = new_index(tmp_path, 16, 5, "hp", name="b.db"); b.process_fasta(fasta, 0, 100000)
    assert a.is_equivalent_to(b) and a.combined_minhash_size() == 1603
    # auto filename (index.rs:2426-2448)
    local = tmp_path / BCL2
    shutil.copyfile(fasta, local)
    auto = host.ProteomeIndex.new_with_auto_filename(str(local), 16, 5, "hp", False)
    assert auto.path() == str(local) + ".hp.k16.scaled5.kmerseek.rocksdb"
    auto.process_fasta(local, 0, 1000)
    assert auto.is_equivalent_to(b)


# src/rust/index.rs:1975-2076, 2079-2250 (amino-acid validation, ambiguity)
def test_amino_acid_validation(tmp_path, index_kats, hash_kats):
    ix = new_index(tmp_path, 5, 1, "protein")
    for s in index_kats["single"]:
        sig = ix.create_protein_signature(s["sequence"], "test_protein")
        assert sig["md5sum"] == s["key"] and len(sig["kmer_infos"]) == s["n_kmers"]
    assert len(ix.create_protein_signature("ACDEFXBZJ", "t")["kmer_infos"]) == 5
    for case in index_kats["invalid"]:
        with pytest.raises(RuntimeError) as e:
            ix.create_protein_signature(case["sequence"], "test_protein")
        assert case["message"] in str(e.value) and "position 18" in str(e.value)
    for seq in ("PLANTANDANIMALGENBMES", "PLANTANDANIMALGENZMES", "PLANTANDANIMALGENJMES"):
        assert len(ix.create_protein_signature(seq, "t")["kmer_infos"]) == 17
    # lower case is only accepted on the FASTA path (index.rs:1000)
    with pytest.raises(RuntimeError):
        ix.create_protein_signature("plantandanimal", "t")
    for a in hash_kats["ambiguity"]:
        jx = new_index(tmp_path, 5, 1, a["moltype"], name=f"amb.{a['moltype']}.db")
        sig = jx.create_protein_signature(a["sequence"], "test_protein")
        assert len(sig["kmer_infos"]) == a["n_kmers"]
        assert sig["kmer_infos"][a["hash"]]["encoded_kmer"] == a["encoded"]


# src/rust/index.rs:2251-2300 (test_process_fasta_amino_acid_validation): one bad record fails the whole file
def test_process_fasta_invalid_record_aborts(tmp_path):
    bad = tmp_path / "bad.fasta"
    bad.write_text(">ok1\nPLANTANDANIMALGENQMES\n>bad\nPLANTANDANIMALGEN1MES\n>ok2\nLIVINGALIVE\n")
    ix = new_index(tmp_path, 5, 1, "protein")
    with pytest.raises(RuntimeError) as e:
        ix.process_fasta(bad, 0, 1000)
    assert "Invalid amino acid '1'" in str(e.value)
    assert ix.signature_count() == 0
    good = tmp_path / "good.fasta"
    good.write_text(">a\nPLANTANDANIMALGENQMES\n>b\nLIVINGALIVE\n>c\nACDEFGHIKLMNPQRSTVWY\n>d\nMKVLAAGIVGLCAK\n")
    ix.process_fasta(good, 0, 2)
    assert ix.signature_count() == 4


# src/rust/index.rs:2337-2368, 2543-2593 (equivalence)
def test_index_equivalence(tmp_path):
    recs = [("PLANTANDANIMALGENQMES", "p1"), ("LIVINGALIVE", "p2"), ("ACDEFGHIKLMNPQRSTVWY", "p3")]
    a = new_index(tmp_path, 5, 1, "protein", name="1.db"); a.add_records(recs)
    b = new_index(tmp_path, 5, 1, "protein", name="2.db"); b.add_records(recs)
    assert a.is_equivalent_to(b) and a.signature_count() == b.signature_count() == 3
    assert a.combined_minhash_size() == b.combined_minhash_size() == 40
    c = new_index(tmp_path, 10, 1, "protein", name="3.db"); c.add_records(recs)
    assert not a.is_equivalent_to(c)
    d = new_index(tmp_path, 5, 1, "protein", name="4.db"); d.add_records(recs[:2] + [("MKVLAAGIVGLCAKWWW", "p3")])
    assert not a.is_equivalent_to(d)


# src/rust/index.rs:2454-2477, 2596-2637 (filename generation) + builder errors (:3021-3060)
def test_filenames_and_builder(tmp_path, index_kats):
    for k, s, mol, want in index_kats["filenames"]:
        ix = host.ProteomeIndex.new_with_auto_filename(str(tmp_path / "test.fasta"), k, s, mol, False)
        assert os.path.basename(ix.path()) == want
        assert ix.generate_filename("test.fasta") == want
        assert (ix.ksize(), ix.scaled(), ix.moltype()) == (k, s, mol)
    b = host.ProteomeIndex.builder().path(tmp_path / "b.db").ksize(5).scaled(1).moltype("protein").build()
    assert b.ksize() == 5 and not b.store_raw_sequences()
    for build, msg in (
        (lambda: host.ProteomeIndex.builder().ksize(5).scaled(1).moltype("hp").build(), "Database path is required"),
        (lambda: host.ProteomeIndex.builder().path("x").scaled(1).moltype("hp").build(), "K-mer size is required"),
        (lambda: host.ProteomeIndex.builder().path("x").ksize(5).moltype("hp").build(), "Scaled value is required"),
        (lambda: host.ProteomeIndex.builder().path("x").ksize(5).scaled(1).build(), "Molecular type is required"),
        (lambda: host.ProteomeIndex.builder().ksize(5).scaled(1).moltype("hp").build_with_auto_filename(), "Base path is required"),
    ):
        with pytest.raises(RuntimeError) as e:
            build()
        assert msg in str(e.value) and e.value.kind == "BuilderError"
    with pytest.raises(RuntimeError) as e:
        host.ProteomeIndex(str(tmp_path / "m.db"), 5, 1, "dna", False)
    assert "Invalid moltype: dna" in str(e.value)


# src/rust/index.rs:2713-2845 (raw-sequence storage), :2847-2934 (mixed case), save/load round trip
def test_raw_sequences_mixed_case_and_persistence(tmp_path, index_kats):
    ix = new_index(tmp_path, 5, 1, "protein", raw=True)
    sig = ix.create_protein_signature("ACDEFGHIKLMNPQRSTVWY", "t", store=True)
    assert sig["raw_sequence"] == "ACDEFGHIKLMNPQRSTVWY" and ix.store_raw_sequences() and ix.signature_count() == 1
    jx = new_index(tmp_path, 5, 1, "protein", raw=False, name="noraw.db")
    assert jx.create_protein_signature("ACDEFGHIKLMNPQRSTVWY", "t")["raw_sequence"] is None
    mc = index_kats["mixed_case"]
    fasta = tmp_path / "mixed.fasta"
    fasta.write_text("".join(f">{n}\n{s}\n" for n, s in mc["records"]))
    mx = new_index(tmp_path, mc["ksize"], mc["scaled"], mc["moltype"], raw=True, name="mixed.db")
    mx.process_fasta(fasta, 0, 1000)
    sigs = mx.get_signatures()
    counts = [len(s["kmer_infos"]) for s in sigs.values()]
    assert len(sigs) == 2 and mc["short_kmers"] in counts and any(c > mc["min_long_kmers"] for c in counts)
    assert all(s["raw_sequence"] == s["raw_sequence"].upper() for s in sigs.values())
    # process_fasta saved the state: loading gives an equivalent index; an empty path has no saved state
    back = host.ProteomeIndex.load(mx.path())
    assert back.is_equivalent_to(mx) and back.store_raw_sequences()
    with pytest.raises(RuntimeError) as e:
        host.ProteomeIndex.load(str(tmp_path / "nothing.db"))
    assert "No saved state found in database" in str(e.value)


# src/rust/lib.rs:28-103
def test_pyo3_surface(tmp_path):
    assert host.sum_as_string(2, 3) == "5"
    assert str(host.PyProteinEncoding.raw()) == "protein" and str(host.PyProteinEncoding.dayhoff()) == "dayhoff"
    assert str(host.PyProteinEncoding.hp()) == "hp"
    p = host.PyProteomeIndex(16, 5, host.PyProteinEncoding.hp(), str(tmp_path / "py.db"))
    assert p.index.moltype() == "hp"
    with pytest.raises(RuntimeError):
        host.PyProteomeIndex(16, 5, "nope", str(tmp_path / "py2.db"))


# SURVEY 8(b) "add sketch_* / search_*": the object API's search — BCL2-25 indexed through process_fasta, ced9 searched through
# ProteomeIndex.search_fasta / PyProteomeIndex.search_fasta -> exactly the 5 x 22 expected rows of the reference's
# tests/test_search.py:33-39 (src/python/kmerseek/search.py:125-141 over src/rust/index.rs:642-652)
def _assert_rows_equal_expected(rows, search_expected):
    got = sorted(rows, key=lambda r: r["match_name"])
    exp = sorted(search_expected["manysearch_rows"], key=lambda r: r["match_name"])
    assert len(got) == len(exp) == 5
    assert list(got[0].keys()) == search_expected["manysearch_columns"] == host.SEARCH_COLUMNS
    for g, w in zip(got, exp):
        for col in search_expected["manysearch_columns"]:
            if col in ("query_name", "query_md5", "match_name", "match_md5", "moltype"):
                assert str(g[col]) == w[col], col
            else:
                assert math.isclose(float(g[col]), float(w[col]), rel_tol=1e-12, abs_tol=1e-15), (col, g[col], w[col])


def test_proteome_index_search_equals_expected(tmp_path, search_expected):
    ix = new_index(tmp_path, 16, 5, "hp")
    ix.process_fasta(os.path.join(GOLDEN, BCL2), 0, 1000)
    assert ix.signature_count() == 25
    rows = ix.search_fasta(os.path.join(GOLDEN, "ced9.fasta"))
    _assert_rows_equal_expected(rows, search_expected)
    assert all(r["query_md5"] == "fe3714626e8180caf90f78091563aae6" for r in rows)
    # the same through (sequence, name) records, and again: the device-resident index is built once and reused
    name, seq = wire.read_fasta(os.path.join(GOLDEN, "ced9.fasta"))[0]
    for _ in range(2):
        _assert_rows_equal_expected(ix.search([(seq.decode(), name)]), search_expected)
    # two queries in one batch: rows grouped by query, in batch order; a query that shares nothing has no rows
    rows2 = ix.search([("ACDEFGHIK", "no_match"), (seq.decode(), name), (seq.decode().lower(), "lower")], upper=True)
    assert [r["query_name"] for r in rows2] == [name] * 5 + ["lower"] * 5
    _assert_rows_equal_expected(rows2[:5], search_expected)
    # storing more signatures invalidates the device index: the next search sees them
    ix.add_records([(seq.decode(), "ced9_copy")])
    rows3 = ix.search([(seq.decode(), name)])
    assert len(rows3) == 6
    me = [r for r in rows3 if r["match_name"] == "ced9_copy"][0]
    assert me["containment"] == 1.0 and me["jaccard"] == 1.0 and me["intersect_hashes"] == 49 and me["max_containment_ani"] == 1.0
    assert me["match_md5"] == me["query_md5"]
    # queries are validated like create_protein_signature's input: the reference's message, first bad record aborts
    with pytest.raises(host.IndexError_) as e:
        ix.search([(seq.decode(), name), ("PLANT1ANDANIMAL", "bad")])
    assert "Invalid amino acid '1'" in str(e.value) and e.value.kind == "InvalidAminoAcid"
    # an empty index, an empty batch
    assert new_index(tmp_path, 16, 5, "hp", name="empty.db").search([(seq.decode(), name)]) == []
    assert ix.search([]) == []


def test_pyo3_sketch_and_search_methods(tmp_path, search_expected):
    p = host.PyProteomeIndex(16, 5, host.PyProteinEncoding.hp(), str(tmp_path / "py.db"))
    assert p.sketch_fasta(os.path.join(GOLDEN, BCL2)) == 25 == p.signature_count()
    out = tmp_path / "search.csv"
    rows = p.search_fasta(os.path.join(GOLDEN, "ced9.fasta"), output=str(out))
    _assert_rows_equal_expected(rows, search_expected)
    back = list(csv.DictReader(open(out)))
    assert list(back[0].keys()) == search_expected["manysearch_columns"]
    _assert_rows_equal_expected(back, search_expected)
    # sketch_sequences / search_sequences on records
    q = host.PyProteomeIndex(5, 1, "protein", str(tmp_path / "py2.db"))
    assert q.sketch_sequences([(TEST_PROTEIN, "a"), ("LIVINGALIVE", "b")]) == 2
    rows = q.search_sequences([("PLANTANDANIMAL", "q")])
    assert len(rows) == 1 and rows[0]["match_name"] == "a" and rows[0]["intersect_hashes"] == 10 and rows[0]["containment"] == 1.0
    assert rows[0]["ksize"] == 15 and rows[0]["scaled"] == 1 and rows[0]["moltype"] == "protein"


# tests/test_entity.py:9-22 (the sketch artifact equals the committed golden .sig.zip)
@pytest.mark.parametrize("key,ksize", [("hp.k24.scaled5", 24), ("hp.k16.scaled5", 16), ("hp.k15.scaled5", 15)])
def test_sketch_sig_zip_equals_golden(tmp_path, golden_sketches, key, ksize):
    fasta = tmp_path / BCL2
    shutil.copyfile(os.path.join(GOLDEN, BCL2), fasta)
    sig = wire.sketch(str(fasta), "hp", ksize, 5)
    assert sig == f"{fasta}.hp.k{ksize}.scaled5.sig.zip" and os.path.exists(sig)
    assert os.path.exists(f"{fasta}.manysketch.csv")
    names, offs, mins, abunds, k, sc, mol = wire.read_sig_zip(sig)
    assert (k, sc, mol) == (ksize, 5, "hp")
    gold = {s["name"]: s for s in golden_sketches[key]["signatures"]}
    assert len(names) == 25
    import zipfile, json
    z = zipfile.ZipFile(sig)
    for i, n in enumerate(names):
        g = gold[n]
        assert mins[int(offs[i]):int(offs[i + 1])].tolist() == g["mins"]
        assert abunds[int(offs[i]):int(offs[i + 1])].tolist() == g["abundances"]
        doc = json.loads(gzip.decompress(z.read(f"signatures/{g['md5sum']}.sig.gz")))
        s = doc[0]["signatures"][0]
        for f in ("num", "ksize", "seed", "max_hash", "md5sum", "molecule"):
            assert s[f] == g[f]
        assert doc[0]["hash_function"] == g["hash_function"]


# tests/test_search.py:9-60 (the manysearch CSV equals the 5 expected rows, all 22 columns)
def test_manysearch_csv_equals_expected(tmp_path, search_expected):
    q = tmp_path / "ced9.fasta"
    t = tmp_path / BCL2
    shutil.copyfile(os.path.join(GOLDEN, "ced9.fasta"), q)
    shutil.copyfile(os.path.join(GOLDEN, BCL2), t)
    with ks.Context(0) as ctx:
        qs = wire.sketch(str(q), "hp", 16, 5, ctx)
        ts = wire.sketch(str(t), "hp", 16, 5, ctx)
        out = tmp_path / "search.csv"
        n = wire.do_manysearch(qs, ts, str(out), 16, 5, "hp", ctx)
    assert n == 5
    got = sorted(csv.DictReader(open(out)), key=lambda r: r["match_name"])
    exp = sorted(search_expected["manysearch_rows"], key=lambda r: r["match_name"])
    assert list(got[0].keys()) == search_expected["manysearch_columns"]
    for g, w in zip(got, exp):
        for col in search_expected["manysearch_columns"]:
            if col in ("query_name", "query_md5", "match_name", "match_md5", "moltype"):
                assert g[col] == w[col], col
            else:
                assert math.isclose(float(g[col]), float(w[col]), rel_tol=1e-12, abs_tol=1e-15), (col, g[col], w[col])


# tests/test_index.py:61-71, tests/test_entity.py:48-59 (k-mer table equals the golden parquet, shape (1712, 5))
@pytest.mark.parametrize("key,ksize", [("hp.k24.scaled5", 24), ("hp.k16.scaled5", 16)])
def test_kmer_table_equals_golden(tmp_path, golden_kmer_tables, bcl2_records, key, ksize):
    with ks.Context(0) as ctx:
        rows = wire.extract_kmers(ctx, bcl2_records, ksize, 5, "hp", "bcl2.fasta.gz")
    gold = golden_kmer_tables[key]
    assert len(rows) == gold["n_rows"]
    got = {}
    for r in rows:
        got.setdefault(r["sequence_name"], []).append([r["start"], r["kmer"], r["encoded"], r["hashval"]])
    assert {n: sorted(v) for n, v in got.items()} == {n: sorted(v) for n, v in gold["rows"].items()}
    out = tmp_path / "kmers.pq"
    wire.write_kmers_parquet(rows, str(out))
    import pyarrow.parquet as pq
    t = pq.read_table(out)
    assert t.num_rows == gold["n_rows"] and t.column_names == ["sequence_file", "sequence_name", "kmer", "hashval", "encoded", "start"]
    if ksize == 24:
        assert t.num_rows == 1712


# tests/test_search.py:63-138 (search --extract-kmers: stitched rows and the printed alignment blocks)
def test_search_extract_kmers_equals_expected(tmp_path, search_expected):
    rows = wire.search_extract_kmers(os.path.join(GOLDEN, "ced9.fasta"), os.path.join(GOLDEN, BCL2), 16, 5, "hp")
    exp = sorted(search_expected["stitched_rows"], key=lambda r: r["match_name"])
    got = sorted(rows, key=lambda r: r["match_name"])
    assert len(got) == len(exp) == 5
    for g, w in zip(got, exp):
        for col in search_expected["stitched_columns"]:
            assert str(g[col]) == str(w[col]), (col, g[col], w[col])
    printed = "\n".join(r["to_print"] for r in rows)
    assert "query: MSIGESIDGKINDWEEPGIVGVVVCGRMMFSLK (59-92)" in printed
    assert "alpha: hphhpphphphpphpphhhhhhhhphphhhphp" in printed
    assert "match: HQQEQEAEGVAAPADP (42-58)" in printed
    # rows come out sorted by (query_start, query_end), as the reference prints them
    assert [r["query_start"] for r in rows] == sorted(r["query_start"] for r in rows)


# ---------------------------------------------------------------------------------------------------------
# pipelined FASTA ingest: same sketches as one batched call, whatever the batching
# ---------------------------------------------------------------------------------------------------------
def _batch_sketch(path, k, scaled, mol):
    recs = wire.read_fasta(str(path))
    res, offs = ks.pack([s for _, s in recs])
    ctx = ks.Context(0)
    try:
        o, m, a = ctx.sketch_batch(res, offs, k, scaled, mol).to_host()
    finally:
        ctx.close()
    return [n for n, _ in recs], o, m, a


@pytest.mark.gpu
@pytest.mark.parametrize("fname", [BCL2, "ced9.fasta", "test_compression.fasta"])
def test_sketch_fasta_pipeline_equals_batch_sketch(fname):
    path = os.path.join(GOLDEN, fname)
    for k, scaled, mol in ((10, 1, "protein"), (16, 5, "hp")):
        names, o, m, a = _batch_sketch(path, k, scaled, mol)
        for pipeline, batch in ((True, 0), (True, 700), (False, 700), (True, 1)):
            n2, o2, m2, a2, stats = host.sketch_fasta(path, k, scaled, mol, validate=False, batch_residues=batch, pipeline=pipeline)
            assert n2 == names
            assert np.array_equal(o2, o) and np.array_equal(m2, m) and np.array_equal(a2, a)
            assert stats["residues"] == sum(len(s) for _, s in wire.read_fasta(path))
            if batch == 1:
                assert stats["batches"] == len(names)  # one record per device batch


@pytest.mark.gpu
def test_sketch_fasta_validate_path_and_errors(tmp_path):
    f = tmp_path / "mixed.fasta"
    f.write_text(">lower\nplantandanimalgenqmes\n>stop\nLIVINGALIVE*IGNORED\n>wrapped\nACDEFGHIKL\nMNPQRSTVWY\n\n>empty\n>last\nMKVLAAGIVGLCAK\r\n")
    names, o, m, a, stats = host.sketch_fasta(f, 5, 1, "protein", validate=True, batch_residues=16)
    assert names == ["lower", "stop", "wrapped", "empty", "last"]
    want = [b"PLANTANDANIMALGENQMES", b"LIVINGALIVE*", b"ACDEFGHIKLMNPQRSTVWY", b"", b"MKVLAAGIVGLCAK"]
    res, offs = ks.pack(want)
    wo, wm, wa = oracle.sketch_batch(res, offs, 5, 1, "protein")
    assert np.array_equal(o, wo) and np.array_equal(m, wm) and np.array_equal(a, wa)
    # raw (manysketch) mode hashes the bytes as given: lower case is folded by the hash LUT, '*' is just a residue
    names, o, m, a, _ = host.sketch_fasta(f, 5, 1, "protein", validate=False)
    res, offs = ks.pack([b"plantandanimalgenqmes", b"LIVINGALIVE*IGNORED", b"ACDEFGHIKLMNPQRSTVWY", b"", b"MKVLAAGIVGLCAK"])
    wo, wm, wa = oracle.sketch_batch(res, offs, 5, 1, "protein")
    assert np.array_equal(o, wo) and np.array_equal(m, wm) and np.array_equal(a, wa)
    bad = tmp_path / "bad.fasta"
    bad.write_text(">ok1\nPLANTANDANIMALGENQMES\n>bad\nPLANTANDANIMALGEN1MES\n")
    with pytest.raises(RuntimeError) as e:
        host.sketch_fasta(bad, 5, 1, "protein", validate=True)
    assert "Invalid amino acid '1' found at position 18" in str(e.value)
    with pytest.raises(RuntimeError) as e:
        host.sketch_fasta(tmp_path / "missing.fasta", 5, 1, "protein")
    assert "cannot open" in str(e.value)
    with pytest.raises(RuntimeError):
        host.sketch_fasta(f, 5, 1, "nucleotide")
    nohdr = tmp_path / "nohdr.fasta"
    nohdr.write_text("ACDEFG\n>x\nACDEFG\n")
    with pytest.raises(RuntimeError) as e:
        host.sketch_fasta(nohdr, 5, 1, "protein")
    assert "does not start with '>'" in str(e.value)


# src/rust/index.rs:1734-1788 (test_process_fasta_zstd_moltype_protein) and :1790-1845 (gz): the same two records
# through plain, gzip and zstd give the same index — 2 signatures, keys f7661cd829e75c0d / 7641839ad508ab8, union 24
@pytest.mark.gpu
def test_process_fasta_zstd_and_gz(tmp_path, index_kats):
    case = index_kats["small_fasta"]["cases"][0]
    assert (case["ksize"], case["scaled"], case["moltype"]) == (5, 1, "protein")
    plain = os.path.join(GOLDEN, "test_compression.fasta")
    gz = tmp_path / "test_compression.fasta.gz"
    gz.write_bytes(gzip.compress(open(plain, "rb").read()))
    made = {}
    for tag, path in (("plain", plain), ("zst", os.path.join(GOLDEN, "test_compression.fasta.zst")), ("gz", str(gz))):
        ix = new_index(tmp_path, 5, 1, "protein", name=f"fasta_{tag}_protein_test.db")
        ix.process_fasta(path, 0, 1000)
        sigs = ix.get_signatures()
        assert len(sigs) == 2, tag
        assert {k: len(v["kmer_infos"]) for k, v in sigs.items()} == {"f7661cd829e75c0d": 7, "7641839ad508ab8": 17}, tag
        assert ix.combined_minhash_size() == 24, tag
        made[tag] = ix
    assert made["zst"].is_equivalent_to(made["plain"]) and made["gz"].is_equivalent_to(made["plain"])
    # the pipelined ingest (manysketch path) reads all three the same way
    ref = host.sketch_fasta(plain, 5, 1, "protein", validate=True)
    for path in (os.path.join(GOLDEN, "test_compression.fasta.zst"), str(gz)):
        got = host.sketch_fasta(path, 5, 1, "protein", validate=True)
        assert got[0] == ref[0] and all(np.array_equal(x, y) for x, y in zip(got[1:4], ref[1:4]))
    # a cut-off archive fails the whole file (no shorter proteome)
    big = b"".join(b">r%d\n" % i + b"ACDEFGHIKLMNPQRSTVWY" * 50 + b"\n" for i in range(2000))
    z = gzip.compress(big)
    cut = tmp_path / "cut.fasta.gz"
    cut.write_bytes(z[:len(z) // 2])
    with pytest.raises(RuntimeError) as e:
        host.sketch_fasta(cut, 5, 1, "protein", validate=True)
    assert "truncated" in str(e.value)
    ix = new_index(tmp_path, 5, 1, "protein", name="cut.db")
    with pytest.raises(RuntimeError) as e:
        ix.process_fasta(cut, 0, 1000)
    assert "truncated" in str(e.value) and e.value.kind == "ParseError"


# ProteomeIndex::load must not trust the length fields of the file
@pytest.mark.gpu
def test_load_rejects_truncated_and_garbage_files(tmp_path):
    ix = new_index(tmp_path, 5, 1, "protein", raw=True, name="ok.db")
    ix.add_records([("PLANTANDANIMALGENQMES", "p1"), ("LIVINGALIVE", "p2")])
    ix.save_state()
    blob = open(ix.path(), "rb").read()
    assert host.ProteomeIndex.load(ix.path()).is_equivalent_to(ix)
    for cut in (9, 20, len(blob) // 3, len(blob) // 2, len(blob) - 1):
        p = tmp_path / f"cut{cut}.db"
        p.write_bytes(blob[:cut])
        with pytest.raises(RuntimeError) as e:
            host.ProteomeIndex.load(str(p))
        assert "corrupt index file" in str(e.value) and e.value.kind == "Io", cut
    # a length field blown up to 2^60 must fail the bounds check, not allocate
    import struct
    for at in (8, 8 + 8 + len("protein") + 24):   # moltype length; combined-sketch size
        bad = bytearray(blob)
        bad[at:at + 8] = struct.pack("<Q", 1 << 60)
        p = tmp_path / f"garbage{at}.db"
        p.write_bytes(bytes(bad))
        with pytest.raises(RuntimeError) as e:
            host.ProteomeIndex.load(str(p))
        assert "corrupt index file" in str(e.value), at
    p = tmp_path / "trailing.db"
    p.write_bytes(blob + b"\0" * 8)
    with pytest.raises(RuntimeError) as e:
        host.ProteomeIndex.load(str(p))
    assert "corrupt index file" in str(e.value)
