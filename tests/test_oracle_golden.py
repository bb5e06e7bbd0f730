"""Pins the CPU oracle against every golden vector the reference's own tests hold for the hot path
(SURVEY.md §8c items 1-8).  CPU only."""
import math

import numpy as np
import pytest

from oracle import oracle

MOLS = ("protein", "dayhoff", "hp")


def test_hash_and_encode_kats(hash_kats):
    seq = hash_kats["sequence"].encode()
    k = hash_kats["ksize"]
    for mol in MOLS:
        rows = hash_kats[mol]
        mins, abunds = oracle.sketch_protein(seq, k, 1, mol)
        assert sorted(r["hash"] for r in rows) == mins.tolist()
        infos = oracle.kmer_infos(seq, k, mol, mins)
        assert len(infos) == len(rows)
        for r in rows:
            assert oracle.hash_murmur(r["encoded"].encode()) == r["hash"]
            info = infos[r["hash"]]
            assert info["encoded_kmer"] == r["encoded"]
            assert info["original_kmer_to_position"] == {o: p for o, p in r["originals"].items()}
            for o in r["originals"]:
                assert oracle.encode(o.encode(), mol).decode() == r["encoded"]
        # abundance = number of windows per hash
        for h, a in zip(mins.tolist(), abunds.tolist()):
            assert a == sum(len(p) for p in infos[h]["original_kmer_to_position"].values())
    e = hash_kats["encode"]
    for mol in MOLS:
        assert oracle.encode(e["sequence"].encode(), mol).decode() == e[mol]


def test_ambiguity_kats(hash_kats):
    for a in hash_kats["ambiguity"]:
        for choice in (b"\x00", b"\x01"):
            seq = oracle.validate_and_resolve(a["sequence"].encode(), choice)
            assert len(seq) == len(a["sequence"]) and not set(seq) & set(b"BZJ")
            mins, _ = oracle.sketch_protein(seq, 5, 1, a["moltype"])
            infos = oracle.kmer_infos(seq, 5, a["moltype"], mins)
            assert len(infos) == a["n_kmers"]
            assert infos[a["hash"]]["encoded_kmer"] == a["encoded"]


def test_max_hash(golden_sketches):
    assert oracle.max_hash(1) == 2**64 - 1
    assert oracle.max_hash(5) == 3689348814741910528
    assert oracle.max_hash(0) == 0
    for key, grp in golden_sketches.items():
        for s in grp["signatures"]:
            assert oracle.max_hash(s["manifest_scaled"]) == s["max_hash"]


@pytest.mark.parametrize("key,ksize", [("hp.k15.scaled5", 15), ("hp.k16.scaled5", 16), ("hp.k24.scaled5", 24)])
def test_golden_sketches_bit_exact(golden_sketches, bcl2_records, key, ksize):
    sigs = {s["name"]: s for s in golden_sketches[key]["signatures"]}
    assert len(sigs) == 25 == len(bcl2_records)
    res, offs = oracle.pack([s for _, s in bcl2_records])
    o, mins, abunds = oracle.sketch_batch(res, offs, ksize, 5, "hp", n_threads=3)
    for i, (name, seq) in enumerate(bcl2_records):
        g = sigs[name]
        m = mins[int(o[i]):int(o[i + 1])]
        a = abunds[int(o[i]):int(o[i + 1])]
        assert g["ksize"] == 3 * ksize and g["seed"] == 42 and g["num"] == 0 and g["molecule"] == "hp"
        assert m.tolist() == g["mins"]
        assert a.tolist() == g["abundances"]
        assert len(m) == g["manifest_n_hashes"]
        assert oracle.sourmash_md5(m, ksize) == g["md5sum"]
        # single-protein entry point agrees with the batch one
        m1, a1 = oracle.sketch_protein(seq, ksize, 5, "hp")
        assert m1.tolist() == g["mins"] and a1.tolist() == g["abundances"]


@pytest.mark.parametrize("key,ksize", [("hp.k24.scaled5", 24), ("hp.k16.scaled5", 16), ("hp.k15.scaled5", 15)])
def test_golden_kmer_tables(golden_kmer_tables, bcl2_records, key, ksize):
    tab = golden_kmer_tables[key]
    seqs = dict(bcl2_records)
    n = 0
    for name, rows in tab["rows"].items():
        seq = seqs[name]
        mins, _ = oracle.sketch_protein(seq, ksize, 5, "hp")
        starts, hashes = oracle.kmer_positions(seq, ksize, "hp", mins)
        got = [[int(s), seq[s:s + ksize].decode(), oracle.encode(seq[s:s + ksize], "hp").decode(), int(h)]
               for s, h in zip(starts.tolist(), hashes.tolist())]
        assert got == sorted(rows)
        n += len(got)
    assert n == tab["n_rows"]
    if key == "hp.k24.scaled5":
        assert n == 1712  # tests/test_entity.py:58


def test_index_kats(index_kats, bcl2_records):
    def run(records, case):
        keys, union = {}, set()
        for _, seq in records:
            seq = oracle.validate_and_resolve(seq.upper())
            mins, _ = oracle.sketch_protein(seq, case["ksize"], case["scaled"], case["moltype"])
            infos = oracle.kmer_infos(seq, case["ksize"], case["moltype"], mins)
            keys[oracle.pseudo_md5(mins)] = len(infos)
            union.update(mins.tolist())
        return keys, union

    small = [(n, s.encode()) for n, s in index_kats["small_fasta"]["records"]]
    for case in index_kats["small_fasta"]["cases"]:
        keys, union = run(small, case)
        assert keys == case["keys"] and len(union) == case["combined"]
    for case in index_kats["bcl2_first25"]["cases"]:
        keys, union = run(bcl2_records, case)
        assert len(keys) == 25
        for k, n in case["keys"].items():
            assert keys[k] == n
        assert len(union) == case["combined"]
    for s in index_kats["single"]:
        mins, _ = oracle.sketch_protein(s["sequence"].encode(), s["ksize"], s["scaled"], s["moltype"])
        assert oracle.pseudo_md5(mins) == s["key"] and len(mins) == s["n_kmers"]
    mc = index_kats["mixed_case"]
    counts = []
    for _, seq in mc["records"]:
        seq = oracle.validate_and_resolve(seq.upper().encode())
        mins, _ = oracle.sketch_protein(seq, mc["ksize"], mc["scaled"], mc["moltype"])
        counts.append(len(oracle.kmer_infos(seq, mc["ksize"], mc["moltype"], mins)))
    assert mc["short_kmers"] in counts and any(c > mc["min_long_kmers"] for c in counts)


def test_validation_errors(index_kats):
    for case in index_kats["invalid"]:
        with pytest.raises(oracle.InvalidAminoAcid) as e:
            oracle.validate_and_resolve(case["sequence"].encode())
        assert case["message"] in str(e.value)
        assert e.value.pos == 18
    # lower case is NOT upper-cased by create_protein_signature (only by the FASTA path)
    with pytest.raises(oracle.InvalidAminoAcid):
        oracle.validate_and_resolve(b"plant")
    # '*' truncates, inclusive (aminoacid.rs:79-83)
    assert oracle.validate_and_resolve(b"ACD*EFG") == b"ACD*"
    assert oracle.validate_and_resolve(b"ACDEFXUO") == b"ACDEFXUO"
    assert oracle.validate_and_resolve(b"BZJ", b"\x00\x00\x00") == b"DEI"
    assert oracle.validate_and_resolve(b"BZJ", b"\x01\x01\x01") == b"NQL"


def test_manysearch_rows(search_expected, ced9_records, bcl2_records):
    k, sc, mol = search_expected["ksize"], search_expected["scaled"], search_expected["moltype"]
    (qname, qseq), = ced9_records
    qm, _ = oracle.sketch_protein(qseq, k, sc, mol)
    assert len(qm) == 49
    rows = []
    for tname, tseq in bcl2_records:
        tm, ta = oracle.sketch_protein(tseq, k, sc, mol)
        r = oracle.manysearch_row(qname, qm, tname, tm, ta.astype(np.uint32), k, sc, mol)
        if r:
            rows.append(r)
    exp = sorted(search_expected["manysearch_rows"], key=lambda r: r["match_name"])
    rows.sort(key=lambda r: r["match_name"])
    assert len(rows) == len(exp) == 5
    assert list(rows[0].keys()) == search_expected["manysearch_columns"] == oracle.MANYSEARCH_COLUMNS
    for got, want in zip(rows, exp):
        for col in search_expected["manysearch_columns"]:
            g, w = got[col], want[col]
            if isinstance(g, str):
                assert g == w, col
            elif isinstance(g, int):
                assert g == int(w), col
            else:
                assert math.isclose(g, float(w), rel_tol=1e-12, abs_tol=1e-15), (col, g, w)
    # COO form from the C all-pairs search agrees
    qo, qmins, _ = oracle.sketch_batch(*oracle.pack([qseq]), k, sc, mol)
    to, tmins, tab = oracle.sketch_batch(*oracle.pack([s for _, s in bcl2_records]), k, sc, mol)
    qid, tid, isect, nw = oracle.manysearch(qo, qmins, to, tmins, tab, n_threads=2)
    names = [n for n, _ in bcl2_records]
    coo = {names[t]: (int(i), int(w)) for t, i, w in zip(tid.tolist(), isect.tolist(), nw.tolist())}
    assert coo == {r["match_name"]: (int(r["intersect_hashes"]), int(r["n_weighted_found"])) for r in exp}
    assert set(qid.tolist()) == {0}
    # older-schema fixture: same pairs and counts
    for r in search_expected["older_schema_rows"]:
        assert coo[r["match_name"]][0] == int(float(r["intersect_hashes"]))
        assert r["query_md5"] == rows[0]["query_md5"]


def test_edges_documented_unpinned():
    # L < k: empty sketch (unpinned)
    m, a = oracle.sketch_protein(b"ACD", 5, 1, "protein")
    assert len(m) == 0
    # X/U/O/* under reduced alphabets map to 'X'
    assert oracle.encode(b"XUO*", "hp") == b"XXXX" and oracle.encode(b"XUO*", "dayhoff") == b"XXXX"
    # sourmash upper-cases inside add_protein
    m1, _ = oracle.sketch_protein(b"plantandanimalgenqmes", 5, 1, "hp")
    m2, _ = oracle.sketch_protein(b"PLANTANDANIMALGENQMES", 5, 1, "hp")
    assert m1.tolist() == m2.tolist()
    # batch threading is deterministic
    rng = np.random.default_rng(0)
    seqs = [bytes(rng.choice(list(b"ACDEFGHIKLMNPQRSTVWY"), size=int(n))) for n in rng.integers(0, 200, 50)]
    r, o = oracle.pack(seqs)
    a1 = oracle.sketch_batch(r, o, 7, 1, "protein", n_threads=1)
    a4 = oracle.sketch_batch(r, o, 7, 1, "protein", n_threads=4)
    for x, y in zip(a1, a4):
        assert np.array_equal(x, y)
