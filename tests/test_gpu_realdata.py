"""Real UniProt sequences — the inputs of the reference's own CLI benchmarks (scripts/benchmark_cli.sh, benches/benchmark_cli.rs:
300 BCL2-family proteins, 2,841 "uncharacterized" proteins; up to 3,881 aa, low-complexity regions, 'X' residues) — through the
HIP path and the oracle: sketches, k-mer positions, all-vs-all search, for the reference's default parameters (hp k=24 scaled=5)
and the other alphabets; file -> sketches through the pipelined ingest."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import kmerseek_amd as ks
from kmerseek_amd import host, wire
from oracle import oracle

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
FILES = ["uniprotkb_BCL2_AND_model_organism_9606_2025_02_06.fasta.gz", "uniprotkb_protein_name_Uncharacterized_2025_04_15.fasta.gz"]
PARAMS = [(24, 5, "hp"), (16, 5, "dayhoff"), (10, 1, "protein"), (7, 1, "hp")]


@pytest.fixture(scope="module")
def ctx():
    c = ks.Context(0, follow_debug_env=True)
    yield c
    c.close()


@pytest.mark.parametrize("fname", FILES)
def test_real_proteins_sketch_positions_search(ctx, fname):
    recs = wire.read_fasta(os.path.join(GOLDEN, fname))
    assert len(recs) in (300, 2841)
    raw = [s for _, s in recs]
    val = [ks.validate_and_resolve(s, upper=True) for s in raw]      # the Rust index path (aminoacid.rs:74-105); no B/Z/J here
    assert any(b"X" in s for s in val)
    res, offs = ks.pack(val)
    for k, scaled, mol in PARAMS:
        S = ctx.sketch_batch(res, offs, k, scaled, mol)
        got = S.to_host()
        want = oracle.sketch_batch(res, offs, k, scaled, mol, n_threads=8)
        for g, w in zip(got, want):
            assert np.array_equal(g, w), (fname, k, scaled, mol)
        if (k, mol) == (7, "hp"):
            continue  # saturated alphabet: hundreds of millions of matched pairs — sketch parity only
        # k-mer positions (process_kmers) on a slice of the file
        sub = slice(0, 120)
        r2, o2 = ks.pack(val[sub])
        ps, pst, ph = ctx.kmer_positions(r2, o2, k, scaled, mol)
        at = 0
        for i, s in enumerate(val[sub]):
            st, hh = oracle.kmer_positions(s, k, mol, want[1][int(want[0][i]):int(want[0][i + 1])])
            assert np.array_equal(pst[at:at + len(st)], st) and np.array_equal(ph[at:at + len(st)], hh) and np.all(ps[at:at + len(st)] == i)
            at += len(st)
        assert at == len(ps)
        # all-vs-all search against the pairwise oracle
        ix = ctx.index_build(S)
        d_res, d_off = ctx.to_device(res), ctx.to_device(offs)
        Q = ctx.sketch_queries_device(ix, d_res.ptr, d_off.ptr, len(val), len(res), max_seq_len=int((offs[1:] - offs[:-1]).max()))
        hits = ctx.search(ix, Q).to_host()
        w = oracle.manysearch(want[0], want[1], want[0], want[1], want[2], n_threads=16)
        for g, x in zip(hits, w):
            assert np.array_equal(g, x), (fname, k, scaled, mol)
        assert len(w[0]) >= len(val) - 5
        d_res.free(); d_off.free()


def test_real_file_through_the_pipelined_ingest():
    path = os.path.join(GOLDEN, FILES[1])
    recs = wire.read_fasta(path)
    names, o, m, a, stats = host.sketch_fasta(path, 24, 5, "hp", validate=True, batch_residues=100_000)
    assert names == [n for n, _ in recs] and stats["batches"] >= 6
    res, offs = ks.pack([ks.validate_and_resolve(s, upper=True) for _, s in recs])
    wo, wm, wa = oracle.sketch_batch(res, offs, 24, 5, "hp", n_threads=8)
    assert np.array_equal(o, wo) and np.array_equal(m, wm) and np.array_equal(a, wa)
