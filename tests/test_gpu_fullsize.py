"""BASELINE.json configs[3] and configs[4] at FULL size (1M-vs-1M protein k=10 scaled=1; 200k all-vs-all hp k=24 scaled=5):
the code paths only these sizes reach — 16-bit join prefix, the S=18 sort prefix of the index build, 256 high digits in the
bucket scatter, look-back chains over ~90k tiles — checked against the oracle on samples spread over the batch (first and
last tiles included) and through size-independent properties; the entry bench.py times (ks_sketch_search_device, called
with max_seq_len as the bench calls it) is held against the two-call result at the same size."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import kmerseek_amd as ks
from kmerseek_amd import dist as ksd, synth
from oracle import oracle


@pytest.fixture(scope="module")
def ctx():
    c = ks.Context(0, follow_debug_env=True)
    yield c
    c.close()


def _dev(ctx, a):
    return ctx.to_device(np.ascontiguousarray(a))


def _sample_ids(n, head=300, tail=300, spread=1500):
    ids = np.unique(np.concatenate([np.arange(min(head, n)), np.arange(max(n - tail, 0), n),
                                    np.linspace(0, n - 1, spread).astype(np.int64)]))
    return ids.astype(np.int64)


def _check_sketch_sample(batch, got, ids, k, scaled, mol):
    res, offs = batch
    o, m, a = got
    sub_res, sub_off = oracle.pack([bytes(res[int(offs[i]):int(offs[i + 1])]) for i in ids])
    wo, wm, wa = oracle.sketch_batch(sub_res, sub_off, k, scaled, mol, n_threads=16)
    for j, i in enumerate(ids):
        b, e = int(o[i]), int(o[i + 1])
        wb, we = int(wo[j]), int(wo[j + 1])
        assert e - b == we - wb, f"sequence {i}: {e - b} hashes, oracle {we - wb}"
        assert np.array_equal(m[b:e], wm[wb:we]) and np.array_equal(a[b:e], wa[wb:we]), f"sequence {i}"


def _oracle_rows(q_sk, t_sk, qids):
    qo, qm, _ = q_sk
    to, tm, ta = t_sk
    sub_off = np.zeros(len(qids) + 1, np.uint64)
    parts = []
    for j, q in enumerate(qids):
        parts.append(qm[int(qo[q]):int(qo[q + 1])])
        sub_off[j + 1] = sub_off[j] + len(parts[-1])
    oq, ot, oi, ow = oracle.manysearch(sub_off, np.concatenate(parts), to, tm, ta, n_threads=16)
    return np.asarray(qids, np.uint32)[oq], ot, oi, ow


def _check_one_call_entry(ctx, index, d_res, d_off, n, n_res, max_len, want_sk, want_rows, want_pairs, want_posting_bytes, defers):
    """The timed entry of bench.py at the size it is timed at: same sketches, same rows as sketch_queries_device + search
    (reference flow: src/python/kmerseek/sketch.py:28-40 then search.py:125-141), and no silent repeat."""
    f0, s0, k0 = ctx.fused_stats(), ctx.search_stats(), ctx.sketch_stats()
    Q1, H1 = ctx.sketch_search_device(index, d_res, d_off, n, n_res, max_seq_len=max_len)
    f1, s1, k1 = ctx.fused_stats(), ctx.search_stats(), ctx.sketch_stats()
    try:
        assert f1["redos"] == f0["redos"], "the one-call entry fell back to the two plain calls"
        assert f1["deferred"] - f0["deferred"] == (1 if defers else 0)
        assert s1 == s0 and k1 == k0, (s0, s1, k0, k1)
        assert Q1.posting_bytes == want_posting_bytes
        assert H1.partition_path == 1 and H1.n_pair_instances == want_pairs and H1.count == len(want_rows[0])
        for g, w in zip(Q1.to_host(), want_sk):
            assert np.array_equal(g, w)
        for g, w in zip(H1.to_host(), want_rows):
            assert np.array_equal(g, w)
    finally:
        H1.free(); Q1.free()


def _csr_properties(o, m, a, n_windows, scaled):
    assert o[0] == 0 and np.all(o[1:] >= o[:-1]) and int(o[-1]) == len(m) == len(a)
    inner = np.ones(len(m), bool)
    inner[o[:-1][o[:-1] < len(m)].astype(np.int64)] = False  # first hash of every non-empty sequence
    assert np.all(m[1:][inner[1:]] > m[:-1][inner[1:]]), "hashes ascend strictly inside every sequence"
    assert m.min() > 0 and m.max() <= oracle.max_hash(scaled)
    if scaled == 1:
        assert int(a.sum()) == n_windows  # every window is kept once


def test_baseline_configs3_1M_vs_1M_protein_k10_full_size(ctx):
    k, scaled, mol = 10, 1, "protein"
    n = 1_000_000
    t = synth.proteome(n, stream=0)
    q = synth.queries(n, t[0], t[1], stream=1000)
    d = [_dev(ctx, x) for x in (t[0], t[1], q[0], q[1])]
    try:
        T = ctx.sketch_batch_device(d[0].ptr, d[1].ptr, n, len(t[0]), k, scaled, mol)
        index = ctx.index_build(T)
        Q = ctx.sketch_queries_device(index, d[2].ptr, d[3].ptr, n, len(q[0]))
        assert Q.has_postings
        H = ctx.search(index, Q)
        assert H.partition_path == 1, "the fused postings + histogram-free bucket scatter is the path bench.py times"
        assert H.bucket_posting_bytes == 9  # (16 join-prefix bits: both prefix bytes are implied behind the scatter)
        qid, tid, isect, nw = H.to_host()
        n_pairs = H.n_pair_instances
        # ---- sketches vs the oracle on >= 2k sequences spread over each batch (first / last tiles included)
        t_sk, q_sk = T.to_host(), Q.to_host()
        _csr_properties(*t_sk, T.n_windows, scaled)
        _csr_properties(*q_sk, Q.n_windows, scaled)
        _check_sketch_sample(t, t_sk, _sample_ids(n), k, scaled, mol)
        _check_sketch_sample(q, q_sk, _sample_ids(n), k, scaled, mol)
        # ---- search vs the oracle: 64 sampled queries against ALL 1M targets (pairwise sorted merge, as manysearch does)
        sample = np.linspace(0, n - 1, 64).astype(np.int64)
        oq, ot, oi, ow = _oracle_rows(q_sk, t_sk, sample)
        sel = np.isin(qid, sample.astype(np.uint32))
        assert np.array_equal(qid[sel], oq) and np.array_equal(tid[sel], ot)
        assert np.array_equal(isect[sel], oi) and np.array_equal(nw[sel], ow)
        assert len(oq) >= 10
        # ---- size-independent properties
        key = qid.astype(np.uint64) << np.uint64(32) | tid.astype(np.uint64)
        assert np.all(key[1:] > key[:-1]), "(qid, tid) strictly ascending"
        assert int(isect.sum()) == n_pairs, "every matched posting pair is counted in exactly one row"
        assert np.all(nw >= isect) and len(qid) > 200_000
        # fused postings == plain search from the CSR (dense partition, path 3)
        Q2 = ctx.sketch_batch_device(d[2].ptr, d[3].ptr, n, len(q[0]), k, scaled, mol)
        assert not Q2.has_postings
        H2 = ctx.search(index, Q2)
        assert H2.partition_path == 3
        for g, w in zip(H2.to_host(), (qid, tid, isect, nw)):
            assert np.array_equal(g, w)
        H2.free(); Q2.free()
        # the entry bench.py times, as bench.py calls it (a 1M batch of <= 3000-aa proteins plans with one round trip: its
        # tile bound is too loose to launch on, so nothing is deferred — DESIGN.md 3.2)
        assert Q.posting_bytes == 10
        max_len = int((q[1][1:] - q[1][:-1]).max())
        _check_one_call_entry(ctx, index, d[2].ptr, d[3].ptr, n, len(q[0]), max_len, q_sk, (qid, tid, isect, nw), n_pairs, 10, defers=False)
        # the 3-pass partitioned index build == the 8-pass LSD sort (forced)
        os.environ["KS_DEBUG_INDEX_LSD"] = "1"
        try:
            index_lsd = ctx.index_build(T)
        finally:
            del os.environ["KS_DEBUG_INDEX_LSD"]
        H3 = ctx.search(index_lsd, Q)
        for g, w in zip(H3.to_host(), (qid, tid, isect, nw)):
            assert np.array_equal(g, w)
        H3.free(); index_lsd.free()
        H.free(); Q.free(); index.free(); T.free()
    finally:
        for b in d:
            b.free()


def test_baseline_configs4_200k_all_vs_all_hp_k24_full_size(ctx):
    k, scaled, mol = 24, 5, "hp"
    n = 200_000
    p = synth.proteome(n, stream=40)
    dr, do = _dev(ctx, p[0]), _dev(ctx, p[1])
    try:
        S = ctx.sketch_batch_device(dr.ptr, do.ptr, n, len(p[0]), k, scaled, mol)
        index = ctx.index_build(S)
        Q = ctx.sketch_queries_device(index, dr.ptr, do.ptr, n, len(p[0]))
        H = ctx.search(index, Q)
        qid, tid, isect, nw = H.to_host()
        sk = S.to_host()
        _csr_properties(*sk, S.n_windows, scaled)
        for g, w in zip(Q.to_host(), sk):
            assert np.array_equal(g, w)  # with and without fused postings: the same sketches
        _check_sketch_sample(p, sk, _sample_ids(n), k, scaled, mol)
        o = sk[0]
        sizes = (o[1:] - o[:-1]).astype(np.int64)
        key = qid.astype(np.uint64) << np.uint64(32) | tid.astype(np.uint64)
        assert np.all(key[1:] > key[:-1])
        assert int(isect.sum()) == H.n_pair_instances
        # self hits carry the sketch size; the matrix is symmetric in `intersect`
        diag = qid == tid
        assert np.array_equal(qid[diag], np.flatnonzero(sizes > 0).astype(np.uint32))
        assert np.array_equal(isect[diag], sizes[qid[diag]].astype(np.uint32))
        rkey = tid.astype(np.uint64) << np.uint64(32) | qid.astype(np.uint64)
        order = np.argsort(rkey, kind="stable")
        assert np.array_equal(rkey[order], key) and np.array_equal(isect[order], isect)
        # oracle on 64 sampled queries against all 200k targets
        sample = np.linspace(0, n - 1, 64).astype(np.int64)
        oq, ot, oi, ow = _oracle_rows(sk, sk, sample)
        sel = np.isin(qid, sample.astype(np.uint32))
        assert np.array_equal(qid[sel], oq) and np.array_equal(tid[sel], ot)
        assert np.array_equal(isect[sel], oi) and np.array_equal(nw[sel], ow)
        # the entry bench.py times (config4_index_sharded), as bench.py calls it: read-back deferred into the search's first wait
        max_len = int((p[1][1:] - p[1][:-1]).max())
        _check_one_call_entry(ctx, index, dr.ptr, do.ptr, n, len(p[0]), max_len, sk, (qid, tid, isect, nw), H.n_pair_instances, 12,
                              defers=True)
        # index sharded 4 ways by target id, hit lists exchanged through the device-resident path: gathered == unsharded
        dev = torch.device("cuda", 0)
        parts = []
        for s0, s1 in ksd.shard_by_residues(p[1], 4):
            sub = ksd.slice_batch(p[0], p[1], s0, s1)
            Ti = ctx.sketch_batch(sub[0], sub[1], k, scaled, mol)
            ix = ctx.index_build(Ti)
            h = ctx.search(ix, Q)
            parts.append(ksd.all_gather_hits_device(h, tid_base=s0, device=dev, sharded="index"))
            h.free(); ix.free(); Ti.free()
        cat = [torch.cat([part[c] for part in parts]) for c in range(4)]
        # what the N-rank gather does with the concatenated rank blocks for order="qid": the library's counting merge
        # (ks_hits_merge_by_qid_device) — each block is (qid, tid)-ordered, the target ranges ascend with the rank
        merged = ksd._order_by_qid(H, True, tuple(cat), [int(part[0].numel()) for part in parts], len(p[1]) - 1, dev,
                                   ctx.synchronize, lambda: torch.cuda.current_stream(dev).synchronize())
        got = [c.cpu().numpy() for c in merged]
        assert np.array_equal(got[0].view(np.uint32), qid) and np.array_equal(got[1].view(np.uint32), tid)
        assert np.array_equal(got[2].view(np.uint32), isect) and np.array_equal(got[3].view(np.uint64), nw)
        order = torch.sort(cat[0], stable=True).indices  # ... and the stable sort it replaces agrees
        assert all(torch.equal(c[order], m) for c, m in zip(cat, merged))
        H.free(); Q.free(); index.free(); S.free()
    finally:
        dr.free(); do.free()
