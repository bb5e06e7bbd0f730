"""N>1 path on CPU: two gloo ranks run the sharding + hit-exchange layer (kmerseek_amd/dist.py).
The per-rank compute is injected from the oracle here (tests only); on a GPU box the same layer is
driven by the HIP path (see test_gpu_parity.py::test_dist_layer_single_rank_gpu)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _oracle_search_fn(k, scaled, mol):
    from oracle import oracle

    def fn(q_res, q_off, t_res, t_off):
        qo, qm, _ = oracle.sketch_batch(q_res, q_off, k, scaled, mol)
        to, tm, ta = oracle.sketch_batch(t_res, t_off, k, scaled, mol)
        return oracle.manysearch(qo, qm, to, tm, ta)
    return fn


def _worker(rank, world, port, mode, out_dir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from kmerseek_amd import dist as ksd, synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        k, scaled, mol = 7, 1, "protein"
        # only rank 0 generates the inputs; everybody receives them by broadcast
        if rank == 0:
            t_res, t_off = synth.proteome(120, stream=200)
            q_res, q_off = synth.queries(90, t_res, t_off, stream=201)
        else:
            t_res = t_off = q_res = q_off = None
        bt_res, bt_off = ksd.broadcast_batch(t_res, t_off, src=0)
        bq_res, bq_off = ksd.broadcast_batch(q_res, q_off, src=0)
        t_res, t_off = bt_res.numpy(), bt_off.numpy().view(np.uint64)
        q_res, q_off = bq_res.numpy(), bq_off.numpy().view(np.uint64)
        fn = _oracle_search_fn(k, scaled, mol)
        if mode == "queries":
            hits = ksd.search_queries_sharded(fn, q_res, q_off, t_res, t_off)
        else:
            hits = ksd.search_index_sharded(fn, q_res, q_off, t_res, t_off)
            # the two-phase form of the same exchange (begin: pack + start the collective, finish: complete it): on host
            # columns it completes at once, with the same rows
            s0, s1 = ksd.shard_by_residues(t_off, world)[rank]
            local = fn(q_res, q_off, *ksd.slice_batch(t_res, t_off, s0, s1))
            pend = ksd.begin_all_gather_hits_device(local, tid_base=s0, sharded="index", order="qid",
                                                    id_counts=(len(q_off) - 1, len(t_off) - 1))
            got = pend.finish()
            assert got is pend.finish()  # (idempotent)
            for g, w in zip(got, hits):
                assert np.array_equal(g.cpu().numpy().view(w.dtype), w)
        np.savez(os.path.join(out_dir, f"hits_{mode}_{rank}.npz"), qid=hits[0], tid=hits[1], isect=hits[2], nw=hits[3],
                 t_res=t_res, t_off=t_off, q_res=q_res, q_off=q_off)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["queries", "index"])
def test_two_rank_search_equals_single_rank(tmp_path, mode):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), mode, str(tmp_path)), nprocs=world, join=True)
    from oracle import oracle
    r0 = np.load(tmp_path / f"hits_{mode}_0.npz")
    r1 = np.load(tmp_path / f"hits_{mode}_1.npz")
    # every rank ends with the same, complete, (qid, tid)-sorted list
    for f in ("qid", "tid", "isect", "nw"):
        assert np.array_equal(r0[f], r1[f])
    want = _oracle_search_fn(7, 1, "protein")(r0["q_res"], r0["q_off"], r0["t_res"], r0["t_off"])
    assert len(want[0]) >= 15
    for g, w in zip((r0["qid"], r0["tid"], r0["isect"], r0["nw"]), want):
        assert np.array_equal(g, w)


def _synthetic_hits(rank, n):
    rng = np.random.default_rng(100 + rank)
    qid = np.sort(rng.integers(0, 1000, n)).astype(np.uint32)
    tid = rng.integers(0, 500, n).astype(np.uint32)
    isect = rng.integers(1, 60, n).astype(np.uint32)
    nw = isect.astype(np.uint64) + rng.integers(0, 5, n).astype(np.uint64)
    wide = rng.choice(n, size=max(1, n // 50), replace=False)  # rows whose values do not fit the transport word
    isect[wide[::2]] = 5000 + rng.integers(0, 1 << 20, len(wide[::2])).astype(np.uint32)
    nw[wide] = (1 << 40) + rng.integers(0, 1 << 20, len(wide)).astype(np.uint64)
    return qid, tid, isect, nw


def _packed_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from kmerseek_amd import dist as ksd
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        h = _synthetic_hits(rank, 3000 + 700 * rank)
        out = {}
        # (1M x 1M ids -> 12 value bits: the wide rows take the escape list); second call: escape list too short -> columns unpacked
        for tag, ids in (("packed", (1 << 20, 1 << 20)), ("plain", None), ("overflow", (1 << 23, 1 << 23))):
            if tag == "overflow":  # 9 value bits: every row with intersect >= 511 escapes... make most rows wide
                h2 = (h[0], h[1], h[2] + np.uint32(600), h[3] + np.uint64(600))
                got = ksd.all_gather_hits(h2, qid_base=rank * 1000, tid_base=7, id_counts=ids)
            else:
                got = ksd.all_gather_hits(h, qid_base=rank * 1000, tid_base=7, id_counts=ids)
            for name, col in zip(("qid", "tid", "isect", "nw"), got):
                out[f"{tag}_{name}"] = col
        np.savez(os.path.join(out_dir, f"packed_{rank}.npz"), **out)
    finally:
        dist.destroy_process_group()


def test_packed_transport_words_and_escape_list(tmp_path):
    """The 64-bit transport word of the hit exchange (8 instead of 20 bytes per row over the links): same rows as the
    unpacked columns, including rows whose values take the escape list and the case where the list is too short."""
    world = 2
    mp.spawn(_packed_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0, r1 = np.load(tmp_path / "packed_0.npz"), np.load(tmp_path / "packed_1.npz")
    want = [np.concatenate([_synthetic_hits(r, 3000 + 700 * r)[c] for r in range(world)]) for c in range(4)]
    want[0] = want[0] + np.repeat(np.arange(world, dtype=np.uint32) * 1000, [3000, 3700])
    want[1] = want[1] + np.uint32(7)
    for r in (r0, r1):
        for tag in ("packed", "plain"):
            for name, w in zip(("qid", "tid", "isect", "nw"), want):
                assert np.array_equal(r[f"{tag}_{name}"], w), (tag, name)
        assert np.array_equal(r["overflow_isect"], want[2] + np.uint32(600)) and np.array_equal(r["overflow_nw"], want[3] + np.uint64(600))
        assert np.array_equal(r["overflow_qid"], want[0])
    assert (want[3] > (1 << 39)).sum() > 50


def test_shard_by_residues_balances_and_covers():
    from kmerseek_amd import dist as ksd
    rng = np.random.default_rng(0)
    lens = rng.integers(0, 3000, 1000)
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    for world in (1, 2, 3, 8):
        sh = ksd.shard_by_residues(offs, world)
        assert sh[0][0] == 0 and sh[-1][1] == 1000
        assert all(a[1] == b[0] for a, b in zip(sh, sh[1:]))
        res = [int(offs[b] - offs[a]) for a, b in sh]
        assert max(res) - min(res) <= 2 * 3000
    # degenerate: fewer sequences than ranks, empty batch
    assert ksd.shard_by_residues(np.array([0, 5], np.uint64), 4)[-1][1] == 1
    assert ksd.shard_by_residues(np.array([0], np.uint64), 2) == [(0, 0), (0, 0)]
    r, o = ksd.slice_batch(np.arange(10, dtype=np.uint8), np.array([0, 3, 7, 10], np.uint64), 1, 3)
    assert r.tolist() == list(range(3, 10)) and o.tolist() == [0, 4, 7]


def test_all_gather_hits_single_rank_shifts_ids():
    # one rank: the (already (qid, tid)-ordered) local list comes back with its ids in global numbering
    from kmerseek_amd import dist as ksd
    h = (np.array([0, 1, 1], np.uint32), np.array([2, 1, 5], np.uint32), np.array([4, 5, 3], np.uint32),
         np.array([4, 6, 3 + (1 << 40)], np.uint64))
    q, t, i, w = ksd.all_gather_hits(h, qid_base=10, tid_base=100)
    assert q.tolist() == [10, 11, 11] and t.tolist() == [102, 101, 105] and i.tolist() == [4, 5, 3]
    assert w.tolist() == [4, 6, 3 + (1 << 40)]
    assert q.dtype == np.uint32 and w.dtype == np.uint64
    e = (np.zeros(0, np.uint32),) * 3 + (np.zeros(0, np.uint64),)
    assert all(len(x) == 0 for x in ksd.all_gather_hits(e))
