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
