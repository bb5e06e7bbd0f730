"""BASELINE.json configs[1], [2], [4] as parity-test cases at (or near) their stated sizes: full oracle comparison where
the pairwise CPU search finishes in seconds, otherwise oracle on a query sample plus size-independent properties
(symmetry of all-vs-all, self hits, (qid, tid) order, N-shard == 1-shard)."""
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


def gpu_search(ctx, q, t, k, scaled, mol):
    T = ctx.sketch_batch(t[0], t[1], k, scaled, mol)
    Q = ctx.sketch_batch(q[0], q[1], k, scaled, mol)
    return Q, T, ctx.search(ctx.index_build(T), Q).to_host()


def oracle_hits_for(q_sk, t_sk, qids):
    qo, qm, _ = q_sk
    to, tm, ta = t_sk
    sub_off = np.zeros(len(qids) + 1, np.uint64)
    parts = []
    for i, q in enumerate(qids):
        parts.append(qm[int(qo[q]):int(qo[q + 1])])
        sub_off[i + 1] = sub_off[i] + len(parts[-1])
    sub = np.concatenate(parts) if parts else np.zeros(0, np.uint64)
    oq, ot, oi, ow = oracle.manysearch(sub_off, sub, to, tm, ta, n_threads=16)
    return np.asarray(qids, np.uint32)[oq], ot, oi, ow


def test_config2_10k_vs_10k_protein_k7(ctx):
    """configs[1]: 10k synthetic proteins vs 10k index, k=7 scaled=1 protein — full pairwise oracle."""
    t = synth.proteome(10000, stream=20)
    q = synth.queries(10000, t[0], t[1], stream=21)
    Q, T, got = gpu_search(ctx, q, t, 7, 1, "protein")
    want_t = oracle.sketch_batch(t[0], t[1], 7, 1, "protein", n_threads=16)
    want_q = oracle.sketch_batch(q[0], q[1], 7, 1, "protein", n_threads=16)
    for g, w in zip(T.to_host() + Q.to_host(), want_t + want_q):
        assert np.array_equal(g, w)
    want = oracle.manysearch(want_q[0], want_q[1], want_t[0], want_t[1], want_t[2], n_threads=16)
    assert len(want[0]) > 2000
    for g, w in zip(got, want):
        assert np.array_equal(g, w)


def test_config3_100k_vs_100k_dayhoff_k16_s5(ctx):
    """configs[2]: Dayhoff k=16 scaled=5, 100k-vs-100k — sketches vs oracle in full, search vs oracle on a query sample."""
    t = synth.proteome(100000, stream=30)
    q = synth.queries(100000, t[0], t[1], stream=31)
    Q, T, got = gpu_search(ctx, q, t, 16, 5, "dayhoff")
    want_t = oracle.sketch_batch(t[0], t[1], 16, 5, "dayhoff", n_threads=16)
    want_q = oracle.sketch_batch(q[0], q[1], 16, 5, "dayhoff", n_threads=16)
    for g, w in zip(T.to_host() + Q.to_host(), want_t + want_q):
        assert np.array_equal(g, w)
    qid, tid, isect, nw = got
    key = qid.astype(np.uint64) << np.uint64(32) | tid.astype(np.uint64)
    assert np.all(key[1:] > key[:-1])
    sample = list(range(0, 100000, 211))  # one query in five is a mutated copy of a target, in random order
    oq, ot, oi, ow = oracle_hits_for(want_q, want_t, sample)
    sel = np.isin(qid, np.asarray(sample, np.uint32))
    assert np.array_equal(qid[sel], oq) and np.array_equal(tid[sel], ot)
    assert np.array_equal(isect[sel], oi) and np.array_equal(nw[sel], ow)
    assert len(oq) > 60


def test_config5_all_vs_all_hp_k24_sharded(ctx):
    """configs[4] shape (hp k=24, all-vs-all, index sharded by target id) at 30k proteins: symmetry, self hits, and
    8 target shards concatenated == unsharded; oracle on a query sample."""
    n = 30000
    p = synth.proteome(n, stream=40)
    k, scaled, mol = 24, 5, "hp"
    S = ctx.sketch_batch(p[0], p[1], k, scaled, mol)
    full = ctx.search(ctx.index_build(S), S).to_host()
    qid, tid, isect, nw = full
    o, m, a = S.to_host()
    sizes = (o[1:] - o[:-1]).astype(np.int64)
    # self hits carry the sketch size
    diag = qid == tid
    assert np.array_equal(np.sort(qid[diag]), np.flatnonzero(sizes > 0).astype(np.uint32))
    assert np.array_equal(isect[diag], sizes[qid[diag]].astype(np.uint32))
    # intersect is symmetric: the (t, q) row exists with the same count
    fwd = dict(zip(zip(qid.tolist(), tid.tolist()), isect.tolist()))
    assert all(fwd[(t_, q_)] == i_ for (q_, t_), i_ in list(fwd.items())[:20000])
    # index sharded by target id, 8 ways, hit lists concatenated (the exchange is a pure concatenation)
    parts = []
    for s0, s1 in ksd.shard_by_residues(p[1], 8):
        sub = ksd.slice_batch(p[0], p[1], s0, s1)
        Ti = ctx.sketch_batch(sub[0], sub[1], k, scaled, mol)
        h = ctx.search(ctx.index_build(Ti), S)
        # the device-resident exchange path (one rank here): columns D2D into the send block, ids shifted on the way
        parts.append(ksd.all_gather_hits(h, tid_base=s0, device=torch.device("cuda", 0), sharded="index"))
    rows = np.concatenate([np.stack([x.astype(np.int64) for x in part], axis=1) for part in parts])
    rows = rows[np.lexsort((rows[:, 1], rows[:, 0]))]
    assert np.array_equal(rows[:, 0], qid) and np.array_equal(rows[:, 1], tid)
    assert np.array_equal(rows[:, 2], isect) and np.array_equal(rows[:, 3].astype(np.uint64), nw)
    # oracle on a sample of queries
    sample = list(range(0, n, 211))
    oq, ot, oi, ow = oracle_hits_for((o, m, a), (o, m, a), sample)
    sel = np.isin(qid, np.asarray(sample, np.uint32))
    assert np.array_equal(qid[sel], oq) and np.array_equal(tid[sel], ot) and np.array_equal(isect[sel], oi)
    assert np.array_equal(nw[sel], ow)


def test_dist_layer_single_rank_gpu(ctx):
    """The sharding layer driven by the HIP search function (world size 1 here; 2-rank logic: tests/test_dist_gloo.py)."""
    t = synth.proteome(800, stream=60)
    q = synth.queries(600, t[0], t[1], stream=61)
    fn = ksd.gpu_search_fn(ctx, 10, 1, "protein")
    a = ksd.search_queries_sharded(fn, q[0], q[1], t[0], t[1])
    b = ksd.search_index_sharded(fn, q[0], q[1], t[0], t[1])
    want = fn(q[0], q[1], t[0], t[1]).to_host()
    assert len(want[0]) > 50
    for x, y, z in zip(a, b, want):
        assert np.array_equal(x, z) and np.array_equal(y, z)
    # device-resident exchange: same rows without the columns ever visiting the host
    dev = torch.device("cuda", 0)
    h = fn(q[0], q[1], t[0], t[1])
    got = ksd.all_gather_hits_device(h, qid_base=7, tid_base=9, device=dev)
    assert all(g.device.type == "cuda" for g in got)
    assert np.array_equal(got[0].cpu().numpy().view(np.uint32), want[0] + 7)
    assert np.array_equal(got[1].cpu().numpy().view(np.uint32), want[1] + 9)
    assert np.array_equal(got[2].cpu().numpy().view(np.uint32), want[2])
    assert np.array_equal(got[3].cpu().numpy().view(np.uint64), want[3])
    ptrs = h.device_ptrs()
    assert all(p != 0 for p in ptrs)
    # one rank, no id shift, but this context launches on a stream of its own: copies, not views (advisor r3: a view's block
    # goes back to the pool when the view dies, and only stream order protects the torch kernels still reading it)
    copies = ksd.all_gather_hits_device(h, device=dev)
    assert copies[0].data_ptr() != ptrs[0] and copies[3].data_ptr() != ptrs[3]
    for g, w_ in zip(copies, want):
        assert np.array_equal(g.cpu().numpy().view(w_.dtype), w_)
    # a context on torch's current stream: the columns themselves come back as torch views (no D2D copy); they keep the hit
    # list alive past an explicit free(), and the context alive past close()
    c2 = ks.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
    h2 = ksd.gpu_search_fn(c2, 10, 1, "protein")(q[0], q[1], t[0], t[1])
    ptrs2 = h2.device_ptrs()
    views = ksd.all_gather_hits_device(h2, device=dev)
    assert views[0].data_ptr() == ptrs2[0] and views[3].data_ptr() == ptrs2[3]
    h2.free()
    assert h2._h is not None  # (deferred: the views pin it)
    c2.close()
    assert c2._h is not None  # (deferred too)
    doubled = [v * 2 for v in views]  # torch kernels queued on the views
    for g, w_ in zip(views, want):
        assert np.array_equal(g.cpu().numpy().view(w_.dtype), w_)
    for g, w_ in zip(doubled, want):
        assert np.array_equal(g.cpu().numpy().view(w_.dtype), w_ * 2)
    del views, g
    assert h2._h is None and c2._h is None  # the last view took the hit list and the context with it
    # an id that does not fit its field is not packed into its neighbour: the escape count says "take the other exchange"
    n = h.count
    small = torch.zeros(n + 1 + 2 * 64, dtype=torch.int64, device=dev)
    torch.cuda.current_stream(dev).synchronize()
    sb = small.data_ptr()
    h.pack64_to_device(sb, sb + 8 * (n + 1), sb + 8 * (n + 1) + 4 * 64, sb + 8 * (n + 1 + 64), sb + 8 * n, 64, 4, 4, qid_base=7, tid_base=9)
    ctx.synchronize()
    assert (int(small[n]) & 0xffffffff) > 64
    # transport words of the multi-GPU exchange (8 instead of 20 bytes per row): packed on the device, unpacked as
    # dist._gather_packed does, rows with wide values through the escape list
    n = h.count
    for qbits, tbits in ((10, 10), (20, 20), (24, 24)):
        v = (64 - qbits - tbits) // 2
        vmax = (1 << v) - 1
        esc_cap = 4096
        buf = torch.zeros(n + 1 + 2 * esc_cap, dtype=torch.int64, device=dev)
        torch.cuda.current_stream(dev).synchronize()  # the fill ran on torch's stream, the pack runs on the context's own
        base = buf.data_ptr()
        h.pack64_to_device(base, base + 8 * (n + 1), base + 8 * (n + 1) + 4 * esc_cap, base + 8 * (n + 1 + esc_cap), base + 8 * n,
                           esc_cap, qbits, tbits, qid_base=7, tid_base=9)
        ctx.synchronize()
        w = buf[:n]
        ne = int(buf[n])
        isect = ((w >> v) & vmax).cpu().numpy().astype(np.uint64)
        nw = (w & vmax).cpu().numpy().astype(np.uint64)
        if ne:
            assert ne <= esc_cap
            rows = buf[n + 1:n + 1 + esc_cap // 2].view(torch.int32)[:ne].cpu().numpy().astype(np.int64)
            isect[rows] = buf[n + 1 + esc_cap // 2:n + 1 + esc_cap].view(torch.int32)[:ne].cpu().numpy().view(np.uint32)
            nw[rows] = buf[n + 1 + esc_cap:n + 1 + 2 * esc_cap][:ne].cpu().numpy().view(np.uint64)
        assert ne == int(((want[2] >= vmax) | (want[3] >= vmax)).sum())
        assert np.array_equal(((w >> (tbits + 2 * v)) & ((1 << qbits) - 1)).cpu().numpy(), want[0].astype(np.int64) + 7)
        assert np.array_equal(((w >> (2 * v)) & ((1 << tbits) - 1)).cpu().numpy(), want[1].astype(np.int64) + 9)
        assert np.array_equal(isect, want[2].astype(np.uint64)) and np.array_equal(nw, want[3])
    assert int(want[2].max()) >= 255  # (24 + 24 id bits leave 8 value bits: the related pairs escape)
    # the receiving side on the device (ks_hits_unpack64_device inside dist._gather_packed), with a stand-in for the
    # collective: two ranks that hold this same shard
    class _TwoRanks:
        @staticmethod
        def all_gather_into_tensor(recv, send):
            half = send.numel()
            recv[:half] = send
            recv[half:2 * half] = send
    real = ksd._dist
    ksd._dist = lambda: _TwoRanks
    try:
        for qbits, tbits in ((12, 12), (24, 24)):
            out = ksd._gather_packed(h, True, n, [n, n], 7, 9, qbits, tbits, dev, ctx.synchronize,
                                         lambda: torch.cuda.current_stream(dev).synchronize())
            assert out is not None
            for col, w_, add, dt in zip(out, want, (7, 9, 0, 0), (np.uint32, np.uint32, np.uint32, np.uint64)):
                got_col = col.cpu().numpy().view(dt)
                assert np.array_equal(got_col[:n], w_.astype(dt) + dt(add)) and np.array_equal(got_col[n:], w_.astype(dt) + dt(add))
    finally:
        ksd._dist = real


def test_pipelined_exchange_on_device_with_escape_overflow(monkeypatch):
    """dist.begin_all_gather_hits_device / PendingGather.finish on DEVICE tensors (advisor r3: the gloo tests pass host tuples
    and take the synchronous path): packed exchange started asynchronously, the hit list freed right after begin, the next
    step's kernels queued in between, unpack + counting merge inside finish() — and, with 8 value bits, more escapes than the
    lists take, so finish() repeats the exchange with unpacked columns by itself.  Two ranks that hold the same shard stand in
    for the collective (one GPU here)."""
    dev = torch.device("cuda", 0)

    class _Work:
        def wait(self):
            return True

    class _TwoRanks:
        calls = []

        @staticmethod
        def all_gather_into_tensor(recv, send, async_op=False):
            half = send.numel()
            recv[:half] = send
            recv[half:2 * half] = send
            _TwoRanks.calls.append((int(half), bool(async_op)))
            return _Work() if async_op else None

    monkeypatch.setattr(ksd, "world_info", lambda: (0, 2))
    monkeypatch.setattr(ksd, "_dist", lambda: _TwoRanks)
    for own_stream in (False, True):
        c = ks.Context(0, stream=None if own_stream else torch.cuda.current_stream(dev).cuda_stream)
        p = synth.proteome(3000, stream=70)
        S = c.sketch_batch(p[0], p[1], 7, 1, "protein")
        ix = c.index_build(S)
        for id_counts, overflow in (((3000, 3000), False), ((1 << 24, 1 << 24), True)):
            h = c.search(ix, S)
            want = h.to_host()
            n = h.count
            qbits, tbits = ksd._bits_for(id_counts[0]), ksd._bits_for(id_counts[1])
            vmax = (1 << ((64 - qbits - tbits) // 2)) - 1
            n_esc = int(((want[2] >= vmax) | (want[3] >= vmax)).sum())
            esc_cap = max(1024, ((n + 63) // 64 * 64) // 64) // 2 * 2
            assert (n_esc > esc_cap) == overflow, (n_esc, esc_cap)
            _TwoRanks.calls.clear()
            pend = ksd.begin_all_gather_hits_device(h, device=dev, sharded="index", order="qid", id_counts=id_counts)
            assert _TwoRanks.calls[-1][1] is True  # the data exchange was started asynchronously
            h.free()                               # allowed right away: the list is pinned until finish()
            assert h._h is not None
            nxt = c.search(ix, S)                  # the next step's kernels run between begin and finish
            got = pend.finish()
            assert h._h is None
            assert nxt.count == n
            nxt.free()
            if overflow:  # the repeat inside finish(): a count exchange and the unpacked columns, both synchronous
                assert [a for _, a in _TwoRanks.calls[-2:]] == [False, False]
            cat = [np.concatenate([w, w]) for w in want]
            order = np.argsort(cat[0], kind="stable")  # by qid, rank blocks in rank order inside a query
            for g, w_ in zip(got, cat):
                assert g.device.type == "cuda"
                assert np.array_equal(g.cpu().numpy().view(w_.dtype), w_[order])
        # an id beyond the declared range has no place in the merged order: the merge fails instead of leaving a gap
        h = c.search(ix, S)
        cols = [torch.empty(h.count, dtype=t, device=dev) for t in (torch.int32, torch.int32, torch.int32, torch.int64)]
        torch.cuda.current_stream(dev).synchronize()
        h.copy_to_device(*[x.data_ptr() for x in cols])
        c.synchronize()
        outs = [torch.empty_like(x) for x in cols]
        torch.cuda.current_stream(dev).synchronize()
        with pytest.raises(ks.KmerseekError) as e:
            c.merge_hits_by_qid_device(*[x.data_ptr() for x in cols], [h.count], 2000, *[x.data_ptr() for x in outs])
        assert "query id" in str(e.value)
        c.merge_hits_by_qid_device(*[x.data_ptr() for x in cols], [h.count], 3000, *[x.data_ptr() for x in outs])
        c.synchronize()
        assert all(torch.equal(a, b) for a, b in zip(cols, outs))
        h.free(); ix.free(); S.free()
        del cols, outs
        c.close()
