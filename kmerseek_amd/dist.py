"""Multi-GPU layer: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on ROCm;
"gloo" on CPU for tests).  The reference has no distributed code at all (its only parallelism is a
rayon par_iter, src/rust/index.rs:990-1005); the sharding below is this framework's own.

Two ways to shard a search, both without any data-path collective inside the hot loop:

* queries sharded, index replicated (BASELINE configs[3]): the index residues are broadcast once and every
  rank sketches + sorts them locally; each rank searches its own query range; hit sets are disjoint by qid.
* index sharded by TARGET id, queries replicated (BASELINE configs[4], all-vs-all): each rank indexes its target
  range and joins all queries against it; hit sets are disjoint by tid.  (Hash-range sharding would need a
  per-pair sum across ranks — a reduce-by-key over the network — and is rejected.)

Either way the only exchange is the concatenation of variable-length COO hit lists: a count all-gather
followed by one padded all-gather of device-resident blocks (``all_gather_hits_device``: the shard's hit columns go D2D
from the ``ks_hits`` object into the send block, no host staging, no host sort).
"""
from __future__ import annotations

import os
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np

Hits = Tuple[np.ndarray, np.ndarray, np.ndarray, np.ndarray]  # qid u32, tid u32, intersect u32, n_weighted u64


def shard_by_residues(offsets: np.ndarray, world: int) -> List[Tuple[int, int]]:
    """Contiguous sequence ranges [s0, s1) per rank with ~equal residue counts (not sequence counts)."""
    offsets = np.asarray(offsets, dtype=np.uint64)
    n = len(offsets) - 1
    total = int(offsets[-1]) if n > 0 else 0
    cuts = [0]
    for r in range(1, world):
        target = total * r // world
        s = int(np.searchsorted(offsets, np.uint64(target), side="left"))
        cuts.append(min(max(s, cuts[-1]), n))
    cuts.append(n)
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def slice_batch(residues: np.ndarray, offsets: np.ndarray, s0: int, s1: int) -> Tuple[np.ndarray, np.ndarray]:
    """Sub-batch of sequences [s0, s1) with offsets rebased to 0."""
    b, e = int(offsets[s0]), int(offsets[s1])
    return residues[b:e], (offsets[s0:s1 + 1] - offsets[s0]).astype(np.uint64)


def _dist():
    import torch.distributed as dist
    return dist


def world_info() -> Tuple[int, int]:
    dist = _dist()
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def _multi(world: int) -> bool:
    """Do the collectives run?  With more than one rank, always.  KS_DIST_FORCE_COLLECTIVES=1 makes a ONE-rank process group
    take the same path (count exchange, packed / unpacked all-gather, broadcasts): on a one-GPU box that is how the RCCL calls
    themselves — dtypes, device buffers, stream order against the context's launches, async_op / wait — are executed
    (tests/test_gpu_rccl.py, KS_BENCH_FORCE_PG); results are the same either way."""
    if world > 1:
        return True
    dist = _dist()
    return os.environ.get("KS_DIST_FORCE_COLLECTIVES") == "1" and dist.is_available() and dist.is_initialized()


def broadcast_batch(residues: Optional[np.ndarray], offsets: Optional[np.ndarray], src: int = 0, device=None):
    """Broadcast a (residues u8, offsets u64) batch from `src` to every rank; returns torch tensors on `device`
    (uint8, int64-viewed offsets).  With one rank it just moves the arrays to the device."""
    import torch
    dist = _dist()
    rank, world = world_info()
    dev = device if device is not None else torch.device("cpu")
    if rank == src:
        meta = torch.tensor([len(residues), len(offsets)], dtype=torch.int64, device=dev)
    else:
        meta = torch.zeros(2, dtype=torch.int64, device=dev)
    multi = _multi(world)
    if multi:
        dist.broadcast(meta, src)
    if rank == src:
        t_res = torch.from_numpy(np.ascontiguousarray(residues, dtype=np.uint8)).to(dev)
        t_off = torch.from_numpy(np.ascontiguousarray(offsets, dtype=np.uint64).view(np.int64)).to(dev)
    else:
        t_res = torch.empty(int(meta[0]), dtype=torch.uint8, device=dev)
        t_off = torch.empty(int(meta[1]), dtype=torch.int64, device=dev)
    if multi:
        dist.broadcast(t_res, src)
        dist.broadcast(t_off, src)
    return t_res, t_off


def _bits_for(n: int) -> int:
    return max(1, int(n - 1).bit_length()) if n > 1 else 1


def all_gather_hits_device(hits, qid_base: int = 0, tid_base: int = 0, device=None, sharded: str = "queries",
                           order: str = "qid", id_counts: Optional[Tuple[int, int]] = None):
    """Concatenate every rank's COO hit list on the device: counts are exchanged first, then ONE padded
    ``all_gather_into_tensor`` moves the rows (RCCL over xGMI with backend "nccl"; gloo on CPU tensors in the tests).

    hits      : a device-resident ``engine.Hits`` — its rows go D2D straight into the send block (no host round trip) —
                or a numpy 4-tuple of host columns (CPU tensors / gloo).
    sharded   : "queries" — ranks own ascending qid ranges, so the concatenation in rank order already IS (qid, tid) order;
                "index"   — ranks own ascending tid ranges: the concatenation is ordered by (shard, qid, tid).
    order     : "qid" (default) — for an index-sharded gather one stable device sort on qid restores global (qid, tid) order;
                "shard" — leave the rank-major concatenation as it is (every block (qid, tid)-ordered; what BASELINE configs[4]
                asks for: "all-gather of per-shard hit lists").
    id_counts : (n_queries, n_targets) of the WHOLE job.  With it the rows travel as 64-bit transport words
                (``ks_hits_pack64_to_device``: ids + both values in one word, rows with wide values on a short escape list):
                8 bytes per row over the links instead of 20 — the exchange, not the kernels, is the step of an all-vs-all
                on 8 GPUs.  Without it (or when the ids need more than 48 bits) the four columns travel as they are.
    Returns (qid i32, tid i32, intersect i32, n_weighted i64) torch tensors on `device`, identical on every rank."""
    import torch
    dist = _dist()
    rank, world = world_info()
    dev = device if device is not None else torch.device("cpu")
    on_device = hasattr(hits, "copy_to_device")
    if on_device and dev.type != "cuda":  # collective buffers on the host (gloo rehearsal): the columns have to come down
        hits, on_device = hits.to_host(), False
    n_local = int(hits.count) if on_device else len(hits[0])
    counts = [n_local]
    multi = _multi(world)
    if multi:
        mine = torch.tensor([n_local], dtype=torch.int64, device=dev)
        allc = torch.zeros(world, dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(allc, mine)
        counts = [int(c) for c in allc.tolist()]

    def own_stream_sync():
        if on_device and hits._ctx.stream != torch.cuda.current_stream(dev).cuda_stream:
            hits._ctx.synchronize()  # the copy ran on the context's own stream; the collective runs on torch's

    def torch_stream_sync():
        # ... and the other way round: a buffer torch allocated (and maybe filled, or whose memory an earlier torch kernel
        # still reads) is written next by a launch on the context's own stream — a non-blocking one, ordered with nothing
        if on_device and hits._ctx.stream != torch.cuda.current_stream(dev).cuda_stream:
            torch.cuda.current_stream(dev).synchronize()

    if id_counts is not None and n_local and (qid_base >= max(id_counts[0], 1) or tid_base >= max(id_counts[1], 1)):
        # (the packed exchange trusts id_counts: per row, ks_hits_pack64_to_device pushes the escape count beyond any capacity
        # when an id does not fit its field, and every rank then takes the unpacked exchange)
        raise ValueError(f"id bases ({qid_base}, {tid_base}) do not fit id_counts {tuple(id_counts)}")

    if (not multi and on_device and qid_base == 0 and tid_base == 0
            and hits._ctx.stream == torch.cuda.current_stream(dev).cuda_stream):
        # one rank, nothing to shift: the columns of the hit list themselves, as torch views (no 20-byte-per-row D2D copy:
        # 0.9 ms of a 3.9 ms step at 31 M rows); each tensor keeps the ks_hits object alive.  Only when the context launches
        # on torch's current stream: the block goes back to the context's pool when the last view dies, and the next ks_*
        # call may hand it out again — stream order is what keeps that call's kernels behind whatever torch kernels the
        # consumer queued on the views.  A context with a stream of its own gets copies (below).
        views = _hit_columns_as_torch(hits, n_local, dev)
        if views is not None:
            return views

    if multi and id_counts is not None:
        qbits, tbits = _bits_for(id_counts[0]), _bits_for(id_counts[1])
        if qbits + tbits <= 48:
            out = _gather_packed(hits, on_device, n_local, counts, qid_base, tid_base, qbits, tbits, dev, own_stream_sync, torch_stream_sync)
            if out is not None:
                qid, tid, isect, nw = out
                if sharded == "index" and order == "qid" and qid.numel():
                    qid, tid, isect, nw = _order_by_qid(hits, on_device, (qid, tid, isect, nw), counts, id_counts[0], dev,
                                                        own_stream_sync, torch_stream_sync)
                return qid, tid, isect, nw
            # (more rows with wide values than the escape list takes: the columns travel unpacked below)

    # one block per rank, SoA: qid[cap] | tid[cap] | intersect[cap] | n_weighted[cap] (as 2 x i32 each); with one rank
    # the block is exact, so its columns are the result (no second pass)
    cap = n_local if not multi else (max(max(counts), 1) + 63) // 64 * 64
    cap += cap & 1  # even: the i64 column of a block stays 8-byte aligned
    send = torch.empty(max(5 * cap, 2), dtype=torch.int32, device=dev)
    if n_local:
        if on_device:
            torch_stream_sync()
            base = send.data_ptr()
            hits.copy_to_device(base, base + 4 * cap, base + 8 * cap, base + 12 * cap, qid_base=qid_base, tid_base=tid_base)
            own_stream_sync()
        else:
            qid, tid, isect, nw = hits
            send[0:n_local] = torch.from_numpy((qid.astype(np.int64) + qid_base).astype(np.int32)).to(dev)
            send[cap:cap + n_local] = torch.from_numpy((tid.astype(np.int64) + tid_base).astype(np.int32)).to(dev)
            send[2 * cap:2 * cap + n_local] = torch.from_numpy(isect.astype(np.int32)).to(dev)
            send[3 * cap:5 * cap].view(torch.int64)[:n_local] = torch.from_numpy(nw.astype(np.int64)).to(dev)

    def columns(blk, c):
        return blk[0:c], blk[cap:cap + c], blk[2 * cap:2 * cap + c], blk[3 * cap:5 * cap].view(torch.int64)[:c]

    if not multi:
        return columns(send, n_local)
    recv = torch.empty(world * 5 * cap, dtype=torch.int32, device=dev)
    dist.all_gather_into_tensor(recv, send)
    cols = list(zip(*[columns(recv[r * 5 * cap:(r + 1) * 5 * cap], c) for r, c in enumerate(counts)]))
    qid, tid, isect, nw = (torch.cat(c) for c in cols)  # one pass: the blocks are padded to the largest shard
    if sharded == "index" and order == "qid" and qid.numel():
        qid, tid, isect, nw = _order_by_qid(hits, on_device, (qid, tid, isect, nw), counts, id_counts[0] if id_counts else None, dev,
                                            own_stream_sync, torch_stream_sync)
    return qid, tid, isect, nw


def _order_by_qid(hits, on_device, cols, counts, n_queries, dev, own_stream_sync, torch_stream_sync):
    """Rank blocks (each ordered by (qid, tid), target ranges ascending with the rank) -> one list ordered by (qid, tid).
    On the device with the library at hand: a counting merge (ks_hits_merge_by_qid_device — run lengths per (query, rank) by
    binary search, one scan, one move); otherwise a stable sort on qid (inside one qid, rank order already is tid order)."""
    import torch
    qid, tid, isect, nw = cols
    if on_device and dev.type == "cuda":
        if n_queries is None:
            n_queries = int(qid.max()) + 1
        out = (torch.empty_like(qid), torch.empty_like(tid), torch.empty_like(isect), torch.empty_like(nw))
        qid, tid, isect, nw = (c.contiguous() for c in cols)
        torch_stream_sync()  # the columns were produced on torch's stream; the merge runs on the context's
        hits._ctx.merge_hits_by_qid_device(qid.data_ptr(), tid.data_ptr(), isect.data_ptr(), nw.data_ptr(), counts, n_queries,
                                           *(o.data_ptr() for o in out))
        own_stream_sync()
        return out
    idx = torch.sort(qid, stable=True).indices
    return qid[idx], tid[idx], isect[idx], nw[idx]


class _DeviceColumn:
    """A device array owned by a library object, seen through __cuda_array_interface__ (torch.as_tensor makes a view of it
    and holds a reference to this object, which holds the owner)."""

    def __init__(self, ptr: int, n: int, typestr: str, owner, dev=None):
        self._owner, self._dev = owner, dev
        owner.pin()  # an explicit owner.free() waits for the views
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": typestr, "data": (int(ptr), False), "version": 2}

    def __del__(self):
        try:
            # The consumer may have queued torch kernels on the view.  They were ordered before the context's next launch when
            # the view was made (same stream); if torch's current stream is another one by now, wait for it before the block
            # can go back to the pool.
            import torch
            cur = torch.cuda.current_stream(self._dev)
            if cur.cuda_stream != self._owner._ctx.stream:
                cur.synchronize()
        except Exception:
            pass
        try:
            self._owner.unpin()
        except Exception:
            pass


def _hit_columns_as_torch(hits, n: int, dev):
    import torch
    if n == 0:
        return (torch.empty(0, dtype=torch.int32, device=dev), torch.empty(0, dtype=torch.int32, device=dev),
                torch.empty(0, dtype=torch.int32, device=dev), torch.empty(0, dtype=torch.int64, device=dev))
    try:
        ptrs = hits.device_ptrs()
        if not all(ptrs):
            return None
        return tuple(torch.as_tensor(_DeviceColumn(p, n, t, hits, dev), device=dev) for p, t in zip(ptrs, ("<i4", "<i4", "<i4", "<i8")))
    except Exception:  # (a torch build without the interface: the copy path below does the same job)
        return None


def sketch_columns_as_torch(sk, dev):
    """(offsets i64[n_seqs + 1], hashes i64[n_hashes], abunds i32[n_hashes]) of a device-resident ``engine.Sketches`` as torch
    views (same lifetime rules as the hit columns: the views pin the object)."""
    import torch
    po, ph, pa = sk.device_ptrs()
    n, m = sk.n_seqs, sk.n_hashes
    offs = torch.as_tensor(_DeviceColumn(po, n + 1, "<i8", sk, dev), device=dev)
    if m == 0:
        return offs, torch.empty(0, dtype=torch.int64, device=dev), torch.empty(0, dtype=torch.int32, device=dev)
    return offs, torch.as_tensor(_DeviceColumn(ph, m, "<i8", sk, dev), device=dev), torch.as_tensor(_DeviceColumn(pa, m, "<i4", sk, dev), device=dev)


class PendingGather:
    """An exchange whose collective has been started (``begin_all_gather_hits_device``): ``finish()`` waits for it and returns
    the gathered columns.  Between the two the caller may launch the next step's kernels — over xGMI the exchange, not the
    kernels, is the step of an index-sharded all-vs-all, and the collective (RCCL runs it on a stream of its own) moves the
    previous step's rows while the next step is sketched and joined."""

    def __init__(self, finish=None, result=None, release=None):
        self._finish, self._result, self._release = finish, result, release

    def finish(self):
        if self._finish is not None:
            try:
                self._result = self._finish()
            finally:
                self._finish = None
                self._drop()
        return self._result

    def _drop(self):
        if self._release is not None:
            rel, self._release = self._release, None
            rel()

    def __del__(self):  # (an exchange nobody completed: the hit list it kept for a repeat is let go)
        try:
            self._drop()
        except Exception:
            pass


def _gather_packed(hits, on_device, n_local, counts, qid_base, tid_base, qbits, tbits, dev, own_stream_sync, torch_stream_sync):
    """The exchange with 64-bit transport words, start to end.  None if some rank had more escapes than the lists take."""
    return _gather_packed_begin(hits, on_device, n_local, counts, qid_base, tid_base, qbits, tbits, dev, own_stream_sync,
                                torch_stream_sync, async_op=False)()


def _gather_packed_begin(hits, on_device, n_local, counts, qid_base, tid_base, qbits, tbits, dev, own_stream_sync, torch_stream_sync,
                         async_op):
    """Packs this rank's rows and STARTS the collective; returns the function that completes the exchange.
    Block of a rank (i64 words): packed[cap] | n_esc | esc_row (u32 x esc_cap) | esc_intersect (u32 x esc_cap) |
    esc_n_weighted (u64 x esc_cap).  The completion returns None if some rank had more escapes than esc_cap."""
    import torch
    dist = _dist()
    world = len(counts)
    v = (64 - qbits - tbits) // 2
    vmax = (1 << v) - 1
    cap = (max(max(counts), 1) + 63) // 64 * 64
    esc_cap = max(1024, cap // 64) // 2 * 2
    words = cap + 1 + 2 * esc_cap  # esc_row + esc_intersect: esc_cap / 2 words each; esc_n_weighted: esc_cap words
    send = torch.zeros(words, dtype=torch.int64, device=dev)
    if n_local:
        if on_device:
            torch_stream_sync()  # (the zero fill above ran on torch's stream)
            base = send.data_ptr()
            hits.pack64_to_device(base, base + 8 * (cap + 1), base + 8 * (cap + 1) + 4 * esc_cap, base + 8 * (cap + 1 + esc_cap),
                                  base + 8 * cap, esc_cap, qbits, tbits, qid_base=qid_base, tid_base=tid_base)
            own_stream_sync()
        else:
            qid, tid, isect, nw = (np.asarray(x).astype(np.uint64) for x in hits)
            esc = np.flatnonzero((isect >= vmax) | (nw >= vmax))
            a, b = isect.copy(), nw.copy()
            a[esc] = vmax; b[esc] = vmax
            w = ((((qid + np.uint64(qid_base)) << np.uint64(tbits)) | (tid + np.uint64(tid_base))) << np.uint64(2 * v)) | (a << np.uint64(v)) | b
            send[:n_local] = torch.from_numpy(w.view(np.int64)).to(dev)
            send[cap] = len(esc)
            k = min(len(esc), esc_cap)
            er = np.zeros(esc_cap, np.uint32); ei = np.zeros(esc_cap, np.uint32); en = np.zeros(esc_cap, np.uint64)
            er[:k] = esc[:k]; ei[:k] = isect[esc[:k]]; en[:k] = nw[esc[:k]]
            send[cap + 1:cap + 1 + esc_cap // 2] = torch.from_numpy(er.view(np.int64)).to(dev)
            send[cap + 1 + esc_cap // 2:cap + 1 + esc_cap] = torch.from_numpy(ei.view(np.int64)).to(dev)
            send[cap + 1 + esc_cap:] = torch.from_numpy(en.view(np.int64)).to(dev)
    recv = torch.empty(world * words, dtype=torch.int64, device=dev)
    work = dist.all_gather_into_tensor(recv, send, async_op=True) if async_op else dist.all_gather_into_tensor(recv, send)
    ctx = hits._ctx if on_device else None

    def finish():
        if work is not None and hasattr(work, "wait"):
            work.wait()  # (RCCL: the current stream waits for the collective; gloo: the host does)
        keep = send  # noqa: F841  (the send block lives until the collective is done)
        blocks = [recv[r * words:(r + 1) * words] for r in range(world)]
        n_esc = [int(x) & 0xffffffff for x in torch.stack([b[cap] for b in blocks]).tolist()]
        if max(n_esc) > esc_cap:
            return None
        total = sum(counts)
        if on_device and total:
            # one native pass per shard straight into the gathered columns (instead of a concatenation and a dozen
            # elementwise kernels over the whole list)
            qid = torch.empty(total, dtype=torch.int32, device=dev); tid = torch.empty(total, dtype=torch.int32, device=dev)
            isect = torch.empty(total, dtype=torch.int32, device=dev); nw = torch.empty(total, dtype=torch.int64, device=dev)
            torch_stream_sync()  # (the collective that filled `recv` ran on torch's side)
            at = 0
            for b, c in zip(blocks, counts):
                if c:
                    ctx.unpack_hits64_device(b.data_ptr(), c, qbits, tbits, qid.data_ptr() + 4 * at, tid.data_ptr() + 4 * at,
                                             isect.data_ptr() + 4 * at, nw.data_ptr() + 8 * at)
                at += c
            own_stream_sync()
        else:
            w = torch.cat([b[:c] for b, c in zip(blocks, counts)])
            qid = ((w >> (tbits + 2 * v)) & ((1 << qbits) - 1)).to(torch.int32)
            tid = ((w >> (2 * v)) & ((1 << tbits) - 1)).to(torch.int32)
            isect = ((w >> v) & vmax).to(torch.int32)
            nw = w & vmax
        off = 0
        for b, c, ne in zip(blocks, counts, n_esc):
            if ne:
                rows = b[cap + 1:cap + 1 + esc_cap // 2].view(torch.int32)[:ne].to(torch.int64) + off
                isect[rows] = b[cap + 1 + esc_cap // 2:cap + 1 + esc_cap].view(torch.int32)[:ne]
                nw[rows] = b[cap + 1 + esc_cap:cap + 1 + 2 * esc_cap][:ne]
            off += c
        return qid, tid, isect, nw
    return finish


def begin_all_gather_hits_device(hits, qid_base: int = 0, tid_base: int = 0, device=None, sharded: str = "queries",
                                 order: str = "qid", id_counts: Optional[Tuple[int, int]] = None) -> PendingGather:
    """``all_gather_hits_device`` in two halves: the rows are packed and the collective is STARTED here; ``finish()`` of the
    returned object delivers the gathered columns.  Only the transport-word exchange of device-resident hits on more than one
    rank really runs in the background (the count exchange before it is a few bytes); everything else completes here.
    ``hits.free()`` may be called as soon as this returns: the object is pinned until ``finish()`` (which repeats the exchange
    with unpacked columns when the escape lists overflow) and released then."""
    import torch
    rank, world = world_info()
    dev = device if device is not None else torch.device("cpu")
    on_device = hasattr(hits, "copy_to_device") and dev.type == "cuda"
    if not _multi(world) or not on_device or id_counts is None or _bits_for(id_counts[0]) + _bits_for(id_counts[1]) > 48:
        return PendingGather(result=all_gather_hits_device(hits, qid_base, tid_base, device, sharded, order, id_counts))
    dist = _dist()
    n_local = int(hits.count)
    mine = torch.tensor([n_local], dtype=torch.int64, device=dev)
    allc = torch.zeros(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(allc, mine)
    counts = [int(c) for c in allc.tolist()]
    ctx = hits._ctx

    def own_stream_sync():
        if ctx.stream != torch.cuda.current_stream(dev).cuda_stream:
            ctx.synchronize()

    def torch_stream_sync():
        if ctx.stream != torch.cuda.current_stream(dev).cuda_stream:
            torch.cuda.current_stream(dev).synchronize()

    qbits, tbits = _bits_for(id_counts[0]), _bits_for(id_counts[1])
    if n_local and (qid_base >= max(id_counts[0], 1) or tid_base >= max(id_counts[1], 1)):
        raise ValueError(f"id bases ({qid_base}, {tid_base}) do not fit id_counts {tuple(id_counts)}")
    # The hit list stays alive until finish(): when some rank has more rows with wide values than the escape lists take, the
    # exchange is repeated right there with the columns unpacked — the caller may still call hits.free() at once (pinned: the
    # release waits), as the docstring says.
    hits.pin()
    fin = _gather_packed_begin(hits, True, n_local, counts, qid_base, tid_base, qbits, tbits, dev, own_stream_sync, torch_stream_sync,
                               async_op=True)

    def finish():
        out = fin()
        if out is None:  # (every rank sees the same escape counts, so every rank takes this branch)
            return all_gather_hits_device(hits, qid_base, tid_base, device, sharded, order, id_counts=None)
        qid, tid, isect, nw = out
        if sharded == "index" and order == "qid" and qid.numel():
            qid, tid, isect, nw = _order_by_qid(hits, True, (qid, tid, isect, nw), counts, id_counts[0], dev, own_stream_sync,
                                                torch_stream_sync)
        return qid, tid, isect, nw
    return PendingGather(finish=finish, release=hits.unpin)


def all_gather_hits(hits, qid_base: int = 0, tid_base: int = 0, device=None, sharded: str = "queries",
                    id_counts: Optional[Tuple[int, int]] = None) -> Hits:
    """``all_gather_hits_device`` with the result copied to host numpy arrays (qid u32, tid u32, intersect u32, n_weighted u64)."""
    qid, tid, isect, nw = all_gather_hits_device(hits, qid_base, tid_base, device, sharded, id_counts=id_counts)
    return (qid.cpu().numpy().view(np.uint32), tid.cpu().numpy().view(np.uint32), isect.cpu().numpy().view(np.uint32),
            nw.cpu().numpy().view(np.uint64))


SearchFn = Callable[[np.ndarray, np.ndarray, np.ndarray, np.ndarray], Hits]


def search_queries_sharded(search_fn: SearchFn, q_res: np.ndarray, q_off: np.ndarray, t_res: np.ndarray,
                           t_off: np.ndarray, device=None, packed: bool = True) -> Hits:
    """Every rank holds all targets, takes its residue-balanced share of the queries, searches, and the
    disjoint-by-qid hit lists are concatenated.  search_fn(q_res, q_off, t_res, t_off) -> local COO."""
    rank, world = world_info()
    s0, s1 = shard_by_residues(q_off, world)[rank]
    lq_res, lq_off = slice_batch(q_res, q_off, s0, s1)
    return all_gather_hits(search_fn(lq_res, lq_off, t_res, t_off), qid_base=s0, device=device, sharded="queries",
                           id_counts=(len(q_off) - 1, len(t_off) - 1) if packed else None)


def search_index_sharded(search_fn: SearchFn, q_res: np.ndarray, q_off: np.ndarray, t_res: np.ndarray,
                         t_off: np.ndarray, device=None, packed: bool = True) -> Hits:
    """Every rank holds all queries and indexes its residue-balanced share of the TARGETS; hit lists are
    disjoint by tid and concatenated."""
    rank, world = world_info()
    s0, s1 = shard_by_residues(t_off, world)[rank]
    lt_res, lt_off = slice_batch(t_res, t_off, s0, s1)
    return all_gather_hits(search_fn(q_res, q_off, lt_res, lt_off), tid_base=s0, device=device, sharded="index",
                           id_counts=(len(q_off) - 1, len(t_off) - 1) if packed else None)


def gpu_search_fn(ctx, ksize: int, scaled: int, moltype: str) -> SearchFn:
    """The product search_fn: sketch both sides + index + search through the HIP library."""
    def fn(q_res, q_off, t_res, t_off) -> Hits:
        T = ctx.sketch_batch(t_res, t_off, ksize, scaled, moltype)
        Q = ctx.sketch_batch(q_res, q_off, ksize, scaled, moltype)
        ix = ctx.index_build(T)
        return ctx.search(ix, Q)  # device-resident: all_gather_hits_device takes the columns D2D
    return fn
