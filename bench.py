#!/usr/bin/env python3
"""bench.py — k-mers hashed+matched per second on MI355X, BASELINE.json's metric.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--scaling strong|weak] [--mode queries-sharded|index-sharded]

Main line (BASELINE.json configs[3], the config the metric is quoted on; it fits one GPU):
1M synthetic query proteins (~300 aa) searched against a 1M-protein index, protein k=10 scaled=1.
A "step" = one pass of the hot path over one query batch already resident in HBM:
    sketch (window -> re-encode -> murmur64 -> FracMinHash -> per-sequence sorted unique + abundance)
    + search (postings partition + join against the prebuilt index + per-pair reduce to COO hits).
The index (sketch + sort of the 1M targets) is built once, untimed — it is the "1M-seq index" of the metric —
and its build time is reported beside the number.

N > 1 (one process per GPU, RCCL): the config as BASELINE states it — "queries sharded across the GPUs, index
broadcast": the index residues are broadcast from rank 0 and every rank sketches + sorts them locally, the SAME 1M
queries are cut into N residue-balanced shards (strong scaling, the default; `--scaling weak` gives every rank its
own 1M-query batch instead).  No data-path collective: hit lists of query shards are disjoint.
value = query k-mer windows of all ranks x K / max-over-ranks time.

`config4_index_sharded` (same JSON line, measured after the main region; `--mode index-sharded` makes it the main
line): BASELINE configs[4] — all-vs-all over 200k proteins, hp k=24 scaled=5, index sharded by target id over the
ranks, every rank joins ALL queries against its shard and the per-shard hit lists are all-gathered on the device
(kmerseek_amd/dist.py: count exchange + one padded all_gather_into_tensor over RCCL/xGMI, no host staging).
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~5 TB/s measured device copy


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--mode", choices=["queries-sharded", "index-sharded"], default="queries-sharded")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong",
                    help="queries-sharded, N > 1: strong = the same --queries cut into N shards (BASELINE configs[3] as "
                         "written); weak = --queries per GPU")
    ap.add_argument("--queries", type=int, default=1_000_000, help="query proteins (total; per GPU with --scaling weak)")
    ap.add_argument("--targets", type=int, default=1_000_000, help="index proteins")
    ap.add_argument("--ksize", type=int, default=10)
    ap.add_argument("--scaled", type=int, default=1)
    ap.add_argument("--moltype", default="protein")
    ap.add_argument("--c4-proteins", type=int, default=200_000, help="proteins of the all-vs-all (configs[4]) workload")
    ap.add_argument("--no-config4", action="store_true", help="skip the configs[4] side measurement")
    ap.add_argument("--two-calls", action="store_true",
                    help="step = ks_sketch_queries_device + ks_search (three host waits) instead of ks_sketch_search_device (two)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-aux", action="store_true", help="skip the side measurements (sketch-only, end-to-end, device ceilings)")
    ap.add_argument("--cpu-sample-queries", type=int, default=0, help="0 = auto (aim at ~15 s of CPU search work)")
    return ap.parse_args()


def relaunch_distributed(args):
    """`python bench.py --gpus N` without a launcher: start the ranks as a child job (never exec) and return its exit code.
    The launcher's stderr is passed through line by line and its tail is shown again when the job failed."""
    import collections
    import threading
    port = 29500 + (os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    child = subprocess.Popen(cmd, stderr=subprocess.PIPE, text=True, errors="replace")
    tail = collections.deque(maxlen=40)

    def pump():
        for line in child.stderr:
            tail.append(line)
            sys.stderr.write(line)
    t = threading.Thread(target=pump, daemon=True)
    t.start()
    rc = child.wait()
    t.join(timeout=5)
    if rc != 0:
        sys.stderr.write(f"[bench] the {args.gpus}-rank job failed (exit code {rc}); last lines of the launcher's stderr:\n" + "".join(tail))
    return rc


class Env:
    """Ranks, devices and the barrier + max-over-ranks clock every timed region uses."""

    def __init__(self, args):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
        # KS_BENCH_REHEARSE=1: run the N-rank path on ONE GPU (all ranks on device 0, gloo instead of RCCL) to rehearse
        # the launcher / collective plumbing on a single-GPU box; numbers from such a run mean nothing.
        self.rehearse = os.environ.get("KS_BENCH_REHEARSE") == "1"
        self.dev_index = 0 if self.rehearse else local_rank
        torch.cuda.set_device(self.dev_index)
        self.dev = torch.device("cuda", self.dev_index)
        # one HIP stream for everything this process queues: torch's current stream for the whole run, and the stream the
        # library's context launches on (ks.Context(stream=...)) — torch ops on views of library-owned arrays, the library's
        # kernels and the collectives torch.distributed orders against the current stream are all in one queue order
        self.stream = torch.cuda.Stream(self.dev)
        torch.cuda.set_stream(self.stream)
        self.cdev = torch.device("cpu") if self.rehearse else self.dev  # where collective buffers live
        self.backend = None
        self.step_ms, self.tail_ms = [], 0.0
        # KS_BENCH_FORCE_PG=1 with ONE rank: a one-rank "nccl" process group, and every collective of the N-rank path is issued
        # (kmerseek_amd/dist.py: KS_DIST_FORCE_COLLECTIVES) — the RCCL calls of the N-GPU job executed on a one-GPU box
        # (tests/test_gpu_rccl.py).  Never the default: the plain N=1 run has no process group and no gather copy.
        self.force_pg = os.environ.get("KS_BENCH_FORCE_PG") == "1" and self.world == 1 and not self.rehearse
        self.multi = self.world > 1 or self.force_pg
        if self.force_pg:
            import socket
            with socket.socket() as s_:
                s_.bind(("127.0.0.1", 0))
                port = s_.getsockname()[1]
            for k_, v_ in (("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0"), ("MASTER_PORT", str(port))):
                os.environ.setdefault(k_, v_)
            os.environ["KS_DIST_FORCE_COLLECTIVES"] = "1"
        if self.multi:
            # Fail fast and say why: a rank that does not come up must not leave the others in the backend's default
            # 10-minute wait (the driver's limit for the whole bench is of that order: a hang would look like a slow run).
            from datetime import timedelta
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            self.backend = "gloo" if self.rehearse else "nccl"
            tmo = timedelta(seconds=int(os.environ.get("KS_BENCH_PG_TIMEOUT_S", "120")))
            what = "init_process_group"
            try:
                if self.rehearse:
                    dist.init_process_group("gloo", timeout=tmo)
                else:
                    dist.init_process_group("nccl", device_id=self.dev, timeout=tmo)
                what = "first collective (all_reduce of one word per rank)"
                t = torch.ones(1, dtype=torch.int64, device=self.cdev)
                dist.all_reduce(t)
                torch.cuda.synchronize(self.dev)
                if int(t[0]) != self.world:
                    raise RuntimeError(f"all_reduce over {self.world} ranks returned {int(t[0])}")
                # Every rank is up and the backend works: the short timeout has done its job.  From here on a rank may wait in
                # a collective for rank 0's untimed side work (CPU baseline, side lines, the unsharded search of the gate).
                try:
                    from torch.distributed.distributed_c10d import _set_pg_timeout
                    _set_pg_timeout(timedelta(seconds=int(os.environ.get("KS_BENCH_RUN_TIMEOUT_S", "900"))))
                except Exception as e:  # (an older torch: the short timeout stays)
                    sys.stderr.write(f"[bench] process-group timeout left at {tmo.total_seconds():.0f} s: {e!r}\n")
            except BaseException as e:  # (a plain exit of this child process with a non-zero code; never a re-exec)
                sys.stderr.write(f"[bench rank {self.rank}/{self.world} device cuda:{self.dev_index} backend {self.backend} "
                                 f"MASTER {os.environ.get('MASTER_ADDR')}:{os.environ.get('MASTER_PORT')}] {what} failed within "
                                 f"{tmo.total_seconds():.0f} s: {type(e).__name__}: {e}\n")
                sys.stderr.flush()
                os._exit(3)

    def barrier(self):
        if self.multi:
            self.dist.barrier()
        self.torch.cuda.synchronize(self.dev)

    def timed(self, fn, steps, finalize=None):
        """EXACTLY `steps` calls of fn between barrier + synchronize on both sides; max over ranks.  `finalize(out)` runs
        inside the timed region after the last call (a pipelined step completes its last exchange there).
        `self.step_ms` = this rank's host wall time of every call (a step returns after its last host wait, so these are
        step times, not enqueue times; one perf_counter read per step is the only thing added to the region)."""
        marks = [0.0] * (steps + 2)
        self.barrier()
        t0 = time.perf_counter()
        marks[0] = t0
        out = None
        for i in range(steps):
            out = fn()
            marks[i + 1] = time.perf_counter()
        if finalize is not None:
            out = finalize(out)
        self.barrier()
        marks[steps + 1] = time.perf_counter()
        el = marks[steps + 1] - t0
        self.step_ms = [(marks[i + 1] - marks[i]) * 1e3 for i in range(steps)]
        self.tail_ms = (marks[steps + 1] - marks[steps]) * 1e3  # finalize + closing barrier / synchronize
        if self.multi:
            t = self.torch.tensor([el], dtype=self.torch.float64, device=self.cdev)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            el = float(t[0])
        return el, out

    def step_times(self):
        """min / median / max of the per-step wall times of the last timed region (the median beside the mean)."""
        v = sorted(self.step_ms)
        if not v:
            return None
        n = len(v)
        med = v[n // 2] if n % 2 else 0.5 * (v[n // 2 - 1] + v[n // 2])
        return {"min": v[0], "median": med, "max": v[-1], "first": self.step_ms[0], "last": self.step_ms[-1],
                "finalize_and_barrier_ms": self.tail_ms, "n": n,
                "all": [round(x, 4) for x in self.step_ms]}

    def sum_ints(self, vals):
        t = self.torch.tensor(list(vals), dtype=self.torch.int64, device=self.cdev)
        if self.multi:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return [int(x) for x in t.tolist()]

    def close(self):
        if self.multi:
            self.dist.destroy_process_group()


def counters(ctx):
    """The library's own record of what it had to repeat or allocate (include/kmerseek_amd.h: ks_ctx_pool_stats,
    ks_ctx_fused_stats, ks_ctx_search_stats, ks_ctx_sketch_stats)."""
    return {"pool": ctx.pool_stats(), "fused": ctx.fused_stats(), "search": ctx.search_stats(), "sketch": ctx.sketch_stats()}


def counters_delta(before, after):
    """What moved inside a timed region: hipMalloc calls of the pool, pool growth, repeats; the steady state is all zeros
    (except `fused.deferred`, which counts the one-call steps that did NOT need a repeat)."""
    d = {"pool_mallocs": after["pool"]["mallocs"] - before["pool"]["mallocs"],
         "pool_bytes_held_before": before["pool"]["bytes_held"], "pool_bytes_held_after": after["pool"]["bytes_held"],
         "pool_blocks_before": before["pool"]["blocks"], "pool_blocks_after": after["pool"]["blocks"]}
    for grp in ("fused", "search", "sketch"):
        for k_, v in after[grp].items():
            d[f"{grp}.{k_}"] = v - before[grp][k_] if k_ != "uses_ticket" else v
    return d


def self_check(env, ksd, ctx, index, d_res, d_off, n, n_res, max_len):
    """Untimed, after the timed region: the entry the steps went through (ks_sketch_search_device) against the two plain calls
    (ks_sketch_queries_device + ks_search) on the same batch — same sketches, same rows, and the rows' intersect column sums
    to the matched posting pairs.  A bench line whose rows are wrong is not evidence (tests/test_gpu_fullsize.py holds both
    against the oracle at this size)."""
    torch = env.torch
    Q2 = ctx.sketch_queries_device(index, d_res, d_off, n, n_res, max_seq_len=max_len)
    H2 = ctx.search(index, Q2)
    Q1, H1 = ctx.sketch_search_device(index, d_res, d_off, n, n_res, max_seq_len=max_len)
    out = {"n_hashes": [Q1.n_hashes, Q2.n_hashes], "hits": [H1.count, H2.count],
           "matched_posting_pairs": [H1.n_pair_instances, H2.n_pair_instances]}
    ok = out["n_hashes"][0] == out["n_hashes"][1] and out["hits"][0] == out["hits"][1]
    if ok:
        r1, r2 = ksd._hit_columns_as_torch(H1, H1.count, env.dev), ksd._hit_columns_as_torch(H2, H2.count, env.dev)
        s1, s2 = ksd.sketch_columns_as_torch(Q1, env.dev), ksd.sketch_columns_as_torch(Q2, env.dev)
        out["rows_equal"] = all(bool(torch.equal(a, b)) for a, b in zip(r1, r2))
        out["sketches_equal"] = all(bool(torch.equal(a, b)) for a, b in zip(s1, s2))
        out["sum_intersect"] = int(r1[2].sum()) if H1.count else 0
        key = (r1[0].to(torch.int64) << 32) | r1[1].to(torch.int64)
        out["rows_strictly_ordered_by_qid_tid"] = bool((key[1:] > key[:-1]).all()) if key.numel() > 1 else True
        ok = (out["rows_equal"] and out["sketches_equal"] and out["sum_intersect"] == out["matched_posting_pairs"][0]
              and out["rows_strictly_ordered_by_qid_tid"])
        del r1, r2, s1, s2, key
    out["ok"] = bool(ok)
    H1.free(); Q1.free(); H2.free(); Q2.free()
    if not ok:
        raise AssertionError(f"bench self-check failed: one-call entry != two-call result: {out}")
    return out


def sharded_equals_unsharded(env, ksd, ctx, gathered, unsharded_hits):
    """SURVEY 8(e) correctness gate, untimed: the N-rank result (gathered columns in global (qid, tid) order, identical on
    every rank) against rank 0's own unsharded search of the whole job, bit for bit."""
    torch = env.torch
    want = ksd._hit_columns_as_torch(unsharded_hits, unsharded_hits.count, env.dev)
    got = [c.to(env.dev) for c in gathered]
    same = all(g.numel() == w.numel() and bool(torch.equal(g, w)) for g, w in zip(got, want))
    out = {"rows_gathered": int(got[0].numel()), "rows_unsharded": int(want[0].numel()), "equal": bool(same)}
    del want, got
    if not same and (env.rehearse or os.environ.get("KS_BENCH_STRICT") == "1"):
        raise AssertionError(f"sharded result != unsharded result: {out}")
    return out


def device_shard(env, res, off, s0, s1):
    """Sequences [s0, s1) of a device-resident batch as a batch of their own (fresh, 16-byte aligned buffers)."""
    torch = env.torch
    b, e = int(off[s0]), int(off[s1])
    r = torch.empty(max(e - b, 1), dtype=torch.uint8, device=env.dev)
    r[:e - b] = res[b:e]
    o = (off[s0:s1 + 1] - off[s0]).contiguous()
    return r, o, e - b


def window_stats(env, off, k):
    lens = (off[1:] - off[:-1])
    w = int(env.torch.clamp(lens - (k - 1), min=0).sum())
    return w, (int(lens.max()) if lens.numel() else 0)


# ---------------------------------------------------------------------------------------------------------------------
# BASELINE configs[3]: queries sharded, index replicated
# ---------------------------------------------------------------------------------------------------------------------
def run_queries_sharded(args, env, ks, synth, ksd):
    import numpy as np
    torch = env.torch
    k, scaled, mol = args.ksize, args.scaled, args.moltype
    rank, world = env.rank, env.world
    strong = args.scaling == "strong"

    # ---- synthetic inputs (seeded).  Index: rank 0 generates the residues, they are broadcast over
    # RCCL/xGMI and every rank sketches + sorts them locally (cheaper than shipping 3.5 GB of postings).
    t0 = time.time()
    t_res_h = t_off_h = q_res_h = q_off_h = None
    if rank == 0:
        t_res_h, t_off_h = synth.proteome(args.targets, stream=0)
    t_res, t_off = ksd.broadcast_batch(t_res_h, t_off_h, src=0, device=env.cdev)
    t_res, t_off = t_res.to(env.dev), t_off.to(env.dev)
    n_t_res = int(t_res.numel())
    if strong:
        # the SAME query set whatever N is: generated once, broadcast, cut into residue-balanced shards on the device
        if rank == 0:
            q_res_h, q_off_h = synth.queries(args.queries, t_res_h, t_off_h, stream=1000)
        qa_res, qa_off = ksd.broadcast_batch(q_res_h, q_off_h, src=0, device=env.cdev)
        qa_res, qa_off = qa_res.to(env.dev), qa_off.to(env.dev)
        off_host = qa_off.cpu().numpy().view(np.uint64)
        s0, s1 = ksd.shard_by_residues(off_host, world)[rank]
        q_res, q_off, n_q_res = device_shard(env, qa_res, qa_off, s0, s1)
        n_q = s1 - s0
        q_base = s0
        if not env.multi or rank != 0:
            del qa_res, qa_off  # (rank 0 of an N-rank job keeps the whole batch for the sharded == unsharded check)
    else:
        if rank != 0:
            t_res_h = t_res.cpu().numpy()
            t_off_h = t_off.cpu().numpy().view(np.uint64)
        q_res_h, q_off_h = synth.queries(args.queries, t_res_h, t_off_h, stream=1000 + rank)
        q_res = torch.from_numpy(q_res_h).to(env.dev)
        q_off = torch.from_numpy(q_off_h.view(np.int64)).to(env.dev)
        n_q, n_q_res = args.queries, len(q_res_h)
    q_windows, q_maxlen = window_stats(env, q_off, k)
    gen_s = time.time() - t0

    stream = torch.cuda.current_stream(env.dev)
    ctx = ks.Context(env.dev_index, stream=stream.cuda_stream)

    # ---- index build (once, untimed region; reported)
    torch.cuda.synchronize(env.dev)
    t0 = time.time()
    T = ctx.sketch_batch_device(t_res.data_ptr(), t_off.data_ptr(), args.targets, n_t_res, k, scaled, mol)
    index = ctx.index_build(T)
    torch.cuda.synchronize(env.dev)
    index_build_s = time.time() - t0
    # ... and once more with the pool warm (the first build pays for every hipMalloc of the context's pool)
    T.free(); index.free()
    torch.cuda.synchronize(env.dev)
    t0 = time.time()
    T = ctx.sketch_batch_device(t_res.data_ptr(), t_off.data_ptr(), args.targets, n_t_res, k, scaled, mol)
    index = ctx.index_build(T)
    torch.cuda.synchronize(env.dev)
    index_build_warm_s = time.time() - t0
    n_t_postings = index.n_postings

    posting_bytes = [12]  # bytes per partitioned query posting (10 against a big index at scaled = 1, else 12)
    bucket_bytes = [12]   # ... and inside the join buckets, behind the bucket scatter (9 when the join prefix has 16 bits)

    def step():
        # sketch for an immediate search: the sketch kernel also writes the postings pre-partitioned for the join; one call
        # (ks_sketch_search_device: the sketch's read-back rides on the search's first wait), sketches AND hits come back
        if args.two_calls:
            Q = ctx.sketch_queries_device(index, q_res.data_ptr(), q_off.data_ptr(), n_q, n_q_res, max_seq_len=q_maxlen)
            H = ctx.search(index, Q)
        else:
            Q, H = ctx.sketch_search_device(index, q_res.data_ptr(), q_off.data_ptr(), n_q, n_q_res, max_seq_len=q_maxlen)
        posting_bytes[0] = Q.posting_bytes or 12
        bucket_bytes[0] = H.bucket_posting_bytes or posting_bytes[0]
        out = (Q.n_hashes, H.count, H.n_pair_instances)
        H.free()
        Q.free()
        return out

    stats = None
    for _ in range(args.warmup):
        stats = step()

    # HIP events on the launch stream bracket the sketch tile kernel during the timed region (mode 2: a bracket around
    # every launch would add ~20 us of idle queue per launch to each step)
    ctx.timing_reset()
    ctx.timing_enable(2)
    c_before = counters(ctx)
    elapsed, stats = env.timed(step, args.steps)
    c_after = counters(ctx)
    step_times = env.step_times()
    ctx.timing_enable(False)
    timing = ctx.timing()
    # untimed extra pass with every launch bracketed, for the complete per-kernel table
    ctx.timing_reset()
    ctx.timing_enable(1)
    for _ in range(2):
        step()
    ctx.timing_enable(False)
    timing_all = {k_: (n_ / 2, ms_ / 2) for k_, (n_, ms_) in ctx.timing().items()}
    check = self_check(env, ksd, ctx, index, q_res.data_ptr(), q_off.data_ptr(), n_q, n_q_res, q_maxlen)
    gate = None
    if env.multi and strong:
        # every rank's hit list of its query shard, all-gathered (qid ranges ascend with the rank: the concatenation IS the
        # global order) == rank 0 searching all the queries by itself.  (A side check: an error in it — other than a mismatch
        # under KS_BENCH_REHEARSE / KS_BENCH_STRICT — is reported in the line, it does not cost the headline.)
        try:
            Q, H = ctx.sketch_search_device(index, q_res.data_ptr(), q_off.data_ptr(), n_q, n_q_res, max_seq_len=q_maxlen)
            got = ksd.all_gather_hits_device(H, qid_base=q_base, device=env.cdev, sharded="queries", id_counts=(args.queries, args.targets))
            H.free(); Q.free()
            if rank == 0:
                Qa, Ha = ctx.sketch_search_device(index, qa_res.data_ptr(), qa_off.data_ptr(), args.queries, int(qa_res.numel()))
                gate = sharded_equals_unsharded(env, ksd, ctx, got, Ha)
                Ha.free(); Qa.free()
                del qa_res, qa_off
            del got
        except AssertionError:
            raise
        except Exception as e:
            gate = {"error": repr(e)}

    # ---- SURVEY §8(d) side lines (rank 0, after the timed region; none of them is `value`)
    aux = None
    if rank == 0 and not args.no_aux:
        aux = {}
        torch.cuda.synchronize(env.dev)
        c0 = time.perf_counter()
        for _ in range(3):
            Q = ctx.sketch_batch_device(q_res.data_ptr(), q_off.data_ptr(), n_q, n_q_res, k, scaled, mol)
            Q.free()
        torch.cuda.synchronize(env.dev)
        aux["kmers_sketched_per_s_device_resident"] = 3 * q_windows / (time.perf_counter() - c0)
        # end to end from host buffers: H2D of residues + offsets, sketch, search, D2H of the hit rows (and of the CSR)
        ctx.sketch_batch(*ks.pack([b"ACDEFGHIKLMNPQRSTVWY" * 40] * 20000), k, scaled, mol).to_host()  # staging buffers exist
        qh_res = q_res[:n_q_res].cpu().numpy()
        qh_off = q_off.cpu().numpy().view(np.uint64)
        for rep in range(2):  # the first pass grows the context's pool for this (posting-free) path: the second one is timed
            if rep:
                H.free(); Q.free()
            c0 = time.perf_counter()
            Q = ctx.sketch_batch(qh_res, qh_off, k, scaled, mol)
            H = ctx.search(index, Q)
            rows = H.to_host()
            c1 = time.perf_counter()
        csr = Q.to_host()                      # pageable numpy arrays: staged through pinned buffers by host threads
        c2p = c2 = time.perf_counter()
        # pinned arrays (ks_host_alloc), allocated beforehand as a caller that streams batches would: one DMA each
        pins = (ctx.pinned_empty(len(csr[0]), np.uint64), ctx.pinned_empty(len(csr[1]), np.uint64), ctx.pinned_empty(len(csr[2]), np.uint32))
        c2 = time.perf_counter()
        ctx._check(ctx._L.ks_sketches_copy_to_host(ctx._h, Q._h, *[p_.ctypes.data_as(ks._lib.C.c_void_p) for p_ in pins]))
        c3 = time.perf_counter()
        del pins
        aux["end_to_end_host_buffers"] = {
            "kmers_per_s_hits_to_host": q_windows / (c1 - c0), "kmers_per_s_hits_and_sketches_to_host": q_windows / (c2p - c0),
            "kmers_per_s_hits_and_sketches_to_pinned_host": q_windows / (c1 - c0 + c3 - c2),
            "sketches_d2h_gb_per_s_pageable": sum(a.nbytes for a in csr) / (c2p - c1) / 1e9,
            "sketches_d2h_gb_per_s_pinned": sum(a.nbytes for a in csr) / (c3 - c2) / 1e9,
            "h2d_bytes": int(qh_res.nbytes + qh_off.nbytes), "d2h_hit_bytes": int(sum(a.nbytes for a in rows)),
            "d2h_sketch_bytes": int(sum(a.nbytes for a in csr)),
            "note": "single call from pageable host arrays; never the headline value"}
        del rows, csr
        H.free()
        Q.free()
        try:
            r = ctx.device_rates()
            # (hipDeviceProp_t gives memoryClockRate x memoryBusWidth; HBM3E moves 4 bits per pin and reported clock, so the
            # double-data-rate product of the two is HALF the pin rate: both printed, the 8 TB/s constant is what frac uses)
            aux["device"] = {"name": torch.cuda.get_device_name(env.dev),
                             "hbm_gb_per_s_from_properties": {"clock_x_buswidth_x2": r["nominal_gb_per_s"],
                                                              "clock_x_buswidth_x4_hbm3e_pin_rate": 2.0 * r["nominal_gb_per_s"]},
                             "d2d_copy_gb_per_s_measured": r["copy_gb_per_s"], "u64_gmul_per_s_measured": r["u64_gmul_per_s"]}
            aux["device"].update(ctx.gather_rates())  # ceiling of a table-lookup hash for the hp alphabet
        except Exception as e:  # a side line must not cost the headline
            aux["device"] = {"error": str(e)}

    all_windows, all_queries, all_hits, all_q_hashes, all_pairs = env.sum_ints(
        [q_windows, n_q, stats[1], stats[0], stats[2]])
    if rank != 0:
        return None, (ctx, index, T)

    n_q_hashes, n_hits, n_pairs = stats
    value = all_windows * args.steps / elapsed
    ms_per_step = elapsed / args.steps * 1e3

    # ---- roofline (HIP-event durations on the launch stream, this process = rank 0's shard)
    # algorithmic bytes EXCLUDE scratch / sort passes.  Sketch, per launch: residues read + 12 B per unique
    # kept hash written + 8 B offset per sequence.  The query launches also write the pre-partitioned postings the search
    # starts from (12 B per kept hash; 10 B against a big index at scaled = 1, where the region implies 8 hash bits): real
    # traffic of this kernel, but a scratch pass of the search — kept apart.
    pb, bb = posting_bytes[0], bucket_bytes[0]
    sketch_bytes = n_q_res + 12 * n_q_hashes + 8 * n_q
    design_bytes = {
        "sketch_tiles": sketch_bytes,
        # scratch passes of the search (not §8(d) bytes): what each launch has to move by design
        "bucket_scatter": (pb + bb) * n_q_hashes,                         # one posting in, one out
        "radix_hist.qpart": 8 * n_q_hashes,
        # query postings read once + the index side once — 4-byte fingerprints for big indexes (>= 2^15 join buckets: the
        # 16-byte posting is fetched per candidate match only), 12-byte postings otherwise — + one 8-B record per match
        "join_buckets": bb * n_q_hashes + ((4 * n_t_postings + 16 * n_pairs) if (n_t_postings >> 14) > 3072 else 12 * n_t_postings) + 8 * n_pairs,
    }
    traffic_tab, traffic_src = load_traffic(profile_key(args, n_q))

    def roof(name):
        # the dominant kernel is bracketed by HIP events inside the timed region; the others come from the untimed pass with
        # every launch bracketed (bracketing them in the timed region would tax every step, see ks_timing_enable)
        live = name in timing and timing[name][0] > 0
        n_l, ms = timing[name] if live else (timing_all[name][0] * args.steps, timing_all[name][1] * args.steps)
        avg_s = ms / n_l / 1e3
        b = design_bytes.get(name)
        ach = (b / avg_s / 1e9) if (b and avg_s > 0) else None
        r = {"kernel": name, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
             "frac": (ach / HBM_PEAK_GBS) if ach else None, "traffic": traffic_tab.get(name), "traffic_source": traffic_src,
             "avg_launch_ms": avg_s * 1e3, "launches_per_step": n_l / args.steps,
             "duration_source": "HIP events inside the timed region" if live else "HIP events, untimed pass after the timed region"}
        if name == "sketch_tiles":
            r["algorithmic_bytes_per_launch"] = b
            r["algorithmic_bytes_formula"] = "n_res + 12*n_hashes + 8*n_seqs (SURVEY 8(d): L + 12*U + 8 per sequence)"
            r["fused_scratch_bytes"] = pb * n_q_hashes  # postings for the join, written by the same launch; not in `achieved`
            r["posting_bytes"] = pb
            r["bucket_posting_bytes"] = bb
            r["achieved_incl_fused_scratch"] = (b + pb * n_q_hashes) / avg_s / 1e9
            # What this kernel actually runs against (informational; `bound` stays the contract's "hbm"): the vector ALU.  Every
            # wave of the query launch issues ~1,620 vector instructions (phase ladder of round 4, DESIGN.md 3.1: hash + placement
            # 613, scan 235 -> ~170, scatter 62, ordering 131, look-back + offsets 80, CSR write 106, postings 318); on gfx950 a SIMD
            # issues one wave64 instruction per ~4.4 cycles whether it is a 32-bit multiply, v_mad_u64_u32 or a three-operand
            # add (tools/gpu/valu_rates.hip: multiplies are FULL rate — rounds 2-4 priced them at half rate), ~2.4 for two-operand ops.
            if (k, scaled, mol) == (10, 1, "protein"):
                props = torch.cuda.get_device_properties(env.dev)
                simds, clk = props.multi_processor_count * 4, 2.4e9
                waves = 8.0 * (n_q / 12.7)  # ~12.7 proteins per packed tile, 8 waves per tile (629k waves per 1M proteins)
                floor_s = waves * 1620 * 4.0 / simds / clk
                r["valu_model"] = {"vector_instructions_per_wave": 1620, "cycles_per_instruction_assumed": 4.0, "simds": simds,
                                   "clock_hz_assumed": clk, "waves_estimated": waves,
                                   "vector_issue_floor_ms": floor_s * 1e3, "launch_over_vector_issue_floor": avg_s / floor_s,
                                   "note": "SQ counters: ~1,620 vector instructions per wave, all full rate (tools/gpu/valu_rates.hip); six waves per SIMD keep "
                                           "the vector ALU ~80 % issued (profiles/r04_sq_counters.md): what the launch has left is instructions, "
                                           "at roughly half a per cent of time per per cent of them (measured, DESIGN.md 3.1)"}
        else:
            r["design_bytes_per_launch"] = b
            r["note"] = "scratch pass of the search: bytes it has to move by design, not SURVEY 8(d) algorithmic bytes"
        return r

    # dominant kernel = largest share of a step (full per-kernel table); on BASELINE's headline config that is the sketch
    # tile kernel, whose launches are the ones bracketed live in the timed region
    dom = max(timing_all.items(), key=lambda kv: kv[1][1])[0] if timing_all else None
    roofline = roof(dom) if dom else None
    roofline_others = [roof(n) for n in design_bytes if n in timing_all and n != dom]
    # whole step against the roofline: §8(d) bytes of sketch + search (12 B per query posting + 12 B per index posting
    # read once + 16 B per COO hit row written) over the measured step time
    step_bytes = sketch_bytes + 12 * n_q_hashes + 12 * n_t_postings + 16 * n_hits
    step_roofline = {"bound": "hbm", "algorithmic_bytes_per_step": step_bytes,
                     "formula": "sketch (n_res + 12*N_Q + 8*n_seqs) + search (12*N_Q + 12*N_T + 16*N_hits), SURVEY 8(d)",
                     "achieved": step_bytes / (ms_per_step / 1e3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": step_bytes / (ms_per_step / 1e3) / 1e9 / HBM_PEAK_GBS}
    per_kernel = {name: {"launches_per_step": n, "ms_per_step": ms} for name, (n, ms) in timing_all.items()}

    cpu = None
    if not args.no_cpu_baseline and args.gpus == 1:
        qh_res = q_res[:n_q_res].cpu().numpy()
        qh_off = q_off.cpu().numpy().view(np.uint64)
        cpu = cpu_baseline(args, T, qh_res, qh_off, k, scaled, mol)

    sharding = (f"the same {args.queries // 1000}k queries cut into {world} residue-balanced shards" if strong
                else f"{args.queries // 1000}k queries per GPU")
    result = {
        "metric": "k-mers hashed+matched/sec", "value": value, "unit": "k-mers/s", "n_gpus": args.gpus,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
        "step_ms": step_times, "timed_region_counters": counters_delta(c_before, c_after), "self_check": check,
        "sharded_equals_unsharded": gate,
        "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "u64",
        "data": "synthetic",
        "config": {"workload": f"{args.queries // 1000}k query proteins vs {args.targets // 1000}k-protein index, {mol} k={k} "
                               f"scaled={scaled}" + (" (BASELINE configs[3]: 1M-vs-1M, k=10 scaled=1 protein, queries "
                                                     "sharded across the GPUs, index replicated by broadcast)"
                                                     if (args.queries, args.targets, k, scaled, mol) ==
                                                     (1_000_000, 1_000_000, 10, 1, "protein") else ""),
                   "queries": all_queries, "targets": args.targets, "ksize": k, "scaled": scaled, "moltype": mol,
                   "entry": "ks_sketch_queries_device + ks_search" if args.two_calls else "ks_sketch_search_device",
                   "parallelism": f"queries sharded x{world} ({sharding}), index replicated",
                   "n_ranks": world, "collective_backend": env.backend},
        "query_proteins_per_s": all_queries * args.steps / elapsed,
        "query_windows": all_windows, "query_hashes": all_q_hashes, "index_postings": n_t_postings,
        "hits": all_hits, "matched_posting_pairs": all_pairs,
        "index_build_s": index_build_s, "index_build_warm_s": index_build_warm_s, "datagen_s": gen_s,
        "roofline": roofline, "step_roofline": step_roofline, "cpu_baseline": cpu,
        "roofline_other_kernels": roofline_others, "kernels": per_kernel, "aux": aux,
    }
    if aux and isinstance(aux.get("device"), dict) and aux["device"].get("u64_gmul_per_s_measured") and "sketch_tiles" in timing:
        # integer-ALU line of the sketch kernel: 64-bit multiplies MurmurHash3 needs per window (8 at k = 10)
        muls = 4 * (k // 16) + (2 if k % 16 > 0 else 0) + (2 if k % 16 > 8 else 0) + 4  # blocks, tail k1 / k2, two fmix64
        n_l, ms = timing["sketch_tiles"]
        ach = q_windows * muls / (ms / n_l / 1e3) / 1e9
        result["alu_roofline_sketch_tiles"] = {"u64_mul_per_window": muls, "achieved_gmul_per_s": ach,
                                               "peak_gmul_per_s": aux["device"]["u64_gmul_per_s_measured"],
                                               "frac": ach / aux["device"]["u64_gmul_per_s_measured"]}
    return result, (ctx, index, T)


def profile_key(args, n_q_local):
    """Name of the profiled workload (tools/gpu/profile_configs.sh) this run matches, or None: PMC traffic is per workload."""
    w = (args.queries, args.targets, args.ksize, args.scaled, args.moltype)
    if args.gpus != 1:
        return None
    return {(1_000_000, 1_000_000, 10, 1, "protein"): "c4_1m_protein_k10_s1", (10_000, 10_000, 7, 1, "protein"): "c2_10k_protein_k7_s1",
            (100_000, 100_000, 16, 5, "dayhoff"): "c3_100k_dayhoff_k16_s5", (125_000, 1_000_000, 10, 1, "protein"): "shard_125k_of_1m"}.get(w)


def load_traffic(key="c4_1m_protein_k10_s1"):
    """HBM bytes per launch from the PMC passes committed under profiles/ (FETCH_SIZE / WRITE_SIZE, separate rocprofv3
    --pmc runs, gfx950 corrections applied by tools/pmc_to_traffic.py).  They are NOT measured by this run: the source
    file, the commit it was collected at and whether the kernel sources changed since are reported beside the number.
    One table per profiled workload (`per_config`); a run on any other workload reports no traffic."""
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if key is None or not os.path.exists(tpath):
        return {}, None
    try:
        doc = json.load(open(tpath))
    except Exception:
        return {}, None
    src = {"file": "profiles/traffic.json", "collected_at_commit": doc.get("commit"), "measured_by_this_run": False}
    want = doc.get("kernel_sources_sha16")
    if want:
        src["kernel_sources_unchanged_since"] = (kernel_sources_sha() == want)
    if "per_config" in doc:
        src["workload"] = key
        return (doc["per_config"].get(key) or {}).get("per_launch_bytes", {}), src
    return (doc.get("per_launch_bytes", {}) if key == "c4_1m_protein_k10_s1" else {}), src


def kernel_sources_sha():
    h = hashlib.sha256()
    d = os.path.join(ROOT, "kmerseek_amd", "csrc")
    for f in ("ks_sketch.hip", "ks_search.hip", "ks_prims.hip", "ks_msd.hip", "ks_device.h"):
        p = os.path.join(d, f)
        if os.path.exists(p):
            h.update(open(p, "rb").read())
    return h.hexdigest()[:16]


def reference_binary_probe(q_res_h, q_off_h, k, scaled, mol, n_records=20000):
    """when the reference's own binary is at hand (`kmerseek-rust` on PATH or $KMERSEEK_RUST_BIN — the
    crate's [[bin]], Cargo.toml:13-15), `kmerseek-rust index` is timed on a FASTA of the same proteins (src/rust/main.rs:17-46).
    Building it here is not possible (no reference sources, registry or network on the GPU box), so `cargo` alone changes
    nothing: the probe records what it found and otherwise stays out of the way."""
    import shutil
    import subprocess
    import tempfile
    exe = os.environ.get("KMERSEEK_RUST_BIN") or shutil.which("kmerseek-rust")
    out = {"probed": True, "binary": exe, "cargo": shutil.which("cargo")}
    if not exe:
        out["note"] = "reference binary not present: the CPU baseline is the oracle (kind 'port')"
        return out
    try:
        n = min(n_records, len(q_off_h) - 1)
        with tempfile.TemporaryDirectory() as d:
            fa = os.path.join(d, "sample.fasta")
            with open(fa, "wb") as f:
                for i in range(n):
                    f.write(b">q%d\n" % i + bytes(q_res_h[int(q_off_h[i]):int(q_off_h[i + 1])]) + b"\n")
            c0 = time.perf_counter()
            r = subprocess.run([exe, "index", "--input", fa, "--output", os.path.join(d, "db"), "--ksize", str(k), "--scaled", str(scaled),
                                "--encoding", mol], capture_output=True, timeout=600)
            dt = time.perf_counter() - c0
        windows = int(sum(max(int(q_off_h[i + 1] - q_off_h[i]) - k + 1, 0) for i in range(n)))
        out.update({"rc": r.returncode, "records": n, "seconds": dt, "kmers_per_s": windows / max(dt, 1e-9) if r.returncode == 0 else None})
    except Exception as e:  # a side line must not cost the headline
        out["error"] = str(e)
    return out


def cpu_baseline(args, T, q_res_h, q_off_h, k, scaled, mol):
    """The oracle (C restatement of the reference CPU path, kind "port") on bounded samples of the same workload:
    sketch half (add_protein, signature.rs:273-282) and the second process_kmers pass (index.rs:749-786) on >= 100k query
    proteins — seconds of work, not milliseconds — and the O(|Q|*|T|) pairwise manysearch (search.py:125-141) on a small,
    evenly spaced query sample against ALL targets.  Rates are combined per k-mer window: 1 / (t_sketch + t_search)."""
    import numpy as np
    from oracle import oracle
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16))  # a 1-GPU box's CPU share is 16 cores
    n_q = len(q_off_h) - 1
    # ---- sketch half + process_kmers pass: the first min(n_q, 200k) query proteins
    n_sk = min(n_q, 200_000)
    sk_res, sk_off = q_res_h[:int(q_off_h[n_sk])], q_off_h[:n_sk + 1]
    sk_windows = int(np.maximum((sk_off[1:] - sk_off[:-1]).astype(np.int64) - k + 1, 0).sum())
    oracle.sketch_batch(sk_res[:int(sk_off[min(n_sk, 64)])], sk_off[:min(n_sk, 64) + 1], k, scaled, mol, n_threads=cores)  # warm
    c0 = time.perf_counter()
    so, sm, _ = oracle.sketch_batch(sk_res, sk_off, k, scaled, mol, n_threads=cores)
    t_sketch = time.perf_counter() - c0
    c0 = time.perf_counter()
    rows = oracle.kmer_positions_batch_count(sk_res, sk_off, k, mol, so, sm, faithful=True, n_threads=cores)
    t_kpos = time.perf_counter() - c0
    del so, sm
    # ---- pairwise search: evenly spaced sample vs all targets
    t_o, t_m, t_a = T.to_host()
    ns = args.cpu_sample_queries
    if ns <= 0:
        pidx = np.linspace(0, n_q - 1, cores).astype(np.int64)
        p_res, p_off = oracle.pack([bytes(q_res_h[int(q_off_h[i]):int(q_off_h[i + 1])]) for i in pidx])
        po, pm, _ = oracle.sketch_batch(p_res, p_off, k, scaled, mol, n_threads=cores)
        c0 = time.perf_counter()
        oracle.manysearch(po, pm, t_o, t_m, t_a, n_threads=cores)
        pilot = max(time.perf_counter() - c0, 1e-3)
        ns = int(max(cores, min(n_q, cores * 15.0 / pilot)))
        ns = max(cores, (ns // cores) * cores)
    ns = min(ns, n_q)
    idx = np.linspace(0, n_q - 1, ns).astype(np.int64)
    s_res, s_off = oracle.pack([bytes(q_res_h[int(q_off_h[i]):int(q_off_h[i + 1])]) for i in idx])
    s_windows = int(np.maximum((s_off[1:] - s_off[:-1]).astype(np.int64) - k + 1, 0).sum())
    so, sm, _ = oracle.sketch_batch(s_res, s_off, k, scaled, mol, n_threads=cores)
    c0 = time.perf_counter()
    cq, _, _, _ = oracle.manysearch(so, sm, t_o, t_m, t_a, n_threads=cores)
    t_search = time.perf_counter() - c0
    per_w_sketch, per_w_kpos, per_w_search = t_sketch / max(sk_windows, 1), t_kpos / max(sk_windows, 1), t_search / max(s_windows, 1)
    return {"value": 1.0 / (per_w_sketch + per_w_search), "unit": "k-mers/s", "cores": cores, "kind": "port",
            "sample": f"sketch: first {n_sk} query proteins ({sk_windows} windows) in {t_sketch:.2f} s; process_kmers second pass "
                      f"(reference's linear `contains` scan) over the same proteins in {t_kpos:.2f} s; search: {ns} evenly spaced "
                      f"queries ({s_windows} windows) pairwise sorted-merge vs all {len(t_o) - 1} target sketches in {t_search:.2f} s; "
                      f"value = 1 / (sketch s/window + search s/window); target sketches are the GPU-built ones (bit-identical "
                      f"to the oracle's by the parity tests)",
            "sketch_kmers_per_s": sk_windows / max(t_sketch, 1e-9),
            "process_kmers_pass_kmers_per_s": sk_windows / max(t_kpos, 1e-9), "process_kmers_rows": rows,
            "sketch_plus_process_kmers_kmers_per_s": 1.0 / (per_w_sketch + per_w_kpos),
            "search_kmers_per_s": s_windows / max(t_search, 1e-9), "hits_in_sample": int(len(cq)),
            "reference_binary": reference_binary_probe(q_res_h, q_off_h, k, scaled, mol)}


# ---------------------------------------------------------------------------------------------------------------------
# BASELINE configs[4]: all-vs-all, index sharded by target id, hit lists all-gathered over RCCL
# ---------------------------------------------------------------------------------------------------------------------
def run_index_sharded(args, env, ks, synth, ksd, ctx=None, n_prot=None, k=24, scaled=5, mol="hp"):
    import numpy as np
    torch = env.torch
    rank, world = env.rank, env.world
    n_prot = n_prot or args.c4_proteins
    t0 = time.time()
    p_res_h = p_off_h = None
    if rank == 0:
        p_res_h, p_off_h = synth.proteome(n_prot, stream=40)
    p_res, p_off = ksd.broadcast_batch(p_res_h, p_off_h, src=0, device=env.cdev)  # queries = all proteins, replicated
    p_res, p_off = p_res.to(env.dev), p_off.to(env.dev)
    n_res = int(p_res.numel())
    off_host = p_off.cpu().numpy().view(np.uint64)
    s0, s1 = ksd.shard_by_residues(off_host, world)[rank]   # this rank's TARGET range
    t_res, t_off, n_t_res = device_shard(env, p_res, p_off, s0, s1)
    q_windows, q_maxlen = window_stats(env, p_off, k)
    gen_s = time.time() - t0
    own = ctx is None
    if own:
        ctx = ks.Context(env.dev_index, stream=torch.cuda.current_stream(env.dev).cuda_stream)
    torch.cuda.synchronize(env.dev)
    t0 = time.time()
    T = ctx.sketch_batch_device(t_res.data_ptr(), t_off.data_ptr(), s1 - s0, n_t_res, k, scaled, mol)
    index = ctx.index_build(T)
    torch.cuda.synchronize(env.dev)
    index_build_s = time.time() - t0

    pending = [None]

    def step():
        if args.two_calls:
            Q = ctx.sketch_queries_device(index, p_res.data_ptr(), p_off.data_ptr(), n_prot, n_res, max_seq_len=q_maxlen)
            H = ctx.search(index, Q)
        else:
            Q, H = ctx.sketch_search_device(index, p_res.data_ptr(), p_off.data_ptr(), n_prot, n_res, max_seq_len=q_maxlen)
        # per-shard hit lists all-gathered on the device; left in rank-major order (each shard's block is (qid, tid)-ordered),
        # as configs[4] states it — a global (qid, tid) order is one counting merge more (order="qid") a consumer may ask for.
        # The rows travel as 64-bit transport words (8 instead of 20 bytes per row over xGMI; rows with wide values on an escape
        # list), and the exchange is PIPELINED: this step's collective is started here and completed while the next step's
        # kernels run (over xGMI the exchange, not the kernels, is the step of this config at 8 GPUs).
        nxt = ksd.begin_all_gather_hits_device(H, tid_base=s0, device=env.cdev, sharded="index", order="shard", id_counts=(n_prot, n_prot))
        rows = pending[0].finish() if pending[0] is not None else None
        pending[0] = nxt
        out = (Q.n_hashes, H.count, H.n_pair_instances)
        H.free()
        Q.free()
        return out, rows

    def drain(out):
        rows = pending[0].finish()
        pending[0] = None
        return out[0], rows

    for _ in range(max(args.warmup, 1)):
        stats, rows = step()
    stats, rows = drain((stats, rows))
    rows = None  # (the warm-up's last hit list must not stay alive beside the three the timed loop keeps: one more list than the
                 # warm-up ever held at once = three fresh blocks = three hipMalloc calls inside the timed region, ~1.3 ms of one step)
    c_before = counters(ctx)
    elapsed, (stats, rows) = env.timed(step, args.steps, finalize=drain)
    c_after = counters(ctx)
    step_times = env.step_times()
    # per-kernel table + roofline of this config (untimed pass with every launch bracketed by HIP events on the launch stream)
    n_t_postings = index.n_postings
    ctx.timing_reset()
    ctx.timing_enable(1)
    for _ in range(2):
        step()
    drain((None, None))
    ctx.timing_enable(False)
    timing_all = {k_: (n_ / 2, ms_ / 2) for k_, (n_, ms_) in ctx.timing().items()}
    check = self_check(env, ksd, ctx, index, p_res.data_ptr(), p_off.data_ptr(), n_prot, n_res, q_maxlen)
    gate = None
    if env.multi:
        # the per-shard hit lists gathered and merged into global (qid, tid) order == rank 0's unsharded all-vs-all
        try:
            Q, H = ctx.sketch_search_device(index, p_res.data_ptr(), p_off.data_ptr(), n_prot, n_res, max_seq_len=q_maxlen)
            got = ksd.all_gather_hits_device(H, tid_base=s0, device=env.cdev, sharded="index", order="qid", id_counts=(n_prot, n_prot))
            H.free(); Q.free()
            if rank == 0:
                Ta = ctx.sketch_batch_device(p_res.data_ptr(), p_off.data_ptr(), n_prot, n_res, k, scaled, mol)
                ixa = ctx.index_build(Ta)
                Qa, Ha = ctx.sketch_search_device(ixa, p_res.data_ptr(), p_off.data_ptr(), n_prot, n_res, max_seq_len=q_maxlen)
                gate = sharded_equals_unsharded(env, ksd, ctx, got, Ha)
                Ha.free(); Qa.free(); ixa.free(); Ta.free()
            del got
        except AssertionError:
            raise
        except Exception as e:
            gate = {"error": repr(e)}
    # the gathered list is complete and holds every (qid, tid) pair once: checked once, outside the timed region, by sorting
    key = torch.sort(rows[0].to(torch.int64) << 32 | rows[1].to(torch.int64)).values
    ordered = bool((key[1:] > key[:-1]).all()) if key.numel() > 1 else True
    n_gathered = int(rows[0].numel())
    local_hits_sum, = env.sum_ints([stats[1]])
    diag = int((rows[0] == rows[1]).sum())
    del rows, key
    T.free(); index.free()
    if own:
        ctx.close()
    if rank != 0:
        return None
    ms_per_step = elapsed / args.steps * 1e3
    # SURVEY 8(d) bytes of this rank's step: sketch of all queries + search against its index shard
    sketch_bytes = n_res + 12 * stats[0] + 8 * n_prot
    step_bytes = sketch_bytes + 12 * stats[0] + 12 * n_t_postings + 16 * stats[1]
    traffic_tab, traffic_src = load_traffic("c5_200k_hp_k24_s5" if (world == 1 and (n_prot, k, scaled, mol) == (200_000, 24, 5, "hp")) else None)
    dom = max(timing_all.items(), key=lambda kv: kv[1][1])[0] if timing_all else None
    design = {"sketch_tiles": sketch_bytes, "bucket_scatter": 24 * stats[0],
              "join_buckets": 12 * stats[0] + ((4 * n_t_postings + 16 * stats[2]) if (n_t_postings >> 14) > 3072 else 12 * n_t_postings) + 8 * stats[2],
              "pair_rows": 8 * stats[2] + 20 * stats[1], "msd_local": 16 * stats[2], "msd_scatter": 16 * stats[2], "msd_hist": 8 * stats[2]}
    roofline = None
    if dom:
        n_l, ms = timing_all[dom]
        b = design.get(dom)
        avg_s = ms / max(n_l, 1) / 1e3
        roofline = {"kernel": dom, "bound": "hbm", "achieved": (b / n_l / avg_s / 1e9) if b and avg_s > 0 else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": (b / n_l / avg_s / 1e9 / HBM_PEAK_GBS) if b and avg_s > 0 else None, "traffic": traffic_tab.get(dom),
                    "traffic_source": traffic_src, "avg_launch_ms": avg_s * 1e3, "launches_per_step": n_l,
                    "bytes_per_step_by_design": b, "duration_source": "HIP events, untimed pass after the timed region",
                    "note": "sketch_tiles: SURVEY 8(d) bytes (n_res + 12*n_hashes + 8*n_seqs); the match-list kernels: bytes they move by design"}
    sk = timing_all.get("sketch_tiles")
    sketch_roofline = None
    if sk and sk[1] > 0:
        sketch_roofline = {"kernel": "sketch_tiles", "algorithmic_bytes_per_launch": sketch_bytes, "avg_launch_ms": sk[1] / sk[0],
                           "achieved": sketch_bytes / (sk[1] / sk[0] / 1e3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": sketch_bytes / (sk[1] / sk[0] / 1e3) / 1e9 / HBM_PEAK_GBS, "traffic": traffic_tab.get("sketch_tiles.compact")}
    return {
        "roofline": roofline, "sketch_roofline": sketch_roofline,
        "step_roofline": {"bound": "hbm", "algorithmic_bytes_per_step": step_bytes, "achieved": step_bytes / (ms_per_step / 1e3) / 1e9,
                          "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": step_bytes / (ms_per_step / 1e3) / 1e9 / HBM_PEAK_GBS,
                          "formula": "sketch (n_res + 12*N_Q + 8*n_seqs) + search (12*N_Q + 12*N_T + 16*N_hits), SURVEY 8(d)"},
        "kernels": {name: {"launches_per_step": n, "ms_per_step": ms} for name, (n, ms) in timing_all.items()},
        "metric": "k-mers hashed+matched/sec", "value": q_windows * args.steps / elapsed, "unit": "k-mers/s",
        "n_gpus": args.gpus, "steps": args.steps, "warmup": max(args.warmup, 1), "ms_per_step": elapsed / args.steps * 1e3,
        "step_ms": step_times, "timed_region_counters": counters_delta(c_before, c_after), "self_check": check,
        "sharded_equals_unsharded": gate,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {"workload": f"all-vs-all containment, {n_prot // 1000}k proteins, {mol} k={k} scaled={scaled} "
                               f"(BASELINE configs[4]: index sharded by target id, per-shard hit lists all-gathered)",
                   "proteins": n_prot, "ksize": k, "scaled": scaled, "moltype": mol,
                   "parallelism": f"index sharded x{world} by target id, queries replicated, hits all-gathered "
                                  f"(count exchange + one padded all_gather_into_tensor on device buffers, rows as 64-bit "
                                  f"transport words)",
                   "n_ranks": world, "collective_backend": env.backend},
        "query_proteins_per_s": n_prot * args.steps / elapsed, "query_windows": q_windows, "query_hashes": stats[0],
        "hits_gathered": n_gathered, "hits_sum_over_shards": local_hits_sum, "gathered_equals_sum_of_shards": n_gathered == local_hits_sum,
        "gathered_pairs_all_distinct": ordered, "self_hits": diag, "matched_posting_pairs_rank0": stats[2],
        "hit_bytes_gathered_per_step": (8 if env.multi else 20) * n_gathered, "index_build_s": index_build_s, "datagen_s": gen_s,
    }


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world == 1:
        sys.exit(relaunch_distributed(args))
    rank = int(os.environ.get("RANK", "0"))

    import __graft_entry__
    if rank == 0:
        __graft_entry__.build()
    else:
        # before the process group: a rank waiting in init_process_group for a rank 0 that is still compiling would run into
        # that call's short timeout (normally the library travelled with the snapshot and nobody builds or waits)
        from kmerseek_amd import build as ks_build
        ks_build.wait_until_built()
    env = Env(args)
    if env.world > 1:
        env.barrier()  # the other ranks load the library rank 0 has just (re)built
    import kmerseek_amd as ks
    from kmerseek_amd import dist as ksd, synth

    if args.mode == "index-sharded":
        result = run_index_sharded(args, env, ks, synth, ksd)
    else:
        result, (ctx, index, T) = run_queries_sharded(args, env, ks, synth, ksd)
        index.free(); T.free()
        c4 = None
        if not args.no_config4:
            try:
                c4 = run_index_sharded(args, env, ks, synth, ksd, ctx=ctx)
            except Exception as e:  # a side line must not cost the headline
                c4 = {"error": repr(e)}
        if result is not None:
            result["config4_index_sharded"] = c4
        ctx.close()
    if rank == 0:
        print(json.dumps(result))
    env.close()


if __name__ == "__main__":
    main()
