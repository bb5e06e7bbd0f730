#!/usr/bin/env python3
"""Kernel-tuning loop for the GPU box: build -D variants of the library, run the bench workload on
cached synthetic data, print ms/step per kernel.  Not part of the product or of the judged bench.

    python tools/microbench.py --variants "base" "RS_IPT=16" "RS_THREADS=256,RS_IPT=16" [--queries N --targets N]
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(args):
    import numpy as np
    import torch
    import kmerseek_amd as ks
    cache = args.cache
    t_res = np.load(cache + ".t_res.npy"); t_off = np.load(cache + ".t_off.npy")
    q_res = np.load(cache + ".q_res.npy"); q_off = np.load(cache + ".q_off.npy")
    dev = torch.device("cuda", 0)
    dt_res = torch.from_numpy(t_res).to(dev); dt_off = torch.from_numpy(t_off.view(np.int64)).to(dev)
    dq_res = torch.from_numpy(q_res).to(dev); dq_off = torch.from_numpy(q_off.view(np.int64)).to(dev)
    ctx = ks.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
    k, sc, mol = args.ksize, args.scaled, args.moltype
    T = ctx.sketch_batch_device(dt_res.data_ptr(), dt_off.data_ptr(), len(t_off) - 1, len(t_res), k, sc, mol)
    ix = ctx.index_build(T)

    def step():
        if args.plain:
            Q = ctx.sketch_batch_device(dq_res.data_ptr(), dq_off.data_ptr(), len(q_off) - 1, len(q_res), k, sc, mol)
        else:
            Q = ctx.sketch_queries_device(ix, dq_res.data_ptr(), dq_off.data_ptr(), len(q_off) - 1, len(q_res))
        H = ctx.search(ix, Q)
        r = (Q.n_hashes, H.count, H.n_pair_instances)
        H.free(); Q.free()
        return r
    for _ in range(2):
        r = step()
    pool0 = ctx.pool_stats()
    ctx.timing_reset(); ctx.timing_enable(not args.no_events)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        r = step()
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / args.steps * 1e3
    ctx.timing_enable(False)
    tm = {k_: round(v[1] / args.steps, 3) for k_, v in ctx.timing().items()}
    out = {"variant": args.variant, "ms_per_step": round(el, 3), "stats": r, "kernels": tm, "pool_before": pool0,
           "pool_after": ctx.pool_stats()}
    import ctypes
    L = ks._lib.load()
    if hasattr(L, "ks_debug_read_stamps"):
        buf = (ctypes.c_ulonglong * 16)()
        L.ks_debug_read_stamps(buf, 0)
        tot = float(sum(buf)) or 1.0
        out["sketch_phase_share"] = [round(v / tot, 4) for v in buf[:12]]
    if hasattr(L, "ks_debug_read_join_stamps"):
        buf = (ctypes.c_ulonglong * 16)()
        L.ks_debug_read_join_stamps(buf, 0)
        tot = float(sum(buf)) or 1.0
        out["join_phase_share"] = [round(v / tot, 4) for v in buf[:9]]
    print(json.dumps(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", nargs="*", default=["base"])
    ap.add_argument("--queries", type=int, default=1_000_000)
    ap.add_argument("--targets", type=int, default=1_000_000)
    ap.add_argument("--ksize", type=int, default=10)
    ap.add_argument("--scaled", type=int, default=1)
    ap.add_argument("--moltype", default="protein")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--cache", default="/tmp/ks_mb")
    ap.add_argument("--child", action="store_true")
    ap.add_argument("--no-events", action="store_true")
    ap.add_argument("--plain", action="store_true", help="sketch without pre-partitioned postings")
    ap.add_argument("--variant", default="base")
    args = ap.parse_args()
    if args.child:
        return child(args)
    import numpy as np
    from kmerseek_amd import synth, build as ks_build
    cache = f"{args.cache}.{args.queries}.{args.targets}"
    if not os.path.exists(cache + ".q_off.npy"):
        t_res, t_off = synth.proteome(args.targets, stream=0)
        q_res, q_off = synth.queries(args.queries, t_res, t_off, stream=1000)
        for n, a in (("t_res", t_res), ("t_off", t_off), ("q_res", q_res), ("q_off", q_off)):
            np.save(f"{cache}.{n}.npy", a)
    for v in args.variants:
        if v.startswith("lib="):  # a prebuilt library (e.g. an earlier commit's, built into kmerseek_amd/variants/)
            so = os.path.abspath(v[4:])
        else:
            defs = [] if v == "base" else v.split(",")
            so = f"/tmp/libks_{abs(hash(v)) % 10**8}.so"
            try:
                ks_build.build(force=True, defs=defs, out=so)
            except subprocess.CalledProcessError as e:
                print(json.dumps({"variant": v, "error": "build failed"}))
                continue
        env = dict(os.environ, KMERSEEK_AMD_LIB=so)
        cmd = [sys.executable, os.path.abspath(__file__), "--child", "--variant", v, "--cache", cache,
               "--ksize", str(args.ksize), "--scaled", str(args.scaled), "--moltype", args.moltype,
               "--steps", str(args.steps)] + (["--no-events"] if args.no_events else []) + (["--plain"] if args.plain else [])
        subprocess.call(cmd, env=env, timeout=300)
        sys.stdout.flush()


if __name__ == "__main__":
    main()
