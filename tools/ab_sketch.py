#!/usr/bin/env python3
"""A/B timing of the sketch tile kernel across library builds on ONE box (boxes of the pool differ by several per cent, and
so do back-to-back runs on one box): every library runs the same cached batch in its own process, rounds interleaved, and
the median launch time over all rounds is reported — plain launches (index side) and launches that also emit postings.

    python tools/ab_sketch.py [--rounds 3] [--reps 20] lib1.so lib2.so ...       ("base" = the in-tree library)
"""
import argparse
import json
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(a):
    import numpy as np
    import kmerseek_amd as ks
    res = np.load(a.cache + ".res.npy"); offs = np.load(a.cache + ".off.npy")
    n = len(offs) - 1
    ctx = ks.Context(0)
    d_res, d_off = ctx.to_device(res), ctx.to_device(offs)
    T = ctx.sketch_batch_device(d_res.ptr, d_off.ptr, n, len(res), a.ksize, a.scaled, a.moltype)
    index = ctx.index_build(T)
    out = {}
    for mode in ("plain", "postings"):
        def once():
            if mode == "postings":
                Q = ctx.sketch_queries_device(index, d_res.ptr, d_off.ptr, n, len(res))
            else:
                Q = ctx.sketch_batch_device(d_res.ptr, d_off.ptr, n, len(res), a.ksize, a.scaled, a.moltype)
            Q.free()
        for _ in range(3):
            once()
        ctx.timing_reset(); ctx.timing_enable(True)
        for _ in range(a.reps):
            once()
        ctx.timing_enable(False)
        t = ctx.timing()
        out[mode] = sum(ms for name, (nl, ms) in t.items() if name.startswith("sketch_")) / a.reps
    print(json.dumps(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="*")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--n", type=int, default=1_000_000)
    ap.add_argument("--ksize", type=int, default=10)
    ap.add_argument("--scaled", type=int, default=1)
    ap.add_argument("--moltype", default="protein")
    ap.add_argument("--cache", default="/tmp/ks_ab")
    ap.add_argument("--child", action="store_true")
    a = ap.parse_args()
    if a.child:
        return child(a)
    import numpy as np
    from kmerseek_amd import synth
    cache = f"{a.cache}.{a.n}"
    if not os.path.exists(cache + ".off.npy"):
        res, offs = synth.proteome(a.n, stream=0)
        np.save(cache + ".res.npy", res); np.save(cache + ".off.npy", offs)
    acc = {lib: {"plain": [], "postings": []} for lib in a.libs}
    for r in range(a.rounds):
        for lib in a.libs:
            env = dict(os.environ)
            if lib != "base":
                env["KMERSEEK_AMD_LIB"] = os.path.abspath(lib)
            cmd = [sys.executable, os.path.abspath(__file__), "--child", "--cache", cache, "--reps", str(a.reps),
                   "--ksize", str(a.ksize), "--scaled", str(a.scaled), "--moltype", a.moltype]
            o = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
            try:
                d = json.loads(o.stdout.strip().splitlines()[-1])
            except Exception:
                print(lib, "FAILED", o.stderr[-400:])
                continue
            for m in acc[lib]:
                acc[lib][m].append(d[m])
    for lib in a.libs:
        print(json.dumps({"lib": os.path.basename(lib), **{m: {"median_ms": round(statistics.median(v), 4), "runs": [round(x, 3) for x in v]}
                                                               for m, v in acc[lib].items() if v}}))


if __name__ == "__main__":
    main()
