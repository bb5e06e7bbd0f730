#!/usr/bin/env python3
"""Device-resident sketch timing of one synthetic batch (tuning aid; the numbers that count come from bench.py).

    python tools/sketch_only.py [n_proteins] [ksize] [scaled] [moltype] [postings 0|1] [reps]

Prints kernel-level HIP-event times of the sketch launches, with and without the fused postings.
Environment knobs of the library apply (KS_DEBUG_SPAN, KS_DEBUG_NO_COMPACT, KS_DEBUG_TILE_R)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import kmerseek_amd as ks
from kmerseek_amd import synth


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    k = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    scaled = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    mol = sys.argv[4] if len(sys.argv) > 4 else "protein"
    postings = int(sys.argv[5]) if len(sys.argv) > 5 else 0
    reps = int(sys.argv[6]) if len(sys.argv) > 6 else 10
    res, offs = synth.proteome(n, stream=0)
    windows = int(np.maximum((offs[1:] - offs[:-1]).astype(np.int64) - k + 1, 0).sum())
    ctx = ks.Context(0)
    d_res, d_off = ctx.to_device(res), ctx.to_device(offs)
    index = None
    if postings:
        T = ctx.sketch_batch_device(d_res.ptr, d_off.ptr, n, len(res), k, scaled, mol)
        index = ctx.index_build(T)

    def once():
        if index is not None:
            Q = ctx.sketch_queries_device(index, d_res.ptr, d_off.ptr, n, len(res))
        else:
            Q = ctx.sketch_batch_device(d_res.ptr, d_off.ptr, n, len(res), k, scaled, mol)
        nh = Q.n_hashes
        Q.free()
        return nh

    for _ in range(3):
        nh = once()
    ctx.timing_reset(); ctx.timing_enable(1)
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        once()
    ctx.synchronize()
    wall = (time.perf_counter() - t0) / reps
    ctx.timing_enable(False)
    t = {name: round(ms / reps, 4) for name, (nl, ms) in ctx.timing().items()}
    print(json.dumps({"n": n, "k": k, "scaled": scaled, "moltype": mol, "postings": postings, "windows": windows, "hashes": nh,
                      "wall_ms": round(wall * 1e3, 4), "sketch_tiles_Gwin_per_s": round(windows / (t.get("sketch_tiles", 1e9) * 1e-3) / 1e9, 2),
                      "kernels_ms": t, "stats": ctx.sketch_stats(), "env": {k_: v for k_, v in os.environ.items() if k_.startswith("KS_DEBUG")}}))
    ctx.close()


if __name__ == "__main__":
    main()
