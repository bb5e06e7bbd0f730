"""Time the k-mer position table (ks_kmer_positions_device) on a resident synthetic proteome.

    python tools/kmerpos_bench.py [--seqs N] [--ksize K] [--scaled S] [--moltype M]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import kmerseek_amd as ks
from kmerseek_amd import synth


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seqs", type=int, default=1_000_000)
    ap.add_argument("--ksize", type=int, default=10)
    ap.add_argument("--scaled", type=int, default=1)
    ap.add_argument("--moltype", default="protein")
    a = ap.parse_args()
    res, offs = synth.proteome(a.seqs, stream=5)
    ctx = ks.Context(0)
    d_res, d_off = ctx.to_device(res), ctx.to_device(offs)
    n = 0
    for _ in range(2):
        n = ctx.kmer_positions_device(d_res.ptr, d_off.ptr, a.seqs, len(res), a.ksize, a.scaled, a.moltype, fetch=False)
    ctx.timing_reset()
    ctx.timing_enable(1)
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        ctx.kmer_positions_device(d_res.ptr, d_off.ptr, a.seqs, len(res), a.ksize, a.scaled, a.moltype, fetch=False)
    ctx.synchronize()
    el = (time.perf_counter() - t0) / 5
    ctx.timing_enable(0)
    lens = (offs[1:] - offs[:-1]).astype(np.int64)
    windows = int(np.maximum(lens - a.ksize + 1, 0).sum())
    print(json.dumps({"seqs": a.seqs, "windows": windows, "kept": n, "ms_per_call": el * 1e3, "windows_per_s": windows / el,
                      "algorithmic_bytes": len(res) + 16 * n, "gb_per_s": (len(res) + 16 * n) / el / 1e9,
                      "kernels": {k: (c, round(ms / 5, 4)) for k, (c, ms) in ctx.timing().items()}}))
    ctx.close()


if __name__ == "__main__":
    main()
