#!/usr/bin/env python3
"""How well do the ALU-bound sketch kernel and the memory-bound search kernels overlap when issued from two streams?
Runs sketch-only and search-only loops alone, then concurrently from two host threads (one context each)."""
import sys, os, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import kmerseek_amd as ks
from kmerseek_amd import synth

N = int(sys.argv[1]) if len(sys.argv) > 1 else 500000
t_res, t_off = synth.proteome(N, stream=0)
q_res, q_off = synth.queries(N, t_res, t_off, stream=1000)
A, B = ks.Context(0), ks.Context(0)
ixA = A.index_build(A.sketch_batch(t_res, t_off, 10, 1, "protein"))
ixB = B.index_build(B.sketch_batch(t_res, t_off, 10, 1, "protein"))
dA = (A.to_device(q_res), A.to_device(q_off))
dB = (B.to_device(q_res), B.to_device(q_off))
QB = B.sketch_queries_device(ixB, dB[0].ptr, dB[1].ptr, len(q_off) - 1, len(q_res))

def sketch_loop(n, out):
    t0 = time.perf_counter()
    for _ in range(n):
        Q = A.sketch_queries_device(ixA, dA[0].ptr, dA[1].ptr, len(q_off) - 1, len(q_res)); Q.free()
    out.append((time.perf_counter() - t0) / n * 1e3)

def search_loop(n, out):
    t0 = time.perf_counter()
    for _ in range(n):
        H = B.search(ixB, QB); H.free()
    out.append((time.perf_counter() - t0) / n * 1e3)

for f in (sketch_loop, search_loop):
    f(3, [])
a, b = [], []
sketch_loop(10, a); search_loop(10, b)
print("alone: sketch %.2f ms, search %.2f ms, sum %.2f" % (a[0], b[0], a[0] + b[0]))
a2, b2 = [], []
ta = threading.Thread(target=sketch_loop, args=(10, a2)); tb = threading.Thread(target=search_loop, args=(10, b2))
t0 = time.perf_counter(); ta.start(); tb.start(); ta.join(); tb.join(); wall = (time.perf_counter() - t0) / 10 * 1e3
print("together: sketch %.2f ms, search %.2f ms, wall per (sketch+search) %.2f ms" % (a2[0], b2[0], wall))
