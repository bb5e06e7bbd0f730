"""Search at 4M x 4M (1.17 G postings a side; manual check, not part of the suite): fused postings == plain partition,
oracle manysearch on a few queries against all targets."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
import kmerseek_amd as ks
from kmerseek_amd import synth
from oracle import oracle

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
k, scaled, mol = 10, 1, "protein"
def big(stream0):
    parts = [synth.proteome(1_000_000, stream=stream0 + i) for i in range(n // 1_000_000)]
    res = np.concatenate([p[0] for p in parts]); offs = np.zeros(n + 1, np.uint64); at = 1; base = np.uint64(0)
    for r, o in parts:
        offs[at:at + len(o) - 1] = o[1:] + base; at += len(o) - 1; base += o[-1]
    return res, offs
t_res, t_off = big(400)
q_res, q_off = synth.queries(n, t_res, t_off, stream=499)
ctx = ks.Context(0)
dt, do, dq, dqo = ctx.to_device(t_res), ctx.to_device(t_off), ctx.to_device(q_res), ctx.to_device(q_off)
t0 = time.time()
T = ctx.sketch_batch_device(dt.ptr, do.ptr, n, len(t_res), k, scaled, mol)
ix = ctx.index_build(T); ctx.synchronize()
print(f"index of {ix.n_postings} postings built in {(time.time() - t0) * 1e3:.0f} ms", flush=True)
def step(fused):
    t0 = time.time()
    Q = ctx.sketch_queries_device(ix, dq.ptr, dqo.ptr, n, len(q_res)) if fused else ctx.sketch_batch_device(dq.ptr, dqo.ptr, n, len(q_res), k, scaled, mol)
    H = ctx.search(ix, Q); ctx.synchronize()
    return Q, H, (time.time() - t0) * 1e3
for _ in range(2):
    Qf, Hf, ms = step(True)
    print(f"fused step {ms:.1f} ms: {Hf.count} hits, {Hf.n_pair_instances} pairs, path {Hf.partition_path}", flush=True)
    f = Hf.to_host(); Hf.free(); Qkeep = Qf
Qp, Hp, ms = step(False)
print(f"plain step {ms:.1f} ms: {Hp.count} hits, path {Hp.partition_path}", flush=True)
p = Hp.to_host()
assert all(np.array_equal(a, b) for a, b in zip(f, p)), "fused != plain"
qo, qm, _ = Qp.to_host(); to, tm, ta = T.to_host()
for qi in (0, n // 3, n - 1, 123457):
    w = oracle.manysearch(np.array([0, qo[qi + 1] - qo[qi]], np.uint64), qm[int(qo[qi]):int(qo[qi + 1])], to, tm, ta, n_threads=16)
    sel = f[0] == qi
    assert np.array_equal(f[1][sel], w[1]) and np.array_equal(f[2][sel], w[2]) and np.array_equal(f[3][sel], w[3]), qi
print("oracle spot checks ok")
