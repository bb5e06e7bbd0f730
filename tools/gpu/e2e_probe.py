import sys, time, numpy as np
sys.path.insert(0, '.')
import kmerseek_amd as ks
from kmerseek_amd import synth
t_res, t_off = synth.proteome(1_000_000, stream=0)
q_res, q_off = synth.queries(1_000_000, t_res, t_off, stream=1000)
ctx = ks.Context(0)
T = ctx.sketch_batch(t_res, t_off, 10, 1, "protein")
ix = ctx.index_build(T)
def tm(f):
    ctx.synchronize(); t0 = time.perf_counter(); r = f(); ctx.synchronize(); return r, (time.perf_counter() - t0) * 1e3
for rep in range(3):
    Q, t1 = tm(lambda: ctx.sketch_batch(q_res, q_off, 10, 1, "protein"))
    H, t2 = tm(lambda: ctx.search(ix, Q))
    rows, t3 = tm(lambda: H.to_host())
    d, t4 = tm(lambda: (ctx.to_device(q_res), ctx.to_device(q_off)))
    Qd, t5 = tm(lambda: ctx.sketch_batch_device(d[0].ptr, d[1].ptr, 1_000_000, len(q_res), 10, 1, "protein", max_seq_len=4000))
    print(f"rep {rep}: sketch_batch(host) {t1:.2f} ms  search(no postings) {t2:.2f}  hits.to_host {t3:.2f}  | upload only {t4:.2f}  sketch_device {t5:.2f}  path {H.partition_path}")
    ctx.timing_reset(); ctx.timing_enable(1)
    H2 = ctx.search(ix, Q); ctx.timing_enable(False)
    print("   ", {k: round(v[1], 3) for k, v in ctx.timing().items() if v[1] > 0.05})
    for o in (Q, H, H2, Qd): o.free()
    d[0].free(); d[1].free()
