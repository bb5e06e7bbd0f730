"""Per-kernel timing of ks_index_build on a resident 1M-protein sketch (KS_DEBUG_INDEX_LSD=1 times the 8-pass fallback)."""
import sys, time, json
sys.path.insert(0, "/root/repo")
import kmerseek_amd as ks
from kmerseek_amd import synth
res, offs = synth.proteome(1_000_000, stream=0)
ctx = ks.Context(0)
d_res, d_off = ctx.to_device(res), ctx.to_device(offs)
T = ctx.sketch_batch_device(d_res.ptr, d_off.ptr, 1_000_000, len(res), 10, 1, "protein")
ix = ctx.index_build(T); ix.free()
ctx.timing_reset(); ctx.timing_enable(1)
t0 = time.perf_counter()
ix = ctx.index_build(T)
ctx.synchronize()
el = time.perf_counter() - t0
ctx.timing_enable(0)
print(json.dumps({"index_build_ms": el * 1e3, "kernels": {k: (c, round(ms, 3)) for k, (c, ms) in ctx.timing().items()}}))
