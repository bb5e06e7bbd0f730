"""Run only the sketch kernels on a resident synthetic query batch (profiling driver; honours KMERSEEK_AMD_LIB)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kmerseek_amd as ks
from kmerseek_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
res, offs = synth.proteome(n, stream=5)
ctx = ks.Context(0)
d_res, d_off = ctx.to_device(res), ctx.to_device(offs)
for _ in range(2):
    S = ctx.sketch_batch_device(d_res.ptr, d_off.ptr, n, len(res), 10, 1, "protein")
    S.free()
ctx.close()
