mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r2_t13.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2_t13.log; tail -4 gpurun_out/r2_t13.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2_b8.json 2> gpurun_out/r2_b8.err; echo "bench rc=$?"
python - <<'PY'
import sys, time, torch, numpy as np
sys.path.insert(0, '.')
import kmerseek_amd as ks
from kmerseek_amd import synth, dist as ksd
dev = torch.device('cuda', 0)
p = synth.proteome(200000, stream=40)
ctx = ks.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
d_res, d_off = ctx.to_device(p[0]), ctx.to_device(p[1])
S = ctx.sketch_batch_device(d_res.ptr, d_off.ptr, 200000, len(p[0]), 24, 5, 'hp')
ix = ctx.index_build(S)
def t(f, n=10):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): r = f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
def search():
    Q = ctx.sketch_queries_device(ix, d_res.ptr, d_off.ptr, 200000, len(p[0])); H = ctx.search(ix, Q); H.free(); Q.free()
Q = ctx.sketch_queries_device(ix, d_res.ptr, d_off.ptr, 200000, len(p[0])); H = ctx.search(ix, Q)
print('search ms', t(search))
print('gather ms', t(lambda: ksd.all_gather_hits_device(H, tid_base=5, device=dev, sharded='index', order='shard')))
n = H.count
buf = torch.empty(5 * (n + (n & 1)), dtype=torch.int32, device=dev)
print('empty ms', t(lambda: torch.empty(5 * (n + (n & 1)), dtype=torch.int32, device=dev)))
base = buf.data_ptr(); cap = n + (n & 1)
print('copy ms', t(lambda: H.copy_to_device(base, base + 4 * cap, base + 8 * cap, base + 12 * cap, tid_base=5)))
PY
