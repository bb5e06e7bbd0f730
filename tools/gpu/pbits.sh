mkdir -p gpurun_out
KS_DEBUG_PBITS_MAX=17 timeout -k 10 600 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py -m gpu -x -q -k "full_size or presorted or search" > gpurun_out/r2_t16.log 2>&1; echo "pytest(17) rc=$?"; tail -3 gpurun_out/r2_t16.log
for pb in 16 17 16 17; do
  KS_DEBUG_PBITS_MAX=$pb python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-aux --no-config4 > gpurun_out/r2_pb$pb.json 2>/dev/null
  python -c "
import json; d=json.load(open('gpurun_out/r2_pb$pb.json')); k=d['kernels']; print('pbits_max $pb', round(d['ms_per_step'],3), {n: round(k[n]['ms_per_step'],3) for n in ('sketch_tiles','bucket_scatter','join_buckets')})"
done
