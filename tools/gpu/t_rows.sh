mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_fullsize.py -m gpu -x -q > gpurun_out/r2_t19.log 2>&1; rc=$?; tail -3 gpurun_out/r2_t19.log; [ $rc -eq 0 ] || exit $rc
for cfg in "1000000 1000000 10 1 protein s1m" "200000 200000 24 5 hp c5"; do
  set -- $cfg
  python bench.py --steps 20 --warmup 5 --queries $1 --targets $2 --ksize $3 --scaled $4 --moltype $5 --no-cpu-baseline --no-aux --no-config4 > gpurun_out/r2_bd_$6.json 2>gpurun_out/r2_bd_$6.err || exit 1
  python -c "
import json; d=json.load(open('gpurun_out/r2_bd_$6.json')); k=d['kernels']; print('$6', round(d['ms_per_step'],3)); print({n: round(v['ms_per_step'],3) for n,v in sorted(k.items(), key=lambda x:-x[1]['ms_per_step'])})"
done
