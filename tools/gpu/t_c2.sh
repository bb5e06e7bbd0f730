mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q > gpurun_out/r2_t22.log 2>&1; rc=$?; tail -3 gpurun_out/r2_t22.log; [ $rc -eq 0 ] || exit $rc
for i in 1 2 3; do python bench.py --steps 30 --warmup 5 --queries 10000 --targets 10000 --ksize 7 --no-cpu-baseline --no-aux --no-config4 2>/dev/null | python -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('c2', round(d['ms_per_step'],3))"; done
