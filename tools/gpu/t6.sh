mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_configs.py -m gpu -x -q > gpurun_out/r2_t6.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2_t6.log; tail -4 gpurun_out/r2_t6.log
( python tools/sketch_only.py 1000000 10 1 protein 1
  python tools/sketch_only.py 1000000 10 1 protein 0
  python tools/sketch_only.py 1000000 16 5 dayhoff 1 ) > gpurun_out/r2_post_early.log 2>&1
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-config4 --no-aux > gpurun_out/r2_b3.json 2> gpurun_out/r2_b3.err; echo "bench rc=$?"
