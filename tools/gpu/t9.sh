mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_fuzz.py -m gpu -x -q > gpurun_out/r2_t9.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2_t9.log; tail -4 gpurun_out/r2_t9.log
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-aux --no-config4 > gpurun_out/r2_b5.json 2> gpurun_out/r2_b5.err; echo "bench rc=$?"
python bench.py --steps 20 --warmup 5 --queries 10000 --targets 10000 --ksize 7 --no-config4 --no-cpu-baseline --no-aux > gpurun_out/r2_c2c.json 2> gpurun_out/r2_c2c.err; echo "c2 rc=$?"
python bench.py --steps 10 --warmup 3 --queries 200000 --targets 200000 --ksize 24 --scaled 5 --moltype hp --no-config4 --no-cpu-baseline --no-aux > gpurun_out/r2_c5c.json 2> gpurun_out/r2_c5c.err; echo "c5 rc=$?"
