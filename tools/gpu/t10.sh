mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "copies or roundtrip or errors" > gpurun_out/r2_t10.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2_t10.log; tail -12 gpurun_out/r2_t10.log
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-config4 > gpurun_out/r2_b6.json 2> gpurun_out/r2_b6.err; echo "bench rc=$?"
