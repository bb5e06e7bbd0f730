mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_configs.py -m gpu -x -q > gpurun_out/r2_t15.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2_t15.log; tail -8 gpurun_out/r2_t15.log
KS_BENCH_REHEARSE=1 timeout -k 10 500 python bench.py --gpus 2 --steps 3 --warmup 1 --mode index-sharded --c4-proteins 50000 2> gpurun_out/r2_reh4.err | grep "^{" > gpurun_out/r2_reh4.json; echo "rehearse rc=$?"
