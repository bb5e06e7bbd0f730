mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "msd or search or golden" > gpurun_out/r2_t7.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2_t7.log; tail -15 gpurun_out/r2_t7.log
