mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fall_back or plan or stride or ragged" > gpurun_out/r2_t12.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2_t12.log; tail -12 gpurun_out/r2_t12.log
