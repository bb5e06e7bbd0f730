mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r2_t11.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2_t11.log; tail -4 gpurun_out/r2_t11.log
python bench.py > gpurun_out/r2_b7.json 2> gpurun_out/r2_b7.err; echo "bench rc=$?"; tail -c 300 gpurun_out/r2_b7.err
