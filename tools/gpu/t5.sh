mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r2_t5.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2_t5.log; tail -7 gpurun_out/r2_t5.log
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r2_b2.json 2> gpurun_out/r2_b2.err; echo "bench rc=$?"; tail -c 300 gpurun_out/r2_b2.err
python bench.py --steps 20 --warmup 5 --queries 10000 --targets 10000 --ksize 7 --no-config4 --no-cpu-baseline --no-aux > gpurun_out/r2_c2.json 2> gpurun_out/r2_c2.err; echo "c2 rc=$?"
python bench.py --steps 20 --warmup 5 --queries 125000 --no-config4 --no-cpu-baseline --no-aux > gpurun_out/r2_q125k.json 2> gpurun_out/r2_q125k.err; echo "q125k rc=$?"
