mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r2_t8.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2_t8.log; tail -4 gpurun_out/r2_t8.log
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-aux > gpurun_out/r2_b4.json 2> gpurun_out/r2_b4.err; echo "bench rc=$?"
python bench.py --steps 20 --warmup 5 --queries 10000 --targets 10000 --ksize 7 --no-config4 --no-cpu-baseline --no-aux > gpurun_out/r2_c2b.json 2> gpurun_out/r2_c2b.err; echo "c2 rc=$?"
python bench.py --steps 10 --warmup 3 --queries 200000 --targets 200000 --ksize 24 --scaled 5 --moltype hp --no-config4 --no-cpu-baseline --no-aux > gpurun_out/r2_c5b.json 2> gpurun_out/r2_c5b.err; echo "c5 rc=$?"
KS_DEBUG_PAIRS_LSD=1 python bench.py --steps 10 --warmup 3 --queries 200000 --targets 200000 --ksize 24 --scaled 5 --moltype hp --no-config4 --no-cpu-baseline --no-aux > gpurun_out/r2_c5lsd.json 2> gpurun_out/r2_c5lsd.err; echo "c5 lsd rc=$?"
