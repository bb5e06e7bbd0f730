mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r2_t14.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2_t14.log; tail -4 gpurun_out/r2_t14.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-aux > gpurun_out/r2_b9.json 2> gpurun_out/r2_b9.err; echo "bench rc=$?"
python bench.py --steps 10 --warmup 3 --queries 200000 --targets 200000 --ksize 24 --scaled 5 --moltype hp --no-config4 --no-cpu-baseline --no-aux > gpurun_out/r2_c5d.json 2> gpurun_out/r2_c5d.err; echo "c5 rc=$?"
python bench.py --steps 20 --warmup 5 --queries 10000 --targets 10000 --ksize 7 --no-config4 --no-cpu-baseline --no-aux > gpurun_out/r2_c2d.json 2> gpurun_out/r2_c2d.err; echo "c2 rc=$?"
python bench.py --steps 10 --warmup 3 --queries 100000 --targets 100000 --ksize 16 --scaled 5 --moltype dayhoff --no-config4 --no-cpu-baseline --no-aux > gpurun_out/r2_c3d.json 2> gpurun_out/r2_c3d.err; echo "c3 rc=$?"
