mkdir -p gpurun_out
set -x
KS_BENCH_REHEARSE=1 timeout -k 10 500 python bench.py --gpus 2 --steps 3 --warmup 1 --queries 200000 --targets 200000 --c4-proteins 50000 --no-cpu-baseline --no-aux > gpurun_out/r2_reh2.json 2> gpurun_out/r2_reh2.err; echo "rc=$?"
tail -c 1500 gpurun_out/r2_reh2.err
KS_BENCH_REHEARSE=1 timeout -k 10 500 python bench.py --gpus 2 --steps 3 --warmup 1 --scaling weak --queries 100000 --targets 200000 --no-config4 --no-cpu-baseline --no-aux > gpurun_out/r2_reh2w.json 2> gpurun_out/r2_reh2w.err; echo "rc=$?"
tail -c 800 gpurun_out/r2_reh2w.err
KS_BENCH_REHEARSE=1 timeout -k 10 500 python bench.py --gpus 2 --steps 3 --warmup 1 --mode index-sharded --c4-proteins 50000 > gpurun_out/r2_reh2i.json 2> gpurun_out/r2_reh2i.err; echo "rc=$?"
tail -c 800 gpurun_out/r2_reh2i.err
