mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > gpurun_out/r2_smoke.log 2>&1; echo "smoke rc=$?"; tail -4 gpurun_out/r2_smoke.log
KS_BENCH_REHEARSE=1 timeout -k 10 500 python bench.py --gpus 2 --steps 3 --warmup 1 --queries 200000 --targets 200000 --c4-proteins 50000 --no-cpu-baseline --no-aux 2> gpurun_out/r2_reh3.err | grep "^{" > gpurun_out/r2_reh3.json; echo "rehearse rc=$?"
tail -c 400 gpurun_out/r2_reh3.err
