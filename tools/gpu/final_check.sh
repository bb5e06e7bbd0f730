mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > gpurun_out/r2_smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/r2_smoke.log
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2_t_final.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/r2_t_final.log
timeout -k 10 900 python tools/fuzz_parity.py --cases 600 --seed 31 > gpurun_out/r2_fuzz_a.log 2>&1; echo "fuzz a rc=$?"; tail -1 gpurun_out/r2_fuzz_a.log
timeout -k 10 900 python tools/fuzz_parity.py --big --cases 14 --seed 32 > gpurun_out/r2_fuzz_big.log 2>&1; echo "fuzz big rc=$?"; tail -1 gpurun_out/r2_fuzz_big.log
python bench.py > gpurun_out/r2_final_default.json 2> gpurun_out/r2_final_default.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r2_final_default.json') if l.startswith('{')][-1])
print({k:d[k] for k in ('value','ms_per_step','n_gpus','scaling')}, 'frac', d['roofline']['frac'], 'launch', d['roofline']['avg_launch_ms'], 'traffic_src', d['roofline']['traffic_source'])
print('config4', d['config4_index_sharded']['ms_per_step'], d['config4_index_sharded']['value'])
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['sketch_kmers_per_s'])
PY
