mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > gpurun_out/r2_smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/r2_smoke.log
python bench.py > gpurun_out/r2_final_default.json 2> gpurun_out/r2_final_default.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r2_final_default.json') if l.startswith('{')][-1])
print({k:d[k] for k in ('metric','value','unit','ms_per_step','n_gpus','scaling','dtype')})
print('roofline', {k:d['roofline'].get(k) for k in ('achieved','frac','avg_launch_ms','traffic','algorithmic_bytes_per_launch')})
print('step_roofline', d.get('step_roofline'))
print('cpu_baseline', d.get('cpu_baseline'))
print('config4', d.get('config4_index_sharded'))
print('kernels', {n: round(v['ms_per_step'],3) for n,v in sorted(d['kernels'].items(), key=lambda x:-x[1]['ms_per_step'])})
print('aux', d.get('aux'))
PY
for cfg in "125000 1000000 10 1 protein s125k" "10000 10000 7 1 protein c2" "100000 100000 16 5 dayhoff c3" "200000 200000 24 5 hp c5"; do
  set -- $cfg
  python bench.py --steps 20 --warmup 5 --queries $1 --targets $2 --ksize $3 --scaled $4 --moltype $5 --no-cpu-baseline --no-aux --no-config4 > gpurun_out/r2_bd_$6.json 2>gpurun_out/r2_bd_$6.err || exit 1
  python -c "
import json; d=json.load(open('gpurun_out/r2_bd_$6.json')); k=d['kernels']; print('$6', round(d['ms_per_step'],3), {n: round(v['ms_per_step'],3) for n,v in sorted(k.items(), key=lambda x:-x[1]['ms_per_step'])[:7]})"
done
KS_BENCH_REHEARSE=1 timeout -k 10 500 python bench.py --gpus 2 --steps 3 --warmup 1 --queries 200000 --targets 200000 --c4-proteins 50000 --no-cpu-baseline --no-aux 2> gpurun_out/r2_reh3.err | grep "^{" > gpurun_out/r2_reh3.json; echo "rehearse rc=$?"
tail -c 300 gpurun_out/r2_reh3.err
