# End-of-round check on the GPU box: smoke, the whole GPU suite, the driver's bench command.  R = round tag of the output files.
set -e -o pipefail
R=${R:-r4}
mkdir -p gpurun_out
python __graft_entry__.py --smoke 2>&1 | tail -4
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/${R}_full.log 2>&1 || { tail -60 gpurun_out/${R}_full.log; exit 1; }
tail -2 gpurun_out/${R}_full.log
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${R}_bench_driver.json 2> gpurun_out/${R}_bench_driver.err
R=$R python - <<'PY'
import json, os
R = os.environ["R"]
d = json.loads([l for l in open(f'gpurun_out/{R}_bench_driver.json') if l.startswith('{')][-1])
c = d['config4_index_sharded']
print('driver-style', round(d['ms_per_step'], 4), d['value'], 'median', round(d['step_ms']['median'], 4), 'roofline', round(d['roofline']['frac'], 4),
      'step', round(d['step_roofline']['frac'], 4), 'counters', d['timed_region_counters'], 'self', d['self_check']['ok'])
print('c4', round(c['ms_per_step'], 4), 'median', round(c['step_ms']['median'], 4), 'max', round(c['step_ms']['max'], 4),
      'kernels', round(sum(v['ms_per_step'] for v in c['kernels'].values()), 4), 'counters', c['timed_region_counters'], 'self', c['self_check']['ok'])
PY
