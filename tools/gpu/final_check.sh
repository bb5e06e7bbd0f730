set -e
mkdir -p gpurun_out
python __graft_entry__.py --smoke 2>&1 | tail -4
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r3_full.log 2>&1 || { tail -60 gpurun_out/r3_full.log; exit 1; }
tail -2 gpurun_out/r3_full.log
python bench.py > gpurun_out/r3_bench_default.json 2> gpurun_out/r3_bench_default.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r3_bench_default.json') if l.startswith('{')][-1])
print('default', round(d['ms_per_step'],4), d['value'], 'roofline', round(d['roofline']['frac'],4), 'step', round(d['step_roofline']['frac'],4), 'c5', round(d['config4_index_sharded']['ms_per_step'],4), 'traffic', d['roofline']['traffic'], d['roofline']['traffic_source']['kernel_sources_unchanged_since'])
PY
