mkdir -p gpurun_out
set -x
KS_BENCH_REHEARSE=1 timeout -k 10 500 python bench.py --gpus 4 --steps 3 --warmup 1 --queries 100000 --targets 400000 --c4-proteins 50000 --no-cpu-baseline --no-aux > gpurun_out/r3_reh4.json 2> gpurun_out/r3_reh4.err; echo "rc=$?"
tail -c 600 gpurun_out/r3_reh4.err
KS_BENCH_REHEARSE=1 timeout -k 10 500 python bench.py --gpus 4 --steps 3 --warmup 1 --scaling weak --queries 100000 --targets 200000 --no-config4 --no-cpu-baseline --no-aux > gpurun_out/r3_reh4w.json 2> gpurun_out/r3_reh4w.err; echo "rc=$?"
tail -c 300 gpurun_out/r3_reh4w.err
KS_BENCH_REHEARSE=1 timeout -k 10 500 python bench.py --gpus 4 --steps 3 --warmup 1 --mode index-sharded --c4-proteins 50000 > gpurun_out/r3_reh4i.json 2> gpurun_out/r3_reh4i.err; echo "rc=$?"
tail -c 300 gpurun_out/r3_reh4i.err
python - <<'PY'
import json
for f in ("r3_reh4","r3_reh4w","r3_reh4i"):
    try:
        d=json.loads([l for l in open(f"gpurun_out/{f}.json") if l.startswith("{")][-1])
        print(f, d["n_gpus"], d["scaling"], round(d["ms_per_step"],3), d["value"], d["config"].get("parallelism","")[:80], (d.get("config4_index_sharded") or {}).get("gathered_equals_sum_of_shards"))
    except Exception as e: print(f, "ERR", e)
PY
