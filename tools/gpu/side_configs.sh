# The BASELINE configs that are not the headline, and the 125k-query shard that decides strong scaling, one bench line each.
R=${R:-r4}
mkdir -p gpurun_out
B="python3 bench.py --no-cpu-baseline --no-config4 --no-aux --steps 20 --warmup 5"
$B --queries 10000 --targets 10000 --ksize 7 > gpurun_out/${R}_c2.json 2> gpurun_out/${R}_c2.err
$B --queries 100000 --targets 100000 --ksize 16 --scaled 5 --moltype dayhoff > gpurun_out/${R}_c3.json 2> gpurun_out/${R}_c3.err
$B --queries 125000 --targets 1000000 > gpurun_out/${R}_shard.json 2> gpurun_out/${R}_shard.err
R=$R python3 - <<'PY'
import json, os
R = os.environ["R"]
for f in ("c2", "c3", "shard"):
    try:
        d = json.loads([l for l in open(f"gpurun_out/{R}_{f}.json") if l.startswith("{")][-1])
        ks = {k: round(v["ms_per_step"], 4) for k, v in d["kernels"].items() if v["ms_per_step"] >= 0.004}
        print(f, "ms/step", round(d["ms_per_step"], 4), "median", round(d["step_ms"]["median"], 4), "self", d["self_check"]["ok"], "mallocs", d["timed_region_counters"]["pool_mallocs"])
        print("   ", ks)
    except Exception as e:
        print(f, "ERR", e)
PY
