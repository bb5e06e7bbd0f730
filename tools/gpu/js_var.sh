mkdir -p gpurun_out
python tools/microbench.py --steps 10 --queries 125000 --variants $JS_VARIANTS > gpurun_out/js_var.log 2>&1
python - <<'PY'
import json
for l in open("gpurun_out/js_var.log"):
    if l.startswith("{"):
        d=json.loads(l); print(d.get("variant"), d.get("ms_per_step"), d.get("kernels",{}).get("join_buckets"), d.get("stats"))
    elif "rror" in l: print(l[:300])
PY
