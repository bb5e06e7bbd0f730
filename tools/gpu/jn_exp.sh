mkdir -p gpurun_out
python tools/microbench.py --steps 10 --variants "JN_EXP=0" "JN_EXP=1" "JN_EXP=2" "base" > gpurun_out/jn_exp_1m.log 2>&1
python tools/microbench.py --steps 10 --queries 125000 --variants "JN_EXP=0" "JN_EXP=1" "JN_EXP=2" "base" > gpurun_out/jn_exp_125k.log 2>&1
python - <<'PY'
import json
for f in ("gpurun_out/jn_exp_1m.log","gpurun_out/jn_exp_125k.log"):
    for l in open(f):
        if l.startswith("{"):
            d=json.loads(l); print(f[-12:], d.get("variant"), d.get("ms_per_step"), d.get("kernels",{}).get("join_buckets"), d.get("stats"))
        elif "rror" in l: print(l[:300])
PY
