mkdir -p gpurun_out
python tools/microbench.py --steps 5 --variants "JN_STAMP" > gpurun_out/jn_stamp_1m.log 2>&1
python tools/microbench.py --steps 5 --queries 125000 --variants "JN_STAMP" > gpurun_out/jn_stamp_125k.log 2>&1
python - <<'PY'
import json
for f in ("gpurun_out/jn_stamp_1m.log","gpurun_out/jn_stamp_125k.log"):
    for l in open(f):
        if l.startswith("{"):
            d=json.loads(l); print(f[-12:], d.get("variant"), d.get("ms_per_step"), d.get("kernels",{}).get("join_buckets"), d.get("join_phase_share"))
        elif "rror" in l: print(l[:300])
PY
