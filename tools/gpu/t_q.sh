mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_realdata.py -m gpu -x -q > gpurun_out/r2_t25.log 2>&1; rc=$?; tail -3 gpurun_out/r2_t25.log; [ $rc -eq 0 ] || exit $rc
python tools/microbench.py --steps 10 --variants "SK_QUEUE_RES=0" "SK_QUEUE_RES=1" > gpurun_out/qres.log 2>&1
python - <<'PY'
import json
for l in open("gpurun_out/qres.log"):
    if l.startswith("{"):
        d=json.loads(l); print(d.get("variant"), d.get("ms_per_step"), d.get("kernels",{}).get("sketch_tiles"), d.get("stats"))
    elif "rror" in l: print(l[:300])
PY
