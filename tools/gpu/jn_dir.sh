mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q -k "search or config" 2>&1 | tail -2 || exit 1
python tools/microbench.py --steps 10 --variants "JN_DIR=0" "JN_DIR=1024" "JN_DIR=2048" "JN_DIR=4096" > gpurun_out/jn_dir_1m.log 2>&1
python tools/microbench.py --steps 10 --queries 125000 --variants "JN_DIR=0" "JN_DIR=2048" "JN_DIR=4096" > gpurun_out/jn_dir_125k.log 2>&1
python - <<'PY'
import json
for f in ("gpurun_out/jn_dir_1m.log","gpurun_out/jn_dir_125k.log"):
    for l in open(f):
        if l.startswith("{"):
            d=json.loads(l); print(f[-12:], d.get("variant"), d.get("ms_per_step"), d.get("kernels",{}).get("join_buckets"), d.get("stats"))
PY
