mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "search or join or row_pass or presorted" 2>&1 | tail -2 || exit 1
for i in 1 2 3; do python bench.py --steps 30 --warmup 5 --queries 10000 --targets 10000 --ksize 7 --no-cpu-baseline --no-aux --no-config4 2>/dev/null | python -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('c2', round(d['ms_per_step'],3))"; done
