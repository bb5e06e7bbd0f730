mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2_t23.log 2>&1; rc=$?; tail -3 gpurun_out/r2_t23.log; [ $rc -eq 0 ] || exit $rc
for e in 0 1; do for i in 1 2; do
  if [ $e = 1 ]; then export KS_DEBUG_PLAN_SYNC=1; else unset KS_DEBUG_PLAN_SYNC; fi
  python bench.py --steps 30 --warmup 5 --queries 10000 --targets 10000 --ksize 7 --no-cpu-baseline --no-aux --no-config4 2>/dev/null | python -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('plan_sync=$e c2', round(d['ms_per_step'],3))"
done; done
unset KS_DEBUG_PLAN_SYNC
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-aux --no-config4 2>/dev/null | python -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('1m', round(d['ms_per_step'],3), round(d['kernels']['sketch_tiles']['ms_per_step'],3))"
python bench.py --steps 20 --warmup 5 --queries 125000 --no-cpu-baseline --no-aux --no-config4 2>/dev/null | python -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('125k', round(d['ms_per_step'],3))"
