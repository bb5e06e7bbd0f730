mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q -k "join or full_size or search" > gpurun_out/r2_t21.log 2>&1; rc=$?; tail -4 gpurun_out/r2_t21.log; [ $rc -eq 0 ] || exit $rc
KS_DEBUG_JOIN_FP=1 KS_DEBUG_JOIN_SPARSE=1 timeout -k 10 600 python tools/fuzz_parity.py --cases 300 --seed 21 > gpurun_out/r2_fuzz_sp.log 2>&1; echo "fuzz sparse rc=$?"; tail -1 gpurun_out/r2_fuzz_sp.log
KS_DEBUG_JOIN_FP=1 KS_DEBUG_JOIN_SPARSE=1 KS_DEBUG_JOIN_SEGS=1 KS_DEBUG_FP_COARSEN=14 timeout -k 10 600 python tools/fuzz_parity.py --cases 200 --seed 22 > gpurun_out/r2_fuzz_sp2.log 2>&1; echo "fuzz sparse2 rc=$?"; tail -1 gpurun_out/r2_fuzz_sp2.log
for v in 0 1; do for nq in 125000 250000 500000; do
  KS_DEBUG_JOIN_SPARSE=$v python bench.py --steps 20 --warmup 5 --queries $nq --no-cpu-baseline --no-aux --no-config4 2>/dev/null | python -c "
import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); k=d['kernels']; print('sparse=$v nq=$nq', round(d['ms_per_step'],3), 'join', round(k['join_buckets']['ms_per_step'],3))"
done; done
