mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2_t18.log 2>&1; rc=$?; tail -3 gpurun_out/r2_t18.log; [ $rc -eq 0 ] || exit $rc
KS_DEBUG_JOIN_FP=1 timeout -k 10 600 python tools/fuzz_parity.py --cases 300 --seed 11 > gpurun_out/r2_fuzz_fp.log 2>&1; echo "fuzz fp rc=$?"; tail -2 gpurun_out/r2_fuzz_fp.log
KS_DEBUG_JOIN_FP=1 KS_DEBUG_JOIN_SEGS=1 KS_DEBUG_FP_COARSEN=12 timeout -k 10 600 python tools/fuzz_parity.py --cases 300 --seed 12 > gpurun_out/r2_fuzz_fp2.log 2>&1; echo "fuzz fp2 rc=$?"; tail -2 gpurun_out/r2_fuzz_fp2.log
timeout -k 10 600 python tools/fuzz_parity.py --cases 300 --seed 13 > gpurun_out/r2_fuzz_c.log 2>&1; echo "fuzz c rc=$?"; tail -2 gpurun_out/r2_fuzz_c.log
bash tools/gpu/shard_breakdown.sh
