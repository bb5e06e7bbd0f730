mkdir -p gpurun_out
timeout -k 10 900 python tools/fuzz_parity.py --cases 700 --seed 5 > gpurun_out/r2_fuzz_a.log 2>&1; echo "fuzz a rc=$?"; tail -3 gpurun_out/r2_fuzz_a.log
timeout -k 10 900 python tools/fuzz_parity.py --cases 400 --seed 77 > gpurun_out/r2_fuzz_b.log 2>&1; echo "fuzz b rc=$?"; tail -3 gpurun_out/r2_fuzz_b.log
timeout -k 10 900 python tools/fuzz_parity.py --big --cases 14 --seed 9 > gpurun_out/r2_fuzz_big.log 2>&1; echo "fuzz big rc=$?"; tail -3 gpurun_out/r2_fuzz_big.log
