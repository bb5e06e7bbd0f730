mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2_t26.log 2>&1; rc=$?; tail -3 gpurun_out/r2_t26.log; [ $rc -eq 0 ] || exit $rc
python tools/sketch_only.py 100000 16 5 dayhoff 0 20 2>/dev/null | python -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('c3 sketch', d['kernels_ms'])"
python tools/sketch_only.py 200000 24 5 hp 0 20 2>/dev/null | python -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('c5 sketch', d['kernels_ms'])"
python tools/sketch_only.py 1000000 16 5 dayhoff 0 10 2>/dev/null | python -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('1M dayhoff sketch', d['kernels_ms'])"
timeout -k 10 600 python tools/fuzz_parity.py --cases 400 --seed 41 > gpurun_out/r2_fuzz_q.log 2>&1; echo "fuzz rc=$?"; tail -1 gpurun_out/r2_fuzz_q.log
