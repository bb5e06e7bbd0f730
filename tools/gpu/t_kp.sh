mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_host_index.py tests/test_gpu_realdata.py -m gpu -x -q -k "kmer or positions or ticket or real" 2>&1 | tail -3 || exit 1
python tools/kmerpos_bench.py 2>&1 | tail -2
python tools/kmerpos_bench.py --ksize 24 --scaled 5 --moltype hp 2>&1 | tail -1
