set -x
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r2_t4.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2_t4.log; tail -7 gpurun_out/r2_t4.log
( for sp in 4096 8192 12288 16384 18944; do
  KS_DEBUG_SPAN=$sp python tools/sketch_only.py 100000 16 5 dayhoff 0
  KS_DEBUG_SPAN=$sp python tools/sketch_only.py 200000 24 5 hp 0
  KS_DEBUG_SPAN=$sp python tools/sketch_only.py 1000000 16 5 dayhoff 0
done
KS_DEBUG_NO_COMPACT=1 python tools/sketch_only.py 1000000 16 5 dayhoff 0
python tools/sketch_only.py 1000000 16 5 dayhoff 1
python tools/sketch_only.py 1000000 10 1 protein 0
python tools/sketch_only.py 1000000 10 2 protein 0
) > gpurun_out/r2_sweep1.log 2>&1
