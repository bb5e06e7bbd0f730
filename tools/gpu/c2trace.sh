set -x
R=$PWD
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/c2_kt -o kt -- python3 $R/bench.py --steps 20 --warmup 5 --queries 10000 --targets 10000 --ksize 7 --no-config4 --no-cpu-baseline --no-aux > $R/gpurun_out/c2_kt.json 2> $R/gpurun_out/c2_kt.err
cd $R
ls gpurun_out/c2_kt
