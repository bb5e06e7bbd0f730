mkdir -p gpurun_out
( for lib in kmerseek_amd/libkmerseek_amd.so kmerseek_amd/variants/libks_nowarm.so kmerseek_amd/variants/libks_warm1536.so; do
  echo "LIB $lib"
  KMERSEEK_AMD_LIB=$PWD/$lib python tools/sketch_only.py 1000000 10 1 protein 0
  KMERSEEK_AMD_LIB=$PWD/$lib python tools/sketch_only.py 1000000 10 1 protein 1
  KMERSEEK_AMD_LIB=$PWD/$lib python tools/sketch_only.py 1000000 16 5 dayhoff 0
done ) > gpurun_out/r2_warm.log 2>&1
python bench.py --steps 20 --warmup 5 --queries 125000 --no-config4 --no-cpu-baseline --no-aux > gpurun_out/r2_q125k_b.json 2> gpurun_out/r2_q125k_b.err; echo "q125k rc=$?"
