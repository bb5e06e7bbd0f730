R=$PWD
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
  rm -rf $R/gpurun_out/pmc_pf$v
  KMERSEEK_AMD_LIB=$R/kmerseek_amd/variants/libks_pf$v.so rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS --kernel-include-regex "k_sketch_tiles" --output-format csv -d $R/gpurun_out/pmc_pf$v -o s -- python3 $R/bench.py --no-cpu-baseline --no-config4 --no-aux --steps 1 --warmup 0 > /dev/null 2> $R/gpurun_out/pmc_pf$v.err
  echo "== SK_POSFIX=$v"; python3 $R/tools/sq_counters.py $(find $R/gpurun_out/pmc_pf$v -name "*counter_collection.csv") | grep -A14 "k_sketch_tiles<0, 0>"
done
