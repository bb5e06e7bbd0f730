R=$PWD
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for n in 0 1 2 3 4 5 6 7; do
  rm -rf $R/gpurun_out/ph_$n
  KMERSEEK_AMD_LIB=$R/kmerseek_amd/variants/libks_stop$n.so rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES --kernel-include-regex "k_sketch_tiles" --output-format csv -d $R/gpurun_out/ph_$n -- python3 $R/tools/sketch_only.py > /dev/null 2> $R/gpurun_out/ph_$n.err || { echo "variant $n failed"; tail -3 $R/gpurun_out/ph_$n.err; exit 1; }
  echo "variant $n done"
done
rm -rf $R/gpurun_out/ph_full
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES --kernel-include-regex "k_sketch_tiles" --output-format csv -d $R/gpurun_out/ph_full -- python3 $R/tools/sketch_only.py > /dev/null 2> $R/gpurun_out/ph_full.err
cd $R
python tools/phase_counters.py gpurun_out | tee gpurun_out/r2_phase_counters.md
