set -x
R=$PWD
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-config4 --no-aux"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-include-regex "k_join_buckets" --output-format csv -d $R/gpurun_out/pmc_jsq1 -o s1 -- $B --steps 1 --warmup 0 > /dev/null 2> $R/gpurun_out/pmc_jsq1.err
rocprofv3 --pmc SQ_WAVES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-include-regex "k_join_buckets" --output-format csv -d $R/gpurun_out/pmc_jsq2 -o s2 -- $B --steps 1 --warmup 0 > /dev/null 2> $R/gpurun_out/pmc_jsq2.err
cd $R
python tools/sq_counters.py $(find gpurun_out/pmc_jsq1 gpurun_out/pmc_jsq2 -name "*counter_collection.csv") | tee gpurun_out/jn_sq.md
