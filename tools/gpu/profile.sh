# rocprofv3 evidence for profiles/: kernel-trace stats of the bench command, HBM traffic (FETCH_SIZE / WRITE_SIZE in separate
# --pmc passes, as MI355X_MICROARCH.md prescribes), SQ counters of the main kernels.  Run through gpurun from the repo root.
set -x
R=$PWD
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-config4 --no-aux"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_kt -o kt -- $B --steps 5 --warmup 2 > $R/gpurun_out/prof_kt.json 2> $R/gpurun_out/prof_kt.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -o f -- $B --steps 1 --warmup 0 > /dev/null 2> $R/gpurun_out/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -o w -- $B --steps 1 --warmup 0 > /dev/null 2> $R/gpurun_out/pmc_write.err
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $R/gpurun_out/pmc_sq1 -o s1 -- $B --steps 1 --warmup 0 > /dev/null 2> $R/gpurun_out/pmc_sq1.err
rocprofv3 --pmc SQ_WAVES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $R/gpurun_out/pmc_sq2 -o s2 -- $B --steps 1 --warmup 0 > /dev/null 2> $R/gpurun_out/pmc_sq2.err
cd $R
find gpurun_out/prof_kt gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_sq1 gpurun_out/pmc_sq2 -name "*.csv" | head -30
