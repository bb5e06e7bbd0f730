# rocprofv3 evidence for every BASELINE config that runs on one GPU (and the 125k-query shard that decides strong scaling):
# per config a kernel-trace stats pass and the two PMC traffic passes (FETCH_SIZE / WRITE_SIZE apart, as
# MI355X_MICROARCH.md prescribes), plus the SQ counters of the headline config.  Run through gpurun from the repo root;
# `python tools/collect_profiles.py rNN` copies the summaries into profiles/.
#   gpurun --timeout 1200 -- 'bash tools/gpu/profile_configs.sh'
R=$PWD
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
run_cfg() { # name, bench args...
  name=$1; shift
  B="python3 $R/bench.py --no-cpu-baseline --no-config4 --no-aux $*"
  rm -rf $R/gpurun_out/prof_$name
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$name/kt -o kt -- $B --steps 5 --warmup 2 > $R/gpurun_out/prof_$name.json 2> $R/gpurun_out/prof_$name.err || { echo "$name: kernel trace failed"; tail -3 $R/gpurun_out/prof_$name.err; return 1; }
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_$name/fetch -o f -- $B --steps 1 --warmup 0 > /dev/null 2>> $R/gpurun_out/prof_$name.err || { echo "$name: FETCH_SIZE pass failed"; return 1; }
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_$name/write -o w -- $B --steps 1 --warmup 0 > /dev/null 2>> $R/gpurun_out/prof_$name.err || { echo "$name: WRITE_SIZE pass failed"; return 1; }
  echo "$name done"
}
run_cfg c4_1m_protein_k10_s1 &&
run_cfg c2_10k_protein_k7_s1 --queries 10000 --targets 10000 --ksize 7 &&
run_cfg c3_100k_dayhoff_k16_s5 --queries 100000 --targets 100000 --ksize 16 --scaled 5 --moltype dayhoff &&
run_cfg c5_200k_hp_k24_s5 --mode index-sharded --c4-proteins 200000 &&
run_cfg shard_125k_of_1m --queries 125000 --targets 1000000 || exit 1
B="python3 $R/bench.py --no-cpu-baseline --no-config4 --no-aux"
rm -rf $R/gpurun_out/pmc_sq1 $R/gpurun_out/pmc_sq2
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $R/gpurun_out/pmc_sq1 -o s1 -- $B --steps 1 --warmup 0 > /dev/null 2> $R/gpurun_out/pmc_sq1.err
rocprofv3 --pmc SQ_WAVES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $R/gpurun_out/pmc_sq2 -o s2 -- $B --steps 1 --warmup 0 > /dev/null 2> $R/gpurun_out/pmc_sq2.err
cd $R
find gpurun_out/prof_* gpurun_out/pmc_sq1 gpurun_out/pmc_sq2 -name "*.csv" | wc -l
