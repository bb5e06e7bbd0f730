# SQ counters of the sketch tile kernel's QUERY launch (fused postings), 1M proteins, for one library build.
#   bash tools/gpu/sk_counters.sh TAG [path/to/lib.so]      -> gpurun_out/skc_TAG.md
R=$PWD
TAG=$1
LIB=$(realpath ${2:-$R/kmerseek_amd/libkmerseek_amd.so})
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
export KMERSEEK_AMD_LIB=$LIB
rm -rf $R/gpurun_out/skc_${TAG}_1 $R/gpurun_out/skc_${TAG}_2
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-include-regex "k_sketch_tiles" --output-format csv -d $R/gpurun_out/skc_${TAG}_1 -o s1 -- python3 $R/tools/sketch_only.py 1000000 10 1 protein 1 3 > /dev/null 2> $R/gpurun_out/skc_${TAG}_1.err || { tail -3 $R/gpurun_out/skc_${TAG}_1.err; exit 1; }
rocprofv3 --pmc SQ_WAVES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-include-regex "k_sketch_tiles" --output-format csv -d $R/gpurun_out/skc_${TAG}_2 -o s2 -- python3 $R/tools/sketch_only.py 1000000 10 1 protein 1 3 > /dev/null 2> $R/gpurun_out/skc_${TAG}_2.err || { tail -3 $R/gpurun_out/skc_${TAG}_2.err; exit 1; }
cd $R
python tools/sq_counters.py $(find gpurun_out/skc_${TAG}_1 gpurun_out/skc_${TAG}_2 -name "*counter_collection.csv") > gpurun_out/skc_${TAG}.md
cat gpurun_out/skc_${TAG}.md
