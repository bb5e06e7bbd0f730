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
# ... and the query launch (fused postings) of the full library, for the posting phases' share
cd /tmp
rm -rf $R/gpurun_out/ph_fullq
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES --kernel-include-regex "k_sketch_tiles" --output-format csv -d $R/gpurun_out/ph_fullq -- python3 $R/tools/sketch_only.py 1000000 10 1 protein 1 3 > /dev/null 2> $R/gpurun_out/ph_fullq.err
cd $R
python - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/ph_fullq/*/*_counter_collection.csv")[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    if r["Kernel_Name"].startswith("void k_sketch_tiles<0, 0"):
        acc[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
for d, c in sorted(acc.items(), key=lambda x: int(x[0])):
    w = c["SQ_WAVES"]
    print("dispatch", d, "waves", int(w), {k: round(v / w, 1) for k, v in c.items() if k != "SQ_WAVES"})
PY
