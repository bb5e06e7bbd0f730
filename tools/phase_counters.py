"""Cumulative instruction mix of k_sketch_tiles by phase, from PMC passes over -DSK_STOP_AFTER=n diagnostic builds.

    # build the variants in-tree (they travel to the GPU box), then on the box, per variant n = 0..7 and the full library:
    #   KMERSEEK_AMD_LIB=kmerseek_amd/variants/libks_stop$n.so rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU \
    #       SQ_INSTS_LDS SQ_WAVE_CYCLES --output-format csv -d gpurun_out/ph_$n -- python3 tools/sketch_only.py
    python tools/phase_counters.py --build            # writes kmerseek_amd/variants/libks_stop{0..7}.so
    python tools/phase_counters.py gpurun_out         # prints the table from gpurun_out/ph_*/
"""
import collections
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PHASES = ["0 prologue (ticket, plan, offsets)", "1 residues -> LUT -> LDS", "2 hash 8 windows + bucket/slot", "3 bucket scan",
          "4 scatter to bucket order", "5 rank inside buckets", "6 distinct ranks, publish, stage representatives",
          "7 look-back, CSR offsets", "8 CSR write (full kernel, no postings)"]


def main():
    if "--build" in sys.argv:
        from kmerseek_amd import build
        os.makedirs(os.path.join(ROOT, "kmerseek_amd", "variants"), exist_ok=True)
        for n in range(8):
            print(build.build(force=True, defs=[f"SK_STOP_AFTER={n}"], out=os.path.join(ROOT, "kmerseek_amd", "variants", f"libks_stop{n}.so")))
        return
    base = sys.argv[1]
    prev = None
    print("| phase | VALU | SALU | LDS | LDS active cycles | of them bank conflicts | cumulative wave quad-cycles |\n|---|---|---|---|---|---|---|")
    for i, n in enumerate(["0", "1", "2", "3", "4", "5", "6", "7", "full"]):
        f = glob.glob(os.path.join(base, f"ph_{n}", "*", "*_counter_collection.csv"))
        if not f:
            continue
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f[0])):
            if r["Kernel_Name"].startswith("void k_sketch_tiles<0, 0"):
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        w = sum(acc["SQ_WAVES"]) / len(acc["SQ_WAVES"])
        d = {k: sum(v) / len(v) / w for k, v in acc.items() if k != "SQ_WAVES"}
        cols = ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT")
        dv = [d.get(c, 0) - (prev.get(c, 0) if prev else 0) for c in cols]
        print(f"| {PHASES[i]} | {dv[0]:.0f} | {dv[1]:.0f} | {dv[2]:.0f} | {dv[3]:.0f} | {dv[4]:.0f} | {d['SQ_WAVE_CYCLES']:.0f} |")
        prev = d


if __name__ == "__main__":
    main()
