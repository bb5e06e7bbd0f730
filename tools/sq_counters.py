"""Per-wave SQ counter table from rocprofv3 --pmc passes (counter_collection.csv files), for the main kernels.

    python tools/sq_counters.py pass1.csv [pass2.csv ...] > profiles/rNN_sq_counters.md

Each kernel's most frequent grid size is taken (the 1M-query launch); counters are divided by SQ_WAVES.
"""
import collections
import csv
import sys

KERNELS = ["void k_sketch_tiles<0, 0", "void k_sketch_tiles<0, 1", "void k_bucket_scatter<", "void k_join_buckets<", "k_join_buckets_keys", "k_msd_local(",
           "void k_msd_scatter<256>", "void k_msd_scatter<512>", "k_kmerpos_tiles"]


def main():
    data = collections.defaultdict(lambda: collections.defaultdict(list))  # kernel -> counter -> [(value, grid)]
    for path in sys.argv[1:]:
        for r in csv.DictReader(open(path)):
            for k in KERNELS:
                if r["Kernel_Name"].startswith(k):
                    data[k][r["Counter_Name"]].append((float(r["Counter_Value"]), int(r["Grid_Size"])))
    print("# SQ counters per wave (rocprofv3 --pmc; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are quad-cycles)\n")
    for k in KERNELS:
        if k not in data:
            continue
        c = data[k]
        grid = collections.Counter(g for _, g in c.get("SQ_WAVES", [])).most_common(1)
        if not grid:
            continue
        grid = grid[0][0]
        avg = {n: sum(v for v, g in rows if g == grid) / max(1, sum(1 for _, g in rows if g == grid)) for n, rows in c.items()}
        waves = avg.get("SQ_WAVES", 0)
        print(f"## `{k}`  (grid {grid}, {waves:.0f} waves)\n\n| counter | per wave |\n|---|---|")
        for n in sorted(avg):
            if n != "SQ_WAVES" and waves:
                print(f"| {n} | {avg[n] / waves:.1f} |")
        print()


if __name__ == "__main__":
    main()
