#!/usr/bin/env python3
"""Copies the rocprofv3 summaries that `tools/gpu/profile.sh` left under gpurun_out/ (scratch) into profiles/ (tracked),
named per round, and regenerates profiles/traffic.json and the SQ counter table from them.

    gpurun -- 'bash tools/gpu/profile.sh'        # on the MI355X box
    python tools/collect_profiles.py r02         # here
"""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
    g, p = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
    pairs = [("prof_kt/kt_kernel_stats.csv", f"{tag}_bench_kernel_stats.csv"),
             ("pmc_fetch/f_counter_collection.csv", f"{tag}_pmc_fetch_size.csv"),
             ("pmc_write/w_counter_collection.csv", f"{tag}_pmc_write_size.csv")]
    for src, dst in pairs:
        shutil.copyfile(os.path.join(g, src), os.path.join(p, dst))
        print("copied", dst)
    # the bench line of the profiled run (HIP-event durations to set beside the profiler's)
    line = [l for l in open(os.path.join(g, "prof_kt.json")) if l.startswith("{")][-1]
    open(os.path.join(p, f"{tag}_bench_line_under_rocprof.json"), "w").write(line)
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "pmc_to_traffic.py"), os.path.join(p, f"{tag}_pmc_fetch_size.csv"),
                           os.path.join(p, f"{tag}_pmc_write_size.csv")])
    md = subprocess.check_output([sys.executable, os.path.join(ROOT, "tools", "sq_counters.py"), os.path.join(g, "pmc_sq1", "s1_counter_collection.csv"),
                                  os.path.join(g, "pmc_sq2", "s2_counter_collection.csv")], text=True)
    open(os.path.join(p, f"{tag}_sq_counters.md"), "w").write(md)
    print("wrote", f"{tag}_sq_counters.md")


if __name__ == "__main__":
    main()
