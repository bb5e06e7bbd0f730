#!/usr/bin/env python3
"""Copies the rocprofv3 summaries that `tools/gpu/profile_configs.sh` left under gpurun_out/ (scratch) into profiles/ (tracked),
named per round and per BASELINE config, and regenerates profiles/traffic.json (one entry per config) and the SQ counter table.

    gpurun --timeout 1200 -- 'bash tools/gpu/profile_configs.sh'     # on the MI355X box
    python tools/collect_profiles.py r03                             # here
"""
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
CONFIGS = ["c4_1m_protein_k10_s1", "c2_10k_protein_k7_s1", "c3_100k_dayhoff_k16_s5", "c5_200k_hp_k24_s5", "shard_125k_of_1m"]


def one(pattern):
    f = glob.glob(pattern, recursive=True)  # (gpurun merges into gpurun_out/: files of earlier runs may still lie there)
    return max(f, key=os.path.getmtime) if f else None


def main():
    import pmc_to_traffic
    tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
    g, p = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
    traffic = {}
    for cfg in CONFIGS:
        d = os.path.join(g, "prof_" + cfg)
        kt = one(os.path.join(d, "kt", "**", "*kernel_stats.csv"))
        fe = one(os.path.join(d, "fetch", "**", "*counter_collection.csv"))
        wr = one(os.path.join(d, "write", "**", "*counter_collection.csv"))
        if not (kt and fe and wr):
            print("missing passes for", cfg)
            continue
        shutil.copyfile(kt, os.path.join(p, f"{tag}_{cfg}_kernel_stats.csv"))
        shutil.copyfile(fe, os.path.join(p, f"{tag}_{cfg}_pmc_fetch_size.csv"))
        shutil.copyfile(wr, os.path.join(p, f"{tag}_{cfg}_pmc_write_size.csv"))
        # the bench line of the profiled run (HIP-event durations to set beside the profiler's)
        line = [l for l in open(os.path.join(g, f"prof_{cfg}.json")) if l.startswith("{")][-1]
        open(os.path.join(p, f"{tag}_{cfg}_bench_line_under_rocprof.json"), "w").write(line)
        traffic[cfg] = pmc_to_traffic.traffic(fe, wr)
        print("collected", cfg)
    out = pmc_to_traffic.stamp()
    out["per_config"] = traffic
    if CONFIGS[0] in traffic:  # (bench.py reads the headline config's figures from the top level)
        out["per_launch_bytes"] = traffic[CONFIGS[0]]["per_launch_bytes"]
        out["detail"] = traffic[CONFIGS[0]]["detail"]
    json.dump(out, open(os.path.join(p, "traffic.json"), "w"), indent=1)
    s1, s2 = one(os.path.join(g, "pmc_sq1", "**", "*counter_collection.csv")), one(os.path.join(g, "pmc_sq2", "**", "*counter_collection.csv"))
    if s1 and s2:
        md = subprocess.check_output([sys.executable, os.path.join(ROOT, "tools", "sq_counters.py"), s1, s2], text=True)
        open(os.path.join(p, f"{tag}_sq_counters.md"), "w").write(md)
        print("wrote", f"{tag}_sq_counters.md")


if __name__ == "__main__":
    main()
