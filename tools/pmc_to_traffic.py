#!/usr/bin/env python3
"""Turn the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; collected separately as the MI355X guide prescribes)
into profiles/traffic.json: HBM bytes per launch of the byte-moving kernels.

gfx950 correction (MI355X_MICROARCH.md §HBM): FETCH_SIZE reports exactly half the bytes of a wide coalesced read, so it
is doubled; WRITE_SIZE is exact.  Both counters are in KiB.

    python tools/pmc_to_traffic.py gpurun_out/pmc_fetch/f_counter_collection.csv gpurun_out/pmc_write/w_counter_collection.csv
"""
import collections
import csv
import json
import os
import sys

NAMES = {  # profiler kernel name prefix -> library timing name
    "void k_sketch_tiles<0, 0": "sketch_tiles",          # (any k: the third template argument is the folded-in k-mer size)
    "void k_sketch_tiles<0, 1": "sketch_tiles.compact",
    "k_msd_local(": "msd_local",
    "void k_msd_scatter<256>": "msd_scatter.level1",
    "void k_msd_scatter<512>": "msd_scatter.level2",
    "void k_msd_scatter<1024>": "msd_scatter.level2",
    "k_bucket_scatter": "bucket_scatter",
    "void k_bucket_scatter<": "bucket_scatter",          # (template on the posting format since round 3)
    "void k_radix_scatter<unsigned int, 1>": "radix_scatter.qpart",
    "void k_radix_hist<1>": "radix_hist.qpart",
    "k_join_buckets": "join_buckets",
    "k_join_sparse": "join_buckets",
    "void k_join_buckets<": "join_buckets",
    "void k_join_sparse<": "join_buckets",
    "void k_radix_scatter<unsigned long, 0>": "radix_scatter.index",
}


def load(path, counter):
    out = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            out[r["Kernel_Name"]].append((float(r["Counter_Value"]), int(r["Grid_Size"])))
    return out


def traffic(fetch_csv, write_csv):
    f = load(fetch_csv, "FETCH_SIZE")
    w = load(write_csv, "WRITE_SIZE")
    res = {}
    for kname, rows in f.items():
        for pref, lib in NAMES.items():
            if kname.startswith(pref):
                # the per-step (query-side) launches: the grid size that occurs most often (the one-off index build differs)
                grids = collections.Counter(x[1] for x in rows)
                g = max(grids, key=lambda v: (grids[v], v))
                fetch = [x[0] for x in rows if x[1] == g]
                write = [x[0] for x in w.get(kname, []) if x[1] == g]
                fb = 2.0 * 1024.0 * sum(fetch) / len(fetch)
                wb = 1024.0 * sum(write) / max(len(write), 1)
                res[lib] = {"fetch_bytes": fb, "write_bytes": wb, "launches_sampled": len(fetch)}
    return {"per_launch_bytes": {k: v["fetch_bytes"] + v["write_bytes"] for k, v in res.items()}, "detail": res}


def stamp():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import subprocess
    import bench  # kernel_sources_sha: bench.py reports whether the kernel sources changed since this collection
    try:
        commit = subprocess.check_output(["git", "-C", root, "rev-parse", "--short", "HEAD"], text=True).strip()
        dirty = bool(subprocess.check_output(["git", "-C", root, "status", "--porcelain", "--", "kmerseek_amd/csrc"], text=True).strip())
    except Exception:
        commit, dirty = None, None
    return {"commit": commit, "kernel_sources_dirty_vs_commit": dirty, "kernel_sources_sha16": bench.kernel_sources_sha(),
            "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over `bench.py --steps 1 --warmup 0`",
            "correction": "FETCH_SIZE x2 (gfx950 reports half of a wide coalesced read), WRITE_SIZE exact; KiB -> bytes"}


def main():
    out = stamp()
    out.update(traffic(sys.argv[1], sys.argv[2]))
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "traffic.json")
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps(out["per_launch_bytes"], indent=1))


if __name__ == "__main__":
    main()
