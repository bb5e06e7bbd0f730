"""End-to-end FASTA ingest: file on disk -> sketches in host memory, stages overlapped vs back to back.

    python tools/ingest_bench.py [--seqs N] [--scaled S] [--gz]

Writes a synthetic proteome (kmerseek_amd.synth, 60 residues per line) to $TMPDIR, then runs
host.sketch_fasta(pipeline=False) and (pipeline=True) on it and prints one JSON line per run.
"""
import argparse
import gzip
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from kmerseek_amd import host, synth


def write_fasta(path, res, offs, gz):
    op = gzip.open if gz else open
    with op(path, "wb") as f:
        chunks = []
        for i in range(len(offs) - 1):
            s = res[int(offs[i]):int(offs[i + 1])].tobytes()
            chunks.append(b">sp|Q%07d|SYN_%d synthetic protein %d\n" % (i, i, i))
            chunks.extend(s[j:j + 60] + b"\n" for j in range(0, len(s), 60))
            if len(chunks) > 100000:
                f.write(b"".join(chunks))
                chunks = []
        f.write(b"".join(chunks))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seqs", type=int, default=1_000_000)
    ap.add_argument("--ksize", type=int, default=10)
    ap.add_argument("--scaled", type=int, default=1)
    ap.add_argument("--moltype", default="protein")
    ap.add_argument("--gz", action="store_true")
    ap.add_argument("--validate", action="store_true")
    ap.add_argument("--batch-mib", type=int, default=16)
    a = ap.parse_args()
    res, offs = synth.proteome(a.seqs, stream=9)
    path = os.path.join(tempfile.gettempdir(), f"ks_ingest_{a.seqs}.fasta" + (".gz" if a.gz else ""))
    t0 = time.time()
    write_fasta(path, res, offs, a.gz)
    size = os.path.getsize(path)
    print(json.dumps({"fasta": path, "bytes": size, "write_s": round(time.time() - t0, 2)}), flush=True)
    try:
        for pipeline in (False, True, True):
            t0 = time.perf_counter()
            names, o, m, ab, st = host.sketch_fasta(path, a.ksize, a.scaled, a.moltype, validate=a.validate,
                                                    batch_residues=a.batch_mib << 20, pipeline=pipeline)
            el = time.perf_counter() - t0
            st = {k: (round(v, 4) if isinstance(v, float) else v) for k, v in st.items()}
            print(json.dumps({"pipeline": pipeline, "records": len(names), "hashes": int(len(m)), "call_s": round(el, 3),
                              "residues_per_s": st["residues"] / st["wall_s"], "file_mb_per_s": size / st["wall_s"] / 1e6,
                              "stats": st}), flush=True)
            del names, o, m, ab
    finally:
        os.unlink(path)


if __name__ == "__main__":
    main()
