#!/usr/bin/env python3
"""Build a side copy of the library for A/B runs on one box (tools/ab_sketch.py, KMERSEEK_AMD_LIB):

    python tools/build_variant.py NAME [--rev GITREV] [-D MACRO[=V] ...]

--rev: take kmerseek_amd/csrc and include/ as they were at that revision (default: the working tree).
The result is kmerseek_amd/variants/libks_NAME.so (git-ignored; it travels to the GPU box with the snapshot)."""
import argparse
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kmerseek_amd import build as B  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("name")
    ap.add_argument("--rev")
    ap.add_argument("-D", action="append", default=[])
    a = ap.parse_args()
    out_dir = os.path.join(ROOT, "kmerseek_amd", "variants")
    os.makedirs(out_dir, exist_ok=True)
    out = os.path.join(out_dir, f"libks_{a.name}.so")
    with tempfile.TemporaryDirectory() as d:
        if a.rev:
            subprocess.check_call(f"git archive {a.rev} kmerseek_amd/csrc include | tar -x -C {d}", shell=True, cwd=ROOT)
        else:
            shutil.copytree(os.path.join(ROOT, "kmerseek_amd", "csrc"), os.path.join(d, "kmerseek_amd", "csrc"))
            shutil.copytree(os.path.join(ROOT, "include"), os.path.join(d, "include"))
        csrc = os.path.join(d, "kmerseek_amd", "csrc")
        srcs = [os.path.join(csrc, f) for f in B.SOURCES if os.path.exists(os.path.join(csrc, f))]
        cmd = [B.HIPCC] + B.FLAGS + [f"-D{x}" for x in a.D] + ["-o", out] + srcs
        subprocess.check_call(cmd, cwd=csrc)
    print(out)


if __name__ == "__main__":
    main()
