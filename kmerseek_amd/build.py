"""Builds libkmerseek_amd.so (HIP, gfx950 only) in-tree with hipcc.

    python -m kmerseek_amd.build [--force]

hipcc cross-compiles without a GPU; the .so travels to the GPU box with the repo snapshot.
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SO = os.path.join(HERE, "libkmerseek_amd.so")
SOURCES = ["ks_ctx.hip", "ks_prims.hip", "ks_msd.hip", "ks_copy.hip", "ks_sketch.hip", "ks_search.hip", "ks_api.hip", "ks_host.cpp", "ks_ingest.cpp", "ks_input.cpp"]
HEADERS = ["ks_common.h", "ks_device.h", "ks_input.h", os.path.join("..", "..", "include", "kmerseek_amd.h"),
           os.path.join("..", "..", "include", "kmerseek_host.hpp"), os.path.join("..", "..", "include", "kmerseek_host_c.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-shared", "-std=c++17", "-Wall", "-Wno-unused-function",
         "-pthread", "-lz", "-ldl"]


def stale() -> bool:
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False, defs=(), out: str = SO) -> str:
    """defs: extra -D macros (kernel-tuning variants, written to `out` instead of the default .so)."""
    if not force and not defs and out == SO and not stale():
        return SO
    srcs = [os.path.join(CSRC, f) for f in SOURCES if os.path.exists(os.path.join(CSRC, f))]
    cmd = [HIPCC] + FLAGS + [f"-D{d}" for d in defs] + ["-o", out] + srcs
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd, cwd=CSRC)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
