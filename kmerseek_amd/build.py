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
    tmp = f"{out}.tmp.{os.getpid()}"  # (linked beside the target and renamed: nobody ever sees half a library, see wait_until_built)
    cmd = [HIPCC] + FLAGS + [f"-D{d}" for d in defs] + ["-o", tmp] + srcs
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    try:
        subprocess.check_call(cmd, cwd=CSRC)
        os.replace(tmp, out)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)
    return out


def wait_until_built(timeout_s: float = 900.0, poll_s: float = 0.5) -> str:
    """For the ranks of a multi-process job that do NOT build: returns once the library is up to date (at once when it
    travelled with the snapshot), raises after `timeout_s`.  Rank 0 builds before it joins the process group; a rank that waited
    for it inside `init_process_group` would run into that call's (deliberately short) timeout instead."""
    import time
    t0 = time.time()
    while stale():
        if time.time() - t0 > timeout_s:
            raise TimeoutError(f"{SO} was not (re)built within {timeout_s:.0f} s")
        time.sleep(poll_s)
    return SO


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
