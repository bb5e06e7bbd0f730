"""The collectives of kmerseek_amd/dist.py through RCCL itself (backend "nccl"), on the one GPU of the test box.

The N-rank job (bench.py --gpus N) runs broadcasts and all-gathers over RCCL / xGMI; the CPU suite covers the same
code with gloo on host tensors (tests/test_dist_gloo.py).  What neither shows is the backend itself: device buffers, dtypes,
`async_op` + `wait()`, and the order of RCCL's stream against the stream the library's context launches on.  A ONE-rank "nccl"
process group with KS_DIST_FORCE_COLLECTIVES=1 runs all of that (tests/rccl_world1_worker.py, a child process so that a
communicator that cannot come up costs this test and not the run), and bench.py's N-rank path is taken the same way
(KS_BENCH_FORCE_PG=1: process group, broadcasts, the sharded == unsharded gates of both splits).
"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(cmd, env_extra, timeout):
    env = dict(os.environ)
    env.update(env_extra)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, f"rc={r.returncode}\n--- stdout\n{r.stdout[-3000:]}\n--- stderr\n{r.stderr[-6000:]}"
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert lines, r.stdout[-2000:]
    return json.loads(lines[-1])


def test_dist_collectives_run_through_rccl_with_one_rank():
    d = _run([sys.executable, os.path.join(ROOT, "tests", "rccl_world1_worker.py")], {}, 600)
    assert d["ok"] and d["backend"] == "nccl"
    for what in ("all_reduce", "broadcast", "all_gather[shared]", "pipelined[shared]", "all_gather[own]", "pipelined[own]"):
        assert what in d["checks"], d


def test_bench_takes_its_n_rank_path_over_rccl_with_one_rank():
    d = _run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--queries", "40000",
              "--targets", "60000", "--c4-proteins", "20000", "--no-cpu-baseline", "--no-aux"],
             {"KS_BENCH_FORCE_PG": "1", "KS_BENCH_STRICT": "1"}, 900)
    assert d["config"]["collective_backend"] == "nccl" and d["n_gpus"] == 1
    assert d["self_check"]["ok"] is True
    assert d["sharded_equals_unsharded"]["equal"] is True, d["sharded_equals_unsharded"]
    c4 = d["config4_index_sharded"]
    assert "error" not in c4, c4
    assert c4["config"]["collective_backend"] == "nccl"
    assert c4["sharded_equals_unsharded"]["equal"] is True, c4["sharded_equals_unsharded"]
    assert c4["self_check"]["ok"] is True
