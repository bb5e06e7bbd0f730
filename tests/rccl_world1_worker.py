"""Runs the multi-GPU layer's collectives through RCCL itself on ONE GPU (tests/test_gpu_rccl.py starts it as a child process).

backend "nccl" with world_size 1 + KS_DIST_FORCE_COLLECTIVES=1: every `torch.distributed` call of kmerseek_amd/dist.py —
the broadcasts of the residues (uint8 / int64 device tensors), the count exchange, the packed and the unpacked
`all_gather_into_tensor`, the asynchronous exchange of `begin_all_gather_hits_device` with the next step's kernels queued behind
it — goes through RCCL's communicator, its stream and its ordering against the stream the library's context launches on.  One
rank moves no bytes over xGMI; what this checks is that the calls are well formed for the backend the N-GPU job uses and that the
rows that come out equal the hit list that went in (reference semantics: `src/python/kmerseek/search.py:125-141`, one COO row per
(query, target) pair).  Prints one JSON line; exit code 0 = all checks passed.
"""
import json
import os
import socket
import sys
from datetime import timedelta

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def main():
    import torch
    import torch.distributed as dist
    import kmerseek_amd as ks
    from kmerseek_amd import dist as ksd, synth

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    stream = torch.cuda.Stream(dev)  # as bench.py: one stream for torch, the library's context and the collectives' ordering
    torch.cuda.set_stream(stream)
    os.environ["KS_DIST_FORCE_COLLECTIVES"] = "1"
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{free_port()}", rank=0, world_size=1, device_id=dev,
                            timeout=timedelta(seconds=90))
    done = []
    try:
        t = torch.ones(1, dtype=torch.int64, device=dev)
        dist.all_reduce(t)
        f = torch.tensor([1.5], dtype=torch.float64, device=dev)
        dist.all_reduce(f, op=dist.ReduceOp.MAX)
        dist.barrier()
        torch.cuda.synchronize(dev)
        assert int(t[0]) == 1 and float(f[0]) == 1.5
        done.append("all_reduce")

        # the residues of the index, as bench.py ships them
        p_res, p_off = synth.proteome(3000, stream=70)
        b_res, b_off = ksd.broadcast_batch(p_res, p_off, src=0, device=dev)
        assert b_res.dtype == torch.uint8 and b_off.dtype == torch.int64
        assert np.array_equal(b_res.cpu().numpy(), p_res) and np.array_equal(b_off.cpu().numpy().view(np.uint64), p_off)
        done.append("broadcast")

        for own_stream in (False, True):
            c = ks.Context(0, stream=None if own_stream else torch.cuda.current_stream(dev).cuda_stream)
            S = c.sketch_batch(p_res, p_off, 7, 1, "protein")
            ix = c.index_build(S)
            n_res = int(p_off[-1])
            tag = "own" if own_stream else "shared"

            def same(got, want, what):
                got = [g.cpu().numpy() if hasattr(g, "cpu") else g for g in got]
                assert len(got[0]) == len(want[0]), (what, len(got[0]), len(want[0]))
                for g, w in zip(got, want):
                    assert np.array_equal(g.astype(np.int64), w.astype(np.int64)), what

            # queries sharded (one shard): packed transport words, then the four columns as they are
            h = c.search(ix, S)
            want = h.to_host()
            assert h.count > 3000
            same(ksd.all_gather_hits_device(h, device=dev, sharded="queries", id_counts=(3000, 3000)), want, "packed")
            same(ksd.all_gather_hits_device(h, device=dev, sharded="queries", id_counts=None), want, "unpacked")
            # index sharded: the counting merge behind the gather
            same(ksd.all_gather_hits_device(h, device=dev, sharded="index", order="qid", id_counts=(3000, 3000)), want, "merge")
            same(ksd.all_gather_hits_device(h, device=dev, sharded="index", order="shard", id_counts=(3000, 3000)), want, "shard")
            # wide ids: most rows escape, the exchange repeats unpacked inside the call
            same(ksd.all_gather_hits_device(h, device=dev, sharded="index", order="qid", id_counts=(1 << 24, 1 << 24)), want, "wide")
            done.append(f"all_gather[{tag}]")

            # pipelined: the collective is started, the hit list freed, the next step queued, then finish()
            for id_counts in ((3000, 3000), (1 << 24, 1 << 24)):
                pend = ksd.begin_all_gather_hits_device(h, device=dev, sharded="index", order="qid", id_counts=id_counts)
                h.free()
                Q, h2 = c.sketch_search_device(ix, b_res.data_ptr(), b_off.data_ptr(), 3000, n_res)
                same(pend.finish(), want, f"pipelined {id_counts}")
                same(h2.to_host(), want, "next step")
                Q.free()
                h = h2
            done.append(f"pipelined[{tag}]")
            h.free(); ix.free(); S.free()
            c.close()
        torch.cuda.synchronize(dev)
    finally:
        dist.destroy_process_group()
    print(json.dumps({"ok": True, "backend": "nccl", "world_size": 1, "checks": done,
                      "rccl": ".".join(str(x) for x in torch.cuda.nccl.version())}))


if __name__ == "__main__":
    main()
