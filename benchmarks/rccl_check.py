"""RCCL on a ONE-GPU box: a fresh process sets up a world-size-1 `nccl` (= RCCL) process group and pushes DEVICE tensors
through every exchange of qgtc_ppopp22_amd/dist.py - the calls BASELINE.json configs[4] depends on
(all_gather_into_tensor of ragged per-batch outputs and of summaries, all_reduce(MAX) of the elapsed time, barrier) - with
the world == 1 short-cuts bypassed (a live group always gets the real collective). Prints one `RCCL_WORLD1 {...}` line.

    python benchmarks/rccl_check.py [nccl|gloo]

Run by tests/test_aa_rccl_world1.py (which compares the nccl result with the gloo one) and by bench.py (extras). It must be
its own process: the process group is set up before anything else touches the GPU and torn down at exit."""
from __future__ import annotations

import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def run(backend: str) -> dict:
    import socket

    import torch

    from qgtc_ppopp22_amd import dist as D

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    t0 = time.perf_counter()
    rank, world, local = D.init_from_env(backend=backend, force_group=True)
    dev = torch.device("cuda", 0) if backend == "nccl" else torch.device("cpu")
    init_s = time.perf_counter() - t0
    assert torch.distributed.is_initialized() and torch.distributed.get_backend() == backend
    D.barrier()
    # ragged per-batch float outputs as an epoch leaves them: batch i is [1190 + 7 i mod 50, 10], every element = i + 0.25 j
    g = torch.Generator().manual_seed(11)
    outs = [(torch.full((1190 + (7 * i) % 50, 10), float(i)) + 0.25 * torch.arange(10.0)).to(dev) for i in range(75)]
    t0 = time.perf_counter()
    allout, nodes = D.gather_batch_outputs(outs, 75, rank, world, device=dev, classes=10)
    if dev.type == "cuda":
        torch.cuda.synchronize()
    gather_ms = (time.perf_counter() - t0) * 1e3
    rep, rep_nodes = D.gather_batch_outputs(outs[:3], 3, rank, world, replicas=True, device=dev, classes=10)
    summ = torch.rand((75, 2), generator=g, dtype=torch.float64).to(dev)
    s1 = D.gather_batch_summaries(summ, 75, rank, world)
    s2 = D.gather_replica_summaries(summ, world)
    mx = D.max_over_ranks(1.25, dev)
    ok = (allout.shape == (75, 1239, 10) and all(int(nodes[i]) == 1190 + (7 * i) % 50 for i in range(75))
          and all(torch.equal(allout[i, :int(nodes[i])], outs[i]) and float(allout[i, int(nodes[i]):].abs().sum()) == 0.0 for i in range(75))
          and torch.equal(rep[:3, :int(rep_nodes.max())], allout[:3, :int(rep_nodes.max())])
          and torch.equal(s1, summ) and torch.equal(s2, summ) and mx == 1.25)
    res = {"backend": torch.distributed.get_backend(), "ranks_seen": D.world_size(), "device": str(allout.device), "ok": bool(ok),
           "init_s": round(init_s, 2), "gather_outputs_ms": round(gather_ms, 2), "gathered_bytes": int(allout.numel() * 4),
           "checksum": float(allout.double().sum().item()), "summaries_checksum": float(s1.sum().item()), "max_over_ranks": mx}
    D.shutdown()
    return res


if __name__ == "__main__":
    print("RCCL_WORLD1 " + json.dumps(run(sys.argv[1] if len(sys.argv) > 1 else "nccl")), flush=True)
