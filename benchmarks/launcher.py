"""`python bench.py --gpus N` without a launcher, and the GPU-free dry run of the rank plumbing."""
from __future__ import annotations

import json
import os
import socket
import subprocess
import sys
import time

import torch


def launch_ranks(args, script) -> int:
    """Start N fresh rank processes (one per GPU, the layout torch.distributed.run would give them: all devices visible,
    LOCAL_RANK picks one) BEFORE this process makes any GPU call, wait for them, and fail if any of them fails. Rank 0
    prints the JSON line straight to our stdout."""
    n = args.gpus
    if not args.dry_run and os.environ.get("QGTC_BENCH_SHARE_GPU", "0") in ("", "0"):
        have = torch.cuda.device_count()       # counting devices does not initialise the GPU
        if have < n:
            print(f"bench.py: --gpus {n} but only {have} GPU(s) visible", file=sys.stderr)
            return 2
    with socket.socket() as s:                 # a free rendezvous port on the loopback interface
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs on this driver
        procs.append(subprocess.Popen([sys.executable, script] + sys.argv[1:], env=env))
    rc = 0
    try:
        pending = list(procs)
        while pending:
            for p in list(pending):
                code = p.poll()
                if code is None:
                    continue
                pending.remove(p)
                if code != 0 and rc == 0:      # one rank failed: the others would wait in a collective forever
                    rc = code
                    for q in pending:
                        q.terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


def dry_run(args):
    """Launcher / collective plumbing without a GPU (CPU tests: `--dry-run --backend gloo`): every rank takes part in the
    same barrier, max-over-ranks and gathers as the real run - the end-of-epoch exchange with the real payload's shape
    (ragged per-batch float outputs [n_i, 10], n_i = 1190 + 7 i mod 50, every element = the batch id, round-robin shards,
    gathered padded; and the weak-scaled form) - and rank 0 prints the line's skeleton."""
    from qgtc_ppopp22_amd import dist as D

    rank, world, local = D.init_from_env(backend=args.backend or "gloo")
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    if os.environ.get("QGTC_BENCH_FAIL_RANK") == str(rank):   # test hook: a rank that dies before the collectives
        sys.exit(3)
    dev = torch.device("cpu")
    D.barrier()
    wall = D.max_over_ranks(0.001 * (rank + 1), dev)
    csum = torch.tensor([[float(1000 + rank)]], dtype=torch.float64)
    sums = D.gather_batch_summaries(csum, world, rank, world)
    ids = D.shard_round_robin(75, rank, world)
    counts = D.gather_batch_summaries(torch.tensor([[float(len(ids))]], dtype=torch.float64), world, rank, world)
    fake = lambda i: torch.full((1190 + (7 * i) % 50, 10), float(i))     # noqa: E731
    allout, nodes = D.gather_batch_outputs([fake(i) for i in ids], 75, rank, world, device=dev, classes=10)
    rep_out, rep_nodes = D.gather_batch_outputs([fake(100 * rank + j) for j in range(3)], 3, rank, world, replicas=True, device=dev, classes=10)
    rep_sums = D.gather_replica_summaries(torch.tensor([[float(rank), float(j)] for j in range(3)], dtype=torch.float64), world)
    # a rank that owns NO batch (fewer batches than ranks) still enters every collective with the common shape
    few, few_nodes = D.gather_batch_outputs([fake(i) for i in D.shard_round_robin(1, rank, world)], 1, rank, world, device=dev, classes=10)
    if rank == 0:
        ex = {"rank_checksums": [float(v) for v in sums.view(-1).tolist()],
              "batches_per_rank": [int(v) for v in counts.view(-1).tolist()],
              "gathered_output_shape": list(allout.shape),
              "gathered_output_nodes": [int(v) for v in nodes.tolist()],
              "gathered_output_first_values": [float(allout[i, 0, 0]) for i in range(allout.size(0))],
              "gathered_output_padding_is_zero": bool(all(float(allout[i, int(nodes[i]):].abs().sum()) == 0.0 for i in range(allout.size(0)))),
              "replica_output_shape": list(rep_out.shape),
              "replica_output_first_values": [float(rep_out[i, 0, 0]) for i in range(rep_out.size(0))],
              "replica_nodes": [int(v) for v in rep_nodes.tolist()],
              "replica_summaries": rep_sums.tolist(),
              "one_batch_shape": list(few.shape), "one_batch_nodes": [int(v) for v in few_nodes.tolist()]}
        print(json.dumps({"metric": "dry run (no GPU work)", "n_gpus": world, "ranks_seen": D.world_size(), "steps": args.steps,
                          "warmup": args.warmup, "max_wall_s": wall, "extras": ex}), flush=True)
    D.shutdown()
