"""Multi-GPU sharding of the cluster batches: one process per GPU, no collective on the data path.

Cluster batches are independent (main_qgtc.py:113-154 carries nothing from one batch to the next),
so batch i runs on rank i mod world_size with the (tiny) packed weights replicated. The only
exchange is an end-of-epoch gather of per-batch results — RCCL over xGMI on the GPU box
(backend "nccl"), gloo in the CPU tests. The reference has no distributed code at all.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def init_from_env(backend: str | None = None, force_group: bool = False):
    """Initialise torch.distributed from RANK/WORLD_SIZE/LOCAL_RANK/MASTER_* (torchrun). Returns
    (rank, world_size, local_rank). A single process needs no process group and gets none - unless `force_group`
    (or QGTC_FORCE_COLLECTIVES=1) asks for one: with a group alive every gather / reduce below runs its real collective
    even at world size 1, which is how the RCCL path is exercised on a one-GPU box (tests/test_aa_rccl_world1.py)."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    force_group = force_group or os.environ.get("QGTC_FORCE_COLLECTIVES", "0") not in ("", "0")
    if (world > 1 or force_group) and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def collectives_live(world: int) -> bool:
    """The exchange functions below skip their collective only when there is nobody to talk to AND no process group: a
    group that was set up at world size 1 (force_group) still gets the real call."""
    return world > 1 or dist.is_initialized()


def world_size() -> int:
    """Ranks the process group actually has (1 without a group): bench.py prints it as `ranks_seen`."""
    return dist.get_world_size() if dist.is_initialized() else 1


def shutdown():
    if dist.is_initialized():
        dist.destroy_process_group()


def shard_round_robin(n_batches: int, rank: int, world: int):
    """Batch ids owned by `rank`: i with i % world == rank (BASELINE.json config 5)."""
    return list(range(rank, n_batches, world))


def owner_of(batch_id: int, world: int) -> int:
    return batch_id % world


def gather_batch_summaries(local: torch.Tensor, n_batches: int, rank: int, world: int) -> torch.Tensor:
    """End-of-epoch gather. `local` is [n_local, D] (one row of D summary values — checksum, shape,
    timing — per batch this rank owns, in shard order). Returns [n_batches, D] on every rank, rows
    in global batch order. One all_gather on buffers padded to the largest shard."""
    if not collectives_live(world):
        return local
    if local.is_cuda and dist.get_backend() == "gloo":
        # (two ranks sharing one GPU in the tests: RCCL refuses duplicate devices, gloo gathers host buffers)
        return gather_batch_summaries(local.cpu(), n_batches, rank, world).to(local.device)
    d = local.size(1)
    per = (n_batches + world - 1) // world
    buf = torch.zeros((per, d), dtype=local.dtype, device=local.device)
    buf[: local.size(0)] = local
    out = torch.empty((world * per, d), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, buf)
    out = out.view(world, per, d)
    rows = [out[owner_of(i, world), i // world] for i in range(n_batches)]
    return torch.stack(rows)


def gather_replica_summaries(local: torch.Tensor, world: int) -> torch.Tensor:
    """Weak scaling (every rank ran ALL its own batches): [n, D] per rank -> [world * n, D], rank-major, on every rank."""
    if not collectives_live(world):
        return local
    if local.is_cuda and dist.get_backend() == "gloo":
        return gather_replica_summaries(local.cpu(), world).to(local.device)
    out = torch.empty((world * local.size(0), local.size(1)), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local.contiguous())
    return out


def gather_batch_outputs(outs, n_batches: int, rank: int, world: int, replicas: bool = False, device=None, classes: int | None = None):
    """SURVEY.md 8e's exchange with the REAL payload: the per-batch float32 outputs of an epoch ([n_i, C] each), gathered
    to every rank with one all_gather_into_tensor on buffers padded to the largest batch (and, sharded, to the largest
    shard), plus the node counts (one more small all_gather: the batches are ragged).
    Sharded (replicas False): rank r holds batches r, r + world, ..; returns ([n_batches, max_n, C], nodes[n_batches]) in
    global batch order. Replicas (weak scaling): every rank holds n_batches of its own; returns [world * n_batches, ..].
    `device` / `classes`: where the buffers live and the outputs' column count - pass them whenever a rank may own NO batch
    (fewer batches than ranks): such a rank must still enter the collectives with a zero buffer of the common shape and
    placement (a rank that guessed cpu / 1 column would hang RCCL)."""
    dev = torch.device(device) if device is not None else (outs[0].device if outs else torch.device("cpu"))
    C = int(classes) if classes is not None else (outs[0].size(1) if outs else 1)
    assert all(o.size(1) == C for o in outs), "every batch output must have `classes` columns"
    if dist.is_initialized() and dev.type == "cuda" and dist.get_backend() == "gloo":
        g, nodes = gather_batch_outputs([o.cpu() for o in outs], n_batches, rank, world, replicas, torch.device("cpu"), C)
        return g.to(dev), nodes.to(dev)
    live = collectives_live(world)
    per = n_batches if replicas else (n_batches + world - 1) // world
    local_n = torch.zeros(per, dtype=torch.int64, device=dev)
    if outs:
        local_n[: len(outs)] = torch.tensor([o.size(0) for o in outs], dtype=torch.int64, device=dev)
    if not live:
        all_n = local_n.view(1, per)
    else:
        all_n = torch.empty((world, per), dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(all_n.view(-1), local_n)
    max_n = int(all_n.max().item()) if all_n.numel() else 0
    buf = torch.zeros((per, max_n, C), dtype=torch.float32, device=dev)
    for i, o in enumerate(outs):
        buf[i, : o.size(0)] = o
    if not live:
        allbuf = buf.view(1, per, max_n, C)
    else:
        allbuf = torch.empty((world, per, max_n, C), dtype=torch.float32, device=dev)
        dist.all_gather_into_tensor(allbuf.view(-1), buf.view(-1))
    if replicas:
        return allbuf.view(world * per, max_n, C), all_n.view(-1)
    order = [(owner_of(i, world), i // world) for i in range(n_batches)]
    return torch.stack([allbuf[r, j] for r, j in order]), torch.stack([all_n[r, j] for r, j in order])


def max_over_ranks(value: float, device) -> float:
    if not dist.is_initialized():
        return value
    if dist.get_backend() == "gloo":
        device = torch.device("cpu")
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    if dist.is_initialized():
        dist.barrier()
