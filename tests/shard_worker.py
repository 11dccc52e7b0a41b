"""One rank of the two-process sharded-epoch test (tests/test_aa_two_ranks_one_gpu.py): a fresh process that joins a
gloo group from RANK / WORLD_SIZE / MASTER_*, binds to cuda:0 and runs bench.py's epoch leg on its round-robin shard of
the 'tiny' graph's batches - the sharded branch + the end-of-epoch gather with real device tensors - and prints what it
gathered as one JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch

    import bench
    from qgtc_ppopp22_amd import dist as D

    rank, world, _ = D.init_from_env(backend="gloo")
    torch.cuda.set_device(0)
    import QGTC as Q

    gin = os.environ.get("QGTC_SHARD_TEST_GIN") == "1"
    kw = dict(dataset="tiny", bits=4 if gin else 2, hidden=64, gin=gin, full=False, psize=40, batch_size=4, only=("batched_correct_chain",))
    res, _ = bench.epoch_leg(Q, rank, world, 0, **kw)
    # the same exchange with the REAL payload (--gather outputs: the per-batch float outputs, padded), and the weak-scaled leg
    # (every rank all ten batches of its own graph, seed 2 + rank)
    res_out, _ = bench.epoch_leg(Q, rank, world, 0, gather="outputs", **kw)
    res_weak, _ = bench.epoch_leg(Q, rank, world, 0, weak=True, **kw)
    res_weak_out, _ = bench.epoch_leg(Q, rank, world, 0, weak=True, gather="outputs", **kw)
    print("SHARD_RESULT " + json.dumps({"rank": rank, "world": world, "res": res, "res_outputs": res_out, "res_weak": res_weak,
                                        "res_weak_outputs": res_weak_out}), flush=True)
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
