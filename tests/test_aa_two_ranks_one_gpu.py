"""The sharded epoch + end-of-epoch gather in TWO LIVE PROCESSES on the 1-GPU box (bench.py::epoch_leg's
`world > 1` branch). Both ranks bind to cuda:0 and exchange over gloo (RCCL refuses two ranks on one device; the
summaries go through host buffers, dist.gather_batch_summaries). The file sorts first and tests/conftest.py keeps it
first: the two children are started BEFORE this process makes any GPU call (a process that has initialised the GPU
must not be the one that execs others on this pool); the unsharded run it compares with happens afterwards."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _start_ranks(world, gin):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   QGTC_SHARD_TEST_GIN="1" if gin else "0")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "shard_worker.py")], cwd=ROOT, env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    return procs


@pytest.fixture(scope="module")
def two_rank_runs():
    """Starts 2 x 2 rank processes (GCN and GIN) while this process has not touched the GPU yet."""
    import torch

    if torch.cuda.device_count() < 1:          # (counting devices does not initialise the GPU)
        pytest.skip("no GPU visible")
    if torch.cuda.is_initialized():
        pytest.skip("this process has already initialised the GPU: run this file first (pytest's default order does)")
    out = {}
    for gin in (False, True):                   # one pair at a time: four processes would share the GPU for no reason
        procs = _start_ranks(2, gin)
        res = []
        for p in procs:
            try:
                so, se = p.communicate(timeout=600)
            except subprocess.TimeoutExpired:
                for q in procs:
                    q.kill()
                raise
            assert p.returncode == 0, se[-3000:]
            lines = [l for l in so.splitlines() if l.startswith("SHARD_RESULT ")]
            assert len(lines) == 1, so[-2000:]
            res.append(json.loads(lines[0][len("SHARD_RESULT "):]))
        out[gin] = res
    return out


@pytest.mark.parametrize("gin", [False, True])
def test_two_live_ranks_gather_the_unsharded_epoch(two_rank_runs, gin):
    """Every rank's gathered table (10 rows: sum and element count of each batch's float32 output, global batch order)
    equals the unsharded run's, computed afterwards in this process on the same GPU."""
    import torch

    import bench
    import QGTC as Q

    res = two_rank_runs[gin]
    assert sorted(r["rank"] for r in res) == [0, 1] and all(r["world"] == 2 for r in res)
    single, _ = bench.epoch_leg(Q, 0, 1, 0, dataset="tiny", bits=4 if gin else 2, hidden=64, gin=gin, full=False, psize=40, batch_size=4,
                                only=("batched_correct_chain",))
    assert "gathered_summaries" not in single
    from qgtc_ppopp22_amd import driver

    args = driver.build_parser().parse_args(["--dataset", "tiny", "--psize", "40", "--batch-size", "4", "--n-hidden", "64", "--n-classes", "10",
                                             "--bit_width", "4" if gin else "2", "--n-epochs", "1", "--use_QGTC", "--quiet", "--batched",
                                             "--chain", "correct"] + (["--run_GIN"] if gin else []))
    outs = driver.run(args, Q=Q)["outs"]
    want = bench.batch_summaries(outs).cpu().tolist()
    assert len(want) == 10 and all(w[0] > 0 for w in want)
    for r in res:
        assert r["res"]["gathered_batches"] == 10
        assert r["res"]["gathered_summaries"] == want, f"rank {r['rank']}"
        assert r["res"]["batched_correct_chain_ms"] > 0
    assert res[0]["res"]["batched_correct_chain_ms"] == res[1]["res"]["batched_correct_chain_ms"]   # the MAX over ranks
    # --gather outputs: the float outputs themselves travelled (padded to the largest batch): the same (sum, numel) table from them
    max_n = max(o.size(0) for o in outs)
    for r in res:
        ro = r["res_outputs"]
        assert ro["gathered_batches"] == 10 and ro["gathered_output_bytes"] == 10 * max_n * outs[0].size(1) * 4
        assert ro["gathered_summaries"] == want, f"rank {r['rank']} (outputs)"
    # weak scaling: every rank ran all ten batches of ITS OWN graph (seed 2 + rank); 20 rows, rank-major, rank 0's = the unsharded run's
    from qgtc_ppopp22_amd import graph as G
    outs1 = driver.run(args, Q=Q, graph=G.make_graph("tiny", 40, seed=3))["outs"]
    want1 = bench.batch_summaries(outs1).cpu().tolist()
    assert want1 != want
    for r in res:
        for key in ("res_weak", "res_weak_outputs"):
            rw = r[key]
            assert rw["gathered_batches"] == 20, key
            assert rw["gathered_summaries"][:10] == want and rw["gathered_summaries"][10:] == want1, f"rank {r['rank']} ({key})"
