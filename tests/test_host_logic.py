"""CPU tests of the host-side logic: graph generator, partition/batch selection, sharding, the
end-of-epoch gather over gloo with world_size 2, driver flags."""
import os
import socket

import numpy as np
import pytest
import torch


def test_graph_is_deterministic_and_simple():
    from qgtc_ppopp22_amd import graph as G

    a, b = G.make_graph("tiny", 40), G.make_graph("tiny", 40)
    assert (a.src == b.src).all() and (a.dst == b.dst).all() and (a.feat == b.feat).all()
    assert (a.src != a.dst).all()
    key = a.src * a.n_nodes + a.dst
    assert np.unique(key).size == key.size
    inside = (a.block_of[a.src] == a.block_of[a.dst]).mean()
    assert 0.8 < inside < 0.99


def test_partitions_and_batches():
    from qgtc_ppopp22_amd import graph as G

    g = G.make_graph("tiny", 40)
    par = G.partition_list(g, 40)
    assert len(par) == 40 and sum(p.size for p in par) == g.n_nodes
    assert np.unique(np.concatenate(par)).size == g.n_nodes
    nodes = G.batch_nodes(par, 3, 40, 4)
    assert nodes.size == sum(par[s].size for s in range(12, 16))
    r, c = G.induced_edges(g, nodes)
    assert r.size and r.max() < nodes.size and c.max() < nodes.size
    # every induced edge is a real edge between batch nodes
    s = set(zip(g.src.tolist(), g.dst.tolist()))
    assert all((int(nodes[i]), int(nodes[j])) in s for i, j in zip(r[:200], c[:200]))


def test_shapes_helpers():
    from qgtc_ppopp22_amd import cols_shape, rows_shape

    assert rows_shape(1213, 1213, 1) == (1216, 40)
    assert cols_shape(1213, 128, 2) == (2 * 40, 128)
    assert cols_shape(128, 10, 2, True) == (2 * 4, 16)
    assert rows_shape(4096, 64, 1) == (4096, 4)


def test_round_robin_sharding():
    from qgtc_ppopp22_amd import dist as D

    for n, w in ((75, 8), (75, 1), (10, 3), (5, 8)):
        shards = [D.shard_round_robin(n, r, w) for r in range(w)]
        assert sorted(i for s in shards for i in s) == list(range(n))
        assert all(D.owner_of(i, w) == r for r, s in enumerate(shards) for i in s)
        assert max(len(s) for s in shards) - min(len(s) for s in shards) <= 1


def test_driver_flags_match_the_reference():
    from qgtc_ppopp22_amd import driver

    a = driver.build_parser().parse_args([])
    # main_qgtc.py:23-41 defaults
    assert (a.gpu, a.n_epochs, a.batch_size, a.psize, a.dim, a.n_hidden, a.n_classes, a.n_layers,
            a.bit_width) == (0, 20, 20, 1500, 10, 16, 10, 1, 2)
    assert not (a.use_pp or a.regular or a.run_GIN or a.use_QGTC or a.zerotile_jump)
    b = driver.build_parser().parse_args("--dataset ppi --use_QGTC --run_GIN --bit_width 4 --n-hidden 64".split())
    assert b.run_GIN and b.use_QGTC and b.bit_width == 4 and b.n_hidden == 64


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _gather_worker(rank, world, port, n_batches, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from qgtc_ppopp22_amd import dist as D

    r, w, _ = D.init_from_env(backend="gloo")
    ids = D.shard_round_robin(n_batches, r, w)
    local = torch.tensor([[float(i), float(i * i), float(r)] for i in ids], dtype=torch.float64).reshape(len(ids), 3)
    D.barrier()
    allrows = D.gather_batch_summaries(local, n_batches, r, w)
    tmax = D.max_over_ranks(10.0 + r, torch.device("cpu"))
    q.put((r, allrows.tolist(), tmax))
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("n_batches", [7, 8])
def test_gather_over_gloo_world2(n_batches):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gather_worker, args=(r, 2, port, n_batches, q)) for r in range(2)]
    [p.start() for p in procs]
    results = [q.get(timeout=120) for _ in procs]
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    expect = [[float(i), float(i * i), float(i % 2)] for i in range(n_batches)]
    for r, rows, tmax in results:
        assert rows == expect      # every rank sees every batch's row, in global batch order
        assert tmax == 11.0        # max over ranks
