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


def test_presets_cover_the_datasets_the_reference_scripts_run():
    """Every name of 0_7a_eval_QGTC_cluster_GCN.py:12-16,38-42 / README.md:84-89 has a synthetic stand-in with the script's own --dim,
    and the four of the published epoch table cut into 75 batches of a size the grouped kernels take (<= 8192 nodes)."""
    from qgtc_ppopp22_amd import graph as G

    dims = {"Proteins": 29, "artist": 100, "soc-BlogCatalog": 128, "ppi": 50, "ogbn-arxiv": 128, "ogbn-products": 100}
    for name, f in dims.items():
        assert G.PRESETS[name][2] == f
    for name in ("artist", "soc-BlogCatalog"):
        g = G.make_graph(name, 1500)
        assert g.n_nodes == G.PRESETS[name][0] and g.feat.shape == (g.n_nodes, dims[name]) and (g.src != g.dst).all()
        par = G.partition_list(g, 1500)
        sizes = [G.batch_nodes(par, c, 1500, 20).size for c in range(75)]
        assert sum(sizes) == g.n_nodes and max(sizes) <= 8192
    with pytest.raises(ValueError):
        G.make_graph("no-such-dataset")


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


def test_npz_loader_and_locality_partitioner(tmp_path):
    """The reference's .npz edge-list format (dataset.py:48-53: arrays src_li / dst_li, node count = largest id + 1)
    and the METIS-free partitioner standing in for partition_utils.py:11-18: on a planted-block graph whose node
    ids were shuffled, contiguous id ranges keep almost no edge inside a partition, the locality partitioner keeps
    a large share - so the cluster batches get their block structure back."""
    from qgtc_ppopp22_amd import graph as G

    g = G.make_sbm_graph("t", 6000, 100, 8.0, 16, seed=5)
    rng = np.random.default_rng(1)
    perm = rng.permutation(g.n_nodes)
    path = str(tmp_path / "g.npz")
    np.savez(path, src_li=perm[g.src], dst_li=perm[g.dst])
    loc = G.load_npz_graph(path, dim=12, psize=100)
    con = G.load_npz_graph(path, dim=12, psize=100, partitioner="contiguous")
    for h in (loc, con):
        assert h.n_nodes == int(max(perm[g.src].max(), perm[g.dst].max())) + 1
        assert h.feat.shape == (h.n_nodes, 12) and h.feat.dtype == np.float32
        np.testing.assert_array_equal(h.src, perm[g.src])
        np.testing.assert_array_equal(h.dst, perm[g.dst])
        assert h.block_of.shape == (h.n_nodes,) and h.block_of.min() >= 0 and h.block_of.max() < 100
        parts = G.partition_list(h, 100)
        assert sorted(np.concatenate(parts).tolist()) == list(range(h.n_nodes))     # a partition of the node set
    assert np.bincount(loc.block_of, minlength=100).max() <= int(np.ceil(1.05 * loc.n_nodes / 100))
    assert np.bincount(loc.block_of, minlength=100).min() >= loc.n_nodes // 200      # departures are capped: no part below half the mean
    assert G.edge_locality(con) < 0.05
    assert G.edge_locality(loc) > 0.4 and G.edge_locality(loc) > 10 * G.edge_locality(con)
    # deterministic
    np.testing.assert_array_equal(G.load_npz_graph(path, dim=12, psize=100).block_of, loc.block_of)
    # malformed files are rejected
    bad = str(tmp_path / "bad.npz")
    np.savez(bad, src_li=np.array([0, 1, 2]), dst_li=np.array([1, 2]))
    with pytest.raises(ValueError):
        G.load_npz_graph(bad, 4, 2)
    np.savez(bad, src_li=np.array([0, -1]), dst_li=np.array([1, 2]))
    with pytest.raises(ValueError):
        G.load_npz_graph(bad, 4, 2)


def test_locality_partitioner_never_empties_a_part():
    """A star-heavy graph where every node of many parts would rather be in the hubs' parts: arrivals are capped by
    the target's room and departures by the source's floor, so every part keeps at least half the mean size."""
    from qgtc_ppopp22_amd import graph as G

    rng = np.random.default_rng(7)
    n, psize = 3000, 60
    hubs = rng.integers(0, 50, size=12000)                 # all edges point into 50 hub nodes
    src = rng.integers(0, n, size=12000)
    part = G.locality_partition(src.astype(np.int64), hubs.astype(np.int64), n, psize)
    sizes = np.bincount(part, minlength=psize)
    assert sizes.min() >= n // (2 * psize) and sizes.max() <= int(np.ceil(1.05 * n / psize))
    assert sizes.sum() == n


def test_zerotile_row_is_what_parse_counter_computes():
    """driver.zerotile_row against a restatement of parse_counter.py:10-34 run over the log lines the extension
    prints (`counter_global: %d` / `counter: %d`, cumulative)."""
    from qgtc_ppopp22_amd import driver

    per_call_total, per_call_nz = [100, 80, 120], [10, 40, 0]
    cum_g = np.cumsum(per_call_total).tolist()
    cum_c = np.cumsum(per_call_nz).tolist()
    log = ["Namespace(batch_size=20, dataset='ppi', dim=10)"]
    for g_, c_ in zip(cum_g, cum_c):
        log += [f"counter_global: {g_}", f"counter: {c_}"]
    log += ["Namespace(batch_size=20, dataset='next', dim=10)"]
    # parse_counter.py's loop: a line with 'dataset' closes the previous block
    rows, gcs, cs, name = [], [], [], None
    for line in log:
        if "dataset" in line:
            if name is not None:
                rows.append((name, sum(gcs), sum(cs), sum(cs) / sum(gcs)))
            name = line.split(",")[1].split("=")[1].strip("'")
            gcs, cs = [], []
        if "counter_global:" in line:
            gcs.append(int(line.split(":")[1]))
            continue
        if "counter:" in line:
            cs.append(int(line.split(":")[1]))
    row = driver.zerotile_row("ppi", cum_g, cum_c)
    assert (row["dataset"], row["non_jumping"], row["jumping"]) == rows[0][:3]
    assert row["line"] == "{} , {} , {} , {:.3f}".format(*rows[0])
    assert row["per_epoch_non_jumping"] == 300 and row["per_epoch_jumping"] == 50
    assert driver.ZEROTILE_HEADER.split(" , ")[0] == "dataset" and "ratio" in driver.ZEROTILE_HEADER


def _run_bench(extra, env_extra=None, timeout=300):
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(root, "bench.py")] + extra, cwd=root, env=env,
                          capture_output=True, text=True, timeout=timeout)


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher in the environment must run TWO ranks (the parent spawns them
    before touching a GPU) and report n_gpus = 2 with one checksum per rank - here in dry mode over gloo."""
    import json

    out = _run_bench(["--gpus", "2", "--dry-run", "--backend", "gloo", "--steps", "3", "--warmup", "1"])
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                   # rank 0 only
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 3
    assert line["extras"]["rank_checksums"] == [1000.0, 1001.0]
    assert line["extras"]["batches_per_rank"] == [38, 37]    # 75 cluster batches round-robin
    assert line["max_wall_s"] == pytest.approx(0.002)        # the MAX over ranks


def test_bench_dry_run_world8_gathers_75_ragged_batches():
    """BASELINE.json configs[4] without hardware: eight gloo ranks, 75 ragged cluster batches round-robin (10 on ranks 0-2, 9 on
    the others), the per-batch float outputs gathered padded into global batch order on every rank - and the weak-scaled
    form (every rank its own batches, rank-major)."""
    import json

    out = _run_bench(["--gpus", "8", "--dry-run", "--backend", "gloo", "--steps", "2", "--warmup", "1"], timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    ex = line["extras"]
    assert line["n_gpus"] == 8 and ex["batches_per_rank"] == [10, 10, 10, 9, 9, 9, 9, 9]
    nodes = [1190 + (7 * i) % 50 for i in range(75)]
    assert ex["gathered_output_shape"] == [75, max(nodes), 10]
    assert ex["gathered_output_nodes"] == nodes
    assert ex["gathered_output_first_values"] == [float(i) for i in range(75)]          # global batch order
    assert ex["gathered_output_padding_is_zero"] is True
    assert ex["replica_output_shape"][0] == 24
    assert ex["replica_output_first_values"] == [float(100 * r + j) for r in range(8) for j in range(3)]   # rank-major
    assert ex["replica_summaries"] == [[float(r), float(j)] for r in range(8) for j in range(3)]
    assert ex["rank_checksums"] == [1000.0 + r for r in range(8)]
    # ONE batch over eight ranks - seven ranks own nothing and still enter the collectives with the common shape
    assert ex["one_batch_shape"] == [1, 1190, 10] and ex["one_batch_nodes"] == [1190] and line["ranks_seen"] == 8


def test_bench_fails_loudly_on_a_rank_count_mismatch():
    """A launcher that started a different number of ranks than --gpus says is an error, never a silent 1-GPU run."""
    out = _run_bench(["--gpus", "4", "--dry-run", "--backend", "gloo"],
                     env_extra={"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0"})
    assert out.returncode != 0 and "WORLD_SIZE=1" in out.stderr


def test_bench_parent_fails_when_a_rank_fails():
    out = _run_bench(["--gpus", "2", "--dry-run", "--backend", "gloo"], env_extra={"QGTC_BENCH_FAIL_RANK": "1"})
    assert out.returncode != 0


def _worst_case_line():
    """bench.py's line assembled from the same functions main() uses, with every optional block present and wide values."""
    import argparse

    import bench
    from benchmarks import epochs, headline

    args = argparse.Namespace(steps=100000, warmup=1000, engine="popcount", issue="eager", streams=1)
    rf = headline.roofline_block(4096, 4096, 64, 1, {"hip_events_eager": 3.623456e-6, "hip_events_graph_replay": 3.073456e-6}, 3.654321e-6, True)
    rf.update({"traffic": 123456789, "traffic_source": "profiles/r05/summary_headline.json (pmc, sources 0123456789ab)",
               "rocprof": {"file": "profiles/r05/kernel_stats_headline.csv", "avg_us": 3594.681, "min_us": 2640.123}})
    ep = {"batched_correct_chain_ms": 0.0226123, "per_batch_reference_chain_ms": 1.8511123,
          "roofline_of_the_grouped_correct_chain": {
              "kernel_us_per_epoch": 17.87123, "launches_per_epoch": 4, "algorithmic_bytes_per_epoch": 72270000123,
              "loader_us_per_iterator_hip_events": 145.123, "host_weight_pack_and_plan_bind_ms": 0.04123,
              "roofline": {"traffic": 43800000123, "frac": 0.50512, "frac_on_traffic": 0.30612}}}
    rf.update(epochs.flat_epoch_scalars("gcn", ep))
    rf.update(epochs.flat_epoch_scalars("gin", ep))
    line = bench.compose_line(args, 8, 8, 4096, 4096, 64, 1, 12345.678, 0.123456, True, rf)
    line["cpu_baseline"] = {"value": 0.9312, "unit": "TOPS", "cores": 256, "kind": "port",
                            "sample": "full 4096x4096x64 1-bit call, 3 x 400 reps (10 s), median; OpenMP C oracle, 32 of 256 threads",
                            "dgl_style_fp32_epoch_ms": 12345.67, "dgl_cores": 128,
                            "dgl_sample": "ogbn-arxiv-sized graph, 15 of 75 batches x5; torch-CPU GraphSAGE-sum x3"}
    line["parity_vs_oracle"] = True
    line["rccl_world1"] = {"ok": True, "backend": "nccl", "ranks_seen": 1, "gather_outputs_ms": 123.45}
    line["extras_file"] = "gpurun_out/bench_extras.json"
    return line


def test_bench_line_stays_small_and_machine_readable():
    """the driver could not parse a 25 KB line. The printed line carries the contract's fields + roofline (with both
    epochs as flat scalars) + cpu_baseline (with the DGL-style epoch inside) in under 4096 bytes, strictly JSON, no prose."""
    import json

    import bench

    line = bench.shrink(_worst_case_line())
    text = json.dumps(line)
    assert len(text) < bench.LINE_LIMIT, len(text)
    back = json.loads(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in back, k
    assert set(back["config"]) == {"workload", "inputs", "parallelism", "engine", "issue"} and back["config"]["issue"] == "eager"
    for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "algorithmic_bytes_per_launch",
              "avg_launch_us", "rocprof", "epoch_gcn_ms", "epoch_gcn_kernel_us", "epoch_gcn_frac", "epoch_gin_ms", "epoch_gin_kernel_us"):
        assert k in back["roofline"], k
    assert back["roofline"]["frac"] == pytest.approx(back["roofline"]["achieved"] / back["roofline"]["peak"], rel=1e-3)
    assert back["roofline"]["avg_launch_from"] == "hip_events_graph_replay" and back["roofline"]["avg_launch_us"] == pytest.approx(3.073, abs=1e-3)
    assert {"value", "unit", "cores", "kind", "sample", "dgl_style_fp32_epoch_ms", "dgl_cores"} <= set(back["cpu_baseline"])
    assert all(len(v) < 120 for v in back["config"].values())
    # a line that would not fit loses optional keys, never the contract's
    fat = _worst_case_line()
    fat["rccl_world1"]["pad"] = "x" * 5000
    slim = bench.shrink(fat)
    assert len(json.dumps(slim)) < bench.LINE_LIMIT and "rccl_world1" not in slim and "roofline" in slim and "cpu_baseline" in slim


def test_bench_default_issue_is_the_references_loop():
    """QGTC_device.cu:407-418 issues its launches eagerly; so does bench.py unless told otherwise (no probing, no best-of-two)."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = "import sys; sys.argv=['bench.py']; import bench; a = bench.parse(); print(a.issue, a.gpus, a.engine)"
    out = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=120)
    assert out.stdout.split() == ["eager", "1", "auto"], out.stderr[-500:]


def test_quiet_fd1_keeps_c_printf_off_stdout():
    """The counter operators print with C printf (qgtc_torch.cpp, as kernel.h:19,27 do); a pipe-buffered printf surfaces at exit,
    BEHIND the JSON line, unless file descriptor 1 itself is parked and C stdio flushed - what benchmarks.common.quiet_fd1 does."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import ctypes, sys; sys.path.insert(0, %r)\n"
            "from benchmarks.common import quiet_fd1\n"
            "libc = ctypes.CDLL(None)\n"
            "print('first')\n"
            "with quiet_fd1():\n"
            "    libc.printf(b'counter_global: 7\\n'); print('python inside')\n"
            "print('{\"last\": 1}')\n") % root
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-500:]
    assert out.stdout.splitlines() == ["first", '{"last": 1}']
