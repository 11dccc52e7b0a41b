"""Whole cluster batches through the driver (per-batch and grouped launches, both chains, GCN and
GIN) against the oracle's restatement of the same six-operator chains."""
import random

import numpy as np
import pytest

from helpers import integer_gcn_reference, oracle_batch_inputs, oracle_chain, oracle_weights, to_np_u32

pytestmark = pytest.mark.gpu

PSIZE, BS = 40, 4   # 10 batches of ~200 nodes from the 2000-node 'tiny' graph


def _args(extra):
    from qgtc_ppopp22_amd import driver

    return driver.build_parser().parse_args(
        ["--dataset", "tiny", "--psize", str(PSIZE), "--batch-size", str(BS), "--n-hidden", "64",
         "--n-classes", "10", "--n-epochs", "2", "--use_QGTC", "--quiet"] + extra)


@pytest.mark.parametrize("chain", ["reference", "correct"])
@pytest.mark.parametrize("gin", [False, True])
@pytest.mark.parametrize("bits", [1, 2, 3, 4, 5, 8, 16, 32])   # (0_7a_eval_QGTC_cluster_GCN.py:10 is checked in with bitwidth = 32, 0_7b:10 with 2)
@pytest.mark.parametrize("batched", [False, True])
def test_epoch_outputs_match_oracle(qgtc, oracle, chain, gin, bits, batched):
    from qgtc_ppopp22_amd import driver, graph as G

    extra = ["--chain", chain, "--bit_width", str(bits)] + (["--run_GIN"] if gin else []) + (["--batched"] if batched else [])
    res = driver.run(_args(extra), Q=qgtc)
    graph = G.make_graph("tiny", PSIZE)
    random.seed(2)                     # the driver's seed: same partition shuffle
    par = G.partition_list(graph, PSIZE)
    random.shuffle(par)
    W = oracle_weights(oracle, graph.feat.shape[1], 64, 10, bits)
    assert len(res["outs"]) == PSIZE // BS
    for cid in range(PSIZE // BS):
        bi = oracle_batch_inputs(oracle, graph, par, cid, PSIZE, BS, bits)
        ct = res["iter"].cTensor_li[cid]
        np.testing.assert_array_equal(to_np_u32(ct.bit_A), bi["bit_A"])
        np.testing.assert_array_equal(to_np_u32(ct.bit_X), bi["bit_X"])
        expect = oracle_chain(oracle, bi, W, bits, chain, gin)[-1]
        np.testing.assert_array_equal(res["outs"][cid].cpu().numpy(), expect)
        if chain == "correct" and not gin and bits <= 8:
            np.testing.assert_array_equal(expect, integer_gcn_reference(bi["A"], bi["X"], 64, 10, bits, oracle))


@pytest.mark.parametrize("engine", ["mfma", "popcount"])
@pytest.mark.parametrize("gin", [False, True])
@pytest.mark.parametrize("batched", [False, True])
def test_engines_give_the_same_epoch(qgtc, engine, gin, batched):
    """--engine mfma / popcount (per-batch operators and grouped launches, reference and layout-correct
    chains) reproduce the default engine's outputs - which the test above pins to the oracle."""
    import torch
    from qgtc_ppopp22_amd import driver

    for chain in ("reference", "correct"):
        extra = ["--chain", chain] + (["--run_GIN", "--bit_width", "4"] if gin else []) + (["--batched"] if batched else [])
        base = driver.run(_args(extra), Q=qgtc)["outs"]
        other = driver.run(_args(extra + ["--engine", engine]), Q=qgtc)["outs"]
        assert qgtc.get_engine() == "auto"              # the driver restores the switch
        assert len(base) == len(other) and all(torch.equal(x, y) for x, y in zip(base, other))


@pytest.mark.parametrize("extra", [[], ["--chain", "correct"], ["--run_GIN", "--bit_width", "4"], ["--non-resident"]])
def test_pack_on_the_fly_matches_packed_once(qgtc, capfd, extra):
    """cluster_gcn.py's structure - pack inside the epoch loop - gives the outputs of the packed-once runs and
    prints the reference's `Trans (ms): .., Compute (ms): ..` line ahead of `Avg. Epoch`."""
    import re
    import torch
    from qgtc_ppopp22_amd import driver

    base = driver.run(_args(extra), Q=qgtc)["outs"]
    a = _args(extra + ["--pack-on-the-fly"])
    a.quiet = False
    res = driver.run(a, Q=qgtc)
    assert len(base) == len(res["outs"]) and all(torch.equal(x, y) for x, y in zip(base, res["outs"]))
    out = capfd.readouterr().out
    assert re.search(r"^Trans \(ms\): \d+\.\d{3}, Compute \(ms\): \d+\.\d{3}$", out, re.M)
    assert re.search(r"^Avg\. Epoch: \d+\.\d{3} ms$", out, re.M)
    assert res["compute_ms"] > 0.0


def test_non_resident_matches_resident(qgtc):
    import torch
    from qgtc_ppopp22_amd import driver

    a = driver.run(_args([]), Q=qgtc)
    b = driver.run(_args(["--non-resident"]), Q=qgtc)
    assert all(torch.equal(x, y) for x, y in zip(a["outs"], b["outs"]))
    assert b["iter"].cTensor_li[0].bit_A.device.type == "cpu"


def test_sharded_batches_cover_the_epoch(qgtc):
    """Round-robin shards of the batch list reproduce the unsharded results batch by batch."""
    import torch
    from qgtc_ppopp22_amd import dist as D, driver

    full = driver.run(_args(["--batched"]), Q=qgtc)["outs"]
    nb = PSIZE // BS
    for world in (2, 3):
        seen = {}
        for rank in range(world):
            ids = D.shard_round_robin(nb, rank, world)
            outs = driver.run(_args(["--batched"]), Q=qgtc, batch_ids=ids)["outs"]
            seen.update(dict(zip(ids, outs)))
        assert sorted(seen) == list(range(nb))
        assert all(torch.equal(seen[i], full[i]) for i in range(nb))


def test_zerotile_counters_mode(qgtc, capfd):
    from qgtc_ppopp22_amd import driver

    qgtc.reset_counters()
    res = driver.run(_args(["--zerotile_jump"]), Q=qgtc)
    out = capfd.readouterr().out
    assert out.count("counter_global:") == PSIZE // BS and out.count("counter:") == PSIZE // BS
    total, nz = res["counters"]
    assert 0 < nz < total


def test_avg_epoch_line_format(qgtc, capfd):
    import re
    from qgtc_ppopp22_amd import driver

    driver.main(["--dataset", "tiny", "--psize", str(PSIZE), "--batch-size", str(BS), "--n-epochs", "1", "--use_QGTC"])
    out = capfd.readouterr().out
    assert "dataset='tiny'" in out                      # parse_time.py:9-14 greps this
    assert re.search(r"Avg\. Epoch: \d+\.\d{3} ms", out)  # parse_time.py:15-17


def test_zerotile_row_from_the_driver_log(qgtc, capfd):
    """`--zerotile_jump` without --quiet: the counter lines plus the `dataset , non-jumping , jumping , ratio` row of
    parse_counter.py:31-33 computed from them."""
    from qgtc_ppopp22_amd import driver

    qgtc.reset_counters()
    args = driver.build_parser().parse_args(["--dataset", "tiny", "--psize", str(PSIZE), "--batch-size", str(BS),
                                             "--n-hidden", "64", "--use_QGTC", "--zerotile_jump"])
    res = driver.run(args, Q=qgtc)
    cap = capfd.readouterr()
    out, err = cap.out.splitlines(), cap.err.splitlines()
    gl = [int(l.split(":")[1]) for l in out if l.startswith("counter_global:")]
    cn = [int(l.split(":")[1]) for l in out if l.startswith("counter:")]
    assert len(gl) == len(cn) == PSIZE // BS
    row = res["zerotile"]
    assert driver.ZEROTILE_HEADER in err and row["line"] in err     # (the summary goes to stderr: stdout stays what parse_counter.py reads)
    assert not any("dataset" in l for l in out)
    assert row["non_jumping"] == sum(gl) and row["jumping"] == sum(cn)
    assert row["line"] == "tiny , {} , {} , {:.3f}".format(sum(gl), sum(cn), sum(cn) / sum(gl))
    assert (row["per_epoch_non_jumping"], row["per_epoch_jumping"]) == (gl[-1], cn[-1]) == tuple(res["counters"])
    assert 0.0 < row["per_epoch_ratio"] < 1.0


def _full_size(dataset, bits, hidden, gin, chain, psize=1500, bs=20):
    from qgtc_ppopp22_amd import driver, graph as G

    base = ["--dataset", dataset, "--psize", str(psize), "--batch-size", str(bs), "--n-hidden", str(hidden),
            "--n-classes", "10", "--bit_width", str(bits), "--n-epochs", "1", "--use_QGTC", "--quiet", "--chain", chain]
    if gin:
        base.append("--run_GIN")
    graph = G.make_graph(dataset, psize)
    random.seed(2)
    par = G.partition_list(graph, psize)
    random.shuffle(par)
    return driver, base, graph, par


@pytest.mark.parametrize("chain", ["correct", "reference"])
@pytest.mark.parametrize("dataset,bits,hidden,gin", [("ogbn-arxiv", 2, 128, False), ("ppi", 4, 64, True)])
def test_full_size_epoch_matches_oracle_on_every_batch(qgtc, oracle, dataset, bits, hidden, gin, chain):
    """BASELINE.json configs 2 / 3 at the size bench.py times: the ogbn-arxiv-sized graph (75 cluster batches of ~1213 nodes,
    F = H = 128, 2-bit, Cluster-GCN) and the ppi-sized one (75 x ~599, F = 50, H = 64, 4-bit, Batched-GIN), grouped launches
    on the default engine. ALL 75 batches against the oracle's chain - the layout-correct chain AND the reference's literal
    one (main_qgtc.py:147-154 / :131-138 as an unchanged driver issues them: mis-laid operands, over-reads included) - and
    against the per-batch launches (three batches were checked, the rest only HIP against HIP)."""
    import torch

    driver, base, graph, par = _full_size(dataset, bits, hidden, gin, chain)
    grouped = driver.run(driver.build_parser().parse_args(base + ["--batched"]), Q=qgtc, graph=graph)
    assert qgtc.get_engine() == "auto"
    per_batch = driver.run(driver.build_parser().parse_args(base), Q=qgtc, graph=graph)
    assert len(grouped["outs"]) == len(per_batch["outs"]) == 75
    W = oracle_weights(oracle, graph.feat.shape[1], hidden, 10, bits)
    sizes = []
    for cid in range(75):
        bi = oracle_batch_inputs(oracle, graph, par, cid, 1500, 20, bits)
        sizes.append(bi["n"])
        ct = grouped["iter"].cTensor_li[cid]
        np.testing.assert_array_equal(to_np_u32(ct.bit_A), bi["bit_A"], err_msg=f"A of batch {cid}")
        np.testing.assert_array_equal(to_np_u32(ct.bit_X), bi["bit_X"], err_msg=f"X of batch {cid}")
        expect = oracle_chain(oracle, bi, W, bits, chain, gin)[-1]
        np.testing.assert_array_equal(grouped["outs"][cid].cpu().numpy(), expect, err_msg=f"grouped, batch {cid}")
        np.testing.assert_array_equal(per_batch["outs"][cid].cpu().numpy(), expect, err_msg=f"per batch, batch {cid}")
        if chain == "correct":
            assert grouped["outs"][cid].abs().sum().item() > 0
    assert min(sizes) > (1100 if dataset == "ogbn-arxiv" else 500)


@pytest.mark.parametrize("dataset,nodes_per_batch", [("ogbn-products", 2500), ("Proteins", 500)])
def test_other_datasets_of_the_reference_scripts_at_their_sizes(qgtc, oracle, dataset, nodes_per_batch):
    """The other graphs the reference's scripts run (0_7a_eval_QGTC_cluster_GCN.py:12-16,41; 4_8_zero_tile_jumping.py:34) at
    their sizes: ogbn-products-sized cluster batches have n ~ 2.6 k nodes (K > 2048: more than 16 k-quads per row block - the
    loader's one-row-at-a-time path, longer occupancy words), Proteins-sized ones ~580 nodes and 29 features. Grouped,
    layout-correct 2-bit Cluster-GCN epoch, hidden 128: three batches against the oracle's chain, all of them against the
    per-batch launches; and the zero-tile counters of the first batch against the oracle's tile census."""
    import torch

    bits, hidden = 2, 128
    driver, base, graph, par = _full_size(dataset, bits, hidden, False, "correct")
    grouped = driver.run(driver.build_parser().parse_args(base + ["--batched"]), Q=qgtc, graph=graph)
    per_batch = driver.run(driver.build_parser().parse_args(base), Q=qgtc, graph=graph)
    assert len(grouped["outs"]) == 75 and all(torch.equal(x, y) for x, y in zip(grouped["outs"], per_batch["outs"]))
    W = oracle_weights(oracle, graph.feat.shape[1], hidden, 10, bits)
    for cid in (0, 37, 74):
        bi = oracle_batch_inputs(oracle, graph, par, cid, 1500, 20, bits)
        assert bi["n"] > nodes_per_batch
        ct = grouped["iter"].cTensor_li[cid]
        np.testing.assert_array_equal(to_np_u32(ct.bit_A), bi["bit_A"])
        np.testing.assert_array_equal(to_np_u32(ct.bit_X), bi["bit_X"])
        np.testing.assert_array_equal(grouped["outs"][cid].cpu().numpy(), oracle_chain(oracle, bi, W, bits, "correct", False)[-1], err_msg=f"batch {cid}")
    # `--zerotile_jump` (main_qgtc.py:142-145) on the first batch: printed counters = the oracle's census of 8-row x 128-bit tile steps
    bi = oracle_batch_inputs(oracle, graph, par, 0, 1500, 20, bits)
    n = bi["n"]
    total, nonzero = oracle.tile_counters(bi["bit_A"], n, n, hidden, 1, bits)
    ct = grouped["iter"].cTensor_li[0]
    Wd = driver.pack_weights(qgtc, graph.feat.shape[1], hidden, 10, bits, torch.device("cuda"))
    t0 = qgtc.bitMM2Bit(ct.bit_X, Wd["W1"], n, graph.feat.shape[1], hidden, bits, bits, bits)
    qgtc.reset_counters()
    qgtc.bitMM2Bit_base_cnt(ct.bit_A, t0, n, n, hidden, 1, bits, bits)
    qgtc.bitMM2Bit_zerojump_cnt(ct.bit_A, t0, n, n, hidden, 1, bits, bits)
    assert tuple(qgtc.get_counters()) == (total, nonzero) and 0 < nonzero < total
    qgtc.reset_counters()


@pytest.mark.parametrize("gin", [False, True, 2])
def test_layer_entry_routes_give_the_same_epoch(qgtc, gin):
    """--batched --chain correct: the default plan (aggregation stages carrying the next X.W), the layer-entry plan
    (--no-chain) and the six separate grouped launches (--no-fuse): identical outputs."""
    import torch
    from qgtc_ppopp22_amd import driver

    base = ["--batched", "--chain", "correct"] + (["--run_GIN", "--bit_width", "2" if gin == 2 else "4"] if gin else [])   # (the default
    # plan keeps T between its launches in the kernels' own formats: codes at 4 bits, k-quad-major planes at 2)
    ref = driver.run(_args(base + ["--no-fuse"]), Q=qgtc)["outs"]
    for extra in ([], ["--no-chain"]):
        got = driver.run(_args(base + extra), Q=qgtc)["outs"]
        assert len(got) == len(ref) and all(torch.equal(x, y) for x, y in zip(got, ref)), extra


@pytest.mark.parametrize("gin", [False, True])
def test_npz_graph_through_the_locality_partitioner_matches_the_oracle_chain(qgtc, oracle, tmp_path, gin):
    """SURVEY 8 f4 end to end on the GPU: an edge list in the reference's `.npz` format (dataset.py:48-53) with shuffled
    node ids -> graph.load_npz_graph -> graph.locality_partition (the METIS stand-in, partition_utils.py:11-18) ->
    ClusterIter -> the grouped, layout-correct HIP epoch; every batch's packed operands and float output against the
    oracle's chain on the same partition."""
    from qgtc_ppopp22_amd import driver, graph as G

    src_g = G.make_sbm_graph("t", 2400, 48, 7.0, 16, seed=11)
    perm = np.random.default_rng(3).permutation(src_g.n_nodes)
    path = str(tmp_path / "edges.npz")
    np.savez(path, src_li=perm[src_g.src], dst_li=perm[src_g.dst])
    psize, bs, bits, hidden = 48, 6, (4 if gin else 2), 64
    argv = ["--dataset", path, "--dim", "16", "--psize", str(psize), "--batch-size", str(bs), "--n-hidden", str(hidden),
            "--n-classes", "10", "--bit_width", str(bits), "--n-epochs", "1", "--use_QGTC", "--quiet", "--batched",
            "--chain", "correct"] + (["--run_GIN"] if gin else [])
    res = driver.run(driver.build_parser().parse_args(argv), Q=qgtc)
    graph = G.load_npz_graph(path, 16, psize)              # deterministic: the same partition the driver built
    assert G.edge_locality(graph) > 0.3                    # the partitioner found the planted blocks again
    random.seed(2)
    par = G.partition_list(graph, psize)
    random.shuffle(par)
    W = oracle_weights(oracle, 16, hidden, 10, bits)
    assert len(res["outs"]) == psize // bs
    for cid in range(psize // bs):
        bi = oracle_batch_inputs(oracle, graph, par, cid, psize, bs, bits)
        ct = res["iter"].cTensor_li[cid]
        np.testing.assert_array_equal(to_np_u32(ct.bit_A), bi["bit_A"])
        np.testing.assert_array_equal(to_np_u32(ct.bit_X), bi["bit_X"])
        expect = oracle_chain(oracle, bi, W, bits, "correct", gin)[-1]
        np.testing.assert_array_equal(res["outs"][cid].cpu().numpy(), expect, err_msg=f"batch {cid}")
        assert res["outs"][cid].abs().sum().item() > 0


@pytest.mark.parametrize("gin", [False, True])
@pytest.mark.parametrize("bits,hidden,classes", [(8, 64, 10), (5, 128, 40), (6, 33, 10), (7, 16, 10), (2, 256, 10), (4, 200, 130), (1, 160, 256), (3, 256, 256)])
def test_wider_chains_stay_on_the_chain_entries(qgtc, oracle, bits, hidden, classes, gin):
    """VERDICT r4 (missing 4): --bit_width 5 .. 8 and --n-hidden 129 .. 256 (main_qgtc.py:31,37) no longer fall back to six grouped
    GEMM launches: the grouped, layout-correct epoch runs on the chain entries (four launches for Cluster-GCN, three for Batched-GIN)
    and every batch equals the oracle's chain."""
    from qgtc_ppopp22_amd import driver, graph as G

    args = driver.build_parser().parse_args(
        ["--dataset", "tiny", "--psize", str(PSIZE), "--batch-size", str(BS), "--n-hidden", str(hidden), "--n-classes", str(classes), "--n-epochs", "2",
         "--use_QGTC", "--quiet", "--chain", "correct", "--bit_width", str(bits), "--batched"] + (["--run_GIN"] if gin else []))
    res = driver.run(args, Q=qgtc)
    assert res["plan"].n_launches == (3 if gin else 4)
    graph = G.make_graph("tiny", PSIZE)
    random.seed(2)
    par = G.partition_list(graph, PSIZE)
    random.shuffle(par)
    W = oracle_weights(oracle, graph.feat.shape[1], hidden, classes, bits)
    for cid in range(PSIZE // BS):
        bi = oracle_batch_inputs(oracle, graph, par, cid, PSIZE, BS, bits)
        np.testing.assert_array_equal(res["outs"][cid].cpu().numpy(), oracle_chain(oracle, bi, W, bits, "correct", gin)[-1], err_msg=f"batch {cid}")


def test_north_star_alias_names_run_the_same_operators(qgtc, oracle):
    """BASELINE.json's names for the operator surface (bit_qnt / mm_v1 / mm_v2) called on the device."""
    import torch

    rng = np.random.default_rng(5)
    n, F, b = 150, 40, 2
    A = (rng.random((n, n)) < 0.05).astype(np.float32)
    X = rng.uniform(-1, 5, size=(n, F)).astype(np.float32)
    bA = qgtc.bit_qnt(torch.from_numpy(A).cuda(), 1, False, False)
    bX = qgtc.bit_qnt(torch.from_numpy(X).cuda(), b, True, False)
    oA, oX = oracle.val2bit(A, 1), oracle.val2bit(X, b, True)
    np.testing.assert_array_equal(to_np_u32(bA), oA)
    np.testing.assert_array_equal(to_np_u32(qgtc.mm_v1(bA, bX, n, n, F, 1, b, b)), oracle.bitmm2bit(oA, oX, n, n, F, 1, b, b))
    np.testing.assert_array_equal(qgtc.mm_v2(bA, bX, n, n, F, 1, b, True).cpu().numpy(), oracle.bitmm2int(oA, oX, n, n, F, 1, b, True))


@pytest.mark.parametrize("bits", [2, 4])
@pytest.mark.parametrize("dim", [300, 1433])
def test_wide_feature_matrix_stays_on_the_chain_entries(qgtc, oracle, bits, dim):
    """More than 128 features (reddit has 602): the first X . W of the grouped Cluster-GCN plan loops over the k-quads of the feature
    matrix (qgtc_chain_transform, K <= 8192) instead of sending the whole epoch to the six-launch route - four launches, the oracle's outputs."""
    from qgtc_ppopp22_amd import driver, graph as G

    graph = G.make_graph("tiny", PSIZE, dim=dim)
    args = _args(["--chain", "correct", "--bit_width", str(bits), "--batched"])
    res = driver.run(args, Q=qgtc, graph=graph)
    assert res["plan"].n_launches == 4
    random.seed(2)
    par = G.partition_list(graph, PSIZE)
    random.shuffle(par)
    W = oracle_weights(oracle, dim, 64, 10, bits)
    for cid in range(PSIZE // BS):
        bi = oracle_batch_inputs(oracle, graph, par, cid, PSIZE, BS, bits)
        np.testing.assert_array_equal(res["outs"][cid].cpu().numpy(), oracle_chain(oracle, bi, W, bits, "correct", False)[-1])
