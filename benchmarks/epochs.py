"""Epoch legs: the second half of BASELINE.json's metric (Cluster-GCN / Batched-GIN epoch ms on synthetic graphs of the
datasets' sizes; main_qgtc.py:157-159's `Avg. Epoch`), their roofline blocks, zero-tile rows and README's epoch table."""
from __future__ import annotations

import json
import os
import subprocess
import sys
import time

import torch

from .common import FP4_PEAK_TFLOPS, HBM_PEAK_GBS, ROOT, hip_events_us, keep_clock_up, profile_summary, quiet_fd1


def batch_summaries(outs):
    """[n_local, 2] float64 on the outputs' device: (sum, element count) of every batch this rank ran."""
    return torch.stack([torch.stack([o.double().sum(), torch.tensor(float(o.numel()), device=o.device, dtype=torch.float64)])
                        for o in outs])


LEGS = [("per_batch_reference_chain", []),
        ("per_batch_nonresident_reference_chain", ["--non-resident"]),
        ("per_batch_graph_reference_chain", ["--graph"]),
        ("per_batch_2_streams_reference_chain", ["--streams", "2"]),
        ("per_batch_pack_on_the_fly_reference_chain", ["--pack-on-the-fly"]),          # cluster_gcn.py's structure
        ("batched_pack_on_the_fly_correct_chain", ["--batched", "--chain", "correct", "--pack-on-the-fly"]),
        ("batched_reference_chain", ["--batched"]),
        ("batched_correct_chain", ["--batched", "--chain", "correct"]),
        ("batched_correct_chain_engine_popcount", ["--batched", "--chain", "correct", "--engine", "popcount"])]
SHORT = ("per_batch_reference_chain", "batched_reference_chain", "batched_correct_chain", "batched_correct_chain_engine_popcount")


def epoch_leg(Q, rank, world, device_index, dataset="ogbn-arxiv", bits=2, hidden=128, gin=False, full=True, psize=1500,
              batch_size=20, only=None, weak=False, gather="summaries", classes=10):
    """Epoch time (BASELINE.json configs 2 / 3 / 4): a synthetic graph of the dataset's size, psize / batch_size batches.
    EVERY leg: the iterator built once ahead of the clock (main_qgtc.py:74-93), one untimed run, the clock kept up, then
    five runs of 20 epochs whose median `Avg. Epoch` is reported (min / max beside it).
    Sharding (world > 1): `weak` False = the batches round-robin over the ranks (strong scaling, BASELINE.json configs[4]);
    `weak` True = every rank runs all the batches of ITS OWN graph of that size (seed + rank): per-GPU work fixed. The one
    exchange of the path (RCCL over xGMI) follows the last leg, outside every epoch clock."""
    from qgtc_ppopp22_amd import dist as D, driver, graph as G

    base = ["--dataset", dataset, "--n-hidden", str(hidden), "--n-classes", str(classes), "--bit_width", str(bits),
            "--use_QGTC", "--gpu", str(device_index), "--quiet", "--n-epochs", "20", "--psize", str(psize),
            "--batch-size", str(batch_size)] + (["--run_GIN"] if gin else [])
    n_batches = psize // batch_size
    graph = G.make_graph(dataset, psize, seed=2 + (rank if weak else 0))
    ids = list(range(n_batches)) if weak else D.shard_round_robin(n_batches, rank, world)
    legs = [l for l in LEGS if (full or l[0] in SHORT) and (only is None or l[0] in only)]
    dev = torch.device("cuda", device_index)
    res, r = {}, None
    for name, extra in legs:
        args = driver.build_parser().parse_args(base + extra)
        it = driver.make_iter(args, Q, graph, ids)                              # ahead of the clock, once per leg
        r0 = driver.run(args, Q=Q, batch_ids=ids, graph=graph, it=it)          # untimed run (allocator, kernel attributes)
        if "plan" in r0:
            keep_clock_up(r0["plan"].run)
        D.barrier()
        ms = []
        for _ in range(5):
            r = driver.run(args, Q=Q, batch_ids=ids, graph=graph, it=it)
            ms.append(r["avg_epoch_ms"])
        res[name + "_ms_min_max_of_5"] = [round(min(ms), 4), round(max(ms), 4)]
        res[name + "_ms"] = round(D.max_over_ranks(sorted(ms)[2], dev), 4)
    if world > 1 and r is not None:
        total = n_batches * world if weak else n_batches
        if gather == "outputs":       # the per-batch float outputs themselves, padded to the largest batch
            t0 = time.perf_counter()
            allout, nodes = D.gather_batch_outputs(r["outs"], n_batches, rank, world, replicas=weak, device=dev, classes=classes)
            torch.cuda.synchronize()
            res["gather_outputs_ms"] = round((time.perf_counter() - t0) * 1e3, 3)
            res["gathered_batches"] = int(allout.size(0))
            res["gathered_output_bytes"] = int(allout.numel() * allout.element_size())
            res["gathered_summaries"] = [[float(allout[i, :int(nodes[i])].double().sum().item()), float(int(nodes[i]) * allout.size(2))]
                                         for i in range(allout.size(0))]
        else:
            allsum = D.gather_batch_summaries(batch_summaries(r["outs"]), total, rank, world) if not weak else \
                D.gather_replica_summaries(batch_summaries(r["outs"]), world)
            res["gathered_batches"] = int(allsum.size(0))
            res["gathered_summaries"] = allsum.cpu().tolist()       # (sum, numel) per batch in global batch order, on every rank
    return res, graph


def chain_ops(gin, F, H, C, b):
    """(K is the batch's node count?, K, N, a, w, output) per operator of the layout-correct chain (main_qgtc.py:131-154)."""
    if gin:
        return [(True, 0, F, 1, b, "bits"), (False, F, H, b, b, "bits"), (True, 0, H, 1, b, "bits"), (False, H, H, b, b, "bits"),
                (True, 0, H, 1, b, "bits"), (False, H, C, b, b, "f32")]
    return [(False, F, H, b, b, "bits"), (True, 0, H, 1, b, "bits"), (False, H, H, b, b, "bits"), (True, 0, H, 1, b, "bits"),
            (False, H, C, b, b, "bits"), (True, 0, C, 1, b, "f32")]


def algorithmic_work(params, gin, F, H, C, b):
    """SURVEY.md 8d summed over the six operators and every batch: (bytes = a M K / 8 + w K N / 8 + output, FP4 MFMA ops =
    2 M K N x base-4 digit pairs, effective ops = 2 M K N). Logical, unpadded shapes."""
    digits = lambda p: (p + 1) // 2       # noqa: E731
    algo = mfma = eff = 0.0
    for (n, _, _, _) in params:
        for (k_is_n, K, N, a, w, out) in chain_ops(gin, F, H, C, b):
            K = n if k_is_n else K
            algo += a * n * K / 8 + w * K * N / 8 + (4 * n * N if out == "f32" else b * n * N / 8)
            mfma += 2.0 * n * K * N * digits(a) * digits(w)
            eff += 2.0 * n * K * N
    return algo, mfma, eff


def epoch_traffic(gin):
    """HBM-side bytes of one epoch from the committed PMC passes of the same launches (2 x FETCH_SIZE + WRITE_SIZE, KiB per
    dispatch, times the dispatches an epoch makes of each kernel) - only when collected from THESE kernel sources."""
    summ, source = profile_summary("epoch_gin" if gin else "epoch")
    if summ is None:
        return None, source
    calls = {k["name"]: k["calls"] for k in summ.get("kernel_stats", []) if "k_rbw" in k["name"]}
    base = min(calls.values()) if calls else 0
    tot = 0.0
    for name, cs in summ.get("pmc_per_dispatch_mean", {}).items():
        if "k_rbw" not in name or not base:
            continue
        per_epoch = next((c for n_, c in calls.items() if n_[:60] == name[:60]), base) / base
        tot += per_epoch * (2.0 * cs.get("FETCH_SIZE", {}).get("mean", 0.0) + cs.get("WRITE_SIZE", {}).get("mean", 0.0)) * 1024.0
    return (int(tot) if tot > 0 else None), source


def epoch_roofline(Q, graph, device_index, dataset, bits, hidden, gin, classes=10):
    """The grouped, layout-correct epoch (default engine) against ITS floors: HIP events around the launches only - plan
    binding + weight packing (inside main_qgtc.py:96's clock) and the loader's one-off packing are timed separately.
    `frac` divides the DENSE algorithmic bytes, `frac_on_traffic` what the counters say was moved (zero-tile jumping skips
    most of A), both over the same kernel time."""
    from qgtc_ppopp22_amd import driver
    from qgtc_ppopp22_amd.sampler import ClusterIter

    dev = torch.device("cuda", device_index)
    it = ClusterIter(dataset, graph, 1500, 20, bit_width=bits, run_GIN=gin, device=dev, qgtc=Q, with_rows_X=True)
    data = it.epoch_data(Q)            # the data loader's share (one grouped pack of the iterator), ahead of the clock
    torch.cuda.synchronize()
    loader_us = hip_events_us(lambda: it.pack_now(Q), 10)
    t0 = time.perf_counter()
    for _ in range(10):
        it.pack_now(Q)
    torch.cuda.synchronize()
    loader_wall_us = (time.perf_counter() - t0) / 10 * 1e6
    # every operator on its own: a six-launch plan whose outputs are all in the public layouts
    F = graph.feat.shape[1]
    W = driver.pack_weights(Q, F, hidden, classes, bits, dev)
    six = driver.PlannedEpoch(Q, data, it.cluster_param_li, W, bits, "correct", gin, fuse=False)
    stage_us = [round(hip_events_us(lambda i=i: data.run_launch(i)), 2) for i in range(six.n_launches)]
    host = []
    for _ in range(5):                  # what main_qgtc.py:96 puts inside its epoch clock besides the launches
        t0 = time.perf_counter()
        W = driver.pack_weights(Q, F, hidden, classes, bits, dev)
        plan = driver.PlannedEpoch(Q, data, it.cluster_param_li, W, bits, "correct", gin)
        host.append((time.perf_counter() - t0) * 1e3)
        torch.cuda.synchronize()
    keep_clock_up(plan.run)
    epoch_us = sorted(hip_events_us(plan.run) for _ in range(5))[2]
    algo, mfma, eff = algorithmic_work(it.cluster_param_li, gin, F, hidden, classes, bits)
    traffic, traffic_source = epoch_traffic(gin)
    return {"kernel_us_per_epoch": round(epoch_us, 2), "kernel_us_per_operator_alone": stage_us, "launches_per_epoch": plan.n_launches,
            "host_weight_pack_and_plan_bind_ms": round(sorted(host)[2], 4),
            "loader_us_per_iterator_hip_events": round(loader_us, 1), "loader_us_per_iterator_wall": round(loader_wall_us, 1),
            "algorithmic_bytes_per_epoch": int(algo), "effective_ops_per_epoch": eff, "eff_TOPS": round(eff / epoch_us / 1e6, 1),
            "floors_us": {"hbm": round(algo / (HBM_PEAK_GBS * 1e9) * 1e6, 2), "mfma_fp4": round(mfma / (FP4_PEAK_TFLOPS * 1e12) * 1e6, 2),
                          "launch_gaps": round(1.5 * (plan.n_launches - 1), 1)},
            "adjacency_tiles_occupied": round(data.occupied_fraction, 4),
            "roofline": {"bound": "hbm", "achieved": round(algo / epoch_us / 1e3, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(algo / epoch_us / 1e3 / HBM_PEAK_GBS, 4), "frac_mfma": round(mfma / epoch_us / 1e6 / FP4_PEAK_TFLOPS, 4),
                         "traffic": traffic, "traffic_source": traffic_source,
                         "frac_on_traffic": round(traffic / epoch_us / 1e3 / HBM_PEAK_GBS, 4) if traffic else None}}


FLAT_KEYS = ("ms", "kernel_us", "launches", "algo_bytes", "traffic", "frac", "frac_on_traffic", "loader_ms", "bind_ms", "per_batch_ms")


def flat_epoch_scalars(tag, ep):
    """Both halves of BASELINE.json's metric as flat scalars for the line's `roofline`: epoch_<tag>_{ms (driver-style Avg.
    Epoch of the grouped plan), kernel_us (HIP events around the launches), launches, algo_bytes, traffic (counter bytes),
    frac, frac_on_traffic, loader_ms (the iterator's one-off packing, GPU time), bind_ms, per_batch_ms (the unchanged
    driver's loop)}."""
    rf = ep["roofline_of_the_grouped_correct_chain"]
    vals = (ep["batched_correct_chain_ms"], rf["kernel_us_per_epoch"], rf["launches_per_epoch"], rf["algorithmic_bytes_per_epoch"],
            rf["roofline"]["traffic"], rf["roofline"]["frac"], rf["roofline"]["frac_on_traffic"],
            round(rf["loader_us_per_iterator_hip_events"] / 1e3, 4), rf["host_weight_pack_and_plan_bind_ms"],
            ep.get("per_batch_reference_chain_ms"))
    return {f"epoch_{tag}_{k}": v for k, v in zip(FLAT_KEYS, vals)}


def zero_tile_rows(Q, graph_arxiv, device_index):
    """`--zerotile_jump` (main_qgtc.py:142-145) on both synthetic graphs: the row parse_counter.py:31-33 would print and the
    plain per-epoch ratio of 8-row x 128-bit tile steps that survive zero-tile jumping. The counter operators print with C
    printf as the reference does: file descriptor 1 is parked on /dev/null around them."""
    from qgtc_ppopp22_amd import driver, graph as G

    out = {}
    for dataset, bits, hidden, g in (("ogbn-arxiv", 2, 128, graph_arxiv), ("ppi", 4, 64, None)):
        args = driver.build_parser().parse_args(["--dataset", dataset, "--n-hidden", str(hidden), "--bit_width", str(bits), "--use_QGTC",
                                                 "--gpu", str(device_index), "--quiet", "--zerotile_jump"])
        Q.reset_counters()
        with quiet_fd1():
            r = driver.run(args, Q=Q, graph=g if g is not None else G.make_graph(dataset, 1500))
        z = r["zerotile"]
        out[dataset] = {"parse_counter_row": z["line"], "per_epoch_non_jumping": z["per_epoch_non_jumping"],
                        "per_epoch_jumping": z["per_epoch_jumping"], "per_epoch_ratio": round(z["per_epoch_ratio"], 4)}
    Q.reset_counters()
    return out


README_TABLE = (("artist", 12, 263.646), ("soc-BlogCatalog", 39, 209.495), ("ppi", 10, 189.016), ("ogbn-arxiv", 10, 208.616))


def readme_table(Q, rank, world, local):
    """The ONE epoch table the reference publishes (README.md:84-89): Cluster-GCN at the script's settings
    (hidden 16, psize 1500, batch 20, 2 bits, each dataset's --dim / --n-classes: 0_7a_eval_QGTC_cluster_GCN.py:6-16,38-40) on
    synthetic graphs of those datasets' sizes: [unchanged per-batch loop ms, grouped plan ms, reference sm_86 ms]."""
    table = {}
    for ds, cls, ref_ms in README_TABLE:
        e4, _ = epoch_leg(Q, rank, world, local, dataset=ds, bits=2, hidden=16, classes=cls, full=False,
                          only=("per_batch_reference_chain", "batched_correct_chain"))
        table[ds] = [e4["per_batch_reference_chain_ms"], e4["batched_correct_chain_ms"], ref_ms]
    return table


def checked_in_script_settings(Q, rank, world, local):
    """What the reference's CHECKED-IN epoch script runs (0_7a_eval_QGTC_cluster_GCN.py:6-10: hidden 16, bitwidth = 32) on the
    ogbn-arxiv-sized graph, per batch and grouped: 32 x 32-bit X.W products are 1024 plane pairs on the generic AND +
    popcount kernel (beyond every matrix-core form)."""
    e, _ = epoch_leg(Q, rank, world, local, dataset="ogbn-arxiv", bits=32, hidden=16, full=False,
                     only=("per_batch_reference_chain", "batched_reference_chain", "batched_correct_chain"))
    return e


def unchanged_driver_in_a_child(env_extra, timeout=240):
    """The reference's literal per-batch loop (main_qgtc.py:112-155: 75 batches x six extension calls, 20 epochs) in a FRESH
    child process with extra environment - for process-wide HIP runtime settings that cannot be flipped once this process
    has touched the GPU. Must be called BEFORE this process initialises the GPU. Median `Avg. Epoch` (ms) of five runs
    after an untimed one, or None."""
    code = ("import json, sys; sys.path.insert(0, %r)\n"
            "import torch, QGTC as Q\n"
            "from qgtc_ppopp22_amd import driver, graph as G\n"
            "args = driver.build_parser().parse_args(['--dataset', 'ogbn-arxiv', '--n-hidden', '128', '--n-classes', '10', '--bit_width', '2', "
            "'--use_QGTC', '--quiet', '--n-epochs', '20'])\n"
            "g = G.make_graph('ogbn-arxiv', 1500)\n"
            "it = driver.make_iter(args, Q, g)\n"
            "ms = [driver.run(args, Q=Q, graph=g, it=it)['avg_epoch_ms'] for _ in range(6)][1:]\n"
            "print('CHILD_MS ' + json.dumps(sorted(ms)))\n") % ROOT
    try:
        out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env_extra), capture_output=True, text=True, timeout=timeout)
        for ln in out.stdout.splitlines():
            if ln.startswith("CHILD_MS "):
                ms = json.loads(ln[len("CHILD_MS "):])
                return round(ms[len(ms) // 2], 4)
    except Exception:   # noqa: BLE001 - optional leg
        return None
    return None
