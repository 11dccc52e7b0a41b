"""One kernel family per invocation, for rocprofv3 (tools/collect_profiles.sh):

    python3 tools/profile_targets.py headline|popcount|w8|gin_single|epoch|epoch_gin|loader|loader_gin|pack|wide1|wide8k|wide2|wide4|big|big16|big1024 [reps]

Runs the named workload `reps` times after a warm-up and prints one JSON line with the HIP-event time per launch
(events recorded on the launch stream), so that the trace's per-kernel averages can be put beside it."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import QGTC as Q


def events(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def gemm_target(M, K, N, w, reps, engine="auto", a=1):
    Q.set_engine(engine)
    g = torch.Generator(device="cuda").manual_seed(3)
    A = torch.randint(0, 2 ** a, (M, K), generator=g, device="cuda").float()     # (made on the device: 4 GiB of float32 at 32768 x 32768)
    X = torch.randint(0, 2 ** w, (K, N), generator=g, device="cuda").float()
    ba, bx = Q.val2bit(A, a, False, False), Q.val2bit(X, w, True, False)
    out = Q.bitMM2Bit(ba, bx, M, K, N, a, w, w)
    us = events(lambda: Q.bitMM2Bit_enqueue(out, ba, bx, M, K, N, a, w, w, 1), reps)
    algo = a * M * K / 8 + w * K * N / 8 + w * M * N / 8
    del A, X
    return {"workload": f"bitMM2Bit {M}x{K}x{N} a={a} w={w}", "us_per_launch_hip_events": round(us, 3), "launches": reps + 1,
            "algorithmic_bytes": int(algo), "eff_TOPS": round(2.0 * M * K * N / us / 1e6, 1),
            "hbm_frac_of_8TBs": round(algo / (us * 1e-6) / 8e12, 4), "fp4_frac_of_10PF": round(2.0 * M * K * N / (us * 1e-6) / 1e16, 4)}


def epoch_target(gin, reps):
    """The grouped, layout-correct epoch on its device-filled plan (driver.PlannedEpoch): what bench.py's epoch legs run."""
    from qgtc_ppopp22_amd import driver, graph as G
    dataset, b, hidden = ("ppi", 4, 64) if gin else ("ogbn-arxiv", 2, 128)
    graph = G.make_graph(dataset, 1500)
    dev = torch.device("cuda:0")
    args = driver.build_parser().parse_args(["--dataset", dataset, "--n-hidden", str(hidden), "--bit_width", str(b), "--use_QGTC", "--quiet", "--batched",
                                             "--chain", "correct"] + (["--run_GIN"] if gin else []))
    it = driver.make_iter(args, Q, graph)
    data = it.epoch_data(Q)
    W = driver.pack_weights(Q, graph.feat.shape[1], hidden, 10, b, dev)
    plan = driver.PlannedEpoch(Q, data, it.cluster_param_li, W, b, "correct", gin)
    us = events(plan.run, reps)
    launches = [round(events(lambda i=i: data.run_launch(i), reps), 2) for i in range(plan.n_launches)]
    return {"workload": f"{dataset}-sized epoch, 75 cluster batches, layout-correct chain, {plan.n_launches} grouped launches (device-filled plan)",
            "us_per_epoch_hip_events": round(us, 2), "us_per_launch_alone_hip_events": launches, "epochs": (1 + plan.n_launches) * (reps + 1)}


def loader_target(gin, reps):
    """The data loader's packing of the whole iterator (ClusterIter's default: one QGTC.EpochPlan.load call = qgtc_load_batches),
    re-run from the resident raw arrays; beside it the batch-by-batch route (QGTC.pack_edges + QGTC.val2bit x 2 per batch + the epoch
    formats) once."""
    import time
    from qgtc_ppopp22_amd import driver, graph as G
    dataset, b, hidden = ("ppi", 4, 64) if gin else ("ogbn-arxiv", 2, 128)
    graph = G.make_graph(dataset, 1500)
    args = driver.build_parser().parse_args(["--dataset", dataset, "--n-hidden", str(hidden), "--bit_width", str(b), "--use_QGTC", "--quiet", "--batched",
                                             "--chain", "correct"] + (["--run_GIN"] if gin else []))
    it = driver.make_iter(args, Q, graph)
    us = events(lambda: it.pack_now(Q), reps)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        it.pack_now(Q)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / reps * 1e6
    import random
    random.seed(2)
    from qgtc_ppopp22_amd.sampler import ClusterIter
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    old = ClusterIter(dataset, graph, 1500, 20, bit_width=b, run_GIN=gin, device="cuda", qgtc=Q, with_rows_X=True, grouped=False)
    old.epoch_data(Q)
    torch.cuda.synchronize()
    old_ms = (time.perf_counter() - t0) * 1e3
    edges, nodes = sum(it.n_edges), sum(p[0] for p in it.cluster_param_li)
    return {"workload": f"{dataset}-sized iterator: 75 cluster batches, {nodes} nodes, {edges} edges, {graph.feat.shape[1]} features, {b}-bit",
            "us_per_iterator_pack_hip_events": round(us, 1), "us_per_iterator_pack_wall_incl_host": round(wall, 1),
            "batch_by_batch_route_wall_ms_incl_host_graph_slicing": round(old_ms, 2), "packs": 2 * reps + 1}


def pack_target(reps):
    x = torch.rand((4096, 4096), device="cuda")
    us_r = events(lambda: Q.val2bit(x, 1, False, False), reps)
    us_c = events(lambda: Q.val2bit(x, 1, True, False), reps)
    algo = 4 * 4096 * 4096 + 4096 * 4096 / 8
    return {"workload": "val2bit 4096x4096 fp32 -> 1 bit", "us_rows_hip_events": round(us_r, 2), "us_cols_hip_events": round(us_c, 2),
            "algorithmic_bytes": int(algo), "GBs_rows": round(algo / us_r / 1e3, 1), "GBs_cols": round(algo / us_c / 1e3, 1)}


def main():
    t = sys.argv[1]
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    if t in ("loader", "loader_gin"):
        r = loader_target(t == "loader_gin", reps)
    elif t == "headline":
        r = gemm_target(4096, 4096, 64, 1, reps)
    elif t == "popcount":   # the AND + v_bcnt engine BASELINE.json's north star names, same workload
        r = gemm_target(4096, 4096, 64, 1, reps, engine="popcount")
        r["valu_peaks"] = {"pair_rate_measured_lane_instr_per_s": 4.2e13, "survey_8d_lane_instr_per_s": 7.864e13,
                           "note": "bit-ops = 2 M K N a w; one v_and_b32 + v_bcnt_u32_b32 pair = 32 bit-MACs = 64 bit-ops"}
        bitops = 2.0 * 4096 * 4096 * 64
        r["valu_frac_of_measured_pair_rate"] = round(bitops / (r["us_per_launch_hip_events"] * 1e-6) / (4.2e13 * 32), 4)
        r["valu_frac_of_survey_8d_peak"] = round(bitops / (r["us_per_launch_hip_events"] * 1e-6) / (7.864e13 * 32), 4)
    elif t == "w8":
        r = gemm_target(4096, 4096, 64, 8, reps)
    elif t == "gin_single":   # a per-batch 4 x 4-bit product of the Batched-GIN chain (main_qgtc.py:132)
        r = gemm_target(599, 50, 64, 4, reps, a=4)
    elif t == "big":       # the throughput-bound half of 5_9_adjmatrix_size.py: the adjacency is 128 MiB (k_bitmm_fp4_stream)
        r = gemm_target(32768, 32768, 64, 1, reps)
    elif t == "big16":
        r = gemm_target(32768, 32768, 16, 1, reps)
    elif t == "big1024":   # QGTC_module/logs/profile_new.log:26 (k_bitmm_fp4_wide)
        r = gemm_target(32768, 32768, 1024, 1, reps)
    elif t == "wide1":
        r = gemm_target(4096, 4096, 1024, 1, reps)
    elif t == "wide8k":
        r = gemm_target(8192, 4096, 1024, 1, reps)
    elif t == "wide2":
        r = gemm_target(8192, 4096, 1024, 2, reps)
    elif t == "wide4":
        r = gemm_target(4096, 4096, 1024, 4, reps)
    elif t == "epoch":
        r = epoch_target(False, reps)
    elif t == "epoch_gin":
        r = epoch_target(True, reps)
    elif t == "pack":
        r = pack_target(reps)
    else:
        raise SystemExit(f"unknown target {t}")
    r["target"] = t
    r["engine"] = Q.get_engine()
    print(json.dumps(r), flush=True)


if __name__ == "__main__":
    main()
