"""One kernel family per invocation, for rocprofv3 (tools/collect_profiles.sh):

    python3 tools/profile_targets.py headline|wide1|wide8k|wide2|wide4|epoch|epoch_gin|pack [reps]

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


def gemm_target(M, K, N, w, reps):
    g = torch.Generator(device="cpu").manual_seed(3)
    A = (torch.rand((M, K), generator=g) < 0.5).float().cuda()
    X = torch.randint(0, 2 ** w, (K, N), generator=g).float().cuda()
    ba, bx = Q.val2bit(A, 1, False, False), Q.val2bit(X, w, True, False)
    out = Q.bitMM2Bit(ba, bx, M, K, N, 1, w, w)
    us = events(lambda: Q.bitMM2Bit_enqueue(out, ba, bx, M, K, N, 1, w, w, 1), reps)
    algo = M * K / 8 + w * K * N / 8 + w * M * N / 8
    return {"workload": f"bitMM2Bit {M}x{K}x{N} a=1 w={w}", "us_per_launch_hip_events": round(us, 3), "launches": reps + 1,
            "algorithmic_bytes": int(algo), "eff_TOPS": round(2.0 * M * K * N / us / 1e6, 1)}


def epoch_target(gin, reps):
    from qgtc_ppopp22_amd import driver, graph as G
    from qgtc_ppopp22_amd.sampler import ClusterIter
    dataset, b, hidden = ("ppi", 4, 64) if gin else ("ogbn-arxiv", 2, 128)
    graph = G.make_graph(dataset, 1500)
    dev = torch.device("cuda:0")
    it = ClusterIter(dataset, graph, 1500, 20, bit_width=b, run_GIN=gin, device=dev, qgtc=Q, with_rows_X=True)
    W = driver.pack_weights(Q, graph.feat.shape[1], hidden, 10, b, dev)
    plan = driver.BatchedEpoch(Q, it.cTensor_li, it.cluster_param_li, W, b, "correct", gin)
    us = events(plan.run, reps)
    stages = [round(events(g.run, reps), 2) for g in plan.stages]
    return {"workload": f"{dataset}-sized epoch, 75 cluster batches, layout-correct chain, grouped launches",
            "us_per_epoch_hip_events": round(us, 2), "us_per_stage_hip_events": stages, "epochs": 7 * (reps + 1)}


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
    if t == "headline":
        r = gemm_target(4096, 4096, 64, 1, reps)
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
