"""`cpu_baseline` of the line: reported baselines timed on the GPU box's host cores - never the thing shipped or measured as
`value`. The only place outside tests/ and smoke() that touches oracle/."""
from __future__ import annotations

import os
import time

import torch


def oracle_bitgemm(M, K, N, w, A, X, budget_s):
    """The C oracle (a port, not the reference: the reference has no CPU bit path) on the same workload. The thread count is
    the one that gives the best rate in a short probe (a 2 MB problem thrashes on 128 threads); the value is the MEDIAN of
    three timed blocks that share ~budget_s of CPU time. Returns (cpu_baseline block, the oracle's result words)."""
    from oracle.qgtc_oracle import Oracle

    try:
        O = Oracle(native=True, out_dir="/tmp")   # -march=native build made on this host
    except Exception:   # noqa: BLE001
        O = Oracle()
    bx = O.val2bit(A.numpy(), 1, False, False)
    bw = O.val2bit(X.numpy(), w, True, False)
    cores = os.cpu_count() or 1
    probe = {}
    for t in sorted({c for c in (4, 8, 16, 32, 64, cores) if c <= cores}):
        O.set_num_threads(t)
        O.bitmm2bit(bx, bw, M, K, N, 1, w, w)          # first call on this team: thread start-up
        t0 = time.perf_counter()
        for _ in range(3):
            O.bitmm2bit(bx, bw, M, K, N, 1, w, w)
        probe[t] = (time.perf_counter() - t0) / 3
    threads = min(probe, key=probe.get)
    O.set_num_threads(threads)
    reps = max(1, min(400, int(budget_s / 3 / max(probe[threads], 1e-6))))
    rates, total, ref = [], 0.0, None
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(reps):
            ref = O.bitmm2bit(bx, bw, M, K, N, 1, w, w)
        dt = time.perf_counter() - t0
        total += dt
        rates.append(2.0 * M * K * N * reps / dt / 1e12)
    return {"value": round(sorted(rates)[1], 4), "unit": "TOPS", "cores": threads, "kind": "port",
            "sample": f"full {M}x{K}x{N} {w}-bit call, 3 x {reps} reps ({total:.0f} s), median; OpenMP C oracle, {threads} of {cores} threads"}, ref


def dgl_style_epoch(graph, dataset, n_batches=15):
    """The reference's DGL fp32 baseline (cluster_gcn_dgl.py:97-105, modules.py:16-24: three GraphSAGE-sum layers per
    cluster batch) restated on torch-CPU (oracle/dgl_cpu_baseline.py; DGL is not installable): a bounded sample of
    `n_batches` of the 75 batches, scaled to an epoch. Returns {"ms", "cores", "sample"}."""
    from oracle.dgl_cpu_baseline import graphsage_cpu_epoch
    from qgtc_ppopp22_amd import graph as G

    par = G.partition_list(graph, 1500)
    graphsage_cpu_epoch(graph, par, 1500, 20, 128, 10, n_batches=2)
    secs, nb = graphsage_cpu_epoch(graph, par, 1500, 20, 128, 10, n_batches=n_batches)
    return {"ms": round(secs * 1e3 * 75 / nb, 2), "cores": torch.get_num_threads(),
            "sample": f"{dataset}-sized graph, {nb} of 75 batches x{75 / nb:.0f}; torch-CPU GraphSAGE-sum x3"}
