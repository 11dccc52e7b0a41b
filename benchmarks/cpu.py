"""`cpu_baseline` of the line: reported baselines timed on the GPU box's host cores - never the thing shipped or measured as
`value`. The only place outside tests/ and smoke() that touches oracle/."""
from __future__ import annotations

import os
import time

import torch


def physical_cores_of_one_socket():
    """The fixed thread policy of `cpu_baseline` (a probe-chosen count made the figure mean something else on
    every box - 0.30 TOPS on 64 of 256 threads, 0.87 on 32): the PHYSICAL cores of socket 0 as /proc/cpuinfo lists them, one thread
    each; hosts that do not say (containers without topology) count half of os.cpu_count() when SMT siblings show, else all of it."""
    cores, phys, core = set(), None, None
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("physical id"):
                phys = int(ln.split(":")[1])
            elif ln.startswith("core id"):
                core = int(ln.split(":")[1])
            elif not ln.strip():
                if phys == 0 and core is not None:
                    cores.add(core)
                phys = core = None
        if phys == 0 and core is not None:
            cores.add(core)
    except (OSError, ValueError):
        cores = set()
    n = os.cpu_count() or 1
    return max(1, min(len(cores), n)) if cores else n


def oracle_bitgemm(M, K, N, w, A, X, budget_s):
    """The C oracle (a port, not the reference: the reference has no CPU bit path) on the same workload, on ONE thread per physical
    core of socket 0 (physical_cores_of_one_socket: the same meaning on every box); the value is the MEDIAN of three timed blocks that
    share ~budget_s of CPU time. Returns (cpu_baseline block, the oracle's result words)."""
    from oracle.qgtc_oracle import Oracle

    try:
        O = Oracle(native=True, out_dir="/tmp")   # -march=native build made on this host
    except Exception:   # noqa: BLE001
        O = Oracle()
    bx = O.val2bit(A.numpy(), 1, False, False)
    bw = O.val2bit(X.numpy(), w, True, False)
    cores = os.cpu_count() or 1
    threads = physical_cores_of_one_socket()
    O.set_num_threads(threads)
    O.bitmm2bit(bx, bw, M, K, N, 1, w, w)          # first call on this team: thread start-up
    t0 = time.perf_counter()
    for _ in range(3):
        O.bitmm2bit(bx, bw, M, K, N, 1, w, w)
    one = (time.perf_counter() - t0) / 3
    reps = max(1, min(400, int(budget_s / 3 / max(one, 1e-6))))
    rates, total, ref = [], 0.0, None
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(reps):
            ref = O.bitmm2bit(bx, bw, M, K, N, 1, w, w)
        dt = time.perf_counter() - t0
        total += dt
        rates.append(2.0 * M * K * N * reps / dt / 1e12)
    return {"value": round(sorted(rates)[1], 4), "unit": "TOPS", "cores": threads, "kind": "port",
            "sample": f"full {M}x{K}x{N} {w}-bit call, 3 x {reps} reps ({total:.0f} s), median; OpenMP C oracle, one thread per physical "
                      f"core of socket 0 ({threads} of {cores} hardware threads)"}, ref


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
