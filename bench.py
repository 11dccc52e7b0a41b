#!/usr/bin/env python3
"""bench.py — headline benchmark of the QGTC bit-GEMM hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is ONE bitMM2Bit launch over the whole workload named in BASELINE.json's north star and
BASELINE.md §1's bold row: a 1-bit 4096x4096 adjacency times a 1-bit 4096x64 feature matrix
(M=K=4096, N=64, output 1 bit), operands already bit-packed and resident in HBM. The metric is the
reference's own (QGTC_device.cu:420-422): effective tera-ops = 2*M*K*N per launch / time. Inputs
are seeded random bits (the reference's all-ones inputs are reported beside it in `extras`).

With N>1 ranks every GPU runs its own copy of the workload (cluster batches are independent, so
the path shards with no data-path collective): weak scaling, value = all ranks' ops / max time;
RCCL is used only to take the max time and to gather per-rank result checksums.

Rank 0 prints one JSON line (contract fields + `roofline`, `cpu_baseline`, `extras`).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

REF_TFLOPS_4096_64 = {1: 46.768, 2: 26.818, 4: 14.196, 8: 7.324}
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8 TB/s spec
# VALU issue rate of the v_and_b32 + v_bcnt_u32_b32 pair measured on MI355X with
# tools/valu_peak.hip (4.2e13 lane-instr/s at 8 waves/SIMD); one pair = 32 bit-MACs = 64 bit-ops.
VALU_PEAK_BITOPS = 4.2e13 * 32
# SURVEY.md 8(d)'s definition of the same roofline: 256 CUs x 4 SIMDs x 32 lane-instr/clk x 2.4 GHz = 7.864e13 lane-instr/s
# (an issue rate the v_and + v_bcnt pair does not reach: v_bcnt is VOP3, ~4.2 cycles per wave64 instruction). Both are printed.
VALU_PEAK_BITOPS_SURVEY = 7.864e13 * 32
FP4_PEAK_TFLOPS = 10000.0        # MI355X_MICROARCH.md: ~10 PF dense FP4 / FP6 MFMA


def eff_ops_of(M, K, N):
    return 2.0 * M * K * N


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=1000)
    p.add_argument("--warmup", type=int, default=50)
    p.add_argument("--bits", type=int, default=1, help="feature bit width w of the headline workload")
    p.add_argument("--no-extras", action="store_true", help="skip the width sweep / epoch / CPU legs")
    p.add_argument("--streams", type=int, default=1,
                   help="issue the steps round-robin on this many HIP streams (independent launches overlap); "
                        "default 1 = the reference's metric, launches back to back on one stream")
    p.add_argument("--issue", choices=["auto", "eager", "graph"], default="auto",
                   help="how the K steps of the timed region are issued: eager (the reference's way) = K hipLaunchKernel calls inside "
                        "the region; graph = the K launches captured ONCE, ahead of the timed region, in a hipGraph and replayed inside it by one "
                        "hipGraphLaunch (same kernels in stream order). Which is faster is the HOST's property: at K = 20 eager takes 3.6 us per "
                        "step on a box whose hipLaunchKernel costs 3.1-3.6 us (it overlaps the kernel's 3.0 us) and 4.6 on one where it costs "
                        "more; the replay takes 3.97-4.03 on both. auto (default): five untimed eager runs ahead of the timed region; only if "
                        "their median is above what a replay is known to take are the graphs built and both ways compared (five runs each); "
                        "the other way's figure is reported in extras")
    p.add_argument("--cpu-seconds", type=float, default=10.0, help="CPU-baseline budget")
    p.add_argument("--backend", choices=["nccl", "gloo"], default=None, help="torch.distributed backend (default: nccl = RCCL)")
    p.add_argument("--dry-run", action="store_true", help="exercise the rank launcher and the collectives only (no GPU)")
    p.add_argument("--gather", choices=["summaries", "outputs"], default="summaries",
                   help="what the end-of-epoch exchange of the sharded epoch legs moves over RCCL: per-batch (sum, numel) pairs, "
                        "or the per-batch float32 outputs themselves padded to the largest batch (SURVEY.md 8e)")
    p.add_argument("--engine", choices=["popcount", "mfma", "auto"], default="auto",
                   help="engine of the headline launches (same words either way): popcount = AND + v_bcnt kernels, "
                        "auto = the library's choice per call (at this shape the FP4 matrix-core kernel for narrow "
                        "right operands); the other engine's figure is reported in extras")
    return p.parse_args()


import contextlib


@contextlib.contextmanager
def engine(Q, name):
    """Run a block on one engine of the library ("auto" is the shipped default) and put the previous one back."""
    prev = Q.get_engine()
    Q.set_engine(name)
    try:
        yield
    finally:
        Q.set_engine(prev)


def median_of_5(Q, ba, bx, M, K, N, w, reps=200):
    """The reference's measurement (QGTC_device.cu:403-422): `reps` launches between two events; the median of five such
    windows after an untimed one (200 reps per point, median of >= 5 runs)."""
    Q.profile(ba, bx, M, K, N, 1, w, w, max(reps // 10, 1))
    return sorted(Q.profile(ba, bx, M, K, N, 1, w, w, reps) for _ in range(5))[2]


def make_workload(Q, M, K, N, w, device, seed, ones=False):
    g = torch.Generator(device="cpu").manual_seed(seed)
    if ones:      # 2_7c_QGTC_GEMM_INT8.py:6-12
        A = torch.ones((M, K))
        X = torch.ones((K, N))
    else:
        A = (torch.rand((M, K), generator=g) < 0.5).float()
        X = torch.randint(0, 2 ** w, (K, N), generator=g).float()
    bit_A = Q.val2bit(A.to(device), 1, False, False)
    bit_X = Q.val2bit(X.to(device), w, True, False)
    return A, X, bit_A, bit_X


CLOCK_WARMUP_S = 0.3
EVENT_MIN_LAUNCHES = 200     # launches in the HIP-event window behind the timed region (QGTC_device.cu:409 times 200 too)


def replay_worth_probing(eager_us_per_step, steps):
    """--issue auto: graphs are only built (and both ways compared) when the eager issue of the K steps is slower than a replay is known to
    be on this chip - 3.0 us per kernel plus ~20 us per hipGraphLaunch (3.97-4.05 us per step at K = 20 on every box measured)."""
    return eager_us_per_step > 3.05 + 20.0 / steps


def time_steps(Q, out, bit_A, bit_X, M, K, N, w, steps, warmup, barrier, streams=1, issue="eager"):
    """warmup untimed launches, then EXACTLY `steps` launches between barrier+synchronize pairs (the contract's timed
    region: wall seconds), then the SAME `steps` launches once more between two HIP events recorded on the stream the
    kernels are launched on, and max(3, K // 200) event-bracketed windows of 200 launches (the roofline's live launch duration: their
    MEDIAN). Returns (wall seconds, mean stream time per launch in seconds, wall seconds of the event-bracketed K-launch region, the
    issue mode used, the untimed probe of an "auto" choice or None).
    Why two regions: recording the two events INSIDE the timed region costs 11-12 us of its wall clock whatever the host's
    wait policy (tools/steps20c.py: 84 us with them, 72 us without, for 20 launches that take 63 us on the stream) - the
    instrument would be a seventh of the measurement. The second region is issued right behind the first, same buffers,
    same stream; its own wall clock is reported beside `value` (extras) so that the cost of the instrument stays visible."""
    if streams > 1:
        outs = [out] + [torch.empty_like(out) for _ in range(streams - 1)]
        enqueue = lambda n: Q.bitMM2Bit_enqueue_streams(outs, bit_A, bit_X, M, K, N, 1, w, w, n)  # noqa: E731
    else:
        enqueue = lambda n: Q.bitMM2Bit_enqueue(out, bit_A, bit_X, M, K, N, 1, w, w, n)  # noqa: E731
    # The chip drops into a low power state within milliseconds of idling (the host has just spent seconds building the
    # inputs / the CPU baseline): the same 20 launches take 85 us warm and 200-450 us after 50-500 ms of idle
    # (tools/steps20b.py). CLOCK_WARMUP_S of untimed launches first, then the W warmup steps of the contract.
    t_w = time.perf_counter()
    while time.perf_counter() - t_w < CLOCK_WARMUP_S:
        enqueue(200)       # (bursts of the timed region's order, each drained: seconds-long back-to-back queues are another regime)
        torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    ev1.record()           # (created here: the first record of a torch event allocates it)
    # issue = "graph": the K launches of the timed region (and the 200 of an event window) are captured ONCE, here, ahead of everything
    # timed, and replayed by one hipGraphLaunch each. Same kernels, same stream order (a captured single-stream sequence is a chain of
    # dependent kernel nodes), same outputs - what changes is the host's share: 20 eager launches cost the host 63-84 us
    # (tools/host_launch_probe.hip: 3.1-3.5 us per hipLaunchKernel), one graph launch 8 us. For THIS kernel it buys nothing: its 3.0 us
    # per launch on the GPU and the host's issue time overlap when issued eagerly (72 us for 20 steps), the replay takes 79 us.
    run_steps, run_window = (lambda: enqueue(steps)), (lambda: enqueue(EVENT_MIN_LAUNCHES))
    probe = None
    if streams > 1:
        issue = "eager"

    def trials(fn, n=5):   # untimed: median wall clock of n runs of the K steps
        ts = []
        for _ in range(n):
            torch.cuda.synchronize()
            tp = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - tp)
        return sorted(ts)[n // 2]

    if issue == "auto":
        # The host decides. Eager first, in the state a plain eager run has (r04: on one box the eager launches read 6.5 us per step once
        # the graphs below existed and 3.6 in a run that never captured one): only a host whose eager issue is slower than a replay is
        # known to be (3.0 us per kernel + ~20 us per graph launch) gets the graphs built and both ways compared in that state.
        eager_us = trials(run_steps) * 1e6 / steps
        probe = {"eager": round(eager_us, 3)}
        if not replay_worth_probing(eager_us, steps):
            issue = "eager"
    if issue in ("graph", "auto"):
        graphs = []
        for n in (steps, EVENT_MIN_LAUNCHES):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                enqueue(n)
            g.replay()             # (the first replay uploads the graph: untimed)
            graphs.append(g)
        torch.cuda.synchronize()
        if issue == "auto":
            probe["graph"] = round(trials(graphs[0].replay) * 1e6 / steps, 3)
            probe["eager_with_the_graphs_alive"] = round(trials(run_steps) * 1e6 / steps, 3)
            issue = "graph" if probe["graph"] < probe["eager_with_the_graphs_alive"] else "eager"
        if issue == "graph":
            run_steps, run_window = graphs[0].replay, graphs[1].replay
    enqueue(max(warmup, 1))
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_steps()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    barrier()
    # the same region again, bracketed by HIP events on the launch stream
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    ev0.record()
    run_steps()
    ev1.record()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    # The dominant kernel's AVERAGE launch duration: windows of EVENT_MIN_LAUNCHES launches, each between its own pair of events and
    # drained before the next - max(1, K // EVENT_MIN_LAUNCHES) of them, averaged. A window of a few launches is mostly its own
    # start-up (the first launch's latency, the event markers); ONE window of a thousand launches and more between two timing events
    # is an instrument artefact of the other kind (r04: 3.5 / 4.9 / 3.7 us per launch on three runs while the event-free region
    # beside it ran at 3.0; tools/event_windows.py: a box whose chip has idled for 10 ms can stay
    # 20-30 % slower on event-bracketed work for seconds while an event-free region beside it runs at full speed).
    windows = max(3, steps // EVENT_MIN_LAUNCHES)   # (one window alone read 3.6-4.1 us on two of six boxes while the event-free region beside it ran at 3.0-3.1)
    per_window = []
    for _ in range(windows):
        ev0.record()
        run_window()
        ev1.record()
        torch.cuda.synchronize()
        per_window.append(ev0.elapsed_time(ev1) * 1e-3 / EVENT_MIN_LAUNCHES)
    per_launch = sorted(per_window)[len(per_window) // 2]
    return t1 - t0, per_launch, t3 - t2, issue, probe


def cpu_baseline(M, K, N, w, A, X, budget_s):
    """The C oracle (a port, not the reference: the reference has no CPU bit path) on the host cores, same workload. The
    thread count is the one that gives the best rate in a short probe (a 2 MB problem thrashes on 128 threads: r03's figure
    swung 2.7x between runs); the value is the MEDIAN of three timed blocks that share ~budget_s of CPU time."""
    from oracle.qgtc_oracle import Oracle

    try:
        O = Oracle(native=True, out_dir="/tmp")   # -march=native build made on this host
    except Exception:
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
    one = probe[threads]
    reps = max(1, min(400, int(budget_s / 3 / max(one, 1e-6))))
    rates, total = [], 0.0
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(reps):
            ref = O.bitmm2bit(bx, bw, M, K, N, 1, w, w)
        dt = time.perf_counter() - t0
        total += dt
        rates.append(2.0 * M * K * N * reps / dt / 1e12)
    return {"value": round(sorted(rates)[1], 4), "unit": "effective TOPS", "cores": threads, "kind": "port",
            "sample": f"full {M}x{K}x{N} {w}-bit workload, 3 blocks x {reps} reps ({total:.1f} s), median; OpenMP C oracle on {threads} of {cores} host "
                      f"threads (best of a probe over {sorted(probe)})",
            "blocks_TOPS": [round(r, 4) for r in rates]}, ref


PROFILE_DIR = "profiles/r04"


def profile_summary(name):
    """The committed rocprofv3 summary of a target (tools/collect_profiles.sh) - only when it was collected from THESE kernel
    sources: the summary records the hash of csrc/ + include/qgtc.h it ran (qgtc_ppopp22_amd/_build.py::kernel_source_hash).
    Returns (summary or None, a string that says where the counter figures come from or why there are none)."""
    from qgtc_ppopp22_amd._build import kernel_source_hash

    rel = f"{PROFILE_DIR}/summary_{name}.json"
    here = kernel_source_hash()
    try:
        with open(os.path.join(ROOT, rel)) as f:
            summ = json.load(f)
    except (OSError, ValueError):
        return None, f"none: {rel} absent"
    got = summ.get("kernel_source_hash")
    if got != here:
        return None, f"none: {rel} was collected from kernel sources {got}, this tree is {here}"
    return summ, f"{rel} (rocprofv3 --pmc passes, kernel sources {here})"


def batch_summaries(outs):
    """[n_local, 2] float64 on the outputs' device: (sum, element count) of every batch this rank ran."""
    return torch.stack([torch.stack([o.double().sum(), torch.tensor(float(o.numel()), device=o.device, dtype=torch.float64)])
                        for o in outs])


def keep_clock_up(plan_run):
    """The chip drops into a low power state within milliseconds of idling (time_steps has the numbers): CLOCK_WARMUP_S of
    untimed grouped epochs ahead of a measured run."""
    t_w = time.perf_counter()
    while time.perf_counter() - t_w < CLOCK_WARMUP_S:
        for _ in range(20):
            plan_run()
        torch.cuda.synchronize()


def epoch_leg(Q, rank, world, device_index, dataset="ogbn-arxiv", bits=2, hidden=128, gin=False, full=True, psize=1500,
              batch_size=20, only=None, weak=False, gather="summaries", classes=10):
    """Epoch time (BASELINE.json configs 3/4/5): a synthetic graph of the dataset's size, 75 batches. Legs: per-batch launches
    (the reference's structure: six extension calls per batch), the same with the packed batches parked on the CPU and
    uploaded every iteration (main_qgtc.py:115), the same captured in a hipGraph, cluster_gcn.py's pack-inside-the-loop
    structure per batch and grouped, and grouped launches (one launch per operator per epoch) of the reference's literal chain
    and of the layout-correct chain. EVERY leg: one iterator built ahead of the clock (main_qgtc.py:74-93), one untimed run,
    the clock kept up, then five runs whose median is reported (min / max beside it).
    Sharding (world > 1): `weak` False = the 75 batches round-robin over the ranks (strong scaling, BASELINE.json config 5);
    `weak` True = every rank runs all 75 batches of ITS OWN graph of that size (seed + rank): per-GPU work fixed."""
    from qgtc_ppopp22_amd import dist as D, driver, graph as G

    base = ["--dataset", dataset, "--n-hidden", str(hidden), "--n-classes", str(classes), "--bit_width", str(bits),
            "--use_QGTC", "--gpu", str(device_index), "--quiet", "--n-epochs", "20"] + (["--run_GIN"] if gin else [])
    base += ["--psize", str(psize), "--batch-size", str(batch_size)]
    n_batches = psize // batch_size
    graph = G.make_graph(dataset, psize, seed=2 + (rank if weak else 0))
    ids = list(range(n_batches)) if weak else D.shard_round_robin(n_batches, rank, world)
    res = {}
    legs = [("per_batch_reference_chain", []), ("batched_reference_chain", ["--batched"]),
            ("batched_correct_chain", ["--batched", "--chain", "correct"]),
            ("batched_correct_chain_engine_popcount", ["--batched", "--chain", "correct", "--engine", "popcount"])]
    if full:
        legs[1:1] = [("per_batch_nonresident_reference_chain", ["--non-resident"]),
                     ("per_batch_graph_reference_chain", ["--graph"]),
                     ("per_batch_2_streams_reference_chain", ["--streams", "2"]),
                     ("per_batch_pack_on_the_fly_reference_chain", ["--pack-on-the-fly"]),   # cluster_gcn.py's structure
                     ("batched_pack_on_the_fly_correct_chain", ["--batched", "--chain", "correct", "--pack-on-the-fly"])]
    if only is not None:
        legs = [l for l in legs if l[0] in only]
    dev = torch.device("cuda", device_index)
    r = None
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
    if world > 1 and r is not None:   # the one exchange of the path (RCCL over xGMI), outside every epoch clock
        total = n_batches * world if weak else n_batches
        if gather == "outputs":       # the per-batch float outputs themselves, padded to the largest batch
            t0 = time.perf_counter()
            allout, nodes = D.gather_batch_outputs(r["outs"], n_batches, rank, world, replicas=weak)
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


def epoch_roofline(Q, graph, device_index, dataset, bits, hidden, gin):
    """The second BASELINE metric against ITS floors: the grouped, layout-correct epoch (the default engine) timed with HIP
    events around the launches only - plan building and weight packing, which main_qgtc.py:96 puts inside its epoch clock,
    and the data loader's one-off packing of the iterator are timed separately. Algorithmic work is summed over the six
    operators and 75 batches from the logical shapes: bytes = a M K / 8 + w K N / 8 + output, FP4 MFMA ops =
    2 M K N x (base-4 digit pairs). `frac` divides the DENSE algorithmic bytes; `frac_on_traffic` divides what the counters say
    was moved (zero-tile jumping skips most of A) - both over the same kernel time."""
    from qgtc_ppopp22_amd import driver
    from qgtc_ppopp22_amd.sampler import ClusterIter

    dev = torch.device("cuda", device_index)
    it = ClusterIter(dataset, graph, 1500, 20, bit_width=bits, run_GIN=gin, device=dev, qgtc=Q, with_rows_X=True)
    data = it.epoch_data(Q)            # the data loader's share (one grouped pack of the iterator), ahead of the clock
    torch.cuda.synchronize()

    def ev(fn, reps=100):   # (a 20-epoch window is 0.6 ms, of which the queue's start-up is 5 - 10 %)
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / reps

    # the loader: the whole iterator packed again from the resident raw arrays (qgtc_load_batches: HIP events, and host wall clock)
    loader_us = ev(lambda: it.pack_now(Q), 10)
    t0 = time.perf_counter()
    for _ in range(10):
        it.pack_now(Q)
    torch.cuda.synchronize()
    loader_wall_us = (time.perf_counter() - t0) / 10 * 1e6
    # every operator on its own: a six-launch plan whose outputs are all in the public layouts (the chained plan keeps T
    # in the kernels' own formats between its launches - its stages cannot run alone)
    W = driver.pack_weights(Q, graph.feat.shape[1], hidden, 10, bits, dev)
    six = driver.PlannedEpoch(Q, data, it.cluster_param_li, W, bits, "correct", gin, fuse=False)
    stage_us = [round(ev(lambda i=i: data.run_launch(i)), 2) for i in range(six.n_launches)]
    torch.cuda.synchronize()
    host = []
    for _ in range(5):                  # what main_qgtc.py:96 puts inside its epoch clock besides the launches
        t0 = time.perf_counter()
        W = driver.pack_weights(Q, graph.feat.shape[1], hidden, 10, bits, dev)
        plan = driver.PlannedEpoch(Q, data, it.cluster_param_li, W, bits, "correct", gin)
        host.append((time.perf_counter() - t0) * 1e3)
        torch.cuda.synchronize()
    host_ms = sorted(host)[2]
    keep_clock_up(plan.run)
    epoch_us = sorted(ev(plan.run) for _ in range(5))[2]
    F, H, C, b = graph.feat.shape[1], hidden, 10, bits
    digits = lambda p: (p + 1) // 2       # noqa: E731
    ops = [  # (K_is_n, K, N, a, w, out: "bits"|"f32") per operator of the layout-correct chain
        [(False, F, H, b, b, "bits"), (True, 0, H, 1, b, "bits"), (False, H, H, b, b, "bits"), (True, 0, H, 1, b, "bits"),
         (False, H, C, b, b, "bits"), (True, 0, C, 1, b, "f32")],
        [(True, 0, F, 1, b, "bits"), (False, F, H, b, b, "bits"), (True, 0, H, 1, b, "bits"), (False, H, H, b, b, "bits"),
         (True, 0, H, 1, b, "bits"), (False, H, C, b, b, "f32")]][1 if gin else 0]
    algo_bytes = mfma_ops = eff_ops = 0.0
    for (n, _, _, _) in it.cluster_param_li:
        for (k_is_n, K, N, a, w, out) in ops:
            K = n if k_is_n else K
            algo_bytes += a * n * K / 8 + w * K * N / 8 + (4 * n * N if out == "f32" else b * n * N / 8)
            mfma_ops += 2.0 * n * K * N * digits(a) * digits(w)
            eff_ops += 2.0 * n * K * N
    occ = [round(data.occupied_fraction, 4)]
    floors = {"hbm_us": round(algo_bytes / (HBM_PEAK_GBS * 1e9) * 1e6, 2), "mfma_fp4_us": round(mfma_ops / (FP4_PEAK_TFLOPS * 1e12) * 1e6, 2),
              "launch_gaps_us": round(1.5 * (plan.n_launches - 1), 1)}
    chained = 6 - plan.n_launches
    # HBM-side traffic of one epoch from the committed PMC passes of the same launches (FETCH_SIZE x 2 on gfx950 + WRITE_SIZE, KiB
    # per dispatch, times the dispatches an epoch makes of each kernel) - only when the summary was collected from THESE sources
    traffic = None
    summ, traffic_source = profile_summary("epoch_gin" if gin else "epoch")
    if summ is not None:
        calls = {k["name"]: k["calls"] for k in summ.get("kernel_stats", []) if "k_rbw" in k["name"]}
        base = min(calls.values()) if calls else 0
        tot = 0.0
        for name, cs in summ.get("pmc_per_dispatch_mean", {}).items():
            if "k_rbw" not in name or not base:
                continue
            per_epoch = next((c for n_, c in calls.items() if n_[:60] == name[:60]), base) / base
            tot += per_epoch * (2.0 * cs.get("FETCH_SIZE", {}).get("mean", 0.0) + cs.get("WRITE_SIZE", {}).get("mean", 0.0)) * 1024.0
        traffic = int(tot) if tot > 0 else None
    frac = algo_bytes / epoch_us / 1e3 / HBM_PEAK_GBS
    return {"kernel_us_per_epoch": round(epoch_us, 2), "kernel_us_per_operator_alone": stage_us, "calls_per_epoch": plan.n_launches,
            "launches_per_epoch": plan.n_launches,
            "launch_structure": (f"6 operators in {plan.n_launches} launches: {chained} aggregation stages carry the next "
                                 "layer's X.W stage (qgtc_chain_transform / qgtc_chain_aggregate; adjacency "
                                 + ("as 512-byte tiles" if getattr(data, "a_tiles", False) else "in the rows layout") + ")") if chained else "6 grouped launches",
            "host_weight_pack_and_plan_bind_ms": round(host_ms, 4),
            "host_note": "host time (median of 5) of the two calls main_qgtc.py:96 puts inside its clock besides the launches: one fill + one pack "
                         "launch for the three weights, one allocation + ONE launch that fills every stage's descriptors on the device",
            "loader_us_per_iterator_hip_events": round(loader_us, 1), "loader_us_per_iterator_wall": round(loader_wall_us, 1),
            "loader_note": "the data loader's packing of all 75 batches (adjacency rows + tiles + bitmaps from the edge lists, X in the cols / rows / "
                           "chain layouts): ONE qgtc_load_batches call, six launches, ahead of the epoch clock as in main_qgtc.py:74-93 "
                           "(round 3: eight launches per batch, 2.6 ms of kernels)",
            "algorithmic_bytes_per_epoch": int(algo_bytes), "effective_ops_per_epoch": eff_ops,
            "eff_TOPS": round(eff_ops / epoch_us / 1e6, 1), "floors": floors,
            "roofline": {"bound": "hbm", "achieved": round(algo_bytes / epoch_us / 1e3, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(frac, 4),
                         "frac_mfma": round(mfma_ops / epoch_us / 1e6 / FP4_PEAK_TFLOPS, 4),
                         "traffic": traffic, "traffic_source": traffic_source,
                         "frac_on_traffic": round(traffic / epoch_us / 1e3 / HBM_PEAK_GBS, 4) if traffic else None,
                         "note": "a few thousand short workgroups per launch: bound by launch floors, dependent load chains and the epilogues' VALU work, see DESIGN.md section 6"},
            "adjacency_tiles_occupied": occ[:1]}


def epoch_block(ep, key="batched_correct_chain_ms"):
    """The compact per-epoch block that goes INSIDE the line's `roofline` (the part of the line the driver's record keeps): both
    halves of BASELINE.json's metric in one place."""
    rf = ep["roofline_of_the_grouped_correct_chain"]
    return {"ms_driver_style": ep[key], "kernel_us": rf["kernel_us_per_epoch"], "launches": rf["launches_per_epoch"],
            "algorithmic_bytes": rf["algorithmic_bytes_per_epoch"], "traffic_bytes": rf["roofline"]["traffic"],
            "frac": rf["roofline"]["frac"], "frac_on_traffic": rf["roofline"]["frac_on_traffic"],
            "loader_ms_once": round(rf["loader_us_per_iterator_hip_events"] / 1e3, 4),
            "plan_bind_ms": rf["host_weight_pack_and_plan_bind_ms"],
            "per_batch_unchanged_driver_ms": ep.get("per_batch_reference_chain_ms")}


def zero_tile_rows(Q, graph_arxiv, device_index):
    """`--zerotile_jump` (main_qgtc.py:142-145) on both synthetic graphs: the row parse_counter.py:31-33 would print
    (dataset, non-jumping, jumping, ratio over the CUMULATIVE counter lines) and the plain per-epoch ratio of 8-row x
    128-bit tile steps that survive zero-tile jumping."""
    import contextlib
    import io
    from qgtc_ppopp22_amd import driver, graph as G

    out = {}
    for dataset, bits, hidden, g in (("ogbn-arxiv", 2, 128, graph_arxiv), ("ppi", 4, 64, None)):
        args = driver.build_parser().parse_args(["--dataset", dataset, "--n-hidden", str(hidden), "--bit_width", str(bits), "--use_QGTC",
                                                 "--gpu", str(device_index), "--quiet", "--zerotile_jump"])
        Q.reset_counters()
        with contextlib.redirect_stdout(io.StringIO()):
            r = driver.run(args, Q=Q, graph=g if g is not None else G.make_graph(dataset, 1500))
        z = r["zerotile"]
        out[dataset] = {"parse_counter_row": z["line"], "per_epoch_non_jumping": z["per_epoch_non_jumping"],
                        "per_epoch_jumping": z["per_epoch_jumping"], "per_epoch_ratio": round(z["per_epoch_ratio"], 4)}
    Q.reset_counters()
    return out


# the reference's published effective TFLOPs (sm_86) for M=K, N, width
REF_MICRO = {
    (1024, 16): (5.847, 3.934, 2.488, 1.541), (2048, 16): (16.605, 10.086, 6.561, 3.483), (4096, 16): (40.627, 20.764, 12.409, 6.763),
    (1024, 32): (11.724, 7.864, 4.456, 3.074), (2048, 32): (32.666, 19.762, 12.807, 6.816), (4096, 32): (35.032, 20.951, 13.929, 7.366),
    (1024, 64): (23.219, 15.429, 10.683, 5.046), (2048, 64): (37.438, 25.055, 12.328, 6.165), (4096, 64): (46.768, 26.818, 14.196, 7.324),
}


# adjacency-matrix-size study, 1-bit (5_9_adjmatrix_size.py), sm_86 effective TFLOPs
REF_ADJ = {16: (5.831, 16.323, 34.425), 32: (11.717, 32.027, 40.175), 64: (23.158, 37.444, 46.759), 128: (28.417, 40.646, 52.517),
           256: (32.089, 44.151, 59.508), 512: (41.743, 49.687, 64.172), 1024: (37.954, 52.970, 66.490)}


def identical_and_closed_form(Q, words, M, K, N, w):
    """Per-point parity flag of the benchmark tables: the default engine's packed words equal the AND + popcount
    kernels' AND decode to the closed form of the all-ones inputs (C = K everywhere, re-quantised: 2^w - 1 where
    K > 2^w; 2_7c_QGTC_GEMM_INT8.py:30-41). Full-size comparison with the oracle: tests/test_gpu_fullsize.py."""
    same = torch.equal(words["auto"], words["popcount"])
    back = Q.bit2val(words["auto"], w, M, N, False, False)
    want = (2 ** w - 1) if K > 2 ** w else (K & (2 ** w - 1))
    return bool(same and bool((back == want).all().item()))


def adj_size_table(Q, device):
    """The reference's adjacency-size study (5_9_adjmatrix_size.py): 1-bit, M = K in 1024/2048/4096,
    N = 16 .. 1024, all-ones inputs, 200 launches per point, median of 5 windows."""
    out = {}
    for nn, ref in REF_ADJ.items():
        row = {}
        for mi, mk in enumerate((1024, 2048, 4096)):
            _, _, ba, bx = make_workload(Q, mk, mk, nn, 1, device, seed=3, ones=True)
            row[f"M{mk}"] = {"ref_sm86": ref[mi]}
            words = {}
            for eng, key in (("auto", "TOPS"), ("popcount", "TOPS_engine_popcount")):
                with engine(Q, eng):
                    ms = median_of_5(Q, ba, bx, mk, mk, nn, 1)
                    words[eng] = Q.bitMM2Bit(ba, bx, mk, mk, nn, 1, 1, 1)
                row[f"M{mk}"][key] = round(2.0 * mk * mk * nn * 200 / (ms * 1e-3) / 1e12, 2)
            row[f"M{mk}"]["identical"] = identical_and_closed_form(Q, words, mk, mk, nn, 1)
        out[f"N{nn}"] = row
    return out


def micro_bench_table(Q, device):
    """The reference's whole micro-benchmark (2_7c_QGTC_GEMM_INT8.py:13-20: 9 shapes x widths 1 .. 8 - its README publishes 1 / 2 / 4 / 8 -,
    200 launches per point between two events, all-ones inputs as there), median of 5 windows."""
    out = {}
    published = {1: 0, 2: 1, 4: 2, 8: 3}
    for (mk, nn), ref in REF_MICRO.items():
        row = {}
        for ww in range(1, 9):
            _, _, ba, bx = make_workload(Q, mk, mk, nn, ww, device, seed=3, ones=True)
            row[f"w{ww}"] = {"ref_sm86": ref[published[ww]] if ww in published else None}
            words = {}
            for eng, key in (("auto", "TOPS"), ("popcount", "TOPS_engine_popcount")):
                with engine(Q, eng):
                    ms = median_of_5(Q, ba, bx, mk, mk, nn, ww)
                    words[eng] = Q.bitMM2Bit(ba, bx, mk, mk, nn, 1, ww, ww)
                row[f"w{ww}"][key] = round(2.0 * mk * mk * nn * 200 / (ms * 1e-3) / 1e12, 2)
            row[f"w{ww}"]["identical"] = identical_and_closed_form(Q, words, mk, mk, nn, ww)
        out[f"{mk}x{mk}x{nn}"] = row
    return out


def unchanged_driver_in_a_child(env_extra, timeout=240):
    """The reference's literal per-batch loop (main_qgtc.py:112-155: 75 batches x six extension calls, 20 epochs) in a FRESH child
    process with extra environment - for process-wide HIP runtime settings that cannot be flipped once this process has touched
    the GPU. Must be called BEFORE this process initialises the GPU. Returns the median `Avg. Epoch` (ms) of five runs after an
    untimed one, or None."""
    import subprocess

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


def launch_ranks(args) -> int:
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes (one per GPU, the layout
    torch.distributed.run would give them: all devices visible, LOCAL_RANK picks one) BEFORE this process makes any
    GPU call, wait for them, and fail if any of them fails. Rank 0 prints the JSON line straight to our stdout."""
    import socket
    import subprocess

    n = args.gpus
    if not args.dry_run:
        have = torch.cuda.device_count()       # counting devices does not initialise the GPU
        if have < n:
            print(f"bench.py: --gpus {n} but only {have} GPU(s) visible", file=sys.stderr)
            return 2
    with socket.socket() as s:                 # a free rendezvous port on the loopback interface
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs on this driver
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    try:
        pending = list(procs)
        while pending:
            for p in list(pending):
                code = p.poll()
                if code is None:
                    continue
                pending.remove(p)
                if code != 0 and rc == 0:      # one rank failed: the others would wait in a collective forever
                    rc = code
                    for q in pending:
                        q.terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


def dry_run(args):
    """Launcher / collective plumbing without a GPU (CPU tests: `--dry-run --backend gloo`): every rank takes part in
    the same barrier, max-over-ranks and checksum gather as the real run and rank 0 prints the line's skeleton."""
    from qgtc_ppopp22_amd import dist as D

    rank, world, local = D.init_from_env(backend=args.backend or "gloo")
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    if os.environ.get("QGTC_BENCH_FAIL_RANK") == str(rank):   # test hook: a rank that dies before the collectives
        sys.exit(3)
    dev = torch.device("cpu")
    D.barrier()
    wall = D.max_over_ranks(0.001 * (rank + 1), dev)
    csum = torch.tensor([[float(1000 + rank)]], dtype=torch.float64)
    sums = D.gather_batch_summaries(csum, world, rank, world)
    ids = D.shard_round_robin(75, rank, world)
    counts = D.gather_batch_summaries(torch.tensor([[float(len(ids))]], dtype=torch.float64), world, rank, world)
    # the end-of-epoch exchange with the real payload's shape: ragged per-batch float outputs [n_i, 10] (n_i = 1190 + 7 i mod 50 nodes,
    # every element = the batch id), sharded round-robin, gathered padded; and the weak-scaled form (every rank its own 3 batches)
    fake = lambda i: torch.full((1190 + (7 * i) % 50, 10), float(i))     # noqa: E731
    allout, nodes = D.gather_batch_outputs([fake(i) for i in ids], 75, rank, world)
    rep_out, rep_nodes = D.gather_batch_outputs([fake(100 * rank + j) for j in range(3)], 3, rank, world, replicas=True)
    rep_sums = D.gather_replica_summaries(torch.tensor([[float(rank), float(j)] for j in range(3)], dtype=torch.float64), world)
    if rank == 0:
        print(json.dumps({"metric": "dry run (no GPU work)", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "max_wall_s": wall, "extras": {"rank_checksums": [float(v) for v in sums.view(-1).tolist()],
                                                         "batches_per_rank": [int(v) for v in counts.view(-1).tolist()],
                                                         "gathered_output_shape": list(allout.shape),
                                                         "gathered_output_nodes": [int(v) for v in nodes.tolist()],
                                                         "gathered_output_first_values": [float(allout[i, 0, 0]) for i in range(allout.size(0))],
                                                         "gathered_output_padding_is_zero": bool(all(float(allout[i, int(nodes[i]):].abs().sum()) == 0.0
                                                                                                     for i in range(allout.size(0)))),
                                                         "replica_output_shape": list(rep_out.shape),
                                                         "replica_output_first_values": [float(rep_out[i, 0, 0]) for i in range(rep_out.size(0))],
                                                         "replica_nodes": [int(v) for v in rep_nodes.tolist()],
                                                         "replica_summaries": rep_sums.tolist()}}), flush=True)
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))           # (nothing above this line touches the GPU)
    if args.dry_run:
        return dry_run(args)
    from qgtc_ppopp22_amd import dist as D

    rank, world, local = D.init_from_env(backend=args.backend)
    host_kernarg_ms = None
    if rank == 0 and world == 1 and not args.no_extras and torch.cuda.device_count() > 0:
        # (a child process, started before THIS process touches the GPU: the knob is read when the HIP runtime starts)
        host_kernarg_ms = unchanged_driver_in_a_child({"HIP_FORCE_DEV_KERNARG": "0"})
    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP path has no fallback)"
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: one rank per GPU (run `python bench.py --gpus N` " \
                               "or torch.distributed.run with --nproc-per-node N)"
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)
    import QGTC as Q

    M = K = 4096
    N, w = 64, args.bits
    A, X, bit_A, bit_X = make_workload(Q, M, K, N, w, device, seed=3 + rank)
    with engine(Q, "popcount"):
        out = Q.bitMM2Bit(bit_A, bit_X, M, K, N, 1, w, w)   # AND + popcount kernels: the words every engine must produce
    with engine(Q, args.engine):
        out_e = Q.bitMM2Bit(bit_A, bit_X, M, K, N, 1, w, w)
        assert torch.equal(out_e, out), "engines disagree"
        wall, kern, wall_ev, issue_used, issue_probe = time_steps(Q, out_e, bit_A, bit_X, M, K, N, w, args.steps, args.warmup, D.barrier, args.streams, args.issue)
        assert torch.equal(out_e, out), "the timed launches changed the result"      # (the replayed / eager launches wrote out_e again)
        eager_headline = None
        if rank == 0 and world == 1 and not args.no_extras and args.streams <= 1:
            other_issue = "graph" if issue_used == "eager" else "eager"
            e_wall, e_kern, _, _, _ = time_steps(Q, out_e, bit_A, bit_X, M, K, N, w, args.steps, args.warmup, D.barrier, 1, other_issue)
            eager_headline = {"issue": other_issue, "TOPS": round(args.steps * 2.0 * M * K * N / e_wall / 1e12, 3),
                              "ms_per_step": round(e_wall * 1e3 / args.steps, 6), "us_per_launch_hip_events": round(e_kern * 1e6, 3),
                              "note": "the same K steps issued the other way (graph: captured once ahead of the timed region, one hipGraphLaunch inside "
                                      "it; eager: K hipLaunchKernel calls inside it)"}
    # the same headline launches on the other engine (identical words), measured back to back with the headline
    # (before the CPU baseline occupies every host core)
    other_engine, other_headline = ("popcount" if args.engine != "popcount" else "auto"), None
    if rank == 0 and world == 1 and not args.no_extras:
        with engine(Q, other_engine):
            o_wall, o_kern, _, _, _ = time_steps(Q, torch.empty_like(out), bit_A, bit_X, M, K, N, w, args.steps, args.warmup, D.barrier, args.streams, args.issue)
        other_headline = {"TOPS": round(args.steps * 2.0 * M * K * N / o_wall / 1e12, 3), "us_per_launch": round(o_kern * 1e6, 3)}
        if other_engine == "popcount":   # the engine BASELINE.json's north star names, against BOTH statements of its VALU roofline
            other_headline["valu_frac_of_measured_pair_rate_4.2e13"] = round(eff_ops_of(M, K, N) * w / o_kern / VALU_PEAK_BITOPS, 4)
            other_headline["valu_frac_of_survey_8d_peak_7.864e13"] = round(eff_ops_of(M, K, N) * w / o_kern / VALU_PEAK_BITOPS_SURVEY, 4)
            other_headline["rocprof"] = f"{PROFILE_DIR}/summary_popcount.json"
    # what runs at this shape: the FP4 matrix-core kernel for narrow right operands (launch.hip.h: skinny_ok -
    # N <= 64, at most 2 x 8 planes, float32 sums exact: K (2^a - 1)(2^w - 1) < 2^24)
    fp4_kernel = args.engine != "popcount" and w <= 8 and K * (2 ** w - 1) < 2 ** 24
    wall_max = D.max_over_ranks(wall, device)
    eff_ops = 2.0 * M * K * N
    value = world * args.steps * eff_ops / wall_max / 1e12

    # result checksum of every rank, gathered over RCCL (the only collective of the path)
    csum = torch.tensor([[float(out.to(torch.int64).sum().item())]], dtype=torch.float64, device=device)
    sums = D.gather_batch_summaries(csum, world, rank, world) if world > 1 else csum

    algo_bytes = 1 * M * K / 8 + w * K * N / 8 + w * M * N / 8      # a*M*K/8 + w*K*N/8 + ob*M*N/8
    # Evidence kept under profiles/ (tools/collect_profiles.sh, rocprofv3 on the SAME workload through
    # tools/profile_targets.py headline): the kernel's per-dispatch duration under --kernel-trace and, from separate
    # --pmc passes, FETCH_SIZE / WRITE_SIZE in KiB per launch (HBM-side bytes = 2 x FETCH_SIZE + WRITE_SIZE: the gfx950
    # correction for 16-byte-per-lane reads, MI355X_MICROARCH.md). Only the 1-bit workload on the default engine has one.
    traffic, rocprof, traffic_source = None, None, "none: only the 1-bit workload on the default engine is profiled"
    if w == 1 and fp4_kernel:
        prof, traffic_source = profile_summary("headline")
        try:
            if prof is not None:
                ks = [k for k in prof["kernel_stats"] if "k_bitmm_fp4_one" in k["name"]][0]
                rocprof = {"file": f"{PROFILE_DIR}/kernel_stats_headline.csv", "calls": ks["calls"], "avg_us": round(ks["avg_ns"] / 1e3, 3),
                           "min_us": round(ks["min_ns"] / 1e3, 3),
                           "hbm_frac_at_avg": round(algo_bytes / (ks["avg_ns"] * 1e-9) / 1e9 / HBM_PEAK_GBS, 5),
                           "note": "kernel span per dispatch under the tracer (no launch gap; the tracer's completion signals "
                                   f"stretch a dispatch this short, see {PROFILE_DIR}/README.md)"}
                pm = [v for k, v in prof["pmc_per_dispatch_mean"].items() if "k_bitmm_fp4_one" in k][0]
                traffic = int(2 * 1024 * pm["FETCH_SIZE"]["mean"] + 1024 * pm["WRITE_SIZE"]["mean"])
        except (KeyError, ValueError, IndexError):
            traffic, rocprof, traffic_source = None, None, "none: " + traffic_source + " (unreadable)"
    # The launches of the timed region run back to back on one stream: its wall clock per step (host issue and the final drain included) is
    # an UPPER bound of the kernel's average launch duration. On some boxes the event-bracketed windows behind a long timed region read
    # 3.6-4.1 us per launch beside 3.0-3.1 us per step of wall clock (r04, --steps 200): the instrument, not the kernel - the bound wins then.
    kern_events, wall_per_step = kern, wall / args.steps
    kern = min(kern_events, wall_per_step) if args.streams <= 1 else kern_events
    hbm_floor_us, mfma_floor_us = algo_bytes / (HBM_PEAK_GBS * 1e9) * 1e6, eff_ops / (FP4_PEAK_TFLOPS * 1e12) * 1e6
    frac_hbm = round(algo_bytes / kern / 1e9 / HBM_PEAK_GBS, 5)
    if fp4_kernel:
        # dominant kernel: k_bitmm_fp4_one (v_mfma_scale_f32_16x16x128_f8f6f4 on E2M1 codes of the bit planes). Its larger
        # floor is the HBM one (2.16 MB / 8 TB/s = 0.27 us against 2.1 Gop / 10 PF = 0.21 us), so that is the bound named.
        roofline = {"bound": "hbm", "kernel": "k_bitmm_fp4_one<1,%d,0,2,2>" % ({1: 1, 2: 2}.get(w, 4 if w <= 4 else 8)),
                    "achieved": round(algo_bytes / kern / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": frac_hbm,
                    "traffic": traffic, "traffic_source": traffic_source, "algorithmic_bytes_per_launch": int(algo_bytes),
                    "avg_launch_us": round(kern * 1e6, 3), "avg_launch_us_hip_events": round(kern_events * 1e6, 3),
                    "avg_launch_bound": "hip_events" if kern_events <= wall_per_step else "wall clock per step of the timed region (an upper bound of it; the event windows read more)",
                    "avg_launch_window": "%d x %d launches, issued like the timed region (%s)" % (max(3, args.steps // EVENT_MIN_LAUNCHES), EVENT_MIN_LAUNCHES, issue_used),
                    "avg_launch_source": "HIP events on the launch stream around windows of %d of the same launches issued right behind the timed "
                                         "region (inside it the two event records cost 11-12 us of a 72 us window), the median of max(3, K // %d) "
                                         "windows: kernel + dependent-launch gap (~1.5 us), i.e. what a caller gets per launch" % (EVENT_MIN_LAUNCHES, EVENT_MIN_LAUNCHES),
                    "wall_ms_per_step_of_the_event_bracketed_region": round(wall_ev * 1e3 / args.steps, 6),
                    "frac_hbm": frac_hbm, "frac_mfma": round(eff_ops / kern / 1e12 / FP4_PEAK_TFLOPS, 5),
                    "floors_us": {"hbm": round(hbm_floor_us, 3), "mfma_fp4": round(mfma_floor_us, 3)},
                    "rocprof": rocprof,
                    "note": "latency-bound: one memory round trip, 16 MFMAs per wave, one LDS reduction and the epilogue behind a "
                            "dependent-launch gap; DESIGN.md 5.4e"}
    else:
        roofline = {"bound": "hbm", "kernel": "k_bitmm<%d,1,%d,ZS>" % ({1: 4, 2: 4, 4: 2, 8: 1}.get(w, 1), w), "achieved": round(algo_bytes / kern / 1e9, 2),
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": frac_hbm,
                    "traffic": traffic, "algorithmic_bytes_per_launch": int(algo_bytes),
                    "avg_launch_us": round(kern * 1e6, 3), "frac_hbm": frac_hbm,
                    "valu": {"achieved_bitops": round(eff_ops * w / kern, 1), "peak_bitops": VALU_PEAK_BITOPS,
                             "frac": round(eff_ops * w / kern / VALU_PEAK_BITOPS, 4),
                             "peak_bitops_survey_8d": VALU_PEAK_BITOPS_SURVEY,
                             "frac_of_survey_8d_peak": round(eff_ops * w / kern / VALU_PEAK_BITOPS_SURVEY, 4),
                             "note": "binding roofline of the popcount path: v_and_b32+v_bcnt_u32_b32 issue, peak measured by tools/valu_peak.hip"}}

    line = {
        "metric": "effective bit-GEMM TOPS (2*M*K*N/t), 1-bit A x %d-bit X, M=K=4096, N=64" % w,
        "value": round(value, 3), "unit": "TOPS", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(wall_max * 1e3 / args.steps, 6),
        "higher_is_better": True, "scaling": "weak",
        "vs_baseline": round(value / REF_TFLOPS_4096_64[w], 3) if w in REF_TFLOPS_4096_64 else None,
        "dtype": ("fp4 (E2M1 codes of the bit planes, MFMA, float32 sums of exact integers)" if fp4_kernel
                  else "u32 bit-planes (AND+popcount into int32)"), "data": "synthetic",
        "config": {"workload": f"bitMM2Bit {M}x{K}x{N}, a=1, w={w}, ob={w} (BASELINE.json configs[1], 2_7c shape)",
                   "inputs": "seeded Bernoulli(0.5) adjacency, uniform w-bit features, packed and resident in HBM",
                   "parallelism": f"replica-per-GPU x{world}, no data-path collective",
                   "engine": args.engine + (" (library default) -> FP4 matrix-core kernel for narrow right operands" if fp4_kernel else " -> AND + popcount kernels"),
                   "issue": ("back-to-back launches on one stream (the reference's metric)" + (": the K launches captured once in a hipGraph ahead of the "
                             "timed region, ONE hipGraphLaunch inside it (same kernels in stream order; eager issue: extras.headline_other_issue)"
                             if issue_used == "graph" else ": K hipLaunchKernel calls inside the timed region (graph replay: extras.headline_other_issue)")
                             + ("" if issue_probe is None else f"; chosen by an untimed probe ahead of the timed region (medians of five runs of the K steps, "
                                                               f"us per step: {issue_probe}; the graphs are only built when eager is above 3.05 + 20 / K)")) if args.streams <= 1
                            else f"independent launches round-robin on {args.streams} HIP streams",
                   "clock_warmup": f"{CLOCK_WARMUP_S} s of untimed launches ahead of the W warmup steps (the chip idles into a low "
                                   "power state while the host builds inputs; nothing else precedes the timed region)"},
        "roofline": roofline,
    }

    if rank == 0 and world == 1 and not args.no_extras:
        cb, ref = cpu_baseline(M, K, N, w, A, X, args.cpu_seconds)
        line["cpu_baseline"] = cb
        got = out.cpu().numpy().view(np.uint32).reshape(-1)
        line["parity_vs_oracle"] = bool((got == ref).all())
    extras = {}
    try:
        if not args.no_extras:
            if rank == 0 and world == 1:
                extras["headline_on_engine_" + other_engine] = other_headline
                extras["headline_other_issue"] = eager_headline
                sweep = {}
                for ww in (1, 2, 4, 8):
                    for label, ones in (("random", False), ("ones", True)):
                        _, _, ba, bx = make_workload(Q, M, K, N, ww, device, seed=3, ones=ones)
                        sweep[f"w{ww}_{label}"] = {"ref_sm86_TFLOPs": REF_TFLOPS_4096_64[ww]}
                        for eng, key in (("auto", "TOPS"), ("popcount", "TOPS_engine_popcount")):
                            with engine(Q, eng):
                                ms = median_of_5(Q, ba, bx, M, K, N, ww)
                            sweep[f"w{ww}_{label}"][key] = round(eff_ops * 200 / (ms * 1e-3) / 1e12, 2)
                            if eng == "auto":
                                sweep[f"w{ww}_{label}"]["us_per_launch"] = round(ms * 1e3 / 200, 3)
                extras["width_sweep_4096x4096x64"] = sweep
                # Independent launches (different cluster batches in a serving loop) need not be serialised
                # by stream order: the same products issued round-robin on two HIP streams, each launch
                # with its own output buffer. NOT the headline metric (that one is the reference's: launches
                # back to back on one stream); it shows what the launch-to-launch dependency costs.
                ovl = {}
                for ww in (1, 2, 4, 8):
                    _, _, ba, bx = make_workload(Q, M, K, N, ww, device, seed=3)
                    with engine(Q, "popcount"):
                        ref_out = Q.bitMM2Bit(ba, bx, M, K, N, 1, ww, ww)
                    outs2 = [torch.empty_like(ref_out) for _ in range(2)]
                    Q.bitMM2Bit_enqueue_streams(outs2, ba, bx, M, K, N, 1, ww, ww, 50)
                    torch.cuda.synchronize()
                    best = None
                    for _ in range(3):
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        Q.bitMM2Bit_enqueue_streams(outs2, ba, bx, M, K, N, 1, ww, ww, 1000)
                        e1.record()
                        torch.cuda.synchronize()
                        ms = e0.elapsed_time(e1)
                        best = ms if best is None else min(best, ms)
                    ok = all(torch.equal(o, ref_out) for o in outs2)
                    ovl[f"w{ww}"] = {"TOPS": round(eff_ops * 1000 / (best * 1e-3) / 1e12, 2),
                                     "us_per_launch": round(best * 1e3 / 1000, 3), "outputs_identical": bool(ok)}
                extras["independent_launches_on_2_streams_4096x4096x64"] = ovl
                # the reference's Fig. 8a comparison: INT8 GEMM on the matrix cores (its cuBLAS numbers
                # are BASELINE.md §2) beside the 1-bit popcount path on the same nine shapes
                cmp9 = {}
                ref_cublas = {(1024, 16): 0.55, (2048, 16): 2.58, (4096, 16): 3.60, (1024, 32): 3.89, (2048, 32): 5.49,
                              (4096, 32): 6.49, (1024, 64): 4.38, (2048, 64): 6.30, (4096, 64): 6.65}
                ref_1bit = {(1024, 16): 5.847, (2048, 16): 16.605, (4096, 16): 40.627, (1024, 32): 11.724, (2048, 32): 32.666,
                            (4096, 32): 35.032, (1024, 64): 23.219, (2048, 64): 37.438, (4096, 64): 46.768}
                g = torch.Generator(device="cpu").manual_seed(5)
                for nn in (16, 32, 64):
                    for mk in (1024, 2048, 4096):
                        A8 = torch.randint(-128, 128, (mk, mk), generator=g, dtype=torch.int8).to(device)
                        B8 = torch.randint(-128, 128, (nn, mk), generator=g, dtype=torch.int8).to(device)
                        Q.i8gemm_profile(A8, B8, 20, False)
                        ms8 = min(Q.i8gemm_profile(A8, B8, 200, False) for _ in range(3))
                        lib8 = None   # the vendor-library GEMM the reference compares with (cuBLAS there, hipBLASLt here)
                        try:
                            B8kn = B8.t().contiguous()
                            for _ in range(5):
                                torch._int_mm(A8, B8kn)
                            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                            e0.record()
                            for _ in range(200):
                                torch._int_mm(A8, B8kn)
                            e1.record()
                            torch.cuda.synchronize()
                            lib8 = round(2.0 * mk * mk * nn * 200 / (e0.elapsed_time(e1) * 1e-3) / 1e12, 2)
                        except Exception:   # noqa: BLE001 - optional leg
                            lib8 = None
                        _, _, ba, bx = make_workload(Q, mk, mk, nn, 1, device, seed=3)
                        with engine(Q, "popcount"):
                            ms1 = median_of_5(Q, ba, bx, mk, mk, nn, 1)
                        ms1a = median_of_5(Q, ba, bx, mk, mk, nn, 1)
                        ops = 2.0 * mk * mk * nn * 200
                        cmp9[f"{mk}x{mk}x{nn}"] = {"int8_mfma_TOPS": round(ops / (ms8 * 1e-3) / 1e12, 2),
                                                  "int8_hipblaslt_TOPS": lib8,
                                                  "bit1_popcount_TOPS": round(ops / (ms1 * 1e-3) / 1e12, 2),
                                                  "bit1_default_engine_TOPS": round(ops / (ms1a * 1e-3) / 1e12, 2),
                                                  "ref_sm86_cublas_int8_TFLOPS": ref_cublas[(mk, nn)],
                                                  "ref_sm86_qgtc_1bit_TFLOPs": ref_1bit[(mk, nn)]}
                cmp9["note"] = ("comparison path only (the reference's Fig. 8a: INT8 tensor-core GEMM beside the 1-bit path); at N <= 64 "
                                "the int8 GEMMs are bound by operand replication, a few percent of the int8 MFMA peak - DESIGN.md 5.4")
                extras["int8_mfma_vs_1bit_popcount_9_shapes"] = cmp9
            if rank == 0 and world == 1:
                extras["micro_bench_ones_9_shapes_x_8_widths"] = micro_bench_table(Q, device)
                extras["adjacency_size_study_1bit"] = adj_size_table(Q, device)
                # the opt-in matrix-core engine (bit planes expanded to int8 on the fly, exact) beside the
                # popcount engine on wide products, where an expanded operand byte feeds several MFMA tiles
                eng = {}
                for (mm, kk, nn, ww) in ((4096, 4096, 1024, 1), (4096, 4096, 1024, 2), (4096, 4096, 1024, 4),
                                         (8192, 4096, 1024, 1), (8192, 4096, 1024, 2)):
                    _, _, ba, bx = make_workload(Q, mm, kk, nn, ww, device, seed=3)
                    row = {}
                    outs_e = {}
                    for name in ("popcount", "mfma", "auto"):
                        with engine(Q, name):
                            ms = median_of_5(Q, ba, bx, mm, kk, nn, ww, reps=50)
                            outs_e[name] = Q.bitMM2Bit(ba, bx, mm, kk, nn, 1, ww, ww)
                        row[name + "_TOPS"] = round(2.0 * mm * kk * nn * 50 / (ms * 1e-3) / 1e12, 1)
                    row["outputs_identical"] = bool(torch.equal(outs_e["popcount"], outs_e["mfma"]) and torch.equal(outs_e["popcount"], outs_e["auto"]))
                    eng[f"{mm}x{kk}x{nn}_w{ww}"] = row
                extras["mfma_engine_vs_popcount_wide_products"] = eng
            multi = ("per_batch_reference_chain", "batched_correct_chain") if world > 1 else None   # (N > 1: the unchanged-driver and the grouped leg)
            ep, graph = epoch_leg(Q, rank, world, local, only=multi, gather=args.gather)
            extras["cluster_gcn_epoch_ogbn_arxiv_shape"] = ep
            if world > 1:
                ep["scaling"] = "strong: the 75 batches round-robin over the ranks (BASELINE.json configs[4]); epoch = max over ranks"
                # the same epoch weak-scaled: every rank runs 75 batches of its own arxiv-sized graph (per-GPU work fixed, 75 x world batches)
                ep_w, _ = epoch_leg(Q, rank, world, local, only=("batched_correct_chain",), weak=True, gather=args.gather)
                ep_w["scaling"] = "weak: 75 batches per rank (each rank its own ogbn-arxiv-sized graph), %d batches in all; epoch = max over ranks" % (75 * world)
                ep_w["batches_per_second"] = round(75 * world / (ep_w["batched_correct_chain_ms"] * 1e-3), 1)
                extras["cluster_gcn_epoch_ogbn_arxiv_shape_weak_scaled"] = ep_w
            if rank == 0 and world == 1:
                ep["per_batch_reference_chain_host_kernarg_ms"] = host_kernarg_ms
                ep["per_batch_note"] = ("the unchanged driver's six extension calls per batch are host-bound: ~3.7 us of each call is hipLaunchKernel writing the "
                                        "kernel arguments into device memory; with HIP_FORCE_DEV_KERNARG=0 (arguments in host memory: 2.4 us per launch, but "
                                        "every kernel starts ~1.2 us later - the headline would drop to ~490 TOPS) the same loop is the *_host_kernarg figure, "
                                        "measured in a fresh child process ahead of this one's GPU work; DESIGN.md section 6")
                ep["roofline_of_the_grouped_correct_chain"] = epoch_roofline(Q, graph, local, "ogbn-arxiv", 2, 128, False)
                extras["zero_tile_jumping"] = zero_tile_rows(Q, graph, local)
            # BASELINE.json configs[3]: Batched-GIN, ppi-sized graph, 4-bit weights/features, hidden 64 (0_7b's value)
            ep_gin, _ = epoch_leg(Q, rank, world, local, dataset="ppi", bits=4, hidden=64, gin=True, full=False, only=multi, gather=args.gather)
            extras["batched_gin_epoch_ppi_shape_4bit"] = ep_gin
            if rank == 0 and world == 1:
                from qgtc_ppopp22_amd import graph as G2
                ep_gin["roofline_of_the_grouped_correct_chain"] = epoch_roofline(Q, G2.make_graph("ppi", 1500), local, "ppi", 4, 64, True)
                # both halves of BASELINE.json's metric inside the part of the line the driver's record keeps: nested blocks, and the same
                # figures once more as flat scalars (a record that keeps only scalars still has them)
                blocks = {"epoch_cluster_gcn": epoch_block(ep), "epoch_batched_gin": epoch_block(ep_gin)}
                line["roofline"].update(blocks)
                for tag, blk in (("gcn", blocks["epoch_cluster_gcn"]), ("gin", blocks["epoch_batched_gin"])):
                    for k in ("ms_driver_style", "kernel_us", "launches", "algorithmic_bytes", "traffic_bytes", "frac", "frac_on_traffic", "loader_ms_once",
                              "plan_bind_ms", "per_batch_unchanged_driver_ms"):
                        line["roofline"][f"epoch_{tag}_{k}"] = blk[k]
                line["roofline"]["epoch_note"] = ("epoch_gcn_* = Cluster-GCN ogbn-arxiv-sized 2-bit (BASELINE.json configs[2]), epoch_gin_* = Batched-GIN "
                                                  "ppi-sized 4-bit (configs[3]); ms_driver_style = main_qgtc.py:157-159's Avg. Epoch of the grouped plan, "
                                                  "kernel_us = HIP events around the launches, frac = dense algorithmic bytes / kernel time / 8 TB/s, "
                                                  "frac_on_traffic = counter bytes instead; loader_ms_once = the iterator's one-off packing (GPU time)")
            if rank == 0 and world == 1:
                # The ONE epoch table the reference publishes (README.md:84-89): Cluster-GCN, the script's settings -
                # hidden 16, psize 1500, batch 20, 2 bits, each dataset's own --dim / --n-classes (0_7a_eval_QGTC_cluster_GCN.py:6-16,38-40:
                # ppi and ogbn-arxiv run on main_qgtc.py's default 10 classes) - on synthetic graphs of those datasets' sizes
                table = {}
                for ds, cls, ref_ms in (("artist", 12, 263.646), ("soc-BlogCatalog", 39, 209.495), ("ppi", 10, 189.016), ("ogbn-arxiv", 10, 208.616)):
                    e4, _ = epoch_leg(Q, rank, world, local, dataset=ds, bits=2, hidden=16, classes=cls, full=False,
                                      only=("per_batch_reference_chain", "batched_correct_chain"))
                    table[ds] = {"per_batch_unchanged_driver_ms": e4["per_batch_reference_chain_ms"], "grouped_correct_chain_ms": e4["batched_correct_chain_ms"],
                                 "ref_sm86_ms": ref_ms}
                table["note"] = ("README.md:84-89's table (it does not say whether its figures are the QGTC or the DGL run); synthetic SBM graphs with the "
                                 "datasets' node / edge counts and feature widths, 75 cluster batches each")
                extras["readme_cluster_gcn_table_hidden16_2bit"] = table
                # (the same four rows inside the part of the line the driver's record keeps: [unchanged per-batch loop, grouped plan, reference sm_86] ms)
                line["roofline"]["epoch_readme_table_ms"] = {ds: [r["per_batch_unchanged_driver_ms"], r["grouped_correct_chain_ms"], r["ref_sm86_ms"]]
                                                             for ds, r in table.items() if isinstance(r, dict)}
            if rank == 0 and world == 1:
                from oracle.dgl_cpu_baseline import graphsage_cpu_epoch
                from qgtc_ppopp22_amd import graph as G

                par = G.partition_list(graph, 1500)
                graphsage_cpu_epoch(graph, par, 1500, 20, 128, 10, n_batches=2)
                secs, nb = graphsage_cpu_epoch(graph, par, 1500, 20, 128, 10, n_batches=15)
                extras["dgl_style_fp32_cpu_epoch_ms"] = {"value": round(secs * 1e3 * 75 / nb, 2), "cores": torch.get_num_threads(),
                                                          "sample": f"ogbn-arxiv-sized graph, {nb} of 75 batches, scaled x{75 / nb:.0f}",
                                                          "kind": "port (torch-CPU GraphSAGE-sum x3; DGL not installable)"}
                # BASELINE.json configs[0] names ppi: the same stand-in on the ppi-sized graph (1_7a_eval_DGL_cluster_GCN.py's dataset)
                g_ppi = G.make_graph("ppi", 1500)
                par_ppi = G.partition_list(g_ppi, 1500)
                graphsage_cpu_epoch(g_ppi, par_ppi, 1500, 20, 128, 10, n_batches=2)
                secs_p, nb_p = graphsage_cpu_epoch(g_ppi, par_ppi, 1500, 20, 128, 10, n_batches=15)
                extras["dgl_style_fp32_cpu_epoch_ms_ppi"] = {"value": round(secs_p * 1e3 * 75 / nb_p, 2), "cores": torch.get_num_threads(),
                                                              "sample": f"ppi-sized graph (BASELINE.json configs[0]), {nb_p} of 75 batches, scaled x{75 / nb_p:.0f}",
                                                              "kind": "port (torch-CPU GraphSAGE-sum x3; DGL not installable)"}
    except Exception as e:   # noqa: BLE001 - an optional leg must not cost the headline line (single-rank runs; with several ranks the others
        if world > 1:        # would wait in a collective: fail the whole job)
            raise
        import traceback
        extras["failed_leg"] = {"error": repr(e), "trace_tail": traceback.format_exc().splitlines()[-6:]}
    if world > 1:
        extras["rank_checksums"] = [float(v) for v in sums.view(-1).tolist()]
    line["extras"] = extras
    if rank == 0:
        print(json.dumps(line), flush=True)
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
