"""The contract's timed region and the dominant kernel's roofline block.

A step = ONE bitMM2Bit launch on 4096 x 4096 x 64 (BASELINE.md §1's bold row), operands packed and resident in HBM. The
K steps are issued the reference's way (QGTC_device.cu:407-418): K launches back to back on one stream."""
from __future__ import annotations

import time

import torch

from .common import (CLOCK_WARMUP_S, EVENT_MIN_LAUNCHES, FP4_PEAK_TFLOPS, HBM_PEAK_GBS, PROFILE_DIR, VALU_PEAK_BITOPS,
                     VALU_PEAK_BITOPS_SURVEY, profile_summary)


def time_steps(Q, out, bit_A, bit_X, M, K, N, w, steps, warmup, barrier, streams=1, issue="eager"):
    """`warmup` untimed launches, then EXACTLY `steps` launches between barrier + synchronize pairs (wall seconds), then
    max(3, steps // 200) windows of 200 of the same launches, each between its own pair of HIP events on the launch
    stream (their MEDIAN = the kernel's average launch duration: kernel + dependent-launch gap) - issued like the timed region
    and, for the eager issue, once more replayed from a graph captured behind it. The events stay OUT of the timed region:
    recording two of them costs 11-12 us of a 72 us region.
    issue: "eager" (default, the reference's loop) = K hipLaunchKernel calls inside the region; "graph" = the K launches
    captured once AHEAD of the region and replayed by one hipGraphLaunch inside it (same kernels in stream order).
    Returns (wall seconds, {how the window was issued: seconds per launch})."""
    if streams > 1:
        outs = [out] + [torch.empty_like(out) for _ in range(streams - 1)]
        enqueue = lambda n: Q.bitMM2Bit_enqueue_streams(outs, bit_A, bit_X, M, K, N, 1, w, w, n)  # noqa: E731
        issue = "eager"
    else:
        enqueue = lambda n: Q.bitMM2Bit_enqueue(out, bit_A, bit_X, M, K, N, 1, w, w, n)  # noqa: E731
    t_w = time.perf_counter()
    while time.perf_counter() - t_w < CLOCK_WARMUP_S:      # keep the chip's clock up (common.keep_clock_up has the numbers)
        enqueue(200)
        torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    ev1.record()           # (the first record of a torch event allocates it)
    run_steps, run_window = (lambda: enqueue(steps)), (lambda: enqueue(EVENT_MIN_LAUNCHES))
    if issue == "graph":
        graphs = []
        for n in (steps, EVENT_MIN_LAUNCHES):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                enqueue(n)
            g.replay()             # (the first replay uploads the graph: untimed)
            graphs.append(g)
        torch.cuda.synchronize()
        run_steps, run_window = graphs[0].replay, graphs[1].replay
    enqueue(max(warmup, 1))
    torch.cuda.synchronize()
    # ---- the timed region of the contract: nothing but the K steps between the two fences ----
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_steps()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    barrier()
    # ---- the roofline's live launch duration: event-bracketed windows right behind it ----
    def windows(run):
        per_window = []
        for _ in range(max(3, steps // EVENT_MIN_LAUNCHES)):
            ev0.record()
            run()
            ev1.record()
            torch.cuda.synchronize()
            per_window.append(ev0.elapsed_time(ev1) * 1e-3 / EVENT_MIN_LAUNCHES)
        return sorted(per_window)[len(per_window) // 2]

    per_launch = {"hip_events_" + issue: windows(run_window)}
    if issue == "eager" and streams <= 1:
        # The same 200 launches replayed from a hipGraph captured HERE, behind the timed region (a capture ahead of it slowed the eager
        # launches on one box): no host in the loop. On a box whose host issues a launch in 3.6 us the eager windows read 3.6 us per
        # launch for a kernel that takes 3.0 (r05: 3.63 eager, 3.07 replayed) - the host's figure, not the kernel's. Both are upper
        # bounds of the kernel's average launch duration; the roofline takes the smaller.
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            enqueue(EVENT_MIN_LAUNCHES)
        g.replay()
        torch.cuda.synchronize()
        per_launch["hip_events_graph_replay"] = windows(g.replay)
    return t1 - t0, per_launch


def algorithmic_bytes(M, K, N, w):
    """a M K / 8 + w K N / 8 + ob M N / 8 with a = 1, ob = w - unpadded logical sizes."""
    return 1 * M * K / 8 + w * K * N / 8 + w * M * N / 8


def roofline_block(M, K, N, w, per_launch, wall_per_step_s, fp4_kernel, single_stream=True):
    """`roofline` of the line for the dominant kernel. `achieved` = algorithmic bytes per launch / the kernel's average launch
    duration. Every figure we have is an UPPER bound of that duration - HIP-event windows of 200 launches issued eagerly (host in
    the loop) or replayed from a graph, and the timed region's own wall clock per step (its launches run back to back on one
    stream) - so the smallest is used and `avg_launch_from` names it; all of them are in `avg_launch_candidates_us`."""
    algo = algorithmic_bytes(M, K, N, w)
    cands = dict(per_launch)
    if single_stream:
        cands["wall_per_step"] = wall_per_step_s
    src = min(cands, key=cands.get)
    kern = cands[src]
    eff_ops = 2.0 * M * K * N
    frac = round(algo / kern / 1e9 / HBM_PEAK_GBS, 5)
    rf = {"bound": "hbm", "kernel": None, "achieved": round(algo / kern / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
          "frac": frac, "traffic": None, "traffic_source": "none: only the 1-bit workload on the default engine is profiled",
          "algorithmic_bytes_per_launch": int(algo), "avg_launch_us": round(kern * 1e6, 3), "avg_launch_from": src,
          "avg_launch_candidates_us": {k: round(v * 1e6, 3) for k, v in cands.items()}, "rocprof": None}
    if fp4_kernel:
        # k_bitmm_fp4_one: v_mfma_scale_f32_16x16x128_f8f6f4 on E2M1 codes of the bit planes. The larger of its two floors is the
        # HBM one (2.16 MB / 8 TB/s = 0.27 us against 2.1 Gop / 10 PF = 0.21 us), so that is the bound named.
        rf["kernel"] = "k_bitmm_fp4_one<1,%d,0,2,2>" % ({1: 1, 2: 2}.get(w, 4 if w <= 4 else 8))
        rf["frac_mfma"] = round(eff_ops / kern / 1e12 / FP4_PEAK_TFLOPS, 5)
        if w == 1:
            prof, rf["traffic_source"] = profile_summary("headline")
            try:
                if prof is not None:
                    ks = [k for k in prof["kernel_stats"] if "k_bitmm_fp4_one" in k["name"]][0]
                    rf["rocprof"] = {"file": f"{PROFILE_DIR}/kernel_stats_headline.csv", "avg_us": round(ks["avg_ns"] / 1e3, 3),
                                     "min_us": round(ks["min_ns"] / 1e3, 3)}
                    pm = [v for k, v in prof["pmc_per_dispatch_mean"].items() if "k_bitmm_fp4_one" in k][0]
                    # HBM-side bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (KiB): the gfx950 correction for 16-byte-per-lane reads
                    rf["traffic"] = int(2 * 1024 * pm["FETCH_SIZE"]["mean"] + 1024 * pm["WRITE_SIZE"]["mean"])
            except (KeyError, ValueError, IndexError):
                rf["traffic"], rf["rocprof"], rf["traffic_source"] = None, None, "none: summary unreadable"
    else:
        rf["kernel"] = "k_bitmm<%d,1,%d,ZS>" % ({1: 4, 2: 4, 4: 2, 8: 1}.get(w, 1), w)
        rf["frac_valu_measured_pair_rate"] = round(eff_ops * w / kern / VALU_PEAK_BITOPS, 4)
        rf["frac_valu_survey_8d_peak"] = round(eff_ops * w / kern / VALU_PEAK_BITOPS_SURVEY, 4)
    return rf


def big_adjacency_scalars(Q, device):
    """Flat scalars for the line's `roofline`: the throughput-bound half of the reference's adjacency-size study
    (5_9_adjmatrix_size.py:15-18: M = K = 32768, 1 bit; QGTC_module/logs/profile_new.log:26) - the one place of this path where the
    HBM roofline binds: the packed adjacency alone is 128 MiB. 50 launches between two HIP events, median of three windows, seeded
    random bits, default engine (k_bitmm_fp4_stream up to 256 columns, k_bitmm_fp4_wide at 1024)."""
    out = {}
    g = torch.Generator(device=device).manual_seed(3)
    M = K = 32768
    A = (torch.rand((M, K), generator=g, device=device) < 0.5).float()
    ba = Q.val2bit(A, 1, False, False)
    del A
    for N in (16, 64, 1024):
        X = (torch.rand((K, N), generator=g, device=device) < 0.5).float()
        bx = Q.val2bit(X, 1, True, False)
        Q.profile(ba, bx, M, K, N, 1, 1, 1, 5)
        us = sorted(Q.profile(ba, bx, M, K, N, 1, 1, 1, 50) for _ in range(3))[1] * 1e3 / 50
        algo = M * K / 8 + K * N / 8 + M * N / 8
        out[f"adj32768_n{N}_us"] = round(us, 2)
        out[f"adj32768_n{N}_hbm_frac"] = round(algo / (us * 1e-6) / (HBM_PEAK_GBS * 1e9), 4)
        out[f"adj32768_n{N}_fp4_frac"] = round(2.0 * M * K * N / (us * 1e-6) / (FP4_PEAK_TFLOPS * 1e12), 4)
    # HBM-side bytes of the 64-column launch from the committed counters (2 x FETCH_SIZE + WRITE_SIZE KiB, as for the headline), when they
    # were collected from these kernel sources
    prof, _src = profile_summary("big")
    try:
        if prof is not None:
            pm = [v for k, v in prof["pmc_per_dispatch_mean"].items() if "k_bitmm_fp4_stream" in k][0]
            out["adj32768_n64_traffic"] = int(2 * 1024 * pm["FETCH_SIZE"]["mean"] + 1024 * pm["WRITE_SIZE"]["mean"])
            out["adj32768_n64_algo_bytes"] = int(M * K / 8 + K * 64 / 8 + M * 64 / 8)
    except (KeyError, ValueError, IndexError):
        pass
    return out


def other_engine_block(M, K, N, w, wall_s, kern_s, steps, popcount):
    """The same K steps on the other engine (identical words): the AND + popcount kernels against BOTH statements of their VALU
    roofline when that is the other engine."""
    eff_ops = 2.0 * M * K * N
    kern_s = min(kern_s.values()) if isinstance(kern_s, dict) else kern_s
    blk = {"TOPS": round(steps * eff_ops / wall_s / 1e12, 3), "us_per_launch": round(kern_s * 1e6, 3)}
    if popcount:
        blk["valu_frac_of_measured_pair_rate_4.2e13"] = round(eff_ops * w / kern_s / VALU_PEAK_BITOPS, 4)
        blk["valu_frac_of_survey_8d_peak_7.864e13"] = round(eff_ops * w / kern_s / VALU_PEAK_BITOPS_SURVEY, 4)
        blk["rocprof"] = f"{PROFILE_DIR}/summary_popcount.json"
    return blk
