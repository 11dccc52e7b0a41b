#!/usr/bin/env python3
"""bench.py - headline benchmark of the QGTC bit-GEMM hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is ONE bitMM2Bit launch over the workload BASELINE.json's north star and BASELINE.md §1's bold row name: a 1-bit
4096 x 4096 adjacency times a 1-bit 4096 x 64 feature matrix (output 1 bit), operands bit-packed and resident in HBM, issued
the reference's way (QGTC_device.cu:407-418: launches back to back on one stream). Metric = the reference's own
(QGTC_device.cu:420-422): effective tera-ops = 2 M K N per launch / time. Inputs are seeded random bits.

With N > 1 ranks every GPU runs its own copy (cluster batches are independent: no data-path collective) - weak scaling,
value = all ranks' ops / max time; RCCL only takes the max time and gathers per-rank checksums / per-batch outputs.

Rank 0 prints ONE short JSON line LAST on stdout: the contract's fields, `roofline` (dominant kernel + both epochs of
BASELINE.json's metric as flat epoch_gcn_* / epoch_gin_* scalars), `cpu_baseline`. Tables, notes and every other leg go to
the file the line names (`extras_file`). The legs live in benchmarks/ (headline, cpu, epochs, tables, launcher)."""
from __future__ import annotations

import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

from benchmarks.common import REF_TFLOPS_4096_64, engine, make_workload
from benchmarks.epochs import batch_summaries, epoch_leg  # noqa: F401  (tests/shard_worker.py drives the sharded legs through these)

LINE_LIMIT = 4096       # bytes of the printed line (tests/test_host_logic.py holds bench.py to it)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=1000)
    p.add_argument("--warmup", type=int, default=50)
    p.add_argument("--bits", type=int, default=1, help="feature bit width w of the headline workload")
    p.add_argument("--no-extras", action="store_true", help="headline only: skip the epoch / table / CPU legs")
    p.add_argument("--no-tables", action="store_true", help="skip the reference's benchmark tables (extras file only)")
    p.add_argument("--streams", type=int, default=1,
                   help="issue the steps round-robin on this many HIP streams (default 1 = the reference's metric)")
    p.add_argument("--issue", choices=["eager", "graph"], default="eager",
                   help="eager (default, the reference's loop): K hipLaunchKernel calls inside the timed region; graph: the K launches "
                        "captured once ahead of it and replayed by one hipGraphLaunch (reported in the extras file either way)")
    p.add_argument("--cpu-seconds", type=float, default=10.0, help="CPU-baseline budget")
    p.add_argument("--backend", choices=["nccl", "gloo"], default=None, help="torch.distributed backend (default: nccl = RCCL)")
    p.add_argument("--dry-run", action="store_true", help="exercise the rank launcher and the collectives only (no GPU)")
    p.add_argument("--gather", choices=["summaries", "outputs"], default="summaries",
                   help="what the end-of-epoch exchange of the sharded epoch legs moves over RCCL (SURVEY.md 8e)")
    p.add_argument("--engine", choices=["popcount", "mfma", "auto"], default="auto",
                   help="engine of the headline launches (same words either way); the other one's figure goes to the extras file")
    p.add_argument("--extras-file", default=None, help="where the tables / notes go (default gpurun_out/bench_extras.json)")
    return p.parse_args()


def compose_line(args, world, ranks_seen, M, K, N, w, value, wall_max, fp4_kernel, roofline):
    """The contract's fields. Strings are short on purpose: the line is machine-read."""
    return {"metric": "effective bit-GEMM TOPS (2*M*K*N/t), 1-bit A x %d-bit X, M=K=4096, N=64" % w,
            "value": round(value, 3), "unit": "TOPS", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(wall_max * 1e3 / args.steps, 6), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": round(value / REF_TFLOPS_4096_64[w], 3) if w in REF_TFLOPS_4096_64 else None,
            "dtype": "fp4 (E2M1 codes of bit planes on MFMA, exact f32 sums)" if fp4_kernel else "u32 (AND+popcount into int32)",
            "data": "synthetic",
            "config": {"workload": f"bitMM2Bit {M}x{K}x{N} a=1 w={w} ob={w} (BASELINE.json configs[1], 2_7c shape)",
                       "inputs": "seeded Bernoulli(0.5) adjacency, uniform w-bit features, packed, resident in HBM",
                       "parallelism": f"replica-per-GPU x{world}, no data-path collective", "engine": args.engine,
                       "issue": args.issue if args.streams <= 1 else f"eager on {args.streams} streams"},
            "ranks_seen": ranks_seen, "roofline": roofline}


def write_extras(path, extras):
    """The tables / notes file (gpurun_out/ is merged back from the GPU box). Returns the path written, or None."""
    for cand in (path, os.path.join(ROOT, "gpurun_out", "bench_extras.json"), "/tmp/qgtc_bench_extras.json"):
        if not cand:
            continue
        try:
            os.makedirs(os.path.dirname(os.path.abspath(cand)), exist_ok=True)
            with open(cand, "w") as f:
                json.dump(extras, f, indent=1)
            return os.path.relpath(cand, ROOT) if os.path.abspath(cand).startswith(ROOT) else cand
        except OSError:
            continue
    return None


def shrink(line):
    """Never print more than LINE_LIMIT bytes: drop the least important optional keys until the line fits."""
    for key in ("rccl_world1", "parity_vs_oracle", "cpu_baseline.sample", "cpu_baseline.dgl_sample", "roofline.rocprof",
                "roofline.traffic_source", "config.inputs"):
        if len(json.dumps(line)) <= LINE_LIMIT:
            break
        top, _, sub = key.partition(".")
        if sub:
            (line.get(top) or {}).pop(sub, None)
        else:
            line.pop(top, None)
    return line


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        from benchmarks.launcher import launch_ranks
        sys.exit(launch_ranks(args, os.path.abspath(__file__)))           # (nothing above this line touches the GPU)
    if args.dry_run:
        from benchmarks.launcher import dry_run
        return dry_run(args)
    from benchmarks import cpu, epochs, headline, tables
    from benchmarks.common import flush_c_stdio
    from qgtc_ppopp22_amd import dist as D

    rank, world, local = D.init_from_env(backend=args.backend)
    solo = rank == 0 and world == 1 and not args.no_extras
    host_kernarg_ms = rccl = None
    if solo and torch.cuda.device_count() > 0:
        # two child processes, started before THIS process touches the GPU: a process-wide HIP runtime knob, and RCCL at world 1
        host_kernarg_ms = epochs.unchanged_driver_in_a_child({"HIP_FORCE_DEV_KERNARG": "0"})
        rccl = rccl_world1()
    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP path has no fallback)"
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: one rank per GPU (run `python bench.py --gpus N` " \
                               "or torch.distributed.run with --nproc-per-node N)"
    if os.environ.get("QGTC_BENCH_SHARE_GPU", "0") not in ("", "0"):
        local = 0                              # test hook (tests/test_aa_bench_two_ranks.py): every rank on cuda:0, collectives over gloo
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)
    import QGTC as Q

    M = K = 4096
    N, w = 64, args.bits
    A, X, bit_A, bit_X = make_workload(Q, M, K, N, w, device, seed=3 + rank)
    with engine(Q, "popcount"):
        out = Q.bitMM2Bit(bit_A, bit_X, M, K, N, 1, w, w)   # AND + popcount kernels: the words every engine must produce
    extras, pop_block = {}, None
    with engine(Q, args.engine):
        out_e = Q.bitMM2Bit(bit_A, bit_X, M, K, N, 1, w, w)
        assert torch.equal(out_e, out), "engines disagree"
        # ================= the timed region (benchmarks/headline.py::time_steps) =================
        wall, kern = headline.time_steps(Q, out_e, bit_A, bit_X, M, K, N, w, args.steps, args.warmup, D.barrier, args.streams, args.issue)
        assert torch.equal(out_e, out), "the timed launches changed the result"
        if solo and args.streams <= 1:   # the same K steps issued the other way, and on the other engine (identical words)
            other = "graph" if args.issue == "eager" else "eager"
            o_wall, o_kern = headline.time_steps(Q, out_e, bit_A, bit_X, M, K, N, w, args.steps, args.warmup, D.barrier, 1, other)
            extras["headline_issue_" + other] = {"TOPS": round(args.steps * 2.0 * M * K * N / o_wall / 1e12, 3), "us_per_launch": round(min(o_kern.values()) * 1e6, 3)}
    if solo:
        other_engine = "popcount" if args.engine != "popcount" else "auto"
        with engine(Q, other_engine):
            o_wall, o_kern = headline.time_steps(Q, torch.empty_like(out), bit_A, bit_X, M, K, N, w, args.steps, args.warmup, D.barrier, args.streams, args.issue)
        extras["headline_on_engine_" + other_engine] = headline.other_engine_block(M, K, N, w, o_wall, o_kern, args.steps, other_engine == "popcount")
        pop_block = extras["headline_on_engine_" + other_engine] if other_engine == "popcount" else None
    # which kernel runs at this shape: the FP4 matrix-core kernel when the float32 sums stay exact (K (2^a - 1)(2^w - 1) < 2^24)
    fp4_kernel = args.engine != "popcount" and w <= 8 and K * (2 ** w - 1) < 2 ** 24
    wall_max = D.max_over_ranks(wall, device)
    value = world * args.steps * 2.0 * M * K * N / wall_max / 1e12
    csum = torch.tensor([[float(out.to(torch.int64).sum().item())]], dtype=torch.float64, device=device)
    sums = D.gather_batch_summaries(csum, world, rank, world)       # result checksum of every rank (the path's only collective)
    roofline = headline.roofline_block(M, K, N, w, kern, wall / args.steps, fp4_kernel, args.streams <= 1)
    line = compose_line(args, world, D.world_size(), M, K, N, w, value, wall_max, fp4_kernel, roofline)
    if solo and pop_block is not None:
        # the mechanism BASELINE.json's north star names (AND + v_bcnt, engine "popcount") on the same workload, beside the default engine's
        # figure: microseconds per launch and its fraction of SURVEY 8d's VALU roofline / of the pair rate tools/valu_peak.hip measures
        line["roofline"].update({"popcount_us": pop_block["us_per_launch"], "popcount_TOPS": pop_block["TOPS"],
                                 "popcount_valu_frac": pop_block["valu_frac_of_survey_8d_peak_7.864e13"],
                                 "popcount_valu_frac_of_measured_pair_rate": pop_block["valu_frac_of_measured_pair_rate_4.2e13"]})

    try:
        if solo:
            cb, ref = cpu.oracle_bitgemm(M, K, N, w, A, X, args.cpu_seconds)
            line["cpu_baseline"] = cb
            line["parity_vs_oracle"] = bool((out.cpu().numpy().view(np.uint32).reshape(-1) == ref).all())
            if rccl is not None:
                line["rccl_world1"] = {k: rccl.get(k) for k in ("ok", "backend", "ranks_seen", "gather_outputs_ms")}
                extras["rccl_world1"] = rccl
            line["roofline"].update(headline.big_adjacency_scalars(Q, device))
            if not args.no_tables:
                extras.update(tables.all_tables(Q, device, M, K, N))
        if not args.no_extras:
            multi = ("per_batch_reference_chain", "batched_correct_chain") if world > 1 else None
            # BASELINE.json configs[2]: Cluster-GCN, ogbn-arxiv-sized, 2 bits, hidden 128 (configs[4] when sharded: strong + weak)
            ep, graph = epoch_leg(Q, rank, world, local, only=multi, gather=args.gather)
            extras["cluster_gcn_epoch_ogbn_arxiv_shape"] = ep
            if world > 1:
                ep["scaling"] = "strong: 75 batches round-robin over the ranks; epoch = max over ranks"
                ep_w, _ = epoch_leg(Q, rank, world, local, only=("batched_correct_chain",), weak=True, gather=args.gather)
                ep_w["scaling"] = "weak: 75 batches per rank (own graph each), %d in all; epoch = max over ranks" % (75 * world)
                ep_w["batches_per_second"] = round(75 * world / (ep_w["batched_correct_chain_ms"] * 1e-3), 1)
                extras["cluster_gcn_epoch_ogbn_arxiv_shape_weak_scaled"] = ep_w
            # BASELINE.json configs[3]: Batched-GIN, ppi-sized, 4 bits, hidden 64 (0_7b's value)
            ep_gin, _ = epoch_leg(Q, rank, world, local, dataset="ppi", bits=4, hidden=64, gin=True, full=False, only=multi, gather=args.gather)
            extras["batched_gin_epoch_ppi_shape_4bit"] = ep_gin
        if solo:
            from qgtc_ppopp22_amd import graph as G

            ep["per_batch_reference_chain_host_kernarg_ms"] = host_kernarg_ms
            ep["roofline_of_the_grouped_correct_chain"] = epochs.epoch_roofline(Q, graph, local, "ogbn-arxiv", 2, 128, False)
            g_ppi = G.make_graph("ppi", 1500)
            ep_gin["roofline_of_the_grouped_correct_chain"] = epochs.epoch_roofline(Q, g_ppi, local, "ppi", 4, 64, True)
            line["roofline"].update(epochs.flat_epoch_scalars("gcn", ep))
            line["roofline"].update(epochs.flat_epoch_scalars("gin", ep_gin))
            extras["zero_tile_jumping"] = epochs.zero_tile_rows(Q, graph, local)
            extras["readme_cluster_gcn_table_hidden16_2bit_ms"] = epochs.readme_table(Q, rank, world, local)
            extras["checked_in_script_bitwidth32_hidden16"] = epochs.checked_in_script_settings(Q, rank, world, local)
            # the DGL-style fp32 CPU epoch "in the same run next to the throughput" (north star): inside cpu_baseline
            dgl = cpu.dgl_style_epoch(graph, "ogbn-arxiv")
            line["cpu_baseline"].update({"dgl_style_fp32_epoch_ms": dgl["ms"], "dgl_cores": dgl["cores"], "dgl_sample": dgl["sample"]})
            extras["dgl_style_fp32_cpu_epoch_ms_ppi"] = cpu.dgl_style_epoch(g_ppi, "ppi")     # BASELINE.json configs[0] names ppi
    except Exception as e:   # noqa: BLE001 - an optional leg must not cost the headline line (single rank; with several ranks the
        if world > 1:        # others would wait in a collective: fail the whole job)
            raise
        import traceback
        extras["failed_leg"] = {"error": repr(e), "trace_tail": traceback.format_exc().splitlines()[-6:]}
        line["failed_leg"] = repr(e)[:200]
    if world > 1:
        extras["rank_checksums"] = [float(v) for v in sums.view(-1).tolist()]
    if rank == 0:
        extras["line"] = dict(line)
        line["extras_file"] = write_extras(args.extras_file, extras)
        flush_c_stdio()                            # (nothing a C printf buffered may surface behind the line)
        print(json.dumps(shrink(line)), flush=True)
    D.shutdown()


def rccl_world1(timeout=180):
    """benchmarks/rccl_check.py in a child process: RCCL (backend nccl) at world size 1 with device tensors through every exchange
    of dist.py. Returns its result dict, {"ok": False, "error": ..} on failure, never raises."""
    import subprocess

    try:
        out = subprocess.run([sys.executable, os.path.join(ROOT, "benchmarks", "rccl_check.py"), "nccl"], capture_output=True, text=True, timeout=timeout)
        for ln in out.stdout.splitlines():
            if ln.startswith("RCCL_WORLD1 "):
                return json.loads(ln[len("RCCL_WORLD1 "):])
        return {"ok": False, "error": (out.stderr or "no result line")[-300:]}
    except Exception as e:   # noqa: BLE001 - optional leg
        return {"ok": False, "error": repr(e)[:300]}


if __name__ == "__main__":
    main()
