"""The driver-style epoch figures of bench.py (Avg. Epoch with weight packing and plan bind inside the clock, main_qgtc.py:96-159)
and the kernel-only roofline block, for the two BASELINE epochs only - minutes faster than the whole bench."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from benchmarks import epochs as bench
import QGTC as Q

only = ("batched_correct_chain", "batched_reference_chain")
out = {}
ep, graph = bench.epoch_leg(Q, 0, 1, 0, only=only)
ep["roofline"] = bench.epoch_roofline(Q, graph, 0, "ogbn-arxiv", 2, 128, False)
out["arxiv_gcn_2bit"] = ep
ep2, g2 = bench.epoch_leg(Q, 0, 1, 0, dataset="ppi", bits=4, hidden=64, gin=True, full=False, only=only)
ep2["roofline"] = bench.epoch_roofline(Q, g2, 0, "ppi", 4, 64, True)
out["ppi_gin_4bit"] = ep2
for k, v in out.items():
    r = v["roofline"]
    print(k, {kk: vv for kk, vv in v.items() if kk != "roofline"}, "kernel_us", r["kernel_us_per_epoch"], "alone", r["kernel_us_per_operator_alone"],
          "host_ms", r["host_weight_pack_and_plan_bind_ms"], "frac", r["roofline"]["frac"])
json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gpurun_out", "epoch_driver_ms.json"), "w"), indent=1)
