"""Avg. epoch of the two drivers over --bit_width (the sweep of the reference's 0_7a / 0_7b scripts): the unchanged per-batch loop and
the grouped layout-correct plan, ogbn-arxiv-sized Cluster-GCN (hidden 128) and ppi-sized Batched-GIN (hidden 64).
usage: bits_sweep.py [gcn|gin|both] [bits,bits,...] [per-batch|grouped|both]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import QGTC as Q
from qgtc_ppopp22_amd import driver, graph as G

which = sys.argv[1] if len(sys.argv) > 1 else "both"
bits_li = [int(b) for b in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 2, 3, 4, 5, 6, 8]
paths = sys.argv[3] if len(sys.argv) > 3 else "both"
for gin, ds, hid in ((False, "ogbn-arxiv", 128), (True, "ppi", 64)):
    if which not in ("both", "gin" if gin else "gcn"):
        continue
    g = G.make_graph(ds, 1500)
    for bits in bits_li:
        base = ["--dataset", ds, "--n-hidden", str(hid), "--n-classes", "10", "--bit_width", str(bits), "--use_QGTC", "--quiet", "--n-epochs", "20"] + (["--run_GIN"] if gin else [])
        row = {}
        for name, extra in (("per-batch", []), ("grouped", ["--batched", "--chain", "correct"])):
            if paths not in ("both", name):
                continue
            args = driver.build_parser().parse_args(base + extra)
            it = driver.make_iter(args, Q, g)
            row[name] = sorted(driver.run(args, Q=Q, graph=g, it=it)["avg_epoch_ms"] for _ in range(4))[1]
        print(f"{'GIN' if gin else 'GCN'} {ds} {bits}-bit: " + "   ".join(f"{k} {v:.4f} ms" for k, v in row.items()), flush=True)
