"""Avg. epoch of the two drivers over --bit_width (the sweep of the reference's 0_7a / 0_7b scripts): the unchanged per-batch loop and
the grouped layout-correct plan, ogbn-arxiv-sized Cluster-GCN (hidden 128) and ppi-sized Batched-GIN (hidden 64)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import QGTC as Q
from qgtc_ppopp22_amd import driver, graph as G

for gin, ds, hid in ((False, "ogbn-arxiv", 128), (True, "ppi", 64)):
    g = G.make_graph(ds, 1500)
    for bits in (1, 2, 3, 4, 5, 6, 8):
        base = ["--dataset", ds, "--n-hidden", str(hid), "--n-classes", "10", "--bit_width", str(bits), "--use_QGTC", "--quiet", "--n-epochs", "20"] + (["--run_GIN"] if gin else [])
        row = []
        for extra in ([], ["--batched", "--chain", "correct"]):
            args = driver.build_parser().parse_args(base + extra)
            it = driver.make_iter(args, Q, g)
            ms = sorted(driver.run(args, Q=Q, graph=g, it=it)["avg_epoch_ms"] for _ in range(4))[1]
            row.append(ms)
        print(f"{'GIN' if gin else 'GCN'} {ds} {bits}-bit: per-batch {row[0]:.3f} ms   grouped correct chain {row[1]:.4f} ms", flush=True)
