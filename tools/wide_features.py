"""Grouped Cluster-GCN epoch on an ogbn-arxiv-sized graph with MORE than 128 features (reddit: 602): the chain entries with a k-quad loop in
the first X . W (run with QGTC_NO_RBW=1 for the six-launch route beside it)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import QGTC as Q
from qgtc_ppopp22_amd import driver, graph as G
for dim in (128, 300, 602, 1024):
    g = G.make_graph("ogbn-arxiv", 1500, dim=dim)
    args = driver.build_parser().parse_args(["--dataset", "ogbn-arxiv", "--n-hidden", "128", "--n-classes", "10", "--bit_width", "2", "--use_QGTC", "--quiet",
                                             "--n-epochs", "20", "--batched", "--chain", "correct"])
    it = driver.make_iter(args, Q, g)
    rs = [driver.run(args, Q=Q, graph=g, it=it) for _ in range(5)]
    ms = sorted(r["avg_epoch_ms"] for r in rs)
    print(f"F = {dim}: grouped correct chain {ms[2]:.4f} ms ({rs[-1]['plan'].n_launches} launches){' QGTC_NO_RBW' if os.environ.get('QGTC_NO_RBW') else ''}", flush=True)
