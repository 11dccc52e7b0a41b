"""The reference's one published epoch table (README.md:84-89: Cluster-GCN, hidden 16, 2 bits, psize 1500, batch 20) on synthetic graphs of
the four datasets' sizes: the unchanged per-batch loop and the grouped plan. usage: readme_table.py [dataset ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import QGTC as Q
from qgtc_ppopp22_amd import driver, graph as G

ROWS = {"artist": 12, "soc-BlogCatalog": 39, "ppi": 10, "ogbn-arxiv": 10}
for ds in (sys.argv[1:] or list(ROWS)):
    g = G.make_graph(ds, 1500)
    base = ["--dataset", ds, "--n-hidden", "16", "--n-classes", str(ROWS[ds]), "--bit_width", "2", "--use_QGTC", "--quiet", "--n-epochs", "20"]
    row = {}
    for name, extra in (("per-batch", []), ("grouped", ["--batched", "--chain", "correct"])):
        args = driver.build_parser().parse_args(base + extra)
        it = driver.make_iter(args, Q, g)
        ms = sorted(driver.run(args, Q=Q, graph=g, it=it)["avg_epoch_ms"] for _ in range(5))
        row[name] = (ms[2], ms[0], ms[-1])
    print(ds, {k: tuple(round(x, 4) for x in v) for k, v in row.items()}, flush=True)
