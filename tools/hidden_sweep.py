"""Avg. epoch of the grouped layout-correct Cluster-GCN plan (ogbn-arxiv-sized) over --n-hidden at a few widths: the chain entries up
to 256 hidden units (bitmm_fp4_rbx.hip.h beyond 128).  usage: hidden_sweep.py [bits,bits] [hidden,hidden,...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import QGTC as Q
from qgtc_ppopp22_amd import driver, graph as G

bits_li = [int(b) for b in sys.argv[1].split(",")] if len(sys.argv) > 1 else [2, 4]
hid_li = [int(b) for b in sys.argv[2].split(",")] if len(sys.argv) > 2 else [16, 64, 128, 160, 256]
g = G.make_graph("ogbn-arxiv", 1500)
for bits in bits_li:
    for hid in hid_li:
        args = driver.build_parser().parse_args(["--dataset", "ogbn-arxiv", "--n-hidden", str(hid), "--n-classes", "10", "--bit_width", str(bits), "--use_QGTC",
                                                 "--quiet", "--n-epochs", "20", "--batched", "--chain", "correct"])
        it = driver.make_iter(args, Q, g)
        ms, launches = [], 0
        for _ in range(5):     # (each run's plan is dropped before the next: a run that has to hipMalloc its pools reads 1 ms more)
            r = driver.run(args, Q=Q, graph=g, it=it)
            ms.append(r["avg_epoch_ms"])
            launches = r["plan"].n_launches
            del r
        print(f"GCN ogbn-arxiv {bits}-bit hidden {hid}: grouped {sorted(ms)[1]:.4f} ms (all: {' '.join('%.4f' % m for m in ms)}), {launches} launches", flush=True)
