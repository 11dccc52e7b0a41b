"""Occupied k-quads per 32-row block of the cluster batches' adjacencies (what bounds the longest workgroup of an
aggregation stage: one workgroup walks one row block's occupied k-quads in pairs)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import QGTC as Q
from qgtc_ppopp22_amd import graph as G
from qgtc_ppopp22_amd.sampler import ClusterIter
for dataset, b, gin in (("ogbn-arxiv", 2, False), ("ppi", 4, True)):
    graph = G.make_graph(dataset, 1500)
    it = ClusterIter(dataset, graph, 1500, 20, bit_width=b, run_GIN=gin, device=torch.device("cuda:0"), qgtc=Q, with_rows_X=True)
    cnt = []
    for ct, (n, _, _, _) in zip(it.cTensor_li, it.cluster_param_li):
        occ = Q.tile_occupancy(ct.bit_A if hasattr(ct, "bit_A") else ct[0], n, n, 1).cpu().numpy().view(np.uint64)
        cnt += [bin(int(w)).count("1") for w in occ]
    cnt = np.array(cnt)
    print(dataset, "row blocks", cnt.size, "k-quads per batch", (n + 127) // 128, "hist", np.bincount(cnt).tolist(), "mean %.2f" % cnt.mean(), "max", cnt.max())
