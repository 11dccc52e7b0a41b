"""CPU stand-in for the reference's DGL Cluster-GCN fp32 baseline (BASELINE.json config 1) —
BASELINE INFRASTRUCTURE ONLY, used by bench.py's reporting; never by the product path.

cluster_gcn_dgl.py + modules.py:9-45 run three GraphSAGE-sum layers per cluster batch:
``h <- relu(Linear(sum over in-edges of h_src))`` (modules.py:16-24, update_all(copy_src, sum)).
DGL, OGB and the datasets cannot be installed here (no network) and the reference's script is
hard-wired to .cuda() (cluster_gcn_dgl.py:82-100), so the same forward is restated with torch CPU
sparse-mm + nn.Linear on the host cores. The arithmetic lives in third-party DGL in the reference
and no reference test pins its outputs: PARITY UNPINNED for this path (definition only).
"""
from __future__ import annotations

import time

import numpy as np
import torch


def graphsage_cpu_epoch(graph, par_li, psize, batch_size, n_hidden, n_classes, n_batches=None,
                        threads=None, seed=3):
    """One forward epoch over `n_batches` cluster batches on CPU. Returns (seconds, batches run)."""
    from qgtc_ppopp22_amd import graph as G

    if threads:
        torch.set_num_threads(threads)
    torch.manual_seed(seed)
    feat = graph.feat.shape[1]
    lin = [torch.nn.Linear(feat, n_hidden), torch.nn.Linear(n_hidden, n_hidden),
           torch.nn.Linear(n_hidden, n_classes)]
    total = psize // batch_size
    n_batches = total if n_batches is None else min(n_batches, total)
    batches = []
    for cid in range(n_batches):
        nodes = G.batch_nodes(par_li, cid, psize, batch_size)
        row, col = G.induced_edges(graph, nodes)
        n = nodes.size
        # h_dst = sum_{src->dst} h_src  ==  A^T @ h  with A[src, dst] = 1
        At = torch.sparse_coo_tensor(np.vstack((col, row)), torch.ones(row.size), (n, n)).coalesce()
        batches.append((At, torch.from_numpy(graph.feat[nodes])))
    t0 = time.perf_counter()
    with torch.no_grad():
        for At, h in batches:
            for layer in lin:
                h = torch.relu(layer(torch.sparse.mm(At, h)))
    return time.perf_counter() - t0, n_batches
