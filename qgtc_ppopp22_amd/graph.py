"""Graphs and cluster partitions for the Cluster-GCN / Batched-GIN drivers.

The reference loads DGL/OGB datasets and partitions them with METIS (main_qgtc.py:52-72,
partition_utils.py:11-18); neither DGL, OGB, METIS nor the datasets exist in this image (no
network), so the drivers run on a deterministic synthetic stand-in with the same *shape*:
a stochastic-block-model graph whose planted blocks play the role of the METIS partitions
(dense inside a block, sparse across), and N(0,1) features (dataset.py:64 uses torch.randn too).
`load_npz_graph` reads the reference's own `.npz` edge-list format (dataset.py:48-53) for users
who have the files.
"""
from __future__ import annotations

import dataclasses

import numpy as np

# name -> (train nodes, avg out-degree, feature dim); public dataset facts quoted in SURVEY.md §8. The other names are the datasets the
# reference's scripts run (0_7a_eval_QGTC_cluster_GCN.py:12-16,38-42; README.md:84-89 publishes epoch times for artist and
# soc-BlogCatalog): node / edge counts as published with those collections, feature widths = the scripts' own --dim values
PRESETS = {
    "ogbn-arxiv": (90941, 7.0, 128),
    "ppi": (44906, 28.0, 50),
    "artist": (50515, 32.4, 100),            # 1 638 396 edges
    "soc-BlogCatalog": (88784, 23.6, 128),   # 2 093 195 edges
    "Proteins": (43471, 3.7, 29),            # 162 088 edges
    "ogbn-products": (196615, 25.0, 100),    # the training split; ~50 neighbours a node in the full graph, half of them inside the split
    "tiny": (2000, 6.0, 32),
}


@dataclasses.dataclass
class Graph:
    name: str
    n_nodes: int
    src: np.ndarray      # int64 [E]
    dst: np.ndarray      # int64 [E]
    feat: np.ndarray     # float32 [n_nodes, F]
    block_of: np.ndarray  # int32 [n_nodes] planted block id (the partitioner's answer)


def make_sbm_graph(name: str, n_nodes: int, n_blocks: int, avg_degree: float, feat_dim: int,
                   p_in: float = 0.9, seed: int = 2) -> Graph:
    """Directed SBM: every node draws Poisson(avg_degree) out-neighbours, a fraction p_in of them
    inside its own block. No self loops, no duplicate edges (so A stays 0/1)."""
    rng = np.random.default_rng(seed)  # main_qgtc.py:46 seeds numpy with 2
    block_of = (np.arange(n_nodes, dtype=np.int64) * n_blocks // n_nodes).astype(np.int32)
    starts = np.searchsorted(block_of, np.arange(n_blocks), side="left")
    ends = np.searchsorted(block_of, np.arange(n_blocks), side="right")
    deg = rng.poisson(avg_degree, size=n_nodes)
    src = np.repeat(np.arange(n_nodes, dtype=np.int64), deg)
    inside = rng.random(src.size) < p_in
    b = block_of[src]
    lo, hi = starts[b], ends[b]
    dst_in = lo + (rng.random(src.size) * (hi - lo)).astype(np.int64)
    dst_out = rng.integers(0, n_nodes, size=src.size, dtype=np.int64)
    dst = np.where(inside, dst_in, dst_out)
    keep = src != dst
    src, dst = src[keep], dst[keep]
    key = np.unique(src * n_nodes + dst)
    src, dst = key // n_nodes, key % n_nodes
    feat = rng.standard_normal((n_nodes, feat_dim), dtype=np.float32)
    return Graph(name, n_nodes, src, dst, feat, block_of)


def make_graph(name: str, psize: int = 1500, dim: int | None = None, seed: int = 2) -> Graph:
    if name not in PRESETS:
        raise ValueError(f"unknown synthetic dataset {name!r}; choose from {sorted(PRESETS)}")
    n, deg, f = PRESETS[name]
    return make_sbm_graph(name, n, psize, deg, dim or f, seed=seed)


def locality_partition(src: np.ndarray, dst: np.ndarray, n_nodes: int, psize: int, sweeps: int = 12) -> np.ndarray:
    """METIS-free stand-in for get_partition_list's partitioner (partition_utils.py:11-18, dgl metis_partition):
    returns int32 [n_nodes] part ids in [0, psize), parts of (almost) equal size, neighbours mostly together - so
    that a cluster batch's adjacency is block-structured (which is what zero-tile skipping feeds on) instead of the
    uniform scatter that contiguous id ranges give on an arbitrary node numbering.

    1. reverse Cuthill-McKee ordering of the symmetrised graph (bandwidth reduction: neighbours get close ranks),
       cut into psize equal runs;
    2. sweeps of size-capped label propagation: a node moves to the part holding most of its neighbours when that
       part has room (cap = 1.05 x the mean size) and its own part stays above half the mean size (arrivals AND
       departures are counted within the sweep), most-gaining nodes first; stops early when fewer than 0.1 % of the
       nodes still want to move. No part ends up empty.
    On the ogbn-arxiv-sized SBM graph with shuffled node ids: 45 % of the edges inside their partition after 12
    sweeps (planted blocks: 89 %, contiguous id ranges: 0.07 %), ~7 s.
    Deterministic; O(E) per sweep with scipy.sparse."""
    import scipy.sparse as sp
    from scipy.sparse.csgraph import reverse_cuthill_mckee

    psize = max(1, min(int(psize), n_nodes))
    keep = src != dst
    s_, d_ = src[keep], dst[keep]
    A = sp.coo_matrix((np.ones(s_.size, dtype=np.float32), (s_, d_)), shape=(n_nodes, n_nodes)).tocsr()
    A = ((A + A.T) > 0).astype(np.float32).tocsr()
    order = reverse_cuthill_mckee(A, symmetric_mode=True)
    rank = np.empty(n_nodes, dtype=np.int64)
    rank[order] = np.arange(n_nodes)
    part = (rank * psize // n_nodes).astype(np.int32)
    cap = int(np.ceil(1.05 * n_nodes / psize))
    for _ in range(sweeps):
        onehot = sp.csr_matrix((np.ones(n_nodes, dtype=np.float32), (np.arange(n_nodes), part)), shape=(n_nodes, psize))
        votes = (A @ onehot).tocsr()                       # votes[v, p] = neighbours of v in part p
        best = np.asarray(votes.argmax(axis=1)).reshape(-1).astype(np.int32)
        best_cnt = np.asarray(votes.max(axis=1).todense()).reshape(-1)
        here = np.asarray(votes[np.arange(n_nodes), part]).reshape(-1)
        gain = best_cnt - here
        size = np.bincount(part, minlength=psize).astype(np.int64)
        floor = max(1, n_nodes // (2 * psize))            # parts at half the mean size keep their nodes
        movers = np.nonzero((gain > 0) & (best != part) & (size[part] > floor))[0]
        if movers.size < max(1, n_nodes // 1000):
            break
        movers = movers[np.argsort(-gain[movers], kind="stable")]
        # admit movers per target part while it has room (arrival order = gain order); departures are not credited
        # back within the sweep, which only makes the cap conservative
        tgt = best[movers]
        o = np.argsort(tgt, kind="stable")
        tgt_sorted = tgt[o]
        first = np.searchsorted(tgt_sorted, tgt_sorted, side="left")
        arrival = np.arange(tgt_sorted.size) - first      # 0, 1, 2.. within each target part
        ok = arrival < (cap - size[tgt_sorted])
        chosen = movers[o[ok]]
        # departures are capped the same way (gain order within each source part): a part gives up nodes only while it
        # stays above the floor, so no part can be emptied - or left far below half the mean - by one sweep
        src_part = part[chosen]
        o2 = np.argsort(src_part, kind="stable")
        sp_sorted = src_part[o2]
        first2 = np.searchsorted(sp_sorted, sp_sorted, side="left")
        leaving = np.arange(sp_sorted.size) - first2
        chosen = chosen[o2[leaving < (size[sp_sorted] - floor)]]
        part[chosen] = best[chosen]
    assert np.bincount(part, minlength=psize).min() > 0, "locality_partition produced an empty part"
    return part


def load_npz_graph(path: str, dim: int, psize: int, seed: int = 2, partitioner: str = "locality") -> Graph:
    """The reference's `.npz` format: arrays `src_li`, `dst_li` (dataset.py:48-53; the node count is the largest id
    + 1 as DGL's add_edges makes it, dataset.py:52-56); features are random as in dataset.py:64. Partitions come from
    `locality_partition` (no METIS here); partitioner="contiguous" keeps plain id ranges."""
    obj = np.load(path)
    src = np.asarray(obj["src_li"], dtype=np.int64).reshape(-1)
    dst = np.asarray(obj["dst_li"], dtype=np.int64).reshape(-1)
    if src.size != dst.size:
        raise ValueError(f"{path}: src_li and dst_li differ in length ({src.size} vs {dst.size})")
    if src.size and (min(src.min(), dst.min()) < 0):
        raise ValueError(f"{path}: negative node id")
    n = int(max(src.max(), dst.max())) + 1 if src.size else 0
    if n == 0:
        raise ValueError(f"{path}: empty edge list")
    rng = np.random.default_rng(seed)
    feat = rng.standard_normal((n, dim), dtype=np.float32)
    if partitioner == "contiguous":
        block_of = (np.arange(n, dtype=np.int64) * psize // n).astype(np.int32)
    elif partitioner == "locality":
        block_of = locality_partition(src, dst, n, psize)
    else:
        raise ValueError(f"unknown partitioner {partitioner!r}")
    return Graph(path, n, src, dst, feat, block_of)


def edge_locality(g: Graph) -> float:
    """Fraction of edges whose endpoints share a partition (1 - edge cut)."""
    return float(np.mean(g.block_of[g.src] == g.block_of[g.dst])) if g.src.size else 1.0


def partition_list(g: Graph, psize: int):
    """Stand-in for get_partition_list (partition_utils.py:11-18): one node-id array per partition."""
    order = np.argsort(g.block_of, kind="stable")
    bounds = np.searchsorted(g.block_of[order], np.arange(psize + 1), side="left")
    return [order[bounds[i]:bounds[i + 1]].astype(np.int64) for i in range(psize)]


def batch_nodes(par_li, cid: int, psize: int, batch_size: int) -> np.ndarray:
    """get_subgraph's node selection (partition_utils.py:20-24): `batch_size` consecutive partitions."""
    parts = [par_li[s] for s in range(cid * batch_size, (cid + 1) * batch_size) if s < psize]
    return np.concatenate(parts).reshape(-1).astype(np.int64)


def induced_edges(g: Graph, nodes: np.ndarray):
    """Edges of the node-induced subgraph, relabelled in the order of `nodes` (DGL g.subgraph)."""
    local = np.full(g.n_nodes, -1, dtype=np.int64)
    local[nodes] = np.arange(nodes.size)
    ls, ld = local[g.src], local[g.dst]
    keep = (ls >= 0) & (ld >= 0)
    return ls[keep], ld[keep]
