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

# name -> (train nodes, avg out-degree, feature dim); public dataset facts quoted in SURVEY.md §8
PRESETS = {
    "ogbn-arxiv": (90941, 7.0, 128),
    "ppi": (44906, 28.0, 50),
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


def load_npz_graph(path: str, dim: int, psize: int, seed: int = 2) -> Graph:
    """The reference's `.npz` format: arrays `src_li`, `dst_li` (dataset.py:48-53); features are
    random as in dataset.py:64. Partitions are contiguous node ranges (no METIS here)."""
    obj = np.load(path)
    src = np.asarray(obj["src_li"], dtype=np.int64)
    dst = np.asarray(obj["dst_li"], dtype=np.int64)
    n = int(max(src.max(), dst.max())) + 1
    rng = np.random.default_rng(seed)
    feat = rng.standard_normal((n, dim), dtype=np.float32)
    block_of = (np.arange(n, dtype=np.int64) * psize // n).astype(np.int32)
    return Graph(path, n, src, dst, feat, block_of)


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
