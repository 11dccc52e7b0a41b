"""ClusterIter — counterpart of the reference's sampler.py::ClusterIter for the QGTC path.

What it reproduces (sampler.py:67-106): partitions are shuffled with `random` (seeded by the
driver), `psize // batch_size` batches are built from `batch_size` consecutive partitions, each
batch's dense float adjacency (row = src, col = dst, value = edge multiplicity) and features are
packed ONCE with `QGTC.val2bit(A, 1, False, False)` and `QGTC.val2bit(X, bit_width, True, False)`,
and the logical sizes `(A0, A1, X0, X1)` travel with the packed tensors.
By default ALL batches are packed by one library call straight from their edge lists
(`QGTC.EpochPlan.load` -> `qgtc_load_batches`: six launches for the iterator, word-for-word the same tensors);
`grouped=False` packs batch by batch (`QGTC.pack_edges` + `QGTC.val2bit`), `dense_adjacency=True` takes the
reference's dense float detour.

What is MI355X-first: packed batches stay resident in HBM by default (a 1213-node 2-bit batch is
230 KiB; all 75 batches of an ogbn-arxiv-sized graph are 17 MiB of 288 GB) instead of being parked
on the CPU and re-uploaded every iteration (sampler.py:104, main_qgtc.py:115). `resident=False`
restores the reference's behaviour for the non-resident variant of the epoch metric.
"""
from __future__ import annotations

import random

import numpy as np
import torch

from . import graph as G


class ClusterTensor:
    """Packed adjacency + features of one cluster batch (sampler.py:12-19)."""

    def __init__(self, bit_A: torch.Tensor, bit_X: torch.Tensor, bit_X_rows: torch.Tensor | None = None):
        self.bit_A = bit_A
        self.bit_X = bit_X
        self.bit_X_rows = bit_X_rows  # rows-layout copy of X for the layout-correct chain

    def to(self, device, non_blocking: bool = False):
        if self.bit_A.device == torch.device(device):      # resident batches: main_qgtc.py:115's `.cuda()` is a no-op
            return self
        return ClusterTensor(self.bit_A.to(device, non_blocking=non_blocking),
                             self.bit_X.to(device, non_blocking=non_blocking),
                             None if self.bit_X_rows is None else self.bit_X_rows.to(device, non_blocking=non_blocking))

    def cuda(self):
        return self.to("cuda")

    def cpu(self):
        return self.to("cpu")


class ClusterIter:
    def __init__(self, dn, g: G.Graph, psize: int, batch_size: int, bit_width: int = 2,
                 run_GIN: bool = False, device="cuda", resident: bool = True, qgtc=None,
                 batch_ids=None, with_rows_X: bool = False, dense_adjacency: bool = False,
                 keep_raw: bool = False, grouped: bool = True):
        if qgtc is None:
            import QGTC as qgtc  # the HIP extension; there is no fallback
        self.g = g
        self.psize = psize
        self.batch_size = batch_size
        self.bit_width = bit_width
        self.run_GIN = run_GIN
        self.device = torch.device(device)
        self.resident = resident
        self.par_li = G.partition_list(g, psize)
        self.max = int(psize // batch_size)       # sampler.py:67
        random.shuffle(self.par_li)               # sampler.py:68 (driver seeds `random` with 2)
        # round-robin sharding hook: only these batch ids are materialised on this rank
        self.batch_ids = list(range(self.max)) if batch_ids is None else list(batch_ids)
        self.cTensor_li, self.cluster_param_li, self.n_edges = [], [], []
        # keep_raw: the unpacked batch (edge list, float features) for drivers that pack inside the epoch
        # loop as cluster_gcn.py:151-227 does
        self.raw_li = []
        self._epoch_data = None
        self.x_in_chain_format = False
        feat = g.feat.shape[1]
        # the chain entries take one width b per chain: 1 .. 4 bits with up to 256 columns, 5 .. 8 bits with up to 128
        # (driver.chain_entries_cover): Batched-GIN's first product reads X in the chain format, every aggregation the adjacency as
        # 512-byte tiles
        self._x_chain = bit_width if (run_GIN and 1 <= bit_width <= 8 and feat <= (256 if bit_width <= 4 else 128)) else 0
        self._a_tiles = self._x_chain > 0 or (not run_GIN and 1 <= bit_width <= 8)
        self._with_rows = with_rows_X
        if grouped and not dense_adjacency and self.batch_ids:
            self._pack_grouped(qgtc, keep_raw)
            return
        for cid in self.batch_ids:
            nodes = G.batch_nodes(self.par_li, cid, psize, batch_size)
            row, col = G.induced_edges(g, nodes)
            n = nodes.size
            r_dev, c_dev = torch.from_numpy(row).to(self.device), torch.from_numpy(col).to(self.device)
            if dense_adjacency:
                # the reference's route: dense float n x n matrix, then val2bit (sampler.py:80-101);
                # torch.sparse.FloatTensor(i, v).to_dense() sums duplicate edges (sampler.py:87-89)
                A = torch.zeros((n, n), dtype=torch.float32, device=self.device)
                if row.size:
                    A.index_put_((r_dev, c_dev), torch.ones(row.size, dtype=torch.float32, device=self.device),
                                 accumulate=True)
                bit_A = qgtc.val2bit(A, 1, False, False)         # sampler.py:98/101
            else:
                # same words, straight from the edge list: the n x n floats (5.9 MB for n = 1213
                # against 190 KiB packed) are never materialised
                bit_A = qgtc.pack_edges(r_dev, c_dev, n, n, 1, False)   # (our own induced edges: no index check, no sync)
            X = torch.from_numpy(g.feat[nodes]).to(self.device)
            bit_X = qgtc.val2bit(X, bit_width, True, False)      # sampler.py:99/102
            bit_Xr = qgtc.val2bit(X, bit_width, False, False) if with_rows_X else None
            ct = ClusterTensor(bit_A, bit_X, bit_Xr)
            if not resident:
                ct = ct.cpu()                                    # sampler.py:104
            self.cTensor_li.append(ct)
            self.cluster_param_li.append((n, n, X.size(0), X.size(1)))  # sampler.py:92-95,105
            self.n_edges.append(int(row.size))
            if keep_raw:
                self.raw_li.append((r_dev, c_dev, X) if resident else (r_dev.cpu(), c_dev.cpu(), X.cpu()))

    def _pack_grouped(self, qgtc, keep_raw):
        """Every batch of the iterator packed by ONE library call (QGTC.EpochPlan.load -> qgtc_load_batches): the batches' edge
        lists (indices local to each batch) and feature rows are concatenated on the host, uploaded once, and a handful of
        grouped launches write what sampler.py:76-106 builds per batch - plus, from the same registers, the formats the grouped
        epoch reads (adjacency tiles, occupancy bitmaps, rows-layout / chain-format X). The per-batch tensors are views."""
        g = self.g
        rows, cols, feats, ns, ecounts = [], [], [], [], []
        for cid in self.batch_ids:
            nodes = G.batch_nodes(self.par_li, cid, self.psize, self.batch_size)
            row, col = G.induced_edges(g, nodes)
            rows.append(np.asarray(row, dtype=np.int64))
            cols.append(np.asarray(col, dtype=np.int64))
            feats.append(g.feat[nodes])
            ns.append(int(nodes.size))
            ecounts.append(int(row.size))
        # the concatenated raw arrays: on the host always, on the device while they are needed (the pack below; afterwards only when
        # keep_raw asks for per-batch raw views or a driver packs inside its epoch loop: pack_now uploads them again on demand)
        self._raw_host = (np.concatenate(rows), np.concatenate(cols), np.ascontiguousarray(np.concatenate(feats), dtype=np.float32))
        self.raw_nodes, self.raw_edge_counts = ns, ecounts
        self._upload_raw()
        src, dst, X = self.raw_src, self.raw_dst, self.raw_feats
        data = self.pack_now(qgtc)
        F = X.size(1)
        e0 = f0 = 0
        As, Xs, Xrs = data.As, data.Xs, (data.Xrs if self._with_rows else None)   # (views into the pools, made on this first access)
        for i, n in enumerate(ns):
            ct = ClusterTensor(As[i], Xs[i], Xrs[i] if self._with_rows else None)
            if not self.resident:
                ct = ct.cpu()                                    # sampler.py:104
            self.cTensor_li.append(ct)
            self.cluster_param_li.append((n, n, n, F))           # sampler.py:92-95,105
            self.n_edges.append(ecounts[i])
            if keep_raw:
                raw = (src[e0:e0 + ecounts[i]], dst[e0:e0 + ecounts[i]], X[f0:f0 + n])
                self.raw_li.append(raw if self.resident else tuple(t.cpu() for t in raw))
            e0 += ecounts[i]
            f0 += n
        if self.resident:
            self._epoch_data = data
            self.x_in_chain_format = self._x_chain > 0
        if not keep_raw:
            self.raw_src = self.raw_dst = self.raw_feats = None      # (46 MB of float features for the ogbn-arxiv-sized iterator)

    def _upload_raw(self):
        r, c, x = self._raw_host
        self.raw_src, self.raw_dst = torch.from_numpy(r).to(self.device), torch.from_numpy(c).to(self.device)
        self.raw_feats = torch.from_numpy(x).to(self.device)

    def pack_now(self, qgtc=None):
        """Pack every batch again from the resident raw arrays (one EpochPlan.load call): what a driver that packs INSIDE its
        epoch loop (cluster_gcn.py:151-227) pays per epoch in the grouped form. Returns the new EpochPlan."""
        if qgtc is None:
            import QGTC as qgtc
        if getattr(self, "raw_src", None) is None:
            self._upload_raw()       # (kept from here on: a driver that packs inside its epoch loop calls this every epoch)
        return qgtc.EpochPlan.load(self.raw_src, self.raw_dst, self.raw_edge_counts, self.raw_feats, self.raw_nodes, self.bit_width,
                                   self._with_rows, self._x_chain, self._a_tiles, False)

    def epoch_data(self, qgtc=None):
        """The data loader's share of a GROUPED epoch, made once beside the packing (outside the epoch clock, like the
        packing itself: main_qgtc.py:74-93): the per-batch table on the device and the adjacencies' occupancy bitmaps
        (Q.EpochPlan). The epoch then binds weights and outputs to it with one launch (driver.PlannedEpoch). The grouped
        loader (the default) has made it already; the per-batch routes build it here from their tensors."""
        if getattr(self, "_epoch_data", None) is None:
            if qgtc is None:
                import QGTC as qgtc
            assert self.resident, "a grouped epoch needs the packed batches on the device"
            cts = self.cTensor_li
            rows = [c.bit_X_rows for c in cts] if cts and cts[0].bit_X_rows is not None else []
            feat = self.cluster_param_li[0][3] if self.cluster_param_li else 0
            self._epoch_data = qgtc.EpochPlan([c.bit_A for c in cts], [c.bit_X for c in cts], rows, [p[0] for p in self.cluster_param_li], 1, True,
                                              self._x_chain, feat, self._a_tiles)
            self.x_in_chain_format = self._x_chain > 0
        return self._epoch_data

    def __len__(self):
        return len(self.cTensor_li)

    def __iter__(self):
        self.n = 0
        return self

    def __next__(self):
        if self.n < len(self.cTensor_li):
            item, param = self.cTensor_li[self.n], self.cluster_param_li[self.n]
            self.n += 1
            return item, param
        raise StopIteration
