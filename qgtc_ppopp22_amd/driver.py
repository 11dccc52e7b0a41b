"""Cluster-GCN / Batched-GIN inference driver — counterpart of the reference's main_qgtc.py.

Same command-line flags and the same per-batch operator chains (main_qgtc.py:112-155, six QGTC
calls per cluster batch), printing the same ``Avg. Epoch: {:.3f} ms`` line (main_qgtc.py:159) that
parse_time.py consumes. DGL/OGB/METIS are not available here, so the graph is a synthetic one of the
named dataset's size (qgtc_ppopp22_amd/graph.py).

Three ways to run the same math:

* ``--chain reference`` (default): the reference's literal call sequence. Its operand layouts are
  inconsistent (a cols-layout X is fed as a left operand, rows-layout results as
  right operands); the kernels read those operands bounds-safely, so the timing-relevant work is
  the same as the reference's but the numbers are not A·X·W.
* ``--chain correct``: the layout-correct chain (X·W re-packed in the cols layout by
  ``bitMM2Bit_col`` before A·(XW), as unitest.py:100-109 does).
* ``--streams S``: the reference's launch structure (one launch per batch and operator), issued by the
  extension with batch i on HIP stream i % S: independent batches overlap on the GPU.
* ``--graph``: either chain with the reference's launch structure (6 calls per batch) captured in
  one hipGraph per epoch.
* ``--batched``: either chain, with each of the six operators issued ONCE for all cluster batches
  (grouped launch): 6 launches per epoch instead of 6 x 75 — on MI355X the ~1.5 us dependent-launch
  boundary, not the arithmetic, dominates these small products.
"""
from __future__ import annotations

import argparse
import sys
import os
import random
import time

import numpy as np
import torch

from . import graph as G
from .sampler import ClusterIter


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description="QGTC Cluster-GCN / Batched-GIN inference on MI355X")
    # dgl.data.register_data_args equivalent (main_qgtc.py:22)
    p.add_argument("--dataset", type=str, default="ogbn-arxiv",
                   help=f"synthetic stand-in of this dataset's size {sorted(G.PRESETS)} or a .npz edge list")
    # main_qgtc.py:23-41, identical names and defaults
    p.add_argument("--gpu", type=int, default=0, help="gpu")
    p.add_argument("--n-epochs", type=int, default=20, help="number of training epochs")
    p.add_argument("--batch-size", type=int, default=20, help="batch size")
    p.add_argument("--psize", type=int, default=1500, help="number of partitions")
    p.add_argument("--dim", type=int, default=10, help="input dimension of each dataset (default=10)")
    p.add_argument("--n-hidden", type=int, default=16, help="number of hidden gcn units (default=16)")
    p.add_argument("--n-classes", type=int, default=10, help="number of classes (default=10)")
    p.add_argument("--n-layers", type=int, default=1, help="number of hidden gcn layers (default=1)")
    p.add_argument("--bit_width", type=int, default=2, help="bitwidth for QGTC quantization (default=2)")
    p.add_argument("--use-pp", action="store_true", help="whether to use precomputation")
    p.add_argument("--regular", action="store_true", help="whether to use DGL")
    p.add_argument("--run_GIN", action="store_true", help="whether to run GIN model")
    p.add_argument("--use_QGTC", action="store_true", help="whether to use QGTC")
    p.add_argument("--zerotile_jump", action="store_true", help="whether to profile zero-tile jumping")
    # additions
    p.add_argument("--chain", choices=["reference", "correct"], default="reference")
    p.add_argument("--batched", action="store_true", help="one grouped launch per operator per epoch")
    p.add_argument("--no-fuse", action="store_true",
                   help="with --batched --chain correct: issue the two products of a layer as two separate calls "
                        "(default: one call of the library's layer entry per layer)")
    p.add_argument("--no-chain", action="store_true",
                   help="with --batched --chain correct: do not join an aggregation stage with the next layer's X.W stage "
                        "(default: one launch for the pair, qgtc_gcn_chain_batched)")
    p.add_argument("--streams", type=int, default=0,
                   help="one launch per batch and operator (the reference's structure), batches spread over "
                        "this many HIP streams by the extension (no Python in the loop)")
    p.add_argument("--graph", action="store_true",
                   help="capture the epoch's per-batch launches (6 x batches) in one hipGraph and replay it")
    p.add_argument("--non-resident", action="store_true",
                   help="park packed batches on the CPU and upload them every iteration (main_qgtc.py:115)")
    p.add_argument("--pack-on-the-fly", action="store_true",
                   help="cluster_gcn.py's structure (:151-227): every iteration brings the batch's edge list and "
                        "float features to the device (Trans), then packs them and runs the six operators "
                        "(Compute), with synchronisation fences around both; prints its `Trans (ms): .., Compute "
                        "(ms): ..` line too. With --batched: the grouped form - one loader call per epoch packs every "
                        "batch, then the grouped epoch")
    p.add_argument("--engine", choices=["popcount", "mfma", "auto"], default="auto",
                   help="auto (default): per launch the kernel family that measured fastest on MI355X; popcount: AND + "
                        "v_bcnt kernels only (the path BASELINE.json names); mfma: bit planes expanded on the matrix "
                        "cores wherever the plane counts allow. Same results.")
    p.add_argument("--quiet", action="store_true")
    return p


def routing_switches() -> bool:
    """One of the library's ROUTING switches is set (launch_common.hip.h: getenv_flag - set and not starting with '0'): the chain
    entries / chained pairs / row-block kernels are off, so a plan must not keep T in their private formats."""
    def on(k):
        v = os.environ.get(k, "")
        return bool(v) and v[0] != "0"
    return any(on(k) for k in ("QGTC_NO_RBW", "QGTC_NO_ROWS", "QGTC_NO_XWROWS"))


def pack_weights(Q, feat, hidden, classes, bw, device):
    """main_qgtc.py:100-110: all-ones weights, cols layout, W3 with output_layer=True. Inside the reference's epoch clock,
    so it is ONE fill and ONE pack launch here (Q.val2bit_many: the same words as four val2bit calls)."""
    ones = torch.ones(feat * hidden + hidden * hidden + hidden * classes, device=device)
    W1 = ones[:feat * hidden].view(feat, hidden)
    W2 = ones[feat * hidden:feat * hidden + hidden * hidden].view(hidden, hidden)
    W3 = ones[feat * hidden + hidden * hidden:].view(hidden, classes)
    p1, p2, p3, p3h = Q.val2bit_many([W1, W2, W3, W3], bw, [True, True, True, True], [False, False, True, False])
    return {
        "W1": p1, "W2": p2,
        "W3": p3,           # main_qgtc.py:110 (output_layer=True)
        "W3h": p3h,         # hidden-style W3 for the layout-correct GCN chain
        "feat": feat, "hidden": hidden, "classes": classes,
    }


# ---------------------------------------------------------------------------------------------
# per-batch chains (six extension calls each)
# ---------------------------------------------------------------------------------------------
def gcn_reference(Q, ct, param, W, b):
    """main_qgtc.py:147-154, literally."""
    A0, A1, X0, X1 = param
    H, C = W["hidden"], W["classes"]
    t0 = Q.bitMM2Bit(ct.bit_X, W["W1"], X0, X1, H, b, b, b)
    t1 = Q.bitMM2Bit(ct.bit_A, t0, A0, A1, H, 1, b, b)
    t2 = Q.bitMM2Bit(t1, W["W2"], A0, H, H, b, b, b)
    t3 = Q.bitMM2Bit(ct.bit_A, t2, A0, A0, H, 1, b, b)
    t4 = Q.bitMM2Bit(t3, W["W3"], A0, H, C, b, b, b)
    return Q.bitMM2Int(ct.bit_A, t4, A0, A0, H, 1, b, False)


def gin_reference(Q, ct, param, W, b):
    """main_qgtc.py:131-138, literally."""
    A0, A1, X0, X1 = param
    H, C = W["hidden"], W["classes"]
    t0 = Q.bitMM2Bit(ct.bit_A, ct.bit_X, A0, A0, X1, 1, b, b)
    t1 = Q.bitMM2Bit(t0, W["W1"], A0, X1, H, b, b, b)
    t2 = Q.bitMM2Bit(ct.bit_A, t1, A0, A0, H, 1, b, b)
    t3 = Q.bitMM2Bit(t2, W["W2"], A0, H, H, b, b, b)
    t4 = Q.bitMM2Bit(ct.bit_A, t3, A0, A0, H, 1, b, b)
    return Q.bitMM2Int(t4, W["W3"], A0, H, C, b, b, False)


def gcn_correct(Q, ct, param, W, b):
    """A·((A·((A·(X·W1))·W2))·W3) with every right operand in the cols layout."""
    n, _, _, F = param
    H, C = W["hidden"], W["classes"]
    xw = Q.bitMM2Bit_col(ct.bit_X_rows, W["W1"], n, F, H, b, b, b)
    h1 = Q.bitMM2Bit(ct.bit_A, xw, n, n, H, 1, b, b)
    hw = Q.bitMM2Bit_col(h1, W["W2"], n, H, H, b, b, b)
    h2 = Q.bitMM2Bit(ct.bit_A, hw, n, n, H, 1, b, b)
    hw3 = Q.bitMM2Bit_col(h2, W["W3h"], n, H, C, b, b, b)
    return Q.bitMM2Int(ct.bit_A, hw3, n, n, C, 1, b, True)


def gin_correct(Q, ct, param, W, b):
    """(A·((A·((A·X)·W1))·W2))·W3 with every right operand in the cols layout."""
    n, _, _, F = param
    H, C = W["hidden"], W["classes"]
    ax = Q.bitMM2Bit(ct.bit_A, ct.bit_X, n, n, F, 1, b, b)
    h1 = Q.bitMM2Bit_col(ax, W["W1"], n, F, H, b, b, b)
    a1 = Q.bitMM2Bit(ct.bit_A, h1, n, n, H, 1, b, b)
    h2 = Q.bitMM2Bit_col(a1, W["W2"], n, H, H, b, b, b)
    a2 = Q.bitMM2Bit(ct.bit_A, h2, n, n, H, 1, b, b)
    return Q.bitMM2Int(a2, W["W3"], n, H, C, b, b, False)


# parse_counter.py:31-33 - the columns of the reference's zerotile_jumping.csv
ZEROTILE_HEADER = "dataset , non-jumping , jumping , ratio(jumping/non-jumping)"


def zerotile_row(dataset, printed_global, printed_counter):
    """The row parse_counter.py:10-34 computes from a `--zerotile_jump` log: it SUMS the printed lines, and the
    printed values are cumulative (kernel.h:13-28 never resets the device counters), so the sums are of running
    totals; `ratio` = jumping / non-jumping as there. `per_epoch_*` are the plain totals of one pass (the last
    cumulative value when the counters started from zero), whose ratio is the fraction of 8-row x 128-bit tile steps
    that survive zero-tile jumping."""
    gc, c = int(sum(printed_global)), int(sum(printed_counter))
    ratio = c / gc if gc else float("nan")
    last_g = int(printed_global[-1]) if printed_global else 0
    last_c = int(printed_counter[-1]) if printed_counter else 0
    return {"dataset": dataset, "non_jumping": gc, "jumping": c, "ratio": ratio,
            "per_epoch_non_jumping": last_g, "per_epoch_jumping": last_c,
            "per_epoch_ratio": (last_c / last_g) if last_g else float("nan"),
            "line": "{} , {} , {} , {:.3f}".format(dataset, gc, c, ratio)}


CHAINS = {("reference", False): gcn_reference, ("reference", True): gin_reference,
          ("correct", False): gcn_correct, ("correct", True): gin_correct}


# ---------------------------------------------------------------------------------------------
# grouped launches: the same six operators, each issued once for all batches
# ---------------------------------------------------------------------------------------------
class BatchedEpoch:
    """Builds the six grouped GEMMs of an epoch once (outputs preallocated and chained); run()
    issues six launches."""

    def __init__(self, Q, cts, params, W, b, chain: str, run_gin: bool, fuse: bool = True,
                 chain_stages: bool = True, keep_aggregates: bool = False):
        H, C = W["hidden"], W["classes"]
        bitA = [c.bit_A for c in cts]
        bitX = [c.bit_X for c in cts]
        n = [p[0] for p in params]
        F = params[0][3]
        BGraw = Q.BatchedGemm
        occ_cache = {}

        def BG(Xs, Ws, dims_, b1, b2, ob_, mode, pad128, zero_jump=False):
            # the adjacency's occupancy bitmaps are computed by the first A-stage and shared by the rest
            if zero_jump and "A" in occ_cache:
                return BGraw(Xs, Ws, dims_, b1, b2, ob_, mode, pad128, True, occ_cache["A"])
            g = BGraw(Xs, Ws, dims_, b1, b2, ob_, mode, pad128, zero_jump)
            if zero_jump:
                occ_cache["A"] = g.occs
            return g

        def dims(k, c):  # (M=n_i, K=k or n_i, N=c)
            return [(ni, ni if k is None else k, c) for ni in n]

        if chain == "reference" and not run_gin:
            x0x1 = [(p[2], p[3], H) for p in params]
            g0 = BG(bitX, [W["W1"]], x0x1, b, b, b, 0, False)
            g1 = BG(bitA, g0.outs, dims(None, H), 1, b, b, 0, False, True)
            g2 = BG(g1.outs, [W["W2"]], dims(H, H), b, b, b, 0, False)
            g3 = BG(bitA, g2.outs, dims(None, H), 1, b, b, 0, False, True)
            g4 = BG(g3.outs, [W["W3"]], dims(H, C), b, b, b, 0, False)
            g5 = BG(bitA, g4.outs, dims(None, H), 1, b, 1, 2, False, True)
        elif chain == "reference":
            g0 = BG(bitA, bitX, dims(None, F), 1, b, b, 0, False, True)
            g1 = BG(g0.outs, [W["W1"]], dims(F, H), b, b, b, 0, False)
            g2 = BG(bitA, g1.outs, dims(None, H), 1, b, b, 0, False, True)
            g3 = BG(g2.outs, [W["W2"]], dims(H, H), b, b, b, 0, False)
            g4 = BG(bitA, g3.outs, dims(None, H), 1, b, b, 0, False, True)
            g5 = BG(g4.outs, [W["W3"]], dims(H, C), b, b, 1, 2, False)
        elif not run_gin:
            bitXr = [c.bit_X_rows for c in cts]
            g0 = BG(bitXr, [W["W1"]], dims(F, H), b, b, b, 1, False)
            g1 = BG(bitA, g0.outs, dims(None, H), 1, b, b, 0, False, True)
            g2 = BG(g1.outs, [W["W2"]], dims(H, H), b, b, b, 1, False)
            g3 = BG(bitA, g2.outs, dims(None, H), 1, b, b, 0, False, True)
            g4 = BG(g3.outs, [W["W3h"]], dims(H, C), b, b, b, 1, False)
            g5 = BG(bitA, g4.outs, dims(None, C), 1, b, 1, 2, True, True)
        else:
            g0 = BG(bitA, bitX, dims(None, F), 1, b, b, 0, False, True)
            g1 = BG(g0.outs, [W["W1"]], dims(F, H), b, b, b, 1, False)
            g2 = BG(bitA, g1.outs, dims(None, H), 1, b, b, 0, False, True)
            g3 = BG(g2.outs, [W["W2"]], dims(H, H), b, b, b, 1, False)
            g4 = BG(bitA, g3.outs, dims(None, H), 1, b, b, 0, False, True)
            g5 = BG(g4.outs, [W["W3"]], dims(H, C), b, b, 1, 2, False)
        self.stages = [g0, g1, g2, g3, g4, g5]
        self.outs = g5.outs
        # The layout-correct chains are pairs "X.W re-packed in the cols layout, then A.(XW)": each pair is one call of
        # the library's layer entry (Q.FusedLayer -> qgtc_gcn_layer_batched): 3 calls per GCN epoch (4 for GIN), each two
        # grouped launches.
        self.launches = list(self.stages)
        self.discarded = set()   # stages whose outputs are not materialised
        if fuse and chain == "correct" and chain_stages:
            # An aggregation stage and the NEXT layer's X.W stage are one call (Q.ChainedPair -> qgtc_gcn_chain_batched: two grouped
            # launches; the one-launch form of the pair is the chain entries of PlannedEpoch).
            # GCN: X.W1 | A.T1 + X.W2 | A.T2 + X.W3 | A.T3 (four calls); GIN: A.X + X.W1 | A.T1 + X.W2 | A.T2 + X.W3 (three).
            pairs = [(1, 2), (3, 4)] if not run_gin else [(0, 1), (2, 3), (4, 5)]
            first = {i: Q.ChainedPair(self.stages[i], self.stages[j], not keep_aggregates) for i, j in pairs}
            second = {j for _, j in pairs}
            self.launches = [first.get(i, g) for i, g in enumerate(self.stages) if i not in second]
        elif fuse and chain == "correct":
            pairs = [(0, 1), (2, 3), (4, 5)] if not run_gin else [(1, 2), (3, 4)]
            first = {i: Q.FusedLayer(self.stages[i], self.stages[j]) for i, j in pairs}
            second = {j for _, j in pairs}
            self.launches = [first.get(i, g) for i, g in enumerate(self.stages) if i not in second]

    def run(self):
        for g in self.launches:
            g.run()
        return self.outs

    def run_per_batch(self, n_streams: int):
        """The same six operators, launched once per batch (450 launches for 75 batches), batch i on
        stream i % n_streams."""
        for g in self.stages:
            g.run_per_problem(n_streams)
        return self.outs


def stage_recipes(Q, chain: str, run_gin: bool, F: int, H: int, C: int, b: int):
    """The six operators of an epoch (main_qgtc.py:131-154; the layout-correct forms as unitest.py:100-109 builds them) as
    qgtc_stage tuples (left, right, K, N, bit1, bit2, ob, mode, pad128, use_occ, fmt) for Q.EpochPlan.bind; weights are
    [W1, W2, W3, W3h]."""
    A, X, XR, N_ = Q.SRC_A, Q.SRC_X, Q.SRC_XR, Q.DIM_NODES
    W = lambda k: Q.SRC_WEIGHT + k          # noqa: E731
    S = lambda j: Q.SRC_STAGE + j           # noqa: E731
    return [t + (0,) for t in _stage_recipes(Q, chain, run_gin, F, H, C, b)]


def _stage_recipes(Q, chain, run_gin, F, H, C, b):
    A, X, XR, N_ = Q.SRC_A, Q.SRC_X, Q.SRC_XR, Q.DIM_NODES
    W = lambda k: Q.SRC_WEIGHT + k          # noqa: E731
    S = lambda j: Q.SRC_STAGE + j           # noqa: E731
    if chain == "reference" and not run_gin:
        return [(X, W(0), F, H, b, b, b, 0, 0, 0), (A, S(0), N_, H, 1, b, b, 0, 0, 1), (S(1), W(1), H, H, b, b, b, 0, 0, 0),
                (A, S(2), N_, H, 1, b, b, 0, 0, 1), (S(3), W(2), H, C, b, b, b, 0, 0, 0), (A, S(4), N_, H, 1, b, 1, 2, 0, 1)]
    if chain == "reference":
        return [(A, X, N_, F, 1, b, b, 0, 0, 1), (S(0), W(0), F, H, b, b, b, 0, 0, 0), (A, S(1), N_, H, 1, b, b, 0, 0, 1),
                (S(2), W(1), H, H, b, b, b, 0, 0, 0), (A, S(3), N_, H, 1, b, b, 0, 0, 1), (S(4), W(2), H, C, b, b, 1, 2, 0, 0)]
    if not run_gin:
        return [(XR, W(0), F, H, b, b, b, 1, 0, 0), (A, S(0), N_, H, 1, b, b, 0, 0, 1), (S(1), W(1), H, H, b, b, b, 1, 0, 0),
                (A, S(2), N_, H, 1, b, b, 0, 0, 1), (S(3), W(3), H, C, b, b, b, 1, 0, 0), (A, S(4), N_, C, 1, b, 1, 2, 1, 1)]
    return [(A, X, N_, F, 1, b, b, 0, 0, 1), (S(0), W(0), F, H, b, b, b, 1, 0, 0), (A, S(1), N_, H, 1, b, b, 0, 0, 1),
            (S(2), W(1), H, H, b, b, b, 1, 0, 0), (A, S(3), N_, H, 1, b, b, 0, 0, 1), (S(4), W(2), H, C, b, b, 1, 2, 0, 0)]


def chain_entries_cover(b: int, F: int, H: int, C: int, run_gin: bool, max_n: int) -> bool:
    """Can the chain entries (qgtc_chain_transform / qgtc_chain_aggregate) run this epoch? One width b per chain: 1 .. 4 bits with up
    to 256 hidden units / classes, 5 .. 8 bits with up to 128 (include/qgtc.h); cluster batches of at most 8192 nodes; Cluster-GCN's
    first product X . W1 loops over the k-quads of up to 8192 features while its float32 sums stay exact (PAD128(F) (2^b - 1)^2 < 2^24 -
    the library counts whole k-quads, launch_common.hip.h::no_wrap: 256 features at 8 bits); Batched-GIN's X is the right operand of an
    aggregation, so F is bounded like H."""
    if not 1 <= b <= 8 or max_n > 8192:
        return False
    lim = 256 if b <= 4 else 128
    if max(H, C) > lim:
        return False
    if run_gin:
        return F <= lim
    return F <= 8192 and (F + 127) // 128 * 128 * ((1 << b) - 1) ** 2 < (1 << 24)


class PlannedEpoch:
    """The grouped epoch on a device-filled plan (Q.EpochPlan): what BatchedEpoch builds on the host - 6 x 75 descriptors,
    pooled outputs, occupancy bitmaps - is split into the data loader's part (`data`, made once beside the packing:
    ClusterIter.epoch_data) and ONE bind launch inside the epoch clock. Same launches, same words as BatchedEpoch."""

    def __init__(self, Q, data, params, W, b, chain: str, run_gin: bool, fuse: bool = True, chain_stages: bool = True,
                 keep_aggregates: bool = False):
        self._args = (Q, data, params, W, b, chain, run_gin, fuse, chain_stages, keep_aggregates)
        self._bind()

    def _bind(self):
        Q, data, params, W, b, chain, run_gin, fuse, chain_stages, keep_aggregates = self._args
        self._bound_engine = Q.get_engine()     # the route (chain entries / grouped GEMMs, T's format) is fixed per bind
        H, C = W["hidden"], W["classes"]
        F = params[0][3]
        self.data = data
        self.final = 5
        stages = stage_recipes(Q, chain, run_gin, F, H, C, b)
        launches = [(0, i, 0, 0, 0) for i in range(6)]
        expand = []
        self.discarded = set()
        max_n = max(p[0] for p in params)
        # the library's own routing switches that take the chain entries / chained pairs away (perf-only switches such as
        # QGTC_NO_XCD leave the route alone)
        switches = routing_switches()
        covered = chain_entries_cover(b, F, H, C, run_gin, max_n)
        if (fuse and chain == "correct" and chain_stages and not run_gin and not keep_aggregates and covered
                and Q.get_engine() != "popcount" and not switches):
            # The Cluster-GCN chain (1 .. 8 bits; the BASELINE epoch: 2) on the chain entries (qgtc_chain_transform / qgtc_chain_aggregate): one wave per row
            # block for the whole width, T between the launches as finished matrix-core operands, weights pre-expanded once
            # per plan. X.W1 | A.T1 + .W2 | A.T2 + .W3 | A.T3 -> float32: four launches.
            for i in (0, 2, 4):
                stages[i] = stages[i][:10] + (1,)
            expand = [(0, F, H, b, 0), (1, H, H, b, 1), (3, H, C, b, 1)]     # (weight, K, N, bits, order)
            launches = [(3, 0, 0, 0, 0), (4, 1, 2, 0, 1), (4, 3, 4, 0, 2), (4, 5, -1, 0, 0)]
            self.discarded = {0, 1, 2, 3, 4}
            if getattr(data, "a_tiles", False):   # the aggregations read the adjacency as 512-byte tiles
                for i in (1, 3, 5):
                    stages[i] = (Q.SRC_AT,) + stages[i][1:]
        elif (fuse and chain == "correct" and chain_stages and run_gin and not keep_aggregates and covered
                and Q.get_engine() != "popcount" and not switches and getattr(data, "x_chain", False)):
            # Batched-GIN (1 .. 8 bits; the BASELINE epoch: 4) on the same entries: A.X + .W1 | A.T1 + .W2 | A.T2 + .W3 -> float32, three launches; X in the
            # chain format from the data loader (ClusterIter.epoch_data), T between the launches likewise
            stages[0] = (stages[0][0], Q.SRC_XC) + stages[0][2:]
            for i in (1, 3):
                stages[i] = stages[i][:10] + (1,)
            expand = [(0, F, H, b, 1), (1, H, H, b, 1), (2, H, C, b, 1)]
            launches = [(4, 0, 1, 0, 0), (4, 2, 3, 0, 1), (4, 4, 5, 0, 2)]
            self.discarded = {0, 1, 2, 3, 4}
            if getattr(data, "a_tiles", False):
                for i in (0, 2, 4):
                    stages[i] = (Q.SRC_AT,) + stages[i][1:]
        elif fuse and chain == "correct" and chain_stages:
            # an aggregation stage and the NEXT layer's X.W stage are one call (qgtc_gcn_chain_batched)
            pairs = [(1, 2), (3, 4)] if not run_gin else [(0, 1), (2, 3), (4, 5)]
            first = {i: (1, i, j, 0 if keep_aggregates else Q.CHAIN_DISCARD, 0) for i, j in pairs}
            second = {j for _, j in pairs}
            launches = [first.get(i, (0, i, 0, 0, 0)) for i in range(6) if i not in second]
        elif fuse and chain == "correct":
            pairs = [(0, 1), (2, 3), (4, 5)] if not run_gin else [(1, 2), (3, 4)]
            first = {i: (2, i, j, 0, 0) for i, j in pairs}
            second = {j for _, j in pairs}
            launches = [first.get(i, (0, i, 0, 0, 0)) for i in range(6) if i not in second]
        self.n_launches = len(launches)
        data.bind([W["W1"], W["W2"], W["W3"], W["W3h"]], [list(t) for t in stages], [list(l) for l in launches], [list(e) for e in expand])

    def run(self):
        if self._args[0].get_engine() != self._bound_engine:    # set_engine() since the bind: another route, other formats of T
            self._bind()
        self.data.run()

    @property
    def outs(self):
        return self.data.outs(self.final)

    def stage_outs(self, i):
        return self.data.outs(i)


# ---------------------------------------------------------------------------------------------
def uses_planned_epoch(args) -> bool:
    """--batched runs on a device-filled plan (PlannedEpoch); the host-built plan (BatchedEpoch) remains for --streams and
    for callers that hold per-batch tensors of their own."""
    return bool(args.batched) and not args.non_resident and not args.zerotile_jump \
        and not getattr(args, "pack_on_the_fly", False)


def make_iter(args, Q, graph, batch_ids=None):
    """The ClusterIter of a run (main_qgtc.py:74-93 builds it ahead of the epoch clock), under the reference's seeds
    (main_qgtc.py:45-47: the partition shuffle of sampler.py:68 draws from `random`)."""
    torch.manual_seed(3)
    np.random.seed(2)
    random.seed(2)
    return ClusterIter(args.dataset, graph, args.psize, args.batch_size, bit_width=args.bit_width,
                       run_GIN=args.run_GIN, device=torch.device(f"cuda:{args.gpu}"), resident=not args.non_resident, qgtc=Q,
                       batch_ids=batch_ids, with_rows_X=(args.chain == "correct"),
                       keep_raw=getattr(args, "pack_on_the_fly", False))


def run(args, Q=None, batch_ids=None, graph=None, it=None):
    """Runs the epoch loop; returns a dict with avg_epoch_ms, the last epoch's per-batch outputs
    and the iterator. `batch_ids` restricts this process to a shard of the batches; `it` = a ClusterIter built earlier
    for the same arguments (the reference builds it once, ahead of the epoch clock: main_qgtc.py:74-93)."""
    if Q is None:
        import QGTC as Q
    torch.manual_seed(3)      # main_qgtc.py:45-47
    np.random.seed(2)
    random.seed(2)
    device = torch.device(f"cuda:{args.gpu}")
    torch.cuda.set_device(device)

    if graph is None and it is None:
        if args.dataset.endswith(".npz"):
            graph = G.load_npz_graph(args.dataset, args.dim, args.psize)
        else:
            graph = G.make_graph(args.dataset, args.psize)
    feat_size = (graph if graph is not None else it.g).feat.shape[1]
    b = args.bit_width
    if it is None:
        it = make_iter(args, Q, graph, batch_ids)
    if uses_planned_epoch(args):
        it.epoch_data(Q)      # the data loader's share of a grouped epoch (per-batch table, adjacency bitmaps): beside the packing
    torch.cuda.synchronize()

    prev_engine = Q.get_engine()
    Q.set_engine(args.engine)
    try:
        return _run_epochs(args, Q, it, feat_size, b, device)
    finally:
        Q.set_engine(prev_engine)


def _run_epochs(args, Q, it, feat_size, b, device):
    start_time = time.time()  # main_qgtc.py:96 — the clock starts before the weights are packed
    W = pack_weights(Q, feat_size, args.n_hidden, args.n_classes, b, device)
    chain = CHAINS[(args.chain, args.run_GIN)]
    outs = []
    if args.zerotile_jump:    # main_qgtc.py:142-145 / cluster_gcn.py:208-211
        printed_global, printed_counter = [], []      # the cumulative values of the `counter_global:` / `counter:` lines
        for ct, param in it:
            ct = ct.to(device)
            A0, A1 = param[0], param[1]
            t0 = Q.bitMM2Bit(ct.bit_X, W["W1"], param[2], param[3], W["hidden"], b, b, b)
            Q.bitMM2Bit_base_cnt(ct.bit_A, t0, A0, A1, W["hidden"], 1, b, b)
            printed_global.append(Q.get_counters()[0])
            Q.bitMM2Bit_zerojump_cnt(ct.bit_A, t0, A0, A1, W["hidden"], 1, b, b)
            printed_counter.append(Q.get_counters()[1])
        row = zerotile_row(args.dataset, printed_global, printed_counter)
        if not args.quiet:
            # (on stderr: stdout is the log the reference's parse_counter.py reads, and a line with the word `dataset` in it that is not
            # the Namespace line stops that script - parse_counter.py:11-13 indexes the line's second comma field)
            print(ZEROTILE_HEADER, file=sys.stderr)
            print(row["line"], file=sys.stderr)
        return {"avg_epoch_ms": float("nan"), "outs": [], "iter": it, "counters": Q.get_counters(), "zerotile": row}

    if getattr(args, "pack_on_the_fly", False) and args.batched:
        # cluster_gcn.py's structure (:151-227: pack inside the epoch loop) in the grouped form: per epoch ONE loader call packs
        # every batch from the resident edge lists and features (qgtc_load_batches), one launch binds the plan, then the epoch's
        # grouped launches. Everything inside the clock.
        plan = None
        for _ in range(args.n_epochs):
            data = it.pack_now(Q)
            plan = PlannedEpoch(Q, data, it.cluster_param_li, W, b, args.chain, args.run_GIN, fuse=not getattr(args, "no_fuse", False),
                                chain_stages=not getattr(args, "no_chain", False))
            plan.run()
        torch.cuda.synchronize()
        end_time = time.time()
        avg = (end_time - start_time) * 1000 / args.n_epochs
        if not args.quiet:
            print("Avg. Epoch: {:.3f} ms".format(avg))       # cluster_gcn.py:246
        return {"avg_epoch_ms": avg, "outs": plan.outs, "iter": it}

    if getattr(args, "pack_on_the_fly", False):
        from .sampler import ClusterTensor
        transfering = running_time = 0.0          # cluster_gcn.py:121-122
        for _ in range(args.n_epochs):
            outs = []
            for (r_, c_, X_), param in zip(it.raw_li, it.cluster_param_li):
                torch.cuda.synchronize()
                t = time.perf_counter()
                r_d, c_d, X_d = r_.to(device), c_.to(device), X_.to(device)   # cluster_gcn.py:165-167 (no-op when resident)
                torch.cuda.synchronize()
                transfering += time.perf_counter() - t
                t = time.perf_counter()
                n = param[0]
                ct = ClusterTensor(Q.pack_edges(r_d, c_d, n, n, 1, False), Q.val2bit(X_d, b, True, False),
                                   Q.val2bit(X_d, b, False, False) if args.chain == "correct" else None)
                outs.append(chain(Q, ct, param, W, b))
                torch.cuda.synchronize()
                running_time += time.perf_counter() - t
        end_time = time.time()
        avg = (end_time - start_time) * 1000 / args.n_epochs
        if not args.quiet:
            print("Trans (ms): {:.3f}, Compute (ms): {:.3f}".format(transfering / args.n_epochs * 1e3,
                                                                      running_time / args.n_epochs * 1e3))   # cluster_gcn.py:245
            print("Avg. Epoch: {:.3f} ms".format(avg))                                                       # cluster_gcn.py:246
        return {"avg_epoch_ms": avg, "outs": outs, "iter": it, "trans_ms": transfering / args.n_epochs * 1e3,
                "compute_ms": running_time / args.n_epochs * 1e3}

    if uses_planned_epoch(args):
        plan = PlannedEpoch(Q, it.epoch_data(Q), it.cluster_param_li, W, b, args.chain, args.run_GIN, fuse=not getattr(args, "no_fuse", False),
                            chain_stages=not getattr(args, "no_chain", False))
        for _ in range(args.n_epochs):
            plan.run()
        torch.cuda.synchronize()
        end_time = time.time()
        avg = (end_time - start_time) * 1000 / args.n_epochs
        if not args.quiet:
            print(AVG_EPOCH_FORMAT.format(avg))
        return {"avg_epoch_ms": avg, "outs": plan.outs, "iter": it, "plan": plan}   # (the per-batch views are made here, after the clock)
    if args.batched or args.streams > 0:
        cts = [c.to(device) for c in it.cTensor_li]
        plan = BatchedEpoch(Q, cts, it.cluster_param_li, W, b, args.chain, args.run_GIN, fuse=not getattr(args, "no_fuse", False),
                            chain_stages=not getattr(args, "no_chain", False))
        for _ in range(args.n_epochs):
            outs = plan.run() if args.batched else plan.run_per_batch(args.streams)
    elif args.graph:
        # the reference's launch structure (six extension calls per batch), recorded once on a
        # side stream and replayed per epoch: the host issues ONE graph launch per epoch
        cts = [c.to(device) for c in it.cTensor_li]
        side = torch.cuda.Stream(device)
        side.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(side):      # warm-up outside capture (kernel attributes, allocator)
            for ct, param in zip(cts, it.cluster_param_li):
                chain(Q, ct, param, W, b)
        torch.cuda.current_stream(device).wait_stream(side)
        graph_obj = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph_obj):
            outs = [chain(Q, ct, param, W, b) for ct, param in zip(cts, it.cluster_param_li)]
        for _ in range(args.n_epochs):
            graph_obj.replay()
    else:
        for _ in range(args.n_epochs):
            outs = []
            for ct, param in it:
                ct = ct.to(device, non_blocking=True)   # main_qgtc.py:115 (no-op when resident)
                outs.append(chain(Q, ct, param, W, b))
    torch.cuda.synchronize()
    end_time = time.time()
    avg = (end_time - start_time) * 1000 / args.n_epochs
    if not args.quiet:
        print(AVG_EPOCH_FORMAT.format(avg))
    return {"avg_epoch_ms": avg, "outs": outs, "iter": it}


# The reference's drivers `print(args)` (main_qgtc.py:42, cluster_gcn.py:45) and its log parsers pick the dataset's name out of that line BY
# POSITION: parse_time.py:12 takes comma-field 2 of main_qgtc.py's line (and skips lines of fewer than five fields), parse_counter.py:13
# comma-field 1 of cluster_gcn.py's (4_8_zero_tile_jumping.py:20-41 runs that script). Those positions are the ones of the Python the reference
# ran on (3.7 / 3.8: argparse.Namespace printed its keys SORTED: batch_size, bit_width, dataset, .. and batch_size, dataset, ..); today's
# argparse prints them in the order they were added (dataset first). So the line is composed here, from each script's own keys, sorted.
MAIN_QGTC_KEYS = sorted(["dataset", "gpu", "n_epochs", "batch_size", "psize", "dim", "n_hidden", "n_classes", "n_layers", "bit_width",
                         "use_pp", "regular", "run_GIN", "use_QGTC", "zerotile_jump"])                      # main_qgtc.py:21-39
CLUSTER_GCN_KEYS = sorted(["dataset", "gpu", "n_epochs", "batch_size", "psize", "dim", "n_hidden", "n_classes", "n_layers",
                           "use_pp", "regular", "use_PyG", "run_GIN", "use_QGTC", "zerotile_jump"])         # cluster_gcn.py:25-42
AVG_EPOCH_FORMAT = "Avg. Epoch: {:.3f} ms"          # main_qgtc.py:159; parse_time.py:15-17 reads the number between ':' and 'ms'


def args_line(args) -> str:
    """The `Namespace(...)` line of the reference's driver this run stands in for: cluster_gcn.py's with --zerotile_jump (the zero-tile
    profile is that script's, read by parse_counter.py), main_qgtc.py's otherwise (read by parse_time.py)."""
    keys = CLUSTER_GCN_KEYS if args.zerotile_jump else MAIN_QGTC_KEYS
    return "Namespace(" + ", ".join("{}={!r}".format(k, getattr(args, k, False)) for k in keys) + ")"


def extra_flags_line(args) -> str:
    """This driver's own flags, on a line of their own (no word of it is one the reference's parsers look for)."""
    ours = [k for k in vars(args) if k not in MAIN_QGTC_KEYS and k not in CLUSTER_GCN_KEYS]
    return "qgtc-mi355x flags: " + ", ".join("{}={!r}".format(k, getattr(args, k)) for k in ours)


def main(argv=None):
    args = build_parser().parse_args(argv)
    if not args.quiet:
        print(args_line(args))   # main_qgtc.py:42 / cluster_gcn.py:45
        print(extra_flags_line(args))
    return run(args)


if __name__ == "__main__":
    main()
