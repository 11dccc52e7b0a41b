#!/usr/bin/env python3
"""Generates the golden fixtures in this directory.

    python tests/golden/make_golden.py

PROVENANCE: the reference (YukeWang96/QGTC_PPoPP22) holds no recorded outputs and cannot be run in
this image (CUDA-only kernels; Python drivers need dgl/ogb), so these vectors are produced by the
repo's own CPU oracle (oracle/qgtc_oracle.c), after that oracle has been checked against the
unitest.py-derived known answers and the warp-level emulation (tests/test_oracle_*.py). They freeze
the oracle's behaviour (regression guard) and give the GPU tests fixed inputs with edge cases:
x<0, ties .5/1.5/2.5, x == 2^b, NaN/inf, C == 2^ob and 2^ob+1, sizes that are not multiples of
8/32/128, K > 128, and two tiny cluster batches (n = 37 and 130) with every operator of the
reference's and the layout-correct GCN chain.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

from oracle.qgtc_oracle import Oracle  # noqa: E402
from helpers import edge_floats, oracle_chain, oracle_weights, rand_q  # noqa: E402


def main():
    O = Oracle()
    rng = np.random.default_rng(20260101)
    out = {}

    # --- quantiser + packing -------------------------------------------------------------
    for i, (H, W, b) in enumerate([(5, 7, 1), (9, 33, 2), (37, 130, 3), (130, 37, 4), (16, 129, 8)]):
        x = edge_floats(rng, H, W, b)
        out[f"pack{i}_x"] = x
        out[f"pack{i}_meta"] = np.array([H, W, b])
        out[f"pack{i}_q"] = O.quantize(x, b)
        out[f"pack{i}_rows"] = O.val2bit(x, b, False, False)
        out[f"pack{i}_cols"] = O.val2bit(x, b, True, False)
        out[f"pack{i}_cols_out"] = O.val2bit(x, b, True, True)
    out["pack_count"] = np.array(5)

    # --- bit-GEMM ------------------------------------------------------------------------
    cases = [(3, 3, 3, 2, 2, 2), (9, 130, 17, 1, 2, 2), (33, 257, 31, 2, 3, 3), (40, 200, 10, 4, 4, 4),
             (64, 640, 33, 1, 8, 8), (37, 37, 64, 1, 2, 2)]
    for i, (M, K, N, a, w, ob) in enumerate(cases):
        qx, qw = rand_q(rng, M, K, a), rand_q(rng, K, N, w)
        X, Wt, W8 = O.pack(qx, a, False), O.pack(qw, w, True), O.pack(qw, w, True, True)
        out[f"mm{i}_meta"] = np.array([M, K, N, a, w, ob])
        out[f"mm{i}_X"], out[f"mm{i}_W"], out[f"mm{i}_W8"] = X, Wt, W8
        out[f"mm{i}_acc"] = O.acc(X, Wt, M, K, N, a, w)
        out[f"mm{i}_bits"] = O.bitmm2bit(X, Wt, M, K, N, a, w, ob)
        out[f"mm{i}_bits_col"] = O.bitmm2bit(X, Wt, M, K, N, a, w, ob, col=True)
        out[f"mm{i}_f32_pad128"] = O.bitmm2int(X, Wt, M, K, N, a, w, True)
        out[f"mm{i}_f32_pad8"] = O.bitmm2int(X, W8, M, K, N, a, w, False)
    out["mm_count"] = np.array(len(cases))
    # requant boundary: ones inputs, C = K in {3,4,5,6} against 2^2
    for K in (3, 4, 5, 6):
        X, Wt = O.pack(np.ones((5, K), np.int32), 1, False), O.pack(np.ones((K, 6), np.int32), 1, True)
        out[f"rq{K}_bits"] = O.bitmm2bit(X, Wt, 5, K, 6, 1, 1, 2)

    # --- two tiny cluster batches, full GCN chains ------------------------------------------
    for i, (n, F, H, C, b, dens) in enumerate([(37, 20, 16, 10, 2, 0.08), (130, 50, 64, 10, 2, 0.03)]):
        A = (rng.random((n, n)) < dens).astype(np.float32)
        np.fill_diagonal(A, 0)
        Xf = rng.standard_normal((n, F)).astype(np.float32)
        bi = {"n": n, "F": F, "A": A, "X": Xf, "bit_A": O.val2bit(A, 1), "bit_X": O.val2bit(Xf, b, True),
              "bit_X_rows": O.val2bit(Xf, b, False)}
        W = oracle_weights(O, F, H, C, b)
        out[f"cb{i}_meta"] = np.array([n, F, H, C, b])
        out[f"cb{i}_A"], out[f"cb{i}_Xf"] = A, Xf
        for chain in ("reference", "correct"):
            for k, t in enumerate(oracle_chain(O, bi, W, b, chain, False)):
                out[f"cb{i}_{chain}_op{k}"] = t
    out["cb_count"] = np.array(2)
    np.savez_compressed(os.path.join(HERE, "qgtc_golden.npz"), **out)
    print("wrote", os.path.join(HERE, "qgtc_golden.npz"), os.path.getsize(os.path.join(HERE, "qgtc_golden.npz")), "bytes")


if __name__ == "__main__":
    main()
