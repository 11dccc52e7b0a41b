import os, sys, time
sys.path.insert(0, "/root/repo")
import torch
import QGTC as Q
from qgtc_ppopp22_amd import driver, graph as G
from qgtc_ppopp22_amd.sampler import ClusterIter
graph = G.make_graph("ogbn-arxiv", 1500)
dev = torch.device("cuda:0")
it = ClusterIter("ogbn-arxiv", graph, 1500, 20, bit_width=2, run_GIN=False, device=dev, qgtc=Q, with_rows_X=True)
W = driver.pack_weights(Q, graph.feat.shape[1], 128, 10, 2, dev)
cts, params = it.cTensor_li, it.cluster_param_li
n = [p[0] for p in params]
bitA = [c.bit_A for c in cts]; bitXr = [c.bit_X_rows for c in cts]
torch.cuda.synchronize()
def t(f, reps=5):
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); r = f(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best * 1e3, r
dims = [(ni, 128, 128) for ni in n]
ms, g0 = t(lambda: Q.BatchedGemm(bitXr, [W["W1"]], dims, 2, 2, 2, 1, False, False)); print(f"BatchedGemm X.W (no bitmap): {ms:.3f} ms")
dimsA = [(ni, ni, 128) for ni in n]
ms, g1 = t(lambda: Q.BatchedGemm(bitA, g0.outs, dimsA, 1, 2, 2, 0, False, True)); print(f"BatchedGemm A.(XW) with bitmaps: {ms:.3f} ms")
ms, g1b = t(lambda: Q.BatchedGemm(bitA, g0.outs, dimsA, 1, 2, 2, 0, False, True, g1.occs)); print(f"BatchedGemm A.(XW) reusing bitmaps: {ms:.3f} ms")
ms, f = t(lambda: Q.FusedLayer(g0, g1)); print(f"FusedLayer: {ms:.3f} ms")
ms, _ = t(lambda: [c.bit_A for c in cts]); print(f"python list of 75 tensors: {ms:.3f} ms")
ms, _ = t(lambda: driver.BatchedEpoch(Q, cts, params, W, 2, "correct", False)); print(f"whole BatchedEpoch: {ms:.3f} ms")
