"""Host-side cost of building the grouped epoch plan (inside the reference's clock, main_qgtc.py:96)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import QGTC as Q
from qgtc_ppopp22_amd import driver, graph as G
from qgtc_ppopp22_amd.sampler import ClusterIter

graph = G.make_graph("ogbn-arxiv", 1500)
dev = torch.device("cuda:0")
it = ClusterIter("ogbn-arxiv", graph, 1500, 20, bit_width=2, run_GIN=False, device=dev, qgtc=Q, with_rows_X=True)
torch.cuda.synchronize()
for rep in range(4):
    t0 = time.perf_counter()
    W = driver.pack_weights(Q, graph.feat.shape[1], 128, 10, 2, dev)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    plan = driver.BatchedEpoch(Q, it.cTensor_li, it.cluster_param_li, W, 2, "correct", False)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    for _ in range(20):
        plan.run()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    print(f"pack_weights {1e3 * (t1 - t0):6.2f} ms   plan build {1e3 * (t2 - t1):6.2f} ms   20 epochs {1e3 * (t3 - t2):6.2f} ms")
    # one stage alone
    t4 = time.perf_counter()
    g = Q.BatchedGemm([c.bit_A for c in it.cTensor_li], plan.stages[0].outs, [(p[0], p[0], 128) for p in it.cluster_param_li], 1, 2, 2, 0, False, True)
    torch.cuda.synchronize()
    t5 = time.perf_counter()
    g2 = Q.BatchedGemm([c.bit_A for c in it.cTensor_li], plan.stages[0].outs, [(p[0], p[0], 128) for p in it.cluster_param_li], 1, 2, 2, 0, False, False)
    torch.cuda.synchronize()
    t6 = time.perf_counter()
    print(f"   one A-stage BatchedGemm with bitmaps {1e3 * (t5 - t4):6.2f} ms, without {1e3 * (t6 - t5):6.2f} ms")
