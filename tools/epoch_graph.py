"""The grouped layout-correct epoch (six launches) issued eagerly and replayed from a captured graph: does a graph shorten
the gaps between the dependent launches?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import QGTC as Q
from qgtc_ppopp22_amd import driver, graph as G
from qgtc_ppopp22_amd.sampler import ClusterIter
gin = len(sys.argv) > 1 and sys.argv[1] == "gin"
dataset, b, hidden = ("ppi", 4, 64) if gin else ("ogbn-arxiv", 2, 128)
graph = G.make_graph(dataset, 1500)
dev = torch.device("cuda:0")
it = ClusterIter(dataset, graph, 1500, 20, bit_width=b, run_GIN=gin, device=dev, qgtc=Q, with_rows_X=True)
W = driver.pack_weights(Q, graph.feat.shape[1], hidden, 10, b, dev)
plan = driver.BatchedEpoch(Q, it.cTensor_li, it.cluster_param_li, W, b, "correct", gin)
def ev(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
print(f"eager: {ev(plan.run):.1f} us per epoch")
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    plan.run()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    plan.run()
print(f"graph replay (1 epoch per graph): {ev(g.replay):.1f} us per epoch")
g4 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g4):
    for _ in range(4): plan.run()
print(f"graph replay (4 epochs per graph): {ev(g4.replay) / 4:.1f} us per epoch")
