"""The grouped layout-correct epoch with and without joining each aggregation stage to the next layer's X.W stage
(qgtc_gcn_chain_batched): launches, time per epoch, outputs compared."""
import os, sys
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
def ev(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
plans = {}
for name, cs in (("separate", False), ("chained", True)):
    plans[name] = driver.BatchedEpoch(Q, it.cTensor_li, it.cluster_param_li, W, b, "correct", gin, chain_stages=cs)
    for g in plans[name].stages:
        for o in g.outs: o.fill_(-1 if o.dtype == torch.int32 else 7.0)
    plans[name].run()
same = all(torch.equal(x, y) for i, (gs, gc) in enumerate(zip(plans["separate"].stages, plans["chained"].stages)) if i not in plans["chained"].discarded for x, y in zip(gs.outs, gc.outs))
for name in ("separate", "chained", "separate", "chained"):
    print(f"{name:9s}: {len(plans[name].launches)} calls, {ev(plans[name].run):6.1f} us per epoch")
print("every stage's outputs identical:", same)
