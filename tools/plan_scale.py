"""Time per launch of the chained epoch plan against the number of cluster batches in it (is a stage bound by the rounds of
workgroups the chip takes, or by a latency chain that does not care how many run beside it?)."""
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
def ev(fn, reps=200):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
for nb in (5, 10, 20, 30, 40, 50, 60, len(it.cTensor_li)):
    plan = driver.BatchedEpoch(Q, it.cTensor_li[:nb], it.cluster_param_li[:nb], W, b, "correct", gin)
    plan.run()
    per = [ev(l.run) for l in plan.launches]
    print(f"{nb:3d} batches: epoch {ev(plan.run, 50):6.1f} us; launches alone " + " ".join(f"{t:5.1f}" for t in per), flush=True)
