"""Race screen for the four-launch epoch (chained aggregation + X.W kernels, LDS hand-over inside a workgroup, partial-line
stores merging in L2): the epoch repeated many times with other traffic in between, every stage's outputs compared with
the six-launch plan's each time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import QGTC as Q
from qgtc_ppopp22_amd import driver, graph as G
from qgtc_ppopp22_amd.sampler import ClusterIter
bad = 0
for gin in (False, True):
    dataset, b, hidden = ("ppi", 4, 64) if gin else ("ogbn-arxiv", 2, 128)
    graph = G.make_graph(dataset, 1500)
    dev = torch.device("cuda:0")
    it = ClusterIter(dataset, graph, 1500, 20, bit_width=b, run_GIN=gin, device=dev, qgtc=Q, with_rows_X=True)
    W = driver.pack_weights(Q, graph.feat.shape[1], hidden, 10, b, dev)
    ref = driver.BatchedEpoch(Q, it.cTensor_li, it.cluster_param_li, W, b, "correct", gin, chain_stages=False)
    ref.run(); torch.cuda.synchronize()
    want = [[o.clone() for o in g.outs] for g in ref.stages]
    plan = driver.BatchedEpoch(Q, it.cTensor_li, it.cluster_param_li, W, b, "correct", gin)
    junk = torch.empty(96 << 20, dtype=torch.int32, device=dev)
    n_bad = 0
    for i in range(300):
        if i % 2 == 0:
            for g in plan.stages:
                for o in g.outs: o.fill_(-1 if o.dtype == torch.int32 else 7.0)
        if i % 3 == 0: junk.random_()
        plan.run()
        if i % 5 == 4: plan.run()          # back to back as well
        torch.cuda.synchronize()
        ok = all(torch.equal(x, y) for i, (g, ws) in enumerate(zip(plan.stages, want)) if i not in plan.discarded for x, y in zip(g.outs, ws))
        n_bad += 0 if ok else 1
    print(f"{'GIN' if gin else 'GCN'}: 300 epochs of {len(plan.launches)} launches, {n_bad} differ from the six-launch plan's outputs", flush=True)
    bad += n_bad
print("total mismatches:", bad)
