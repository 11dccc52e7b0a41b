"""Per-stage timing of the grouped Cluster-GCN epoch (six launches): which operator dominates."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import QGTC as Q
from qgtc_ppopp22_amd import driver, graph as G
from qgtc_ppopp22_amd.sampler import ClusterIter

Q.set_engine(os.environ.get("ENGINE", "auto"))   # grouped launches follow the engine switch
chain = sys.argv[1] if len(sys.argv) > 1 else "correct"
gin = len(sys.argv) > 2 and sys.argv[2] == "gin"
dataset = "ppi" if gin else "ogbn-arxiv"
b = 4 if gin else 2
hidden = 64 if gin else 128
graph = G.make_graph(dataset, 1500)
dev = torch.device("cuda:0")
it = ClusterIter(dataset, graph, 1500, 20, bit_width=b, run_GIN=gin, device=dev, qgtc=Q, with_rows_X=(chain == "correct"))
W = driver.pack_weights(Q, graph.feat.shape[1], hidden, 10, b, dev)
plan = driver.BatchedEpoch(Q, it.cTensor_li, it.cluster_param_li, W, b, chain, gin, fuse=False)
for _ in range(3):
    plan.run()
torch.cuda.synchronize()
ref_outs = [o.clone() for o in plan.outs]


def timed(fn, reps=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


if chain == "correct":     # one fused launch per layer (qgtc_gcn_layer_batched) against the two grouped launches
    fused = driver.BatchedEpoch(Q, it.cTensor_li, it.cluster_param_li, W, b, chain, gin, fuse=True, chain_stages=False)
    for _ in range(3):
        fused.run()
    torch.cuda.synchronize()
    same = all(torch.equal(x, y) for x, y in zip(fused.outs, ref_outs))
    for i, g in enumerate(fused.launches):
        print(f"fused launch {i} ({type(g).__name__}): {timed(g.run):8.1f} us")
    print(f"epoch ({chain}, {'GIN' if gin else 'GCN'}) fused, {len(fused.launches)} launches: {timed(fused.run):8.1f} us   outputs identical: {same}")
names = ["g0", "g1", "g2", "g3", "g4", "g5"]
for g, nme in zip(plan.stages, names):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        g.run()
    e1.record()
    torch.cuda.synchronize()
    print(f"{nme}: {e0.elapsed_time(e1) * 1e3 / 20:8.1f} us per grouped launch")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    plan.run()
e1.record()
torch.cuda.synchronize()
print(f"epoch ({chain}, {'GIN' if gin else 'GCN'}): {e0.elapsed_time(e1) * 1e3 / 20:8.1f} us")
print("zero-jump per stage:", [(g.zero_jump, round(g.occupied_fraction, 3)) for g in plan.stages])
for S in (1, 2, 3, 4, 8):
    for _ in range(2):
        plan.run_per_batch(S)
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        plan.run_per_batch(S)
    e1.record()
    torch.cuda.synchronize()
    print(f"per-batch launches on {S} stream(s): {e0.elapsed_time(e1) * 1e3 / 10:8.1f} us per epoch (host {1e6 * (time.perf_counter() - t0) / 10:8.1f} us)")
