"""Per-launch time of the bound epoch plan (each launch alone, 200 reps between two events) and of the whole epoch.
CHAIN=reference times the reference's literal call sequence (six grouped launches) instead of the layout-correct chain."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import QGTC as Q
from qgtc_ppopp22_amd import driver, graph as G

def ev(fn, reps=200):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps

which = sys.argv[1:] or ["arxiv", "ppi"]
chain = os.environ.get("CHAIN", "correct")
for name in which:
    dataset, bits, hidden, gin = ("ogbn-arxiv", 2, 128, False) if name == "arxiv" else ("ppi", 4, 64, True)
    g = G.make_graph(dataset, 1500)
    args = driver.build_parser().parse_args(["--dataset", dataset, "--n-hidden", str(hidden), "--bit_width", str(bits), "--use_QGTC", "--quiet", "--batched", "--chain", chain] + (["--run_GIN"] if gin else []))
    it = driver.make_iter(args, Q, g)
    data = it.epoch_data(Q)
    dev = torch.device("cuda:0")
    W = driver.pack_weights(Q, g.feat.shape[1], hidden, 10, bits, dev)
    plan = driver.PlannedEpoch(Q, data, it.cluster_param_li, W, bits, chain, gin)
    t_w = time.perf_counter()
    while time.perf_counter() - t_w < 0.3:
        for _ in range(20):
            plan.run()
        torch.cuda.synchronize()
    per = [round(ev(lambda i=i: data.run_launch(i)), 2) for i in range(plan.n_launches)]
    ep = round(ev(plan.run, 100), 2)
    print(name, chain, "launches", per, "sum", round(sum(per), 2), "epoch", ep, 
          "occupied", round(data.occupied_fraction, 3), flush=True)
