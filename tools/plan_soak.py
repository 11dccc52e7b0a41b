"""Race screen for the epochs on the chain entries (device-filled plan): 600 epochs each of the arxiv-sized Cluster-GCN and the
ppi-sized Batched-GIN plan, other traffic on the chip in between, the final float outputs of EVERY epoch compared with a
six-launch plan's (public layouts, grouped kernels); the chain's private buffers start from poison.
usage: plan_soak.py [epochs] [widths]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import QGTC as Q
from qgtc_ppopp22_amd import driver, graph as G

EPOCHS = int(sys.argv[1]) if len(sys.argv) > 1 else 600
dev = torch.device("cuda:0")
bad_total = 0
CASES = (("ogbn-arxiv", 2, 128, False), ("ppi", 4, 64, True))
if len(sys.argv) > 2 and sys.argv[2] == "widths":   # every width of the chain entries, both models (round 4)
    CASES = tuple((ds, b, h, gin) for ds, h, gin in (("ogbn-arxiv", 128, False), ("ppi", 64, True), ("ppi", 128, True)) for b in (1, 2, 3, 4))
for dataset, bits, hidden, gin in CASES:
    g = G.make_graph(dataset, 1500)
    args = driver.build_parser().parse_args(["--dataset", dataset, "--n-hidden", str(hidden), "--bit_width", str(bits), "--use_QGTC", "--quiet", "--batched",
                                             "--chain", "correct"] + (["--run_GIN"] if gin else []))
    it = driver.make_iter(args, Q, g)
    data = it.epoch_data(Q)
    W = driver.pack_weights(Q, g.feat.shape[1], hidden, 10, bits, dev)
    six = driver.PlannedEpoch(Q, data, it.cluster_param_li, W, bits, "correct", gin, fuse=False)
    six.run()
    torch.cuda.synchronize()
    ref = [o.clone() for o in six.outs]
    plan = driver.PlannedEpoch(Q, data, it.cluster_param_li, W, bits, "correct", gin)   # (re-binds the same data: new pool)
    assert plan.n_launches == (3 if gin else 4)
    noise = torch.empty(64 << 20, dtype=torch.int32, device=dev)
    bad = 0
    for e in range(EPOCHS):
        if e % 50 == 0:
            for o in plan.outs:
                o.fill_(float("nan"))
        noise.random_()                      # other traffic: 256 MB of writes between epochs
        plan.run()
        if e % 7 == 0:
            noise[: 1 << 20].add_(1)
        outs = plan.outs
        if not all(torch.equal(a, b) for a, b in zip(outs, ref)):
            bad += 1
    print(dataset, "bits", bits, "hidden", hidden, "epochs", EPOCHS, "launches per epoch", plan.n_launches, "mismatching epochs", bad, flush=True)
    bad_total += bad
print("mismatches", bad_total)
