import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import QGTC as Q
from qgtc_ppopp22_amd import driver, graph as G
dataset, bits, hidden, gin = "ogbn-arxiv", 2, 128, False
g = G.make_graph(dataset, 1500)
args = driver.build_parser().parse_args(["--dataset", dataset, "--n-hidden", str(hidden), "--bit_width", str(bits), "--use_QGTC", "--quiet", "--batched", "--chain", "correct"])
it = driver.make_iter(args, Q, g)
data = it.epoch_data(Q)
dev = torch.device("cuda:0")
pc = time.perf_counter
W = driver.pack_weights(Q, 128, hidden, 10, bits, dev)
plan = driver.PlannedEpoch(Q, data, it.cluster_param_li, W, bits, "correct", gin)
for variant in ("plain", "dummy", "sleep", "nobusy", "busy_short", "plain"):
    for trial in range(3):
        if variant != "nobusy":
            t_w = pc()
            while pc() - t_w < (0.02 if variant == "busy_short" else 0.3):
                for _ in range(20):
                    plan.run()
                torch.cuda.synchronize()
        if variant == "dummy":
            for _ in range(3):
                torch.empty(16, device=dev).fill_(0)
            torch.cuda.synchronize()
        if variant == "sleep":
            time.sleep(0.002)
        torch.cuda.synchronize()
        t0 = pc()
        W = driver.pack_weights(Q, 128, hidden, 10, bits, dev)
        t1 = pc()
        plan = driver.PlannedEpoch(Q, data, it.cluster_param_li, W, bits, "correct", gin)
        t2 = pc()
        for _ in range(20):
            plan.run()
        t3 = pc()
        torch.cuda.synchronize()
        t4 = pc()
        print(variant, trial, "weights %.1f bind %.1f issue %.1f sync %.1f total %.1f" % ((t1 - t0) * 1e6, (t2 - t1) * 1e6, (t3 - t2) * 1e6, (t4 - t3) * 1e6, (t4 - t0) * 1e6))
