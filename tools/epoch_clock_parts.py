"""What the driver-style epoch clock (main_qgtc.py:96-159: weights, plan bind, 20 epochs, synchronise) is made of."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import QGTC as Q
from qgtc_ppopp22_amd import driver, graph as G

for dataset, bits, hidden, gin in (("ogbn-arxiv", 2, 128, False), ("ppi", 4, 64, True)):
    g = G.make_graph(dataset, 1500)
    args = driver.build_parser().parse_args(["--dataset", dataset, "--n-hidden", str(hidden), "--bit_width", str(bits), "--use_QGTC", "--quiet",
                                             "--batched", "--chain", "correct"] + (["--run_GIN"] if gin else []))
    it = driver.make_iter(args, Q, g)
    data = it.epoch_data(Q)
    dev = torch.device("cuda:0")
    F = g.feat.shape[1]
    for trial in range(4):
        W = driver.pack_weights(Q, F, hidden, 10, bits, dev)
        plan = driver.PlannedEpoch(Q, data, it.cluster_param_li, W, bits, "correct", gin)
        t_w = time.perf_counter()
        while time.perf_counter() - t_w < 0.3:
            for _ in range(20):
                plan.run()
            torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        for e in ev:
            e.record()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ev[0].record()
        W = driver.pack_weights(Q, F, hidden, 10, bits, dev)
        t1 = time.perf_counter()
        ev[1].record()
        plan = driver.PlannedEpoch(Q, data, it.cluster_param_li, W, bits, "correct", gin)
        t2 = time.perf_counter()
        ev[2].record()
        for _ in range(20):
            plan.run()
        t3 = time.perf_counter()
        ev[3].record()
        torch.cuda.synchronize()
        t4 = time.perf_counter()
        print(dataset, "trial", trial, "host us: weights %.1f bind %.1f 20 epochs issued %.1f sync %.1f total %.1f | gpu us: weights %.1f bind %.1f epochs %.1f (%.2f per epoch)" % (
            (t1 - t0) * 1e6, (t2 - t1) * 1e6, (t3 - t2) * 1e6, (t4 - t3) * 1e6, (t4 - t0) * 1e6,
            ev[0].elapsed_time(ev[1]) * 1e3, ev[1].elapsed_time(ev[2]) * 1e3, ev[2].elapsed_time(ev[3]) * 1e3, ev[2].elapsed_time(ev[3]) * 1e3 / 20))
