"""Host microseconds of each call inside the epoch clock's setup (weights + bind)."""
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
F, H, C = 128, 128, 10
pc = time.perf_counter
for trial in range(6):
    torch.cuda.synchronize()
    t = [pc()]
    ones = torch.ones(F * H + H * H + H * C, device=dev); t.append(pc())
    W1 = ones[:F * H].view(F, H); W2 = ones[F * H:F * H + H * H].view(H, H); W3 = ones[F * H + H * H:].view(H, C); t.append(pc())
    p = Q.val2bit_many([W1, W2, W3, W3], bits, [True] * 4, [False, False, True, False]); t.append(pc())
    W = {"W1": p[0], "W2": p[1], "W3": p[2], "W3h": p[3], "feat": F, "hidden": H, "classes": C}
    stages = driver.stage_recipes(Q, "correct", gin, F, H, C, bits); t.append(pc())
    plan = driver.PlannedEpoch(Q, data, it.cluster_param_li, W, bits, "correct", gin); t.append(pc())
    plan.run(); t.append(pc())
    plan.run(); t.append(pc())
    torch.cuda.synchronize(); t.append(pc())
    print("trial", trial, "ones %.1f views %.1f val2bit_many %.1f recipes %.1f PlannedEpoch %.1f run %.1f run %.1f sync %.1f" % tuple((b - a) * 1e6 for a, b in zip(t, t[1:])))
    # inside bind
    t = [pc()]
    x = torch.empty(7_500_000, dtype=torch.int32, device=dev); t.append(pc())
    y = torch.empty(6 * 75 * 72, dtype=torch.uint8, device=dev); t.append(pc())
    print("   empty pool %.1f empty descs %.1f" % tuple((b - a) * 1e6 for a, b in zip(t, t[1:])))
    del x, y
