"""Per-batch Batched-GIN epoch (main_qgtc.py:131-138 literally: 75 batches x six calls) on the default engine, with and without the
single-launch row-block kernel for its 4 x 4-bit products (QGTC_NO_ROWS1=1 in a second process: the popcount kernel). The epoch is bound by the host's six calls per batch: the launch time below is what moves."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import QGTC as Q
from qgtc_ppopp22_amd import driver, graph as G
g = G.make_graph("ppi", 1500)
args = driver.build_parser().parse_args(["--dataset", "ppi", "--n-hidden", "64", "--n-classes", "10", "--bit_width", "4", "--use_QGTC", "--quiet", "--n-epochs", "20", "--run_GIN"])
it = driver.make_iter(args, Q, g)
ms = [driver.run(args, Q=Q, graph=g, it=it)["avg_epoch_ms"] for _ in range(6)][1:]
print("NO_ROWS1" if os.environ.get("QGTC_NO_ROWS1") else "rows1   ", "per-batch GIN epoch ms:", [round(m, 3) for m in ms], flush=True)
n = 599
X = Q.val2bit(torch.rand(n, 50, device="cuda") * 16, 4, False, False)
W = Q.val2bit(torch.rand(50, 64, device="cuda") * 16, 4, True, False)
out = Q.bitMM2Bit(X, W, n, 50, 64, 4, 4, 4)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
Q.bitMM2Bit_enqueue(out, X, W, n, 50, 64, 4, 4, 4, 500)
e1.record()
torch.cuda.synchronize()
print("   599 x 50 x 64 4x4-bit single launch:", round(e0.elapsed_time(e1) * 1e3 / 500, 2), "us")
