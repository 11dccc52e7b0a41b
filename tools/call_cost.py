"""Host cost of one extension call on the unchanged-driver path (main_qgtc.py:128-154: six calls per cluster batch).

Prints (1) the pieces of a bitMM2Bit call timed in C++ (QGTC._host_parts), (2) microseconds per call of the lean entry points
against the pybind11 ("checked") ones in a host-bound loop, (3) the driver's per-batch legs (75 batches x 6 calls, 20 epochs)
on both, resident and non-resident. Results: DESIGN.md section 6."""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import QGTC as Q
from qgtc_ppopp22_amd import driver, graph as G

dev = torch.device("cuda:0")
n, F, H, b = 1213, 128, 128, 2
X = Q.val2bit(torch.randn(n, F, device=dev), b, False, False)
W = Q.val2bit(torch.ones(F, H, device=dev), b, True, False)
print("host_parts us [torch::empty, empty_cuda, guard+stream, C-ABI launch]:", [round(v, 3) for v in Q.host_parts(X, W, n, F, H, b, b, b, 4096)], flush=True)


def per_call(fn, reps=4000):
    for _ in range(200):
        fn(X, W, n, F, H, b, b, b)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(reps):
            fn(X, W, n, F, H, b, b, b)
        t1 = time.perf_counter()          # host-side issue time (the queue is deep enough not to block in 4000 x 3 us)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        best = min(best, (t1 - t0) / reps * 1e6)
        last_total = (t2 - t0) / reps * 1e6
    return round(best, 3), round(last_total, 3)


print("bitMM2Bit lean     us/call (issue, incl. drain):", per_call(Q.bitMM2Bit))
print("bitMM2Bit checked  us/call (issue, incl. drain):", per_call(Q.checked_bitMM2Bit))
print("bitMM2Bit_col lean us/call:", per_call(Q.bitMM2Bit_col), " checked:", per_call(Q.checked_bitMM2Bit_col), flush=True)

checked = types.SimpleNamespace(**{k: getattr(Q, k) for k in dir(Q) if not k.startswith("__")})
checked.val2bit, checked.bitMM2Bit, checked.bitMM2Bit_col, checked.bitMM2Int = (Q.checked_val2bit, Q.checked_bitMM2Bit, Q.checked_bitMM2Bit_col,
                                                                              Q.checked_bitMM2Int)
g = G.make_graph("ogbn-arxiv", 1500)
base = ["--dataset", "ogbn-arxiv", "--n-hidden", "128", "--n-classes", "10", "--bit_width", "2", "--use_QGTC", "--quiet", "--n-epochs", "20"]
for extra in ([], ["--non-resident"], ["--chain", "correct"], ["--pack-on-the-fly"]):
    args = driver.build_parser().parse_args(base + extra)
    for name, mod in (("lean", Q), ("checked", checked)):
        it = driver.make_iter(args, Q, g)
        ms = []
        for _ in range(4):
            ms.append(driver.run(args, Q=mod, graph=g, it=it)["avg_epoch_ms"])
        print("per-batch epoch", extra, name, "ms:", [round(m, 3) for m in ms], flush=True)
