"""Grouped cols-layout stages (75 ragged batches): which kernel family and how long. Run as is and with QGTC_ROWS_FIRST=1."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import QGTC as Q
rng = np.random.default_rng(0)
ns = [int(x) for x in rng.integers(1100, 1300, size=75)]
tag = "rows_first" if os.environ.get("QGTC_ROWS_FIRST") else "default   "
for (a, w, ob) in ((1, 1, 1), (2, 2, 2), (3, 3, 3), (4, 4, 4), (2, 4, 4), (4, 8, 8)):
    for (K, N) in ((128, 128), (128, 10), (128, 64), (50, 64), (256, 64), (300, 128), (600, 40)):
        Xs = [Q.val2bit(torch.rand(n, K, device="cuda") * (1 << a), a, False, False) for n in ns]
        W = Q.val2bit(torch.rand(K, N, device="cuda") * (1 << w), w, True, False)
        bg = Q.BatchedGemm(Xs, [W], [(n, K, N) for n in ns], a, w, ob, 1, True, False)
        bg.run(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(3):
            e0.record()
            for _ in range(50): bg.run()
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / 50)
        print(tag, f"a={a} w={w} ob={ob} K={K} N={N}: {best:7.2f} us", flush=True)
