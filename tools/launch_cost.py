"""Host microseconds of the C-ABI launch alone (QGTC.host_parts) for a tiny and an epoch-sized product."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import QGTC as Q
dev = torch.device("cuda:0")
for (n, F, H, b) in ((8, 8, 8, 1), (1213, 128, 128, 2), (4096, 4096, 64, 1)):
    X = Q.val2bit(torch.randn(n, F, device=dev), b, False, False)
    W = Q.val2bit(torch.ones(F, H, device=dev), b, True, False)
    for eng in ("auto", "popcount"):
        Q.set_engine(eng)
        print(os.environ.get("TAG", ""), (n, F, H, b), eng, [round(v, 3) for v in Q.host_parts(X, W, n, F, H, b, b, b, 4096)], flush=True)
