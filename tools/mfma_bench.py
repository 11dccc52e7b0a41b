"""popcount engine vs the opt-in MFMA engine (bits expanded to int8 on the fly), same products."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, QGTC
M = K = 4096
for N, w in ((64, 1), (128, 1), (256, 1), (1024, 1), (128, 2), (1024, 2), (1024, 4), (256, 7)):
    A = (torch.rand(M, K) < 0.5).float().cuda()
    X = torch.randint(0, 2 ** w, (K, N)).float().cuda()
    bA, bX = QGTC.val2bit(A, 1, False, False), QGTC.val2bit(X, w, True, False)
    res = {}
    for eng in ("popcount", "mfma", "auto"):
        QGTC.set_engine(eng)
        QGTC.profile(bA, bX, M, K, N, 1, w, w, 20)
        ms = min(QGTC.profile(bA, bX, M, K, N, 1, w, w, 100) for _ in range(3))
        res[eng] = (ms * 1e3 / 100, QGTC.bitMM2Bit(bA, bX, M, K, N, 1, w, w))
    QGTC.set_engine("popcount")
    same = torch.equal(res["popcount"][1], res["mfma"][1])
    ops = 2.0 * M * K * N
    print(f"{M}x{K}x{N} w={w}: popcount {res['popcount'][0]:7.2f} us ({ops/res['popcount'][0]/1e6:7.1f} TOPS)   "
          f"mfma {res['mfma'][0]:7.2f} us ({ops/res['mfma'][0]/1e6:7.1f} TOPS)   auto {res['auto'][0]:7.2f} us   identical={same}")
