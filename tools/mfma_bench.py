"""popcount engine vs the opt-in MFMA engine (bits expanded to int8 on the fly), same products."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, QGTC
M = K = 4096
for N, w, a in ((64, 1, 1), (128, 1, 1), (256, 1, 1), (512, 1, 1), (1024, 1, 1), (256, 2, 1), (512, 2, 1), (256, 2, 2), (128, 2, 1), (1024, 2, 1), (1024, 4, 1), (256, 7, 1),
                (256, 8, 1), (1024, 8, 1), (128, 8, 8), (1024, 8, 8)):
    A = torch.randint(0, 2 ** a, (M, K)).float().cuda()
    X = torch.randint(0, 2 ** w, (K, N)).float().cuda()
    bA, bX = QGTC.val2bit(A, a, False, False), QGTC.val2bit(X, w, True, False)
    res = {}
    reps = 100 if a * w * N <= 8192 else 10
    for eng in ("popcount", "mfma", "auto"):
        QGTC.set_engine(eng)
        QGTC.profile(bA, bX, M, K, N, a, w, w, 5)
        ms = min(QGTC.profile(bA, bX, M, K, N, a, w, w, reps) for _ in range(3))
        res[eng] = (ms * 1e3 / reps, QGTC.bitMM2Bit(bA, bX, M, K, N, a, w, w))
    QGTC.set_engine("popcount")
    same = torch.equal(res["popcount"][1], res["mfma"][1])
    ops = 2.0 * M * K * N
    print(f"{M}x{K}x{N} a={a} w={w}: popcount {res['popcount'][0]:7.2f} us ({ops/res['popcount'][0]/1e6:7.1f} TOPS)   "
          f"mfma {res['mfma'][0]:7.2f} us ({ops/res['mfma'][0]/1e6:7.1f} TOPS)   auto {res['auto'][0]:7.2f} us   identical={same}")
