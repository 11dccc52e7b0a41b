"""Randomised agreement check of the engines (not a pytest: run on the GPU box through gpurun)."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch, QGTC
from qgtc_ppopp22_amd.shapes import rows_shape, cols_shape
rng = np.random.default_rng(12345)
bad = 0
for case in range(400):
    M = int(rng.integers(1, 700)); N = int(rng.integers(1, 300))
    K = int(rng.choice([rng.integers(1, 200), rng.integers(200, 3000), rng.integers(3000, 20000)]))
    a = int(rng.choice([1, 1, 2, 2, 3, 4, 8])); w = int(rng.choice([1, 2, 3, 4, 5, 8]))
    ob = int(rng.integers(1, 9))
    g = torch.Generator().manual_seed(case)
    dX = torch.randint(-2**31, 2**31 - 1, rows_shape(M, K, a), dtype=torch.int32, generator=g).cuda()
    dW = torch.randint(-2**31, 2**31 - 1, cols_shape(K, N, w), dtype=torch.int32, generator=g).cuda()
    if case % 3 == 0:
        dX[: dX.size(0) // 2] = 0
    QGTC.set_engine("popcount")
    ref = (QGTC.bitMM2Bit(dX, dW, M, K, N, a, w, ob), QGTC.bitMM2Bit_col(dX, dW, M, K, N, a, w, ob), QGTC.bitMM2Int(dX, dW, M, K, N, a, w, True))
    for eng in ("mfma", "auto"):
        QGTC.set_engine(eng)
        got = (QGTC.bitMM2Bit(dX, dW, M, K, N, a, w, ob), QGTC.bitMM2Bit_col(dX, dW, M, K, N, a, w, ob), QGTC.bitMM2Int(dX, dW, M, K, N, a, w, True))
        for i, (x, y) in enumerate(zip(got, ref)):
            if not torch.equal(x, y):
                bad += 1
                print("MISMATCH", eng, i, M, K, N, a, w, ob)
        for mode in (0, 1, 2):
            for zj in (False, True):
                bg = QGTC.BatchedGemm([dX], [dW], [(M, K, N)], a, w, ob, mode, True, zj)
                bg.run()
                if not torch.equal(bg.outs[0].view(-1), ref[mode].view(-1)):
                    bad += 1
                    print("MISMATCH grouped", eng, mode, zj, M, K, N, a, w, ob)
    QGTC.set_engine("popcount")
print("cases done:", case + 1, "mismatches:", bad)
