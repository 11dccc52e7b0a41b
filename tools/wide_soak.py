"""Race screen for the wide-operand kernel's hand-counted vmcnt / barrier pipeline: the same large
product repeated many times (every result compared with the first), then random shapes against the popcount engine."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, QGTC
torch.manual_seed(5)
bad = 0
for (M, K, N, a, w, reps) in ((8192, 4096, 1024, 1, 1, 300), (4096, 4096, 1024, 1, 1, 300), (8192, 8192, 2048, 1, 1, 60), (8192, 4096, 1024, 2, 2, 100), (4096, 2048, 768, 1, 2, 200)):
    A = torch.randint(0, 2 ** a, (M, K)).float().cuda(); X = torch.randint(0, 2 ** w, (K, N)).float().cuda()
    bA, bX = QGTC.val2bit(A, a, False, False), QGTC.val2bit(X, w, True, False)
    QGTC.set_engine("popcount"); ref = QGTC.bitMM2Int(bA, bX, M, K, N, a, w, True)
    QGTC.set_engine("mfma")
    n_bad = 0
    junk = torch.empty(64 << 20, dtype=torch.int32, device="cuda")
    for i in range(reps):
        if i % 3 == 0: junk.random_()          # other traffic between launches: different cache / timing states
        got = QGTC.bitMM2Int(bA, bX, M, K, N, a, w, True)
        if not torch.equal(got, ref): n_bad += 1
    print(f"{M}x{K}x{N} a={a} w={w}: {reps} launches, {n_bad} differ from the popcount engine's result", flush=True)
    bad += n_bad
g = torch.Generator().manual_seed(9)
for i in range(150):
    M = int(torch.randint(8, 3000, (1,), generator=g)); K = int(torch.randint(1, 9000, (1,), generator=g)); N = int(torch.randint(257, 2100, (1,), generator=g))
    a = int(torch.randint(1, 3, (1,), generator=g)); w = int(torch.randint(1, 3, (1,), generator=g)); ob = int(torch.randint(1, 6, (1,), generator=g))
    A = torch.randint(0, 2 ** a, (M, K), generator=g).float().cuda(); X = torch.randint(0, 2 ** w, (K, N), generator=g).float().cuda()
    bA, bX = QGTC.val2bit(A, a, False, False), QGTC.val2bit(X, w, True, False)
    outs = {}
    for eng in ("popcount", "mfma"):
        QGTC.set_engine(eng)
        outs[eng] = (QGTC.bitMM2Bit(bA, bX, M, K, N, a, w, ob), QGTC.bitMM2Bit_col(bA, bX, M, K, N, a, w, ob), QGTC.bitMM2Int(bA, bX, M, K, N, a, w, True))
    ok = all(torch.equal(x, y) for x, y in zip(outs["popcount"], outs["mfma"]))
    if not ok:
        bad += 1
        print(f"MISMATCH {M}x{K}x{N} a={a} w={w} ob={ob}")
print("random shapes done; total mismatches:", bad)
