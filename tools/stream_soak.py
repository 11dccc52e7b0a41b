"""Race screen for the long-K kernel's hand-counted vmcnt / barrier pipeline (bitmm_fp4_stream.hip.h: fetching waves count their LDS-DMA
pieces, multiplying waves their fragment reads, one barrier a group): the same products repeated many times with other traffic in between
(every result compared with the popcount engine's), both tile heights, dense and sparse left operands, then random shapes.
python tools/stream_soak.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, QGTC
bad = 0
g = torch.Generator(device="cuda").manual_seed(11)
junk = torch.empty(96 << 20, dtype=torch.int32, device="cuda")
for (M, K, N, density, reps, rf) in ((32768, 32768, 64, 0.5, 200, None), (32768, 32768, 16, 0.5, 150, None), (16384, 16384, 256, 0.5, 150, None), (16384, 8192, 64, 0.01, 300, "4"),
                                     (8192, 20000, 200, 0.5, 200, "4"), (8200, 4097, 33, 0.5, 300, None), (65536, 8192, 48, 0.003, 150, None), (300, 32768, 64, 0.5, 300, None)):
    A = (torch.rand(M, K, device="cuda", generator=g) < density).float()
    if density < 0.1:
        A[: M // 2, K // 4: K // 2] = 0     # whole steps of zeros next to occupied ones: the skip path and the multiply path alternate
    X = (torch.rand(K, N, device="cuda", generator=g) < 0.5).float()
    bA, bX = QGTC.val2bit(A, 1, False, False), QGTC.val2bit(X, 1, True, False)
    del A, X
    QGTC.set_engine("popcount")
    ref_f, ref_b = QGTC.bitMM2Int(bA, bX, M, K, N, 1, 1, True), QGTC.bitMM2Bit(bA, bX, M, K, N, 1, 1, 3)
    QGTC.set_engine("mfma")
    if rf:
        os.environ["QGTC_STREAM_RF"] = rf
    n_bad = 0
    for i in range(reps):
        if i % 3 == 0:
            junk.random_()          # other traffic between launches: different cache / timing states
        got_f = QGTC.bitMM2Int(bA, bX, M, K, N, 1, 1, True)
        got_b = QGTC.bitMM2Bit(bA, bX, M, K, N, 1, 1, 3)
        if not (torch.equal(got_f, ref_f) and torch.equal(got_b, ref_b)):
            n_bad += 1
    os.environ.pop("QGTC_STREAM_RF", None)
    print(f"{M}x{K}x{N} density {density} tiles {'128 rows' if rf else 'as routed'}: {reps} x 2 launches, {n_bad} differ from the popcount engine's result", flush=True)
    bad += n_bad
    del bA, bX, ref_f, ref_b
gc = torch.Generator().manual_seed(9)
for i in range(150):
    M = int(torch.randint(1, 20000, (1,), generator=gc)); K = int(torch.randint(4097, 30000, (1,), generator=gc)); N = int(torch.randint(1, 257, (1,), generator=gc))
    ob = int(torch.randint(1, 25, (1,), generator=gc))
    A = (torch.rand(M, K, device="cuda", generator=g) < (0.5 if i % 2 else 0.002)).float(); X = (torch.rand(K, N, device="cuda", generator=g) < 0.5).float()
    bA, bX = QGTC.val2bit(A, 1, False, False), QGTC.val2bit(X, 1, True, False)
    del A, X
    outs = {}
    for eng in ("popcount", "mfma"):
        QGTC.set_engine(eng)
        outs[eng] = (QGTC.bitMM2Bit(bA, bX, M, K, N, 1, 1, ob), QGTC.bitMM2Bit_col(bA, bX, M, K, N, 1, 1, ob), QGTC.bitMM2Int(bA, bX, M, K, N, 1, 1, True))
    if not all(torch.equal(x, y) for x, y in zip(outs["popcount"], outs["mfma"])):
        bad += 1
        print(f"MISMATCH {M}x{K}x{N} ob={ob}")
QGTC.set_engine("auto")
print("stream_soak: random shapes done; total mismatches:", bad)
