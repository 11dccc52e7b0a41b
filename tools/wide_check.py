"""Wide-operand FP4 kernel (bitmm_fp4_wide.hip.h): popcount engine vs mfma engine on ragged shapes in all three output
forms (raw random words, padding bits included where the layout allows), then timings at N = 512 / 1024 / 2048."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, QGTC

torch.manual_seed(1)
bad = 0
shapes = [(128, 1024, 512), (129, 1024, 257), (1000, 1152, 300), (77, 128, 1000), (513, 2176, 1030), (2048, 4096, 1024), (300, 896, 513), (8, 3968, 264)]
if len(sys.argv) > 1 and sys.argv[1] == "time":
    shapes = []
for (M, K, N) in shapes:
  for (a, w) in ((1, 1), (1, 2), (2, 1), (2, 2), (1, 4), (4, 1), (2, 4), (4, 2), (1, 8), (2, 8), (8, 1), (8, 2)):
    A = torch.randint(0, 2 ** a, (M, K)).float().cuda()
    X = torch.randint(0, 2 ** w, (K, N)).float().cuda()
    bA, bX = QGTC.val2bit(A, a, False, False), QGTC.val2bit(X, w, True, False)
    for ob in (1, 3, 10):
        outs = {}
        for eng in ("popcount", "mfma"):
            QGTC.set_engine(eng)
            outs[eng] = (QGTC.bitMM2Bit(bA, bX, M, K, N, a, w, ob), QGTC.bitMM2Bit_col(bA, bX, M, K, N, a, w, ob), QGTC.bitMM2Int(bA, bX, M, K, N, a, w))
        for i, name in enumerate(("rows", "cols", "int")):
            ok = torch.equal(outs["popcount"][i], outs["mfma"][i])
            if not ok:
                bad += 1
                d = (outs["popcount"][i] != outs["mfma"][i])
                print(f"MISMATCH {M}x{K}x{N} a={a} w={w} ob={ob} {name}: {int(d.sum())} of {d.numel()} differ; first at {d.flatten().nonzero()[:4].flatten().tolist()}")
    print(f"{M}x{K}x{N} a={a} w={w} checked", flush=True)
print("mismatches:", bad)
for (M, K, N) in ((4096, 4096, 512), (4096, 4096, 1024), (8192, 4096, 1024), (8192, 8192, 2048), (4096, 4096, 2048)):
  for (a, w) in ((1, 1), (1, 2), (2, 2), (1, 4), (2, 4), (4, 1), (1, 8), (2, 8)):
    A = torch.randint(0, 2 ** a, (M, K)).float().cuda()
    X = torch.randint(0, 2 ** w, (K, N)).float().cuda()
    bA, bX = QGTC.val2bit(A, a, False, False), QGTC.val2bit(X, w, True, False)
    line = f"{M}x{K}x{N} a={a} w={w}:"
    for eng, env in (("mfma", "1"), ("mfma", "")):
        os.environ["QGTC_NO_WIDE"] = env
        QGTC.set_engine(eng)
        QGTC.profile(bA, bX, M, K, N, a, w, w, 5)
        ms = min(QGTC.profile(bA, bX, M, K, N, a, w, w, 50) for _ in range(3))
        us = ms * 1e3 / 50
        line += f"  {'128-tile' if env else 'wide'} {us:7.2f} us ({2.0 * M * K * N / us / 1e6:7.0f} TOPS)"
    print(line, flush=True)
