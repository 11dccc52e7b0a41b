"""Round 6: the 32 x 32 x 64 form of the wide kernel (k_bitmm_fp4_wide32: one-plane operands on 128 x 256 tiles) against the AND + popcount
kernels word for word (ragged shapes, all three outputs; QGTC_WIDE_RF=4 forces the tile shape it serves), then timings."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import QGTC

if "time" not in sys.argv[1:]:
    torch.manual_seed(1)
    bad = 0
    for (M, K, N) in [(128, 1024, 512), (129, 1000, 257), (1000, 1153, 300), (77, 128, 1000), (513, 2176, 1030), (2048, 4096, 1024), (300, 897, 513), (8, 3968, 264), (4100, 5000, 700)]:
        for density in (0.5, 0.01):
            A = (torch.rand(M, K, device="cuda") < density).float()
            X = (torch.rand(K, N, device="cuda") < 0.5).float()
            bA, bX = QGTC.val2bit(A, 1, False, False), QGTC.val2bit(X, 1, True, False)
            for ob in (1, 3, 10, 23):
                outs = {}
                for eng in ("popcount", "mfma"):
                    QGTC.set_engine(eng)
                    outs[eng] = (QGTC.bitMM2Bit(bA, bX, M, K, N, 1, 1, ob), QGTC.bitMM2Bit_col(bA, bX, M, K, N, 1, 1, ob), QGTC.bitMM2Int(bA, bX, M, K, N, 1, 1))
                for i, name in enumerate(("rows", "cols", "int")):
                    if not torch.equal(outs["popcount"][i], outs["mfma"][i]):
                        bad += 1
                        d = outs["popcount"][i] != outs["mfma"][i]
                        print(f"MISMATCH {M}x{K}x{N} ob={ob} {name}: {int(d.sum())} of {d.numel()} differ; first at {d.flatten().nonzero()[:4].flatten().tolist()}")
        print(f"{M}x{K}x{N} checked", flush=True)
    print("mismatches:", bad)
QGTC.set_engine("auto")
for (M, K, N) in ((4096, 4096, 1024), (8192, 4096, 1024), (8192, 8192, 2048), (16384, 16384, 1024), (32768, 32768, 1024), (32768, 32768, 512)):
    A = (torch.rand(M, K, device="cuda") < 0.5).float()
    X = (torch.rand(K, N, device="cuda") < 0.5).float()
    bA, bX = QGTC.val2bit(A, 1, False, False), QGTC.val2bit(X, 1, True, False)
    del A, X
    ms = min(QGTC.profile(bA, bX, M, K, N, 1, 1, 1, 20) for _ in range(3))
    us = ms * 1e3 / 20
    print(f"{M}x{K}x{N}: {us:.1f} us  {2.0 * M * K * N / us / 1e6:.0f} TOPS  fp4_frac {2.0 * M * K * N / (us * 1e-6) / 1e16:.3f}", flush=True)
