"""Round 6: the long-K kernel (k_bitmm_fp4_stream) against the AND + popcount kernels, word for word, on ragged shapes, every output
form, dense / sparse / empty left operands; then timings of 5_9_adjmatrix_size.py's big shapes. python tools/stream_check.py [time]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import QGTC as Q

LIB = Q  # noqa


def operands(M, K, N, density, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    A = (torch.rand(M, K, device="cuda", generator=g) < density).float()
    X = (torch.rand(K, N, device="cuda", generator=g) < 0.5).float()
    return Q.val2bit(A, 1, False, False), Q.val2bit(X, 1, True, False)


def check():
    bad = 0
    shapes = [(70, 4100, 64), (64, 8192, 16), (100, 5000, 17), (257, 9000, 33), (1000, 8200, 64), (130, 4224, 65), (333, 6000, 130),
              (64, 16384, 200), (2049, 4500, 256), (31, 4097, 1), (5000, 12000, 48), (8192, 8192, 64), (16384, 8192, 32)]
    for (M, K, N) in shapes:
        for density in (0.5, 0.001, 0.0):
            ba, bx = operands(M, K, N, density, M + K + N)
            for ob in (1, 3, 14):
                outs = {}
                for eng in ("auto", "popcount"):
                    Q.set_engine(eng)
                    outs[eng] = (Q.bitMM2Bit(ba, bx, M, K, N, 1, 1, ob), Q.bitMM2Bit_col(ba, bx, M, K, N, 1, 1, ob),
                                 Q.bitMM2Int(ba, bx, M, K, N, 1, 1, True))
                Q.set_engine("auto")
                ok = all(torch.equal(a, b) for a, b in zip(outs["auto"], outs["popcount"]))
                if not ok:
                    bad += 1
                    print("MISMATCH", M, K, N, density, ob, [torch.equal(a, b) for a, b in zip(outs["auto"], outs["popcount"])])
    print("stream_check:", "all identical" if bad == 0 else f"{bad} mismatches")
    return bad


def timings():
    for mk in (8192, 16384, 32768):
        for n in (16, 32, 64, 128, 256):
            ba, bx = operands(mk, mk, n, 0.5, 3)
            ms = min(Q.profile(ba, bx, mk, mk, n, 1, 1, 1, 20) for _ in range(3))
            us = ms * 1e3 / 20
            print(f"{mk}x{mk}x{n}: {us:.2f} us  {2.0 * mk * mk * n / us / 1e6:.0f} TOPS  hbm_frac {(mk * mk / 8 + mk * n / 4) / (us * 1e-6) / 8e12:.3f}  "
                  f"fp4_frac {2.0 * mk * mk * max(n, 16) / (us * 1e-6) / 1e16:.3f}", flush=True)


if __name__ == "__main__":
    rc = 0 if "time" in sys.argv[1:] and "check" not in sys.argv[1:] else check()
    if "time" in sys.argv[1:]:
        timings()
    sys.exit(1 if rc else 0)
