"""Exploration: reference-style micro-benchmark sweep (2_7c shapes) through QGTC.profile."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import QGTC

REF = {  # BASELINE.md §1 (sm_86), TFLOPs by (M, N, w)
    (1024, 16): (5.847, 3.934, 2.488, 1.541), (2048, 16): (16.605, 10.086, 6.561, 3.483),
    (4096, 16): (40.627, 20.764, 12.409, 6.763), (1024, 32): (11.724, 7.864, 4.456, 3.074),
    (2048, 32): (32.666, 19.762, 12.807, 6.816), (4096, 32): (35.032, 20.951, 13.929, 7.366),
    (1024, 64): (23.219, 15.429, 10.683, 5.046), (2048, 64): (37.438, 25.055, 12.328, 6.165),
    (4096, 64): (46.768, 26.818, 14.196, 7.324)}

torch.manual_seed(3)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rand = "--ones" not in sys.argv
for wi, w in enumerate([1, 2, 4, 8]):
    for N in [16, 32, 64]:
        for M in [1024, 2048, 4096]:
            K = M
            if rand:
                A = (torch.rand(M, K, device="cuda") < 0.5).float()
                X = torch.randint(0, 2 ** w, (K, N), device="cuda").float()
            else:
                A = torch.ones(M, K, device="cuda")
                X = torch.ones(K, N, device="cuda")
            ba = QGTC.val2bit(A, 1, False, False)
            bx = QGTC.val2bit(X, w, True, False)
            QGTC.profile(ba, bx, M, K, N, 1, w, w, 20)
            ms = min(QGTC.profile(ba, bx, M, K, N, 1, w, w, reps) for _ in range(3))
            us = ms * 1e3 / reps
            tops = 2.0 * M * K * N / (us * 1e-6) / 1e12
            print(f"w={w} M=K={M} N={N}: {us:8.2f} us/launch  eff {tops:9.1f} TOPS   ref {REF[(M, N)][wi]:7.3f}  x{tops / REF[(M, N)][wi]:.1f}", flush=True)
