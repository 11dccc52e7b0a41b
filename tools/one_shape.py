"""Run one micro-benchmark shape (for rocprofv3): python tools/one_shape.py M K N w reps"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import QGTC

M, K, N, w, reps = (int(v) for v in sys.argv[1:6])
torch.manual_seed(3)
A = (torch.rand(M, K, device="cuda") < 0.5).float()
X = torch.randint(0, 2 ** w, (K, N), device="cuda").float()
ba = QGTC.val2bit(A, 1, False, False)
bx = QGTC.val2bit(X, w, True, False)
ms = QGTC.profile(ba, bx, M, K, N, 1, w, w, reps)
print(f"{M}x{K}x{N} w={w}: {ms * 1e3 / reps:.2f} us/launch, {2.0 * M * K * N * reps / (ms * 1e-3) / 1e12:.1f} eff TOPS")
