import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, QGTC
for nn in (16, 32, 64, 256, 1024):
    for mk in (1024, 2048, 4096):
        A = torch.randint(-128, 128, (mk, mk), dtype=torch.int8).cuda()
        B = torch.randint(-128, 128, (nn, mk), dtype=torch.int8).cuda()
        QGTC.i8gemm_profile(A, B, 20, False)
        ms = min(QGTC.i8gemm_profile(A, B, 200, False) for _ in range(3))
        print(f"i8gemm {mk}x{mk}x{nn}: {ms*1e3/200:.2f} us  {2.0*mk*mk*nn*200/(ms*1e-3)/1e12:.1f} TOPS")
