"""Throughput of INDEPENDENT bitMM2Bit launches spread over S HIP streams vs one stream."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, QGTC
M = K = 4096
for N, w in ((64, 1), (64, 2), (64, 8), (16, 1)):
    A = (torch.rand(M, K) < 0.5).float().cuda()
    X = torch.randint(0, 2 ** w, (K, N)).float().cuda()
    bA, bX = QGTC.val2bit(A, 1, False, False), QGTC.val2bit(X, w, True, False)
    ref = QGTC.bitMM2Bit(bA, bX, M, K, N, 1, w, w)
    for S in (1, 2, 3, 4, 8):
        outs = [torch.empty_like(ref) for _ in range(S)]
        QGTC.bitMM2Bit_enqueue_streams(outs, bA, bX, M, K, N, 1, w, w, 50)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(3):
            e0.record()
            QGTC.bitMM2Bit_enqueue_streams(outs, bA, bX, M, K, N, 1, w, w, 1000)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        assert all(torch.equal(o, ref) for o in outs)
        us = best * 1e3 / 1000
        print(f"{M}x{K}x{N} w={w} streams={S}: {us:.2f} us/launch  {2.0*M*K*N/us/1e6:.1f} TOPS")
