"""The 20-step window of the driver's bench run under different host wait policies (ROC_ACTIVE_WAIT_TIMEOUT, set by the caller's
environment before HIP starts): wall and event microseconds of `e0; 20 launches; e1; synchronize`, eight shots after a
0.3 s warm-up."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, QGTC
M = K = 4096; N = 64
A = (torch.rand((M, K)) < 0.5).float().cuda(); X = torch.randint(0, 2, (K, N)).float().cuda()
ba, bx = QGTC.val2bit(A, 1, False, False), QGTC.val2bit(X, 1, True, False)
out = QGTC.bitMM2Bit(ba, bx, M, K, N, 1, 1, 1)
eager = lambda n: QGTC.bitMM2Bit_enqueue(out, ba, bx, M, K, N, 1, 1, 1, n)
def shot():
    t = time.perf_counter()
    while time.perf_counter() - t < 0.3:
        eager(200); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); e1.record()
    eager(5); torch.cuda.synchronize(); torch.cuda.synchronize()
    t0 = time.perf_counter(); e0.record(); eager(20); e1.record(); torch.cuda.synchronize(); t1 = time.perf_counter()
    return (t1 - t0) * 1e6, e0.elapsed_time(e1) * 1e3
r = [shot() for _ in range(8)]
print("ROC_ACTIVE_WAIT_TIMEOUT=%s: wall us " % os.environ.get("ROC_ACTIVE_WAIT_TIMEOUT") + " ".join(f"{w:6.1f}" for w, _ in r) + "   events us " + " ".join(f"{e:6.1f}" for _, e in r))
# no events inside the region
def shot_noev():
    t = time.perf_counter()
    while time.perf_counter() - t < 0.3:
        eager(200); torch.cuda.synchronize()
    eager(5); torch.cuda.synchronize(); torch.cuda.synchronize()
    t0 = time.perf_counter(); eager(20); torch.cuda.synchronize(); t1 = time.perf_counter()
    return (t1 - t0) * 1e6
print("   without the two event records: wall us " + " ".join(f"{shot_noev():6.1f}" for _ in range(8)))
