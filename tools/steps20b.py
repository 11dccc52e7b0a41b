import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, QGTC
M = K = 4096; N = 64
A = (torch.rand((M, K)) < 0.5).float().cuda(); X = torch.randint(0, 2, (K, N)).float().cuda()
ba, bx = QGTC.val2bit(A, 1, False, False), QGTC.val2bit(X, 1, True, False)
out = QGTC.bitMM2Bit(ba, bx, M, K, N, 1, 1, 1)
eager = lambda n: QGTC.bitMM2Bit_enqueue(out, ba, bx, M, K, N, 1, 1, 1, n)
def shot(pre, idle):
    eager(5); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if pre: e0.record(); e1.record()
    torch.cuda.synchronize()
    if idle: time.sleep(idle)
    t0 = time.perf_counter(); e0.record(); eager(20); e1.record(); torch.cuda.synchronize(); t1 = time.perf_counter()
    return (t1 - t0) * 1e6, e0.elapsed_time(e1) * 1e3
for pre in (False, True):
    for idle in (0, 0.001, 0.05, 0.5):
        r = [shot(pre, idle) for _ in range(4)]
        print(f"pre-created events {pre}, idle {idle:5.3f} s before the region: wall us " + " ".join(f"{w:6.1f}" for w, _ in r) + "   events us " + " ".join(f"{e:6.1f}" for _, e in r))
print("--- warm-up duration before (5 warmup launches, sync, 20 timed launches):")
def shot2(warm_s, chunk):
    time.sleep(0.3)                       # the host builds inputs: the chip idles
    t = time.perf_counter()
    while time.perf_counter() - t < warm_s:
        eager(chunk); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); e1.record()
    eager(5); torch.cuda.synchronize(); torch.cuda.synchronize()
    t0 = time.perf_counter(); e0.record(); eager(20); e1.record(); torch.cuda.synchronize(); t1 = time.perf_counter()
    return (t1 - t0) * 1e6, e0.elapsed_time(e1) * 1e3
for warm_s in (0.0, 0.02, 0.1, 0.5, 2.0):
    for chunk in (200, 2000):
        r = [shot2(warm_s, chunk) for _ in range(4)]
        print(f"warm-up {warm_s:4.2f} s in chunks of {chunk:4d}: wall us " + " ".join(f"{w:6.1f}" for w, _ in r) + "   events us " + " ".join(f"{e:6.1f}" for _, e in r))
