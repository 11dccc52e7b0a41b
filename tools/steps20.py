"""What the driver's `bench.py --steps 20 --warmup 5` is made of: wall time of 20 launches between two synchronisations,
with and without the event records, eager and replayed from a captured graph. Run once per setting of the runtime's
wait policy (e.g. ROC_ACTIVE_WAIT_TIMEOUT=1000 python tools/steps20.py)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, QGTC

M = K = 4096; N = 64
A = (torch.rand((M, K)) < 0.5).float().cuda(); X = torch.randint(0, 2, (K, N)).float().cuda()
ba, bx = QGTC.val2bit(A, 1, False, False), QGTC.val2bit(X, 1, True, False)
out = QGTC.bitMM2Bit(ba, bx, M, K, N, 1, 1, 1)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20

def run(fn, events):
    best = 1e9
    for _ in range(20):
        fn(5) if fn is not None else None
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if events: e0.record()
        fn(steps)
        if events: e1.record()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best * 1e6

eager = lambda n: QGTC.bitMM2Bit_enqueue(out, ba, bx, M, K, N, 1, 1, 1, n)
print("env ROC_ACTIVE_WAIT_TIMEOUT =", os.environ.get("ROC_ACTIVE_WAIT_TIMEOUT"), " steps =", steps)
print(f"empty region (sync only)        : {run(lambda n: None, False):7.1f} us")
print(f"empty region + 2 event records  : {run(lambda n: None, True):7.1f} us")
print(f"eager, events                   : {run(eager, True):7.1f} us   ({2.0*M*K*N*steps/run(eager, True)/1e6:6.0f} TOPS)")
print(f"eager, no events                : {run(eager, False):7.1f} us")
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    eager(3)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    eager(steps)
rep = lambda n: g.replay() if n == steps else eager(n)
print(f"graph replay, events            : {run(rep, True):7.1f} us   ({2.0*M*K*N*steps/run(rep, True)/1e6:6.0f} TOPS)")
print(f"graph replay, no events         : {run(rep, False):7.1f} us")
