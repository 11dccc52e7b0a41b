"""Host time of the eager issue loop: 200 launches of the headline kernel through Q.bitMM2Bit_enqueue (C-ABI call per launch), host clock
from the call to its return (nothing waited for), beside the device time of the same 200. Run tools/kernarg_probe.hip on the same box for
the bare hipLaunchKernel figure."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import QGTC as Q

M = K = 4096
N = 64
dev = torch.device("cuda:0")
A = (torch.rand(M, K, device=dev) < 0.5).float()
X = (torch.rand(K, N, device=dev) < 0.5).float()
bit_A = Q.val2bit(A, 1, False, False)
bit_X = Q.val2bit(X, 1, True, False)
out = Q.bitMM2Bit(bit_A, bit_X, M, K, N, 1, 1, 1)
enq = lambda n: Q.bitMM2Bit_enqueue(out, bit_A, bit_X, M, K, N, 1, 1, 1, n)
for _ in range(100):
    enq(200)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
host, devt = [], []
for _ in range(30):
    torch.cuda.synchronize()
    e0.record()
    t0 = time.perf_counter()
    enq(200)
    t1 = time.perf_counter()
    e1.record()
    torch.cuda.synchronize()
    host.append((t1 - t0) / 200)
    devt.append(e0.elapsed_time(e1) * 1e-3 / 200)
host.sort(); devt.sort()
print(f"bitMM2Bit_enqueue: host {host[15]*1e6:.3f} us / launch (min {host[0]*1e6:.3f}), device {devt[15]*1e6:.3f} us / launch (min {devt[0]*1e6:.3f})")
