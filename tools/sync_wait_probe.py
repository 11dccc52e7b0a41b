"""Does the host's wait policy move the K = 20 region? `hipSetDeviceFlags(hipDeviceScheduleSpin / Yield / BlockingSync)` set through ctypes
BEFORE torch touches the GPU (argv[1] = auto | spin | yield | block), then the contract's region: synchronize, 20 launches, synchronize."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mode = sys.argv[1] if len(sys.argv) > 1 else "auto"
flag = {"auto": 0, "spin": 1, "yield": 2, "block": 4}[mode]
hip = ctypes.CDLL("libamdhip64.so")
rc = hip.hipSetDeviceFlags(ctypes.c_uint(flag)) if flag else 0
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
res = {}
for k in (1, 20, 200):
    ts = []
    for _ in range(60):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        enq(k)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    res[k] = ts[30] * 1e6
print(f"wait policy {mode:5s} (hipSetDeviceFlags rc {rc}): K=1 {res[1]:.1f} us, K=20 {res[20]:.1f} us ({2.0*M*K*N*20/res[20]/1e6:.0f} TOPS), K=200 {res[200]:.1f} us")
