"""Single launches over plane counts and shapes on the default engine: time per launch and the kernel that served it (a scan for slow spots).
usage: route_sweep.py [cols]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import QGTC as Q
import qgtc_ppopp22_amd
L = ctypes.CDLL(qgtc_ppopp22_amd.lib_path())
L.qgtc_bitmm_route.restype = ctypes.c_char_p
cols = len(sys.argv) > 1 and sys.argv[1] == "cols"
for a, w in ((1, 1), (1, 4), (2, 2), (2, 8), (3, 3), (4, 4), (5, 5), (6, 2), (8, 1), (8, 8)):
    for M, K, N in ((599, 50, 64), (1213, 128, 128), (1213, 1213, 128), (1213, 1213, 10), (4096, 4096, 64), (4096, 4096, 256), (2048, 2048, 512)):
        X = Q.val2bit(torch.rand(M, K, device="cuda") * (1 << a), a, False, False)
        W = Q.val2bit(torch.rand(K, N, device="cuda") * (1 << w), w, True, False)
        ob = min(w, 8)
        out = (Q.bitMM2Bit_col if cols else Q.bitMM2Bit)(X, W, M, K, N, a, w, ob)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(3):
            e0.record()
            Q.bitMM2Bit_enqueue(out, X, W, M, K, N, a, w, ob, 100, cols)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / 100)
        route = L.qgtc_bitmm_route(M, K, N, a, w, ob, 1 if cols else 0, 0x10).decode()
        print(f"a={a} w={w} {M}x{K}x{N}{' cols' if cols else ''}: {best:7.2f} us  {2.0 * M * K * N / best / 1e6:8.1f} eff TOPS  {route}", flush=True)
