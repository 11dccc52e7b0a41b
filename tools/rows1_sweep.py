"""Single products with at most 256 columns: the row-block kernel (k_bitmm_fp4_rows_single) against the other single-launch routes.
usage: rows1_sweep.py a w [cols]; run as is and with QGTC_NO_ROWS1=1 (the routes behind it)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import QGTC as Q
a, w = int(sys.argv[1]), int(sys.argv[2])
cols = len(sys.argv) > 3 and sys.argv[3] == "cols"
tag = ("NO_ROWS1" if os.environ.get("QGTC_NO_ROWS1") else "rows1   ") + f" a={a} w={w}" + (" cols" if cols else "")
for M in (599, 1213, 4096, 16384):
    for N in (16, 64, 128, 256):
        line = []
        for K in (128, 512, 1024, 2048, 4096, 8192):
            if K > max(M, 1213):
                continue
            X = Q.val2bit(torch.rand(M, K, device="cuda") * (1 << a), a, False, False)
            W = Q.val2bit(torch.rand(K, N, device="cuda") * (1 << w), w, True, False)
            out = (Q.bitMM2Bit_col if cols else Q.bitMM2Bit)(X, W, M, K, N, a, w, w)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            best = 1e9
            for _ in range(3):
                e0.record()
                Q.bitMM2Bit_enqueue(out, X, W, M, K, N, a, w, w, 200, cols)
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) * 1e3 / 200)
            line.append(f"K={K}: {best:.2f}")
        print(tag, f"M={M} N={N}  " + "  ".join(line), flush=True)
