"""Which golden bit-GEMM fixture an engine gets wrong, and where (diagnostic)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import QGTC
from helpers import to_dev, to_np_u32
from qgtc_ppopp22_amd.shapes import cols_shape, rows_shape
G = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "qgtc_golden.npz"))
QGTC.set_engine(sys.argv[1] if len(sys.argv) > 1 else "mfma")
for i in range(int(G["mm_count"])):
    M, K, N, a, w, ob = (int(v) for v in G[f"mm{i}_meta"])
    dX = to_dev(torch, G[f"mm{i}_X"], rows_shape(M, K, a))
    dW = to_dev(torch, G[f"mm{i}_W"], cols_shape(K, N, w))
    dW8 = to_dev(torch, G[f"mm{i}_W8"], cols_shape(K, N, w, True))
    for name, got, want in (("bits", to_np_u32(QGTC.bitMM2Bit(dX, dW, M, K, N, a, w, ob)), G[f"mm{i}_bits"]),
                            ("bits_col", to_np_u32(QGTC.bitMM2Bit_col(dX, dW, M, K, N, a, w, ob)), G[f"mm{i}_bits_col"]),
                            ("f32_pad128", QGTC.bitMM2Int(dX, dW, M, K, N, a, w, True).cpu().numpy(), G[f"mm{i}_f32_pad128"]),
                            ("f32_pad8", QGTC.bitMM2Int(dX, dW8, M, K, N, a, w, False).cpu().numpy(), G[f"mm{i}_f32_pad8"])):
        bad = np.argwhere(got.reshape(want.shape) != want)
        print(i, (M, K, N, a, w, ob), name, "ok" if bad.size == 0 else f"MISMATCH at {bad[:8].tolist()} got {got.reshape(want.shape)[tuple(bad[0])]} want {want[tuple(bad[0])]} ({len(bad)} of {want.size})")
