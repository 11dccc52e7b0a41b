"""Which kernel serves which call: the markdown table of DESIGN.md section 5, generated from the library's own rule functions
(qgtc_bitmm_route / qgtc_bitmm_batched_route = the functions the launchers switch on). No GPU needed.

    python tools/routing_table.py            # prints the table
"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import qgtc_ppopp22_amd

ENGINE = {"popcount": 0x0, "mfma": 0x8, "auto": 0x10}
ZERO_JUMP = 0x4
MODES = {0: "bits, rows", 1: "bits, cols", 2: "float32"}


def lib():
    L = ctypes.CDLL(qgtc_ppopp22_amd.lib_path())
    for f in (L.qgtc_bitmm_route, L.qgtc_bitmm_batched_route):
        f.restype = ctypes.c_char_p
    return L


SINGLE = [  # (what, M, K, N, a, w, ob, mode)
    ("micro-benchmark 2_7c, 1 bit (bench.py's step)", 4096, 4096, 64, 1, 1, 1, 0),
    ("micro-benchmark, 8-bit features", 4096, 4096, 64, 1, 8, 8, 0),
    ("micro-benchmark, smallest shape", 1024, 1024, 16, 1, 1, 1, 0),
    ("5_9 adjacency study, N = 256", 4096, 4096, 256, 1, 1, 1, 0),
    ("5_9 adjacency study, N = 1024", 4096, 4096, 1024, 1, 1, 1, 0),
    ("5_9's largest K (32768: float32 sums still exact at 1 bit)", 32768, 32768, 64, 1, 1, 1, 0),
    ("5_9 at 16384, N = 256 (four column tiles of the long-K kernel)", 16384, 16384, 256, 1, 1, 1, 0),
    ("5_9's largest shape (QGTC_module/logs/profile_new.log:26)", 32768, 32768, 1024, 1, 1, 1, 0),
    ("long K, few rows", 4096, 8192, 64, 1, 1, 1, 0),
    ("long K, two-bit features", 32768, 32768, 64, 1, 2, 2, 0),
    ("per-batch X.W of the arxiv epoch (main_qgtc.py:147)", 1213, 128, 128, 2, 2, 2, 0),
    ("per-batch A.(XW) (main_qgtc.py:148)", 1213, 1213, 128, 1, 2, 2, 0),
    ("per-batch bitMM2Bit_col (layout-correct chain)", 1213, 128, 128, 2, 2, 2, 1),
    ("per-batch bitMM2Int, 10 classes", 1213, 1213, 10, 1, 2, 1, 2),
    ("ppi GIN per-batch, 4 x 4 bits", 599, 50, 64, 4, 4, 4, 0),
    ("the same, layout-correct (bitMM2Bit_col)", 599, 50, 64, 4, 4, 4, 1),
    ("per-batch X.W at --bit_width 8", 1213, 128, 128, 8, 8, 8, 0),
    ("4 x 4 bits, a 128-column right operand", 4096, 4096, 128, 4, 4, 4, 0),
    ("4 x 4 bits, 256 columns on big operands", 16384, 4096, 256, 4, 4, 4, 0),
    ("wide product, 2 x 2 bits", 8192, 4096, 1024, 2, 2, 2, 0),
    ("wide product, 4 x 4 bits (no wide kernel: 96 KB stage)", 4096, 4096, 1024, 4, 4, 4, 0),
    ("3-bit x 5-bit (no fixed-plane kernel)", 1000, 1000, 200, 3, 5, 4, 0),
    ("9 planes (beyond every matrix-core form)", 512, 512, 64, 9, 2, 4, 0),
    ("8 x 8 bits, long K (float32 sums inexact)", 4096, 4096, 64, 8, 8, 8, 0),
]
GROUPED = [  # (what, max_M, max_K, max_N, a, w, ob, mode, extra flags)
    ("epoch X.W stage, 2 bits, cols out", 1213, 128, 128, 2, 2, 2, 1, 0),
    ("epoch A.(XW) stage with bitmaps", 1213, 1213, 128, 1, 2, 2, 0, ZERO_JUMP),
    ("epoch A.(XW) stage, dense (no bitmaps)", 1213, 1213, 128, 1, 2, 2, 0, 0),
    ("class-count stage, rows out", 1213, 128, 10, 2, 2, 2, 0, 0),
    ("float32 aggregation, 10 classes", 1213, 1213, 10, 1, 2, 1, 2, ZERO_JUMP),
    ("ppi X.W stage, 4 x 4 bits", 599, 50, 64, 4, 4, 4, 1, 0),
    ("literal chain: rows-layout X.W (main_qgtc.py:147 grouped)", 1213, 128, 128, 2, 2, 2, 0, 0),
    ("wide grouped stage (N = 512)", 1213, 1213, 512, 1, 2, 2, 0, 0),
    ("8-bit grouped X.W", 1213, 128, 128, 8, 8, 8, 1, 0),
    ("3-bit grouped X.W (cols out, no fixed-shape kernel)", 1213, 128, 128, 3, 3, 3, 1, 0),
]


def table():
    L = lib()
    out = ["| call shape (M x K x N, a x w -> ob, output) | engine `popcount` | engine `auto` (default) | engine `mfma` |", "|---|---|---|---|"]
    for what, M, K, N, a, w, ob, mode in SINGLE:
        r = [L.qgtc_bitmm_route(M, K, N, a, w, ob, mode, f).decode() for f in ENGINE.values()]
        out.append(f"| single: {what}: {M} x {K} x {N}, {a} x {w} -> {ob if mode != 2 else '-'}, {MODES[mode]} | `{r[0]}` | `{r[2]}` | `{r[1]}` |")
    for what, M, K, N, a, w, ob, mode, extra in GROUPED:
        r = [L.qgtc_bitmm_batched_route(M, K, N, a, w, ob, mode, f | extra).decode() for f in ENGINE.values()]
        out.append(f"| grouped: {what}: <= {M} x {K} x {N}, {a} x {w} -> {ob if mode != 2 else '-'}, {MODES[mode]} | `{r[0]}` | `{r[2]}` | `{r[1]}` |")
    return "\n".join(out)


if __name__ == "__main__":
    print(table())
