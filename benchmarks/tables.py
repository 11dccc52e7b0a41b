"""The reference's benchmark tables on MI355X (extras file only): 2_7c micro-benchmark, 5_9 adjacency-size study, the width
sweep of the headline shape, Fig. 8a's int8 comparison, wide products per engine, independent launches on two streams.
200 launches per point between two events (QGTC_device.cu:403-422), median of five windows; README's sm_86 numbers beside
them."""
from __future__ import annotations

import torch

from .common import REF_TFLOPS_4096_64, engine, make_workload, median_of_5

REF_MICRO = {   # effective TFLOPs (sm_86) for (M = K, N) at widths 1 / 2 / 4 / 8
    (1024, 16): (5.847, 3.934, 2.488, 1.541), (2048, 16): (16.605, 10.086, 6.561, 3.483), (4096, 16): (40.627, 20.764, 12.409, 6.763),
    (1024, 32): (11.724, 7.864, 4.456, 3.074), (2048, 32): (32.666, 19.762, 12.807, 6.816), (4096, 32): (35.032, 20.951, 13.929, 7.366),
    (1024, 64): (23.219, 15.429, 10.683, 5.046), (2048, 64): (37.438, 25.055, 12.328, 6.165), (4096, 64): (46.768, 26.818, 14.196, 7.324),
}
REF_ADJ = {16: (5.831, 16.323, 34.425), 32: (11.717, 32.027, 40.175), 64: (23.158, 37.444, 46.759), 128: (28.417, 40.646, 52.517),
           256: (32.089, 44.151, 59.508), 512: (41.743, 49.687, 64.172), 1024: (37.954, 52.970, 66.490)}   # BASELINE.md §3, 1 bit
REF_CUBLAS_INT8 = {(1024, 16): 0.55, (2048, 16): 2.58, (4096, 16): 3.60, (1024, 32): 3.89, (2048, 32): 5.49,
                   (4096, 32): 6.49, (1024, 64): 4.38, (2048, 64): 6.30, (4096, 64): 6.65}


def tops(M, K, N, reps, ms):
    return round(2.0 * M * K * N * reps / (ms * 1e-3) / 1e12, 2)


def identical_and_closed_form(Q, words, M, K, N, w):
    """Per-point parity flag: the default engine's packed words equal the AND + popcount kernels' AND decode to the closed
    form of the all-ones inputs (C = K everywhere, re-quantised: 2^w - 1 where K > 2^w; 2_7c_QGTC_GEMM_INT8.py:30-41).
    Full-size comparison with the oracle: tests/test_gpu_fullsize.py."""
    same = torch.equal(words["auto"], words["popcount"])
    back = Q.bit2val(words["auto"], w, M, N, False, False)
    want = (2 ** w - 1) if K > 2 ** w else (K & (2 ** w - 1))
    return bool(same and bool((back == want).all().item()))


def both_engines(Q, ba, bx, M, K, N, w):
    cell, words = {}, {}
    for eng, key in (("auto", "TOPS"), ("popcount", "TOPS_engine_popcount")):
        with engine(Q, eng):
            cell[key] = tops(M, K, N, 200, median_of_5(Q, ba, bx, M, K, N, w))
            words[eng] = Q.bitMM2Bit(ba, bx, M, K, N, 1, w, w)
    cell["identical"] = identical_and_closed_form(Q, words, M, K, N, w)
    return cell


def micro_bench_table(Q, device):
    """2_7c_QGTC_GEMM_INT8.py:13-20: 9 shapes x widths 1 .. 8 (README publishes 1 / 2 / 4 / 8), all-ones inputs as there."""
    out, published = {}, {1: 0, 2: 1, 4: 2, 8: 3}
    for (mk, nn), ref in REF_MICRO.items():
        row = {}
        for ww in range(1, 9):
            _, _, ba, bx = make_workload(Q, mk, mk, nn, ww, device, seed=3, ones=True)
            row[f"w{ww}"] = dict(ref_sm86=ref[published[ww]] if ww in published else None, **both_engines(Q, ba, bx, mk, mk, nn, ww))
        out[f"{mk}x{mk}x{nn}"] = row
    return out


ADJ_SIZES = (128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768)   # 5_9_adjmatrix_size.py:15: M = K = 2^7 .. 2^15


def adj_size_table(Q, device):
    """5_9_adjmatrix_size.py:15-18: 1 bit, M = K = 2^7 .. 2^15, N = 16 .. 1024, all-ones inputs made on the device (README publishes
    1024 / 2048 / 4096; QGTC_module/logs/profile_new.log:26 records m = k = 32768, n = 1024). Per cell: both engines, and where the
    default engine's launch stands against the two rooflines that can bind it - the HBM one on the call's algorithmic bytes
    (M K / 8 + K N / 8 + M N / 8: it is the bound from M = K = 16384 up at N <= 64, where the adjacency alone is 32 - 128 MiB) and
    the dense FP4 matrix-core one (2 M K N / 10 PFLOP/s: the bound of the wide columns)."""
    from .common import FP4_PEAK_TFLOPS, HBM_PEAK_GBS
    out = {}
    for nn, ref in REF_ADJ.items():
        row = {}
        for mk in ADJ_SIZES:
            ba = Q.val2bit(torch.ones((mk, mk), device=device), 1, False, False)
            bx = Q.val2bit(torch.ones((mk, nn), device=device), 1, True, False)
            published = {1024: 0, 2048: 1, 4096: 2}
            cell = dict(ref_sm86=ref[published[mk]] if mk in published else None, **both_engines(Q, ba, bx, mk, mk, nn, 1))
            us = 2.0 * mk * mk * nn / (cell["TOPS"] * 1e6)
            cell["us_per_launch"] = round(us, 3)
            cell["hbm_frac"] = round((mk * mk / 8 + mk * nn / 8 + mk * nn / 8) / (us * 1e-6) / (HBM_PEAK_GBS * 1e9), 4)
            cell["fp4_frac"] = round(cell["TOPS"] / FP4_PEAK_TFLOPS, 4)
            row[f"M{mk}"] = cell
            del ba, bx
        out[f"N{nn}"] = row
    return out


def width_sweep(Q, device, M, K, N):
    sweep = {}
    for ww in (1, 2, 4, 8):
        for label, ones in (("random", False), ("ones", True)):
            _, _, ba, bx = make_workload(Q, M, K, N, ww, device, seed=3, ones=ones)
            cell = {"ref_sm86_TFLOPs": REF_TFLOPS_4096_64[ww]}
            for eng, key in (("auto", "TOPS"), ("popcount", "TOPS_engine_popcount")):
                with engine(Q, eng):
                    ms = median_of_5(Q, ba, bx, M, K, N, ww)
                cell[key] = tops(M, K, N, 200, ms)
                if eng == "auto":
                    cell["us_per_launch"] = round(ms * 1e3 / 200, 3)
            sweep[f"w{ww}_{label}"] = cell
    return sweep


def two_streams(Q, device, M, K, N):
    """Independent launches (different cluster batches in a serving loop) need not be serialised by stream order: the same
    products round-robin on two HIP streams, each with its own output buffer. NOT the headline metric."""
    ovl = {}
    for ww in (1, 2, 4, 8):
        _, _, ba, bx = make_workload(Q, M, K, N, ww, device, seed=3)
        with engine(Q, "popcount"):
            ref_out = Q.bitMM2Bit(ba, bx, M, K, N, 1, ww, ww)
        outs2 = [torch.empty_like(ref_out) for _ in range(2)]
        Q.bitMM2Bit_enqueue_streams(outs2, ba, bx, M, K, N, 1, ww, ww, 50)
        torch.cuda.synchronize()
        best = None
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            Q.bitMM2Bit_enqueue_streams(outs2, ba, bx, M, K, N, 1, ww, ww, 1000)
            e1.record()
            torch.cuda.synchronize()
            best = e0.elapsed_time(e1) if best is None else min(best, e0.elapsed_time(e1))
        ovl[f"w{ww}"] = {"TOPS": tops(M, K, N, 1000, best), "us_per_launch": round(best, 3),
                         "outputs_identical": bool(all(torch.equal(o, ref_out) for o in outs2))}
    return ovl


def int8_comparison(Q, device):
    """The reference's Fig. 8a: INT8 GEMM on the matrix cores (k_i8gemm, and the vendor library - hipBLASLt through
    torch._int_mm - where the reference used cuBLAS) beside the 1-bit path on the same nine shapes."""
    cmp9 = {}
    g = torch.Generator(device="cpu").manual_seed(5)
    for nn in (16, 32, 64):
        for mk in (1024, 2048, 4096):
            A8 = torch.randint(-128, 128, (mk, mk), generator=g, dtype=torch.int8).to(device)
            B8 = torch.randint(-128, 128, (nn, mk), generator=g, dtype=torch.int8).to(device)
            Q.i8gemm_profile(A8, B8, 20, False)
            ms8 = min(Q.i8gemm_profile(A8, B8, 200, False) for _ in range(3))
            lib8 = None
            try:
                B8kn = B8.t().contiguous()
                for _ in range(5):
                    torch._int_mm(A8, B8kn)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(200):
                    torch._int_mm(A8, B8kn)
                e1.record()
                torch.cuda.synchronize()
                lib8 = tops(mk, mk, nn, 200, e0.elapsed_time(e1))
            except Exception:   # noqa: BLE001 - optional leg
                lib8 = None
            _, _, ba, bx = make_workload(Q, mk, mk, nn, 1, device, seed=3)
            with engine(Q, "popcount"):
                ms1 = median_of_5(Q, ba, bx, mk, mk, nn, 1)
            ms1a = median_of_5(Q, ba, bx, mk, mk, nn, 1)
            cmp9[f"{mk}x{mk}x{nn}"] = {"int8_mfma_TOPS": tops(mk, mk, nn, 200, ms8), "int8_hipblaslt_TOPS": lib8,
                                      "bit1_popcount_TOPS": tops(mk, mk, nn, 200, ms1), "bit1_default_engine_TOPS": tops(mk, mk, nn, 200, ms1a),
                                      "ref_sm86_cublas_int8_TFLOPS": REF_CUBLAS_INT8[(mk, nn)], "ref_sm86_qgtc_1bit_TFLOPs": REF_MICRO[(mk, nn)][0]}
    return cmp9


def wide_products(Q, device):
    """The three engines on wide products, where an expanded operand feeds several MFMA tiles."""
    eng = {}
    for (mm, kk, nn, ww) in ((4096, 4096, 1024, 1), (4096, 4096, 1024, 2), (4096, 4096, 1024, 4), (8192, 4096, 1024, 1), (8192, 4096, 1024, 2)):
        _, _, ba, bx = make_workload(Q, mm, kk, nn, ww, device, seed=3)
        row, outs_e = {}, {}
        for name in ("popcount", "mfma", "auto"):
            with engine(Q, name):
                ms = median_of_5(Q, ba, bx, mm, kk, nn, ww, reps=50)
                outs_e[name] = Q.bitMM2Bit(ba, bx, mm, kk, nn, 1, ww, ww)
            row[name + "_TOPS"] = round(2.0 * mm * kk * nn * 50 / (ms * 1e-3) / 1e12, 1)
        row["outputs_identical"] = bool(torch.equal(outs_e["popcount"], outs_e["mfma"]) and torch.equal(outs_e["popcount"], outs_e["auto"]))
        eng[f"{mm}x{kk}x{nn}_w{ww}"] = row
    return eng


def all_tables(Q, device, M, K, N):
    return {"width_sweep_4096x4096x64": width_sweep(Q, device, M, K, N),
            "independent_launches_on_2_streams_4096x4096x64": two_streams(Q, device, M, K, N),
            "int8_mfma_vs_1bit_popcount_9_shapes": int8_comparison(Q, device),
            "micro_bench_ones_9_shapes_x_8_widths": micro_bench_table(Q, device),
            "adjacency_size_study_1bit": adj_size_table(Q, device),
            "mfma_engine_vs_popcount_wide_products": wide_products(Q, device)}
