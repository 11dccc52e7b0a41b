"""The reference's own (print-only) test script replayed on the HIP path, call for call.

QGTC_module/unitest.py builds all-ones inputs and prints what comes back; it records no expected values, but every result
is derivable in closed form; tests/test_oracle_kat.py pins the ORACLE to those forms, this file
pushes the same literal call sequences through `import QGTC` - same functions, same positional arguments (including the
7-argument bitMM2Int form of unitest.py:72,79,143,146) - and asserts the closed forms AND word-for-word equality with the
oracle. Line numbers refer to QGTC_module/unitest.py. Below that: the wide points of 5_9_adjmatrix_size.py against the
oracle directly."""
import contextlib
import io
import re

import numpy as np
import pytest

from helpers import to_np_u32

pytestmark = pytest.mark.gpu


def _ones(torch, h, w):
    return torch.ones((h, w)).cuda()          # unitest.py:19,46-47,...: torch.ones(..).cuda()


@pytest.mark.parametrize("engine", ["auto", "popcount", "mfma"])
def test_unitest_py_call_sequences(qgtc, oracle, engine, capfd):
    import torch
    QGTC = qgtc
    QGTC.set_engine(engine)

    # ---- test_bitencodingAndDecoding(size=32, nbits=3)                                   unitest.py:18-40
    size, nbits = 32, 3
    A = _ones(torch, size, size)
    height, width = A.size(0), A.size(1)
    for col_major, output_layer in ((False, False), (True, False), (True, True)):          # :23-25, :29-30, :34-38
        bit_a = QGTC.val2bit(A, nbits, col_major, output_layer)
        A_val = QGTC.bit2val(bit_a, nbits, height, width, col_major, output_layer)
        assert A_val.dtype == torch.int32 and tuple(A_val.shape) == (32, 32) and bool((A_val == 1).all())
        np.testing.assert_array_equal(to_np_u32(bit_a), oracle.val2bit(np.ones((32, 32), np.float32), nbits, col_major, output_layer))
    assert tuple(bit_a.shape) == (3 * 1 * 4, 32)                                           # :36-37 prints [nbits * STEP128(H) * 4, PAD8(W)]

    # ---- TEST_bitMM2bit(M=T, K=T, N=T, nbits_a=2, nbits_b=2), T = 32                     unitest.py:45-57, :174-175
    T = 32
    A, B = _ones(torch, T, T), _ones(torch, T, T)
    bit_a = QGTC.val2bit(A, 2, False, False)
    bit_b = QGTC.val2bit(B, 2, True, False)
    bit_c = QGTC.bitMM2Bit(bit_a, bit_b, T, T, T, 2, 2, 2)
    C = QGTC.bit2val(bit_c, 2, T, T, False, False)
    assert bool((C == 3).all())                                                            # C = 32 > 2^2 -> 3 everywhere
    np.testing.assert_array_equal(to_np_u32(bit_c), oracle.bitmm2bit(to_np_u32(bit_a), to_np_u32(bit_b), T, T, T, 2, 2, 2))

    # ---- TEST_bitMM2Int(M=3, K=3, N=3, nbits_a=3, nbits_b=3)                             unitest.py:62-80, :171
    M = K = N = 3
    A, B = _ones(torch, M, K), _ones(torch, K, N)
    bit_a = QGTC.val2bit(A, 3, False, False)
    bit_b = QGTC.val2bit(B, 3, True, False)
    C = QGTC.bitMM2Int(bit_a, bit_b, M, K, N, 3, 3)                                        # :72 - SEVEN arguments
    assert C.dtype == torch.float32 and tuple(C.size()) == (3, 3) and bool((C == 3.0).all())
    bit_a = QGTC.val2bit(A, 3, False, True)                                                # :77-78 "output layer"
    bit_b = QGTC.val2bit(B, 3, True, True)
    C = QGTC.bitMM2Int(bit_a, bit_b, M, K, N, 3, 3)                                        # :79
    assert bool((C == 3.0).all())

    # ---- TEST_GINConv(M=3, K=3, N=3, N1=3, nbits_a=1, nbits_x=2, nbits_w=2)              unitest.py:126-147, :176
    M = K = N = 3
    A, X, W = _ones(torch, M, K), _ones(torch, K, N), _ones(torch, N, 3)
    bit_a = QGTC.val2bit(A, 1, False, False)
    bit_x = QGTC.val2bit(X, 2, True, False)
    bit_w = QGTC.val2bit(W, 2, True, False)
    bit_AX = QGTC.bitMM2Bit(bit_a, bit_x, M, K, N, 1, 2, 2)                                # :140
    assert bool((QGTC.bit2val(bit_AX, 2, M, N, False, False) == 3).all())                  # 3 <= 2^2 is kept
    int_AX = QGTC.bitMM2Int(bit_a, bit_x, M, K, N, 1, 2)                                   # :143 (7 arguments)
    assert bool((int_AX == 3.0).all())
    int_AXW = QGTC.bitMM2Int(bit_AX, bit_w, M, K, N, 2, 2)                                 # :146
    assert bool((int_AXW == 9.0).all())

    # ---- TEST_GCNConv(N=8, D=128, D1=8, nbits_a=1, nbits_x=2, nbits_w=2)                 unitest.py:86-110, :177
    N, D, D1 = 8, 128, 8
    X, W, A = _ones(torch, N, D), _ones(torch, D, D1), _ones(torch, N, N)
    bit_x = QGTC.val2bit(X, 2, False, False)
    bit_w = QGTC.val2bit(W, 2, True, False)
    bit_a = QGTC.val2bit(A, 1, False, False)
    bit_XW_col = QGTC.bitMM2Bit_col(bit_x, bit_w, N, D, D1, 1, 2, 2)                       # :100 - bit1 = nbits_a = 1 on a 2-bit X: plane 0 only
    assert tuple(bit_XW_col.shape) == (2 * 1 * 4, 128)                                     # QGTC_device.cu:456
    val_XW = QGTC.bit2val(bit_XW_col, 2, N, D1, True, False)                               # :106
    assert bool((val_XW == 3).all())                                                       # XW = 128 > 4 -> 3
    int_AXW = QGTC.bitMM2Int(bit_a, bit_XW_col, N, N, D1, 1, 2, True)                      # :109
    assert tuple(int_AXW.shape) == (8, 8) and bool((int_AXW == 24.0).all())
    np.testing.assert_array_equal(to_np_u32(bit_XW_col), oracle.bitmm2bit(to_np_u32(bit_x), to_np_u32(bit_w), N, D, D1, 1, 2, 2, col=True))

    # ---- PROFILE_NonZeroTile(N, N, dim, nbits_x=bitwidth)                                 unitest.py:158-167, :179-183 (one size of its sweep)
    capfd.readouterr()
    QGTC.reset_counters()
    expect_global = expect_nz = 0
    for bitwidth in (1, 2, 4, 8):
        for dim in (16, 32, 64):
            n = 1024
            A, X = _ones(torch, n, n), _ones(torch, n, dim)
            bit_a = QGTC.val2bit(A, 1, False, False)
            bit_x = QGTC.val2bit(X, bitwidth, True, False)
            QGTC.bitMM2Bit_base_cnt(bit_a, bit_x, n, n, dim, 1, bitwidth, bitwidth)         # :166
            QGTC.bitMM2Bit_zerojump_cnt(bit_a, bit_x, n, n, dim, 1, bitwidth, bitwidth)     # :167
            step = (n // 8) * ((dim + 7) // 8) * (n // 128) * bitwidth                      # S8(M) S8(N) S128(K) a w (kernel.h:452)
            expect_global += step
            expect_nz += step                                                               # all-ones A: every tile is non-zero
            assert QGTC.get_counters() == (expect_global, expect_nz)
    printed = capfd.readouterr().out                                                        # the lines parse_counter.py:19-25 greps
    g = [int(v) for v in re.findall(r"^counter_global: (-?\d+)$", printed, flags=re.M)]
    c = [int(v) for v in re.findall(r"^counter: (-?\d+)$", printed, flags=re.M)]
    assert len(g) == 12 and len(c) == 12 and g[-1] == expect_global and c[-1] == expect_nz and g == sorted(g)
    QGTC.reset_counters()


# 5_9_adjmatrix_size.py:9-12,15-18: 1-bit, all-ones, M = K = 2^T, N = dim in 16 .. 1024. The BASELINE tables quote M = 1024 /
# 2048 / 4096; the N <= 64 columns are covered at full size by test_gpu_fullsize.py - here the wide ones, oracle-direct.
@pytest.mark.parametrize("N", [128, 256, 512, 1024])
def test_adjacency_size_study_wide_points_equal_the_oracle(qgtc, oracle, N):
    import torch
    M = K = 4096
    A, X = torch.ones((M, K)).cuda(), torch.ones((K, N)).cuda()
    bit_a = qgtc.val2bit(A, 1, False, False)
    bit_x = qgtc.val2bit(X, 1, True, False)
    with contextlib.redirect_stdout(io.StringIO()):
        out = qgtc.bitMM2Bit_profile(bit_a, bit_x, M, K, N, 1, 1, 1)                      # 5_9_adjmatrix_size.py:12
    want = oracle.bitmm2bit(to_np_u32(bit_a), to_np_u32(bit_x), M, K, N, 1, 1, 1)
    np.testing.assert_array_equal(to_np_u32(out), want)
    # and a non-degenerate operand pair at the same shape (all-ones inputs make every accumulator equal K)
    g = torch.Generator().manual_seed(N)
    A = (torch.rand((M, K), generator=g) < 0.5).float().cuda()
    X = (torch.rand((K, N), generator=g) < 0.5).float().cuda()
    bit_a, bit_x = qgtc.val2bit(A, 1, False, False), qgtc.val2bit(X, 1, True, False)
    got = qgtc.bitMM2Int(bit_a, bit_x, M, K, N, 1, 1, True).cpu().numpy()
    np.testing.assert_array_equal(got, oracle.bitmm2int(to_np_u32(bit_a), to_np_u32(bit_x), M, K, N, 1, 1, True))
