"""Known-answer tests that pin the oracle to the reference's own (print-only) tests.

QGTC_module/unitest.py builds all-ones inputs, so every result it would print is derivable in
closed form. These are the only "golden vectors" the reference holds: it records no
expected outputs. Line numbers below refer to QGTC_module/unitest.py.
"""
import numpy as np
import pytest

from oracle.qgtc_oracle import P8, P128, S8, S128


def ones(h, w):
    return np.ones((h, w), dtype=np.float32)


def test_kat1_encode_decode_roundtrip(oracle):
    """unitest.py:18-40 — ones(32,32), nbits=3: bit2val(val2bit(A)) == 1 for all three layouts."""
    A = ones(32, 32)
    for cm, ol in ((False, False), (True, False), (True, True)):
        bits = oracle.val2bit(A, 3, cm, ol)
        assert (oracle.bit2val(bits, 3, 32, 32, cm, ol) == 1).all()
    # shapes the reference would print at :38-39 for the (col, output) case
    assert oracle.cols_words(32, 32, 3, True) == 3 * S128(32) * 4 * P8(32)


def test_kat2_bitmm2bit_clamps(oracle):
    """unitest.py:45-57 with T=32, 2-bit everything (:175): C = 32 > 4 -> every element decodes to 3."""
    T = 32
    a = oracle.val2bit(ones(T, T), 2, False, False)
    b = oracle.val2bit(ones(T, T), 2, True, False)
    c = oracle.bitmm2bit(a, b, T, T, T, 2, 2, 2)
    assert (oracle.bit2val(c, 2, T, T, False, False) == 3).all()


def test_kat3_bitmm2int(oracle):
    """unitest.py:62-80 with M=K=N=3, 3-bit (:171): bitMM2Int == 3.0 for both packings."""
    a = oracle.val2bit(ones(3, 3), 3, False, False)
    b = oracle.val2bit(ones(3, 3), 3, True, False)
    assert (oracle.bitmm2int(a, b, 3, 3, 3, 3, 3, False) == 3.0).all()  # 7-arg call form: pad_128 defaults False
    a = oracle.val2bit(ones(3, 3), 3, False, True)
    b = oracle.val2bit(ones(3, 3), 3, True, True)
    assert (oracle.bitmm2int(a, b, 3, 3, 3, 3, 3, False) == 3.0).all()


def test_kat4_gin(oracle):
    """unitest.py:126-147 with 3x3, a=1, x=w=2 (:176): AX = 3 (kept by ob=2 since 3 <= 4),
    int_AX = 3.0, int_AXW = 9.0."""
    M = K = N = N1 = 3
    bit_a = oracle.val2bit(ones(M, K), 1, False, False)
    bit_x = oracle.val2bit(ones(K, N), 2, True, False)
    bit_w = oracle.val2bit(ones(N, N1), 2, True, False)
    bit_AX = oracle.bitmm2bit(bit_a, bit_x, M, K, N, 1, 2, 2)
    assert (oracle.bit2val(bit_AX, 2, M, N, False, False) == 3).all()
    assert (oracle.bitmm2int(bit_a, bit_x, M, K, N, 1, 2, False) == 3.0).all()
    assert (oracle.bitmm2int(bit_AX, bit_w, M, K, N, 2, 2, False) == 9.0).all()


def test_kat5_gcn(oracle):
    """unitest.py:86-110 with N=8, D=128, D1=8, a=1, x=w=2 (:177): bitMM2Bit_col is called with
    bit1 = nbits_a = 1 on the 2-bit-packed X, so only plane 0 is used: XW = 128 -> clamp 3;
    bitMM2Int(A, XWcol, 8, 8, 8, 1, 2, True) = 8 * 3 = 24.0."""
    N, D, D1 = 8, 128, 8
    bit_x = oracle.val2bit(ones(N, D), 2, False, False)
    bit_w = oracle.val2bit(ones(D, D1), 2, True, False)
    bit_a = oracle.val2bit(ones(N, N), 1, False, False)
    xw_col = oracle.bitmm2bit(bit_x, bit_w, N, D, D1, 1, 2, 2, col=True)
    assert (oracle.bit2val(xw_col, 2, N, D1, True, False) == 3).all()
    assert (oracle.bitmm2int(bit_a, xw_col, N, N, D1, 1, 2, True) == 24.0).all()


@pytest.mark.parametrize("bitwidth", [1, 2, 4, 8])
@pytest.mark.parametrize("dim", [16, 32, 64])
def test_kat6_counters(oracle, bitwidth, dim):
    """unitest.py:158-183 — per call counter_global grows by S8(M)*S8(N)*S128(K)*a*w and, with an
    all-ones A, counter grows by the same (kernel.h:452,587)."""
    n = 256  # the reference sweeps 1024..4096; the closed form is size-independent
    bit_a = oracle.val2bit(ones(n, n), 1, False, False)
    total, nz = oracle.tile_counters(bit_a, n, n, dim, 1, bitwidth)
    assert total == S8(n) * S8(dim) * S128(n) * bitwidth
    assert nz == total


@pytest.mark.parametrize("w", [1, 2, 4, 8])
def test_kat7_microbench_invariant(oracle, w):
    """2_7c_QGTC_GEMM_INT8.py:6-12 — ones inputs: C = K everywhere, so with ob = w the packed
    output decodes to clamp(K) = 2^w - 1 in range and 0 in the padding."""
    M = K = 384  # > 2^8, so that C = K clamps for every w (C == 2^w would be kept and pack as 0)
    N = 16
    bit_a = oracle.val2bit(ones(M, K), 1, False, False)
    bit_x = oracle.val2bit(ones(K, N), w, True, False)
    out = oracle.bitmm2bit(bit_a, bit_x, M, K, N, 1, w, w).reshape(w, P8(M), S128(N) * 4)
    assert (oracle.bit2val(out, w, M, N, False, False) == 2 ** w - 1).all()
    assert (out[:, :, 1:] == 0).all() and (out[:, :, 0] == 0xFFFF0000).all()


def test_quantizer_table(oracle):
    """kernel.h:39-44,66-68 worked example from SURVEY.md §8 a1 (b = 1)."""
    x = np.array([[-1, -0.0, 0, .5, .51, 1, 1.5, 2, 2.0001, 100]], dtype=np.float32)
    q = oracle.quantize(x, 1)
    assert ((q & 1).reshape(-1) == np.array([1, 0, 0, 0, 1, 1, 0, 0, 1, 1])).all()
    assert oracle.quantize(np.array([[np.nan]], np.float32), 4)[0, 0] == 0
    assert oracle.quantize(np.array([[2.5, 3.5, 16.0, 16.5]], np.float32), 4).tolist() == [[2, 4, 16, 15]]


def test_requant_rule(oracle):
    """kernel.h:31-37 as called at :350: NOT min(c, 2^ob - 1)."""
    assert [oracle.requant(c, 2) for c in (0, 3, 4, 5, 1000, -7)] == [0, 3, 4, 3, 3, 1]
