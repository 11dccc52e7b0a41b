"""The C oracle against two independent restatements: the closed-form NumPy version and the
warp-level emulation of the reference kernels' data movement (oracle/warp_emulation.py)."""
import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

from helpers import edge_floats, rand_q
from oracle import warp_emulation as we
from oracle.qgtc_oracle import (P8, P128, np_acc_from_values, np_pack_cols, np_pack_rows,
                                np_quantize, np_requant, np_unpack_cols, np_unpack_rows)


@settings(max_examples=40, deadline=None)
@given(H=st.integers(1, 70), W=st.integers(1, 140), nbits=st.integers(1, 8), seed=st.integers(0, 2 ** 16))
def test_quantize_and_pack_three_ways(oracle, H, W, nbits, seed):
    rng = np.random.default_rng(seed)
    x = edge_floats(rng, H, W, nbits)
    q = oracle.quantize(x, nbits)
    np.testing.assert_array_equal(q, np_quantize(x, nbits))
    rows = oracle.pack(q, nbits, False)
    cols = oracle.pack(q, nbits, True)
    np.testing.assert_array_equal(rows, np_pack_rows(q, nbits).reshape(-1))
    np.testing.assert_array_equal(cols, np_pack_cols(q, nbits).reshape(-1))
    np.testing.assert_array_equal(rows, oracle.val2bit(x, nbits, False))
    np.testing.assert_array_equal(cols, oracle.val2bit(x, nbits, True))
    qm = q & ((1 << nbits) - 1)
    np.testing.assert_array_equal(oracle.bit2val(rows, nbits, H, W, False), qm)
    np.testing.assert_array_equal(oracle.bit2val(cols, nbits, H, W, True), qm)
    np.testing.assert_array_equal(np_unpack_rows(rows, nbits, H, W), qm)
    np.testing.assert_array_equal(np_unpack_cols(cols, nbits, H, W), qm)
    colo = oracle.pack(q, nbits, True, True)
    np.testing.assert_array_equal(colo, np_pack_cols(q, nbits, True).reshape(-1))
    np.testing.assert_array_equal(oracle.bit2val(colo, nbits, H, W, True, True), qm)


@pytest.mark.parametrize("H,W,nbits", [(5, 7, 1), (37, 130, 3), (130, 37, 2), (8, 128, 8), (129, 129, 4)])
def test_pack_matches_warp_emulation(oracle, H, W, nbits):
    rng = np.random.default_rng(H + W)
    q = rand_q(rng, H, W, nbits)
    np.testing.assert_array_equal(oracle.pack(q, nbits, False), we.emu_pack_rows(q, nbits))
    np.testing.assert_array_equal(oracle.pack(q, nbits, True), we.emu_pack_cols(q, nbits))


@pytest.mark.parametrize("M,K,N,a,w,ob", [(3, 3, 3, 2, 2, 2), (9, 130, 17, 1, 2, 2), (17, 40, 9, 2, 3, 3),
                                           (8, 128, 8, 1, 1, 1), (20, 300, 33, 3, 2, 4), (12, 257, 8, 1, 4, 1)])
def test_bitmm_matches_warp_emulation(oracle, M, K, N, a, w, ob):
    rng = np.random.default_rng(M * K + N)
    qx, qw = rand_q(rng, M, K, a), rand_q(rng, K, N, w)
    X, Wt = oracle.pack(qx, a, False), oracle.pack(qw, w, True)
    acc = oracle.acc(X, Wt, M, K, N, a, w)
    np.testing.assert_array_equal(acc, np_acc_from_values(qx, qw, a, w))
    out = oracle.bitmm2bit(X, Wt, M, K, N, a, w, ob)
    np.testing.assert_array_equal(out, np_pack_rows(np_requant(acc, ob), ob).reshape(-1))
    np.testing.assert_array_equal(out, we.emu_bitmm2bit(X, Wt, M, K, N, a, w, ob))
    np.testing.assert_array_equal(oracle.bitmm2bit(X, Wt, M, K, N, a, w, ob, col=True),
                                  np_pack_cols(np_requant(acc, ob), ob).reshape(-1))
    np.testing.assert_array_equal(oracle.bitmm2int(X, Wt, M, K, N, a, w, True),
                                  we.emu_bitmm2int(X, Wt, M, K, N, a, w, True))
    W8 = oracle.pack(qw, w, True, True)
    np.testing.assert_array_equal(oracle.bitmm2int(X, W8, M, K, N, a, w, False), acc.astype(np.float32))
    if w == 1 or P8(N) == P128(N):  # where the reference's PAD8 path is in-bounds
        np.testing.assert_array_equal(oracle.bitmm2int(X, W8, M, K, N, a, w, False),
                                      we.emu_bitmm2int(X, W8, M, K, N, a, w, False))


@settings(max_examples=25, deadline=None)
@given(M=st.integers(1, 40), K=st.integers(1, 300), N=st.integers(1, 40), a=st.integers(1, 4),
       w=st.integers(1, 4), ob=st.integers(1, 8), seed=st.integers(0, 2 ** 16))
def test_bitmm_is_integer_matmul(oracle, M, K, N, a, w, ob, seed):
    rng = np.random.default_rng(seed)
    qx, qw = rand_q(rng, M, K, a), rand_q(rng, K, N, w)
    X, Wt = oracle.pack(qx, a, False), oracle.pack(qw, w, True)
    c = (qx.astype(np.int64) @ qw.astype(np.int64))
    np.testing.assert_array_equal(oracle.bitmm2int(X, Wt, M, K, N, a, w, True), c.astype(np.float32))
    dec = oracle.bit2val(oracle.bitmm2bit(X, Wt, M, K, N, a, w, ob), ob, M, N, False)
    clamp = np.where(c > 2 ** ob, 2 ** ob - 1, c) & (2 ** ob - 1)
    np.testing.assert_array_equal(dec, clamp)


def test_requant_three_ways(oracle):
    for c in [-5, 0, 1, 3, 4, 5, 255, 256, 257, 2 ** 24 + 1, 2 ** 31 - 1, -2 ** 31]:
        for ob in [1, 2, 8, 24, 25]:
            assert oracle.requant(c, ob) == we._requant_ref(c, ob) == int(np_requant(np.int32(c), ob)), (c, ob)


def test_bounds_safe_reads(oracle):
    """Short / mis-laid operands read as zero past their extent (the reference reads raw memory)."""
    rng = np.random.default_rng(1)
    M, K, N, a, w = 20, 200, 12, 1, 2
    qx, qw = rand_q(rng, M, K, a), rand_q(rng, K, N, w)
    X, Wt = oracle.pack(qx, a, False), oracle.pack(qw, w, True)
    full = oracle.acc(X, Wt, M, K, N, a, w)
    cut = oracle.acc(X, Wt[: Wt.size // 2], M, K, N, a, w)  # plane 1 missing -> only plane 0 counts
    np.testing.assert_array_equal(cut, np_acc_from_values(qx, qw & 1, a, 1))
    assert (full != cut).any()


def test_tile_counters_sparse(oracle):
    rng = np.random.default_rng(2)
    M, K, N, a, w = 50, 700, 20, 2, 3
    qx = rand_q(rng, M, K, a, density=0.002)
    X = oracle.pack(qx, a, False)
    total, nz = oracle.tile_counters(X, M, K, N, a, w)
    planes = np_pack_rows(qx, a)
    cnt = 0
    for p in range(a):
        for bx in range(P8(M) // 8):
            for i in range(planes.shape[2] // 4):
                cnt += int(planes[p, bx * 8:bx * 8 + 8, i * 4:i * 4 + 4].any())
    assert nz == cnt * ((N + 7) // 8) * w and 0 < nz < total
