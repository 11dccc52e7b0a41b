"""GPU parity on the accumulator's EDGE domain (VERDICT r5 missing 2): what the reference's checked-in
script launches (0_7a_eval_QGTC_cluster_GCN.py:10, bitwidth = 32 => a = w = ob = 32).

  * int32 accumulators that wrap (kernel.h:338-341: `c += tmp << b_opt` in 32-bit registers) once K (2^a - 1)(2^w - 1) >= 2^31;
  * plane-pair shifts b_opt = pa + pw of 31, 32, 33 .. 62 (PTX shl clamps: a shift >= 32 contributes 0);
  * the re-quantisation comparing in FLOAT (kernel.h:31-37): a wrapped (negative) sum becomes 1, and for ob >= 24
    c = 2^ob + 1 rounds to 2^ob as a float, is NOT above the bound and is kept (as 2^ob: its low ob bits are 0).

Operands are raw random 32-bit words (padding bits and padding lines included: the kernels AND whole k-quads as the
reference does); every engine setting (beyond eight planes all of them run the generic AND + popcount kernel - the test
pins that too); single launches and grouped ones; all three outputs; against the C oracle, whose requant / wrap rules are
cross-checked against the NumPy restatement in tests/test_oracle_crosscheck.py."""
import numpy as np
import pytest

from helpers import ENGINES, to_dev, to_np_u32, use_engine
from qgtc_ppopp22_amd.shapes import P8, P128, S128, cols_shape, rows_shape

pytestmark = pytest.mark.gpu

WIDE = (12, 16, 24, 32)
OBS = (8, 23, 24, 31, 32)


def _raw(rng, words, density=None):
    v = rng.integers(0, 2 ** 32, size=words, dtype=np.uint64).astype(np.uint32)
    if density is not None:   # thin the set bits: sums that stay small beside sums that wrap
        v &= rng.integers(0, 2 ** 32, size=words, dtype=np.uint64).astype(np.uint32)
        v &= rng.integers(0, 2 ** 32, size=words, dtype=np.uint64).astype(np.uint32)
    return v


def _check_all_outputs(qgtc, oracle, torch, X, Wt, M, K, N, a, w, obs, engines=ENGINES, batched=True):
    dX, dW = to_dev(torch, X, rows_shape(M, K, a)), to_dev(torch, Wt, cols_shape(K, N, w))
    want_f = oracle.bitmm2int(X, Wt, M, K, N, a, w, True)
    for eng in engines:
        with use_engine(qgtc, eng):
            np.testing.assert_array_equal(qgtc.bitMM2Int(dX, dW, M, K, N, a, w, True).cpu().numpy(), want_f, err_msg=f"float {eng}")
            for ob in obs:
                np.testing.assert_array_equal(to_np_u32(qgtc.bitMM2Bit(dX, dW, M, K, N, a, w, ob)),
                                              oracle.bitmm2bit(X, Wt, M, K, N, a, w, ob), err_msg=f"rows ob={ob} {eng}")
                np.testing.assert_array_equal(to_np_u32(qgtc.bitMM2Bit_col(dX, dW, M, K, N, a, w, ob)),
                                              oracle.bitmm2bit(X, Wt, M, K, N, a, w, ob, col=True), err_msg=f"cols ob={ob} {eng}")
            if not batched:
                continue
            for mode in (0, 1, 2):
                for ob in (obs if mode != 2 else obs[:1]):
                    bg = qgtc.BatchedGemm([dX, dX], [dW], [(M, K, N)] * 2, a, w, ob, mode, True, False)
                    bg.run()
                    torch.cuda.synchronize()
                    for o in bg.outs:
                        if mode == 2:
                            np.testing.assert_array_equal(o.cpu().numpy(), want_f, err_msg=f"grouped float {eng}")
                        else:
                            np.testing.assert_array_equal(to_np_u32(o), oracle.bitmm2bit(X, Wt, M, K, N, a, w, ob, col=(mode == 1)),
                                                          err_msg=f"grouped mode={mode} ob={ob} {eng}")


@pytest.mark.parametrize("K", [1, 64, 300])
@pytest.mark.parametrize("a", WIDE)
@pytest.mark.parametrize("w", WIDE)
def test_raw_words_at_wide_plane_counts(qgtc, oracle, a, w, K):
    """Raw random words, a, w in {12, 16, 24, 32}, ob in {8, 23, 24, 31, 32}: accumulators wrap many times (16 x 16 bits with
    K = 64 already reaches 2^38), shifts run to a + w - 2 = 62."""
    import torch
    rng = np.random.default_rng(1000 * a + 10 * w + K)
    M, N = int(rng.integers(1, 71)), int(rng.integers(1, 71))
    X = _raw(rng, a * P8(M) * S128(K) * 4)
    Wt = _raw(rng, w * P128(N) * S128(K) * 4)
    _check_all_outputs(qgtc, oracle, torch, X, Wt, M, K, N, a, w, OBS)


@pytest.mark.parametrize("a,w", [(12, 12), (16, 24), (32, 32)])
def test_thinned_raw_words_mix_small_and_wrapped_sums(qgtc, oracle, a, w):
    """Sparse planes: some sums stay below 2^ob, some land between 2^ob and 2^31, some wrap - all three requant branches per tile."""
    import torch
    rng = np.random.default_rng(77 + a + w)
    M, K, N = 70, 300, 33
    X = _raw(rng, a * P8(M) * S128(K) * 4, density=0.125)
    Wt = _raw(rng, w * P128(N) * S128(K) * 4, density=0.125)
    X.reshape(a, -1)[a // 2:] = 0          # upper planes empty on one side: the kernel's plane-block skips on this domain
    _check_all_outputs(qgtc, oracle, torch, X, Wt, M, K, N, a, w, OBS)


def _pack_values(oracle, qx, qw, a, w):
    return oracle.pack(qx.astype(np.int64).astype(np.uint32).view(np.int32), a, False), oracle.pack(qw.astype(np.int64).astype(np.uint32).view(np.int32), w, True)


def test_all_ones_16x16_bits_k64_wraps_many_times(qgtc, oracle):
    """All 16 planes of both operands set, K = 64: every sum is 64 (2^16 - 1)^2 = 2^38 - ... mod 2^32 - the directed case of VERDICT r5."""
    import torch
    M, K, N, a, w = 40, 64, 24, 16, 16
    qx, qw = np.full((M, K), 2 ** 16 - 1, dtype=np.int64), np.full((K, N), 2 ** 16 - 1, dtype=np.int64)
    X, Wt = _pack_values(oracle, qx, qw, a, w)
    want = ((64 * (2 ** 16 - 1) ** 2) & 0xFFFFFFFF)
    want = want - 2 ** 32 if want >= 2 ** 31 else want
    assert (oracle.bitmm2int(X, Wt, M, K, N, a, w, True) == np.float32(want)).all()   # the closed form pins the oracle here
    _check_all_outputs(qgtc, oracle, torch, X, Wt, M, K, N, a, w, OBS)


def test_an_accumulator_of_exactly_two_to_the_31(qgtc, oracle):
    """K = 128 values 2^12 on the left, 2^12 on the right: 128 * 2^24 = 2^31 exactly -> INT_MIN -> negative -> re-quantised to 1
    (kernel.h:34); one column one short of it (2^31 - 2^24: stays positive, clamps to 2^ob - 1)."""
    import torch
    M, K, N, a, w = 9, 128, 10, 13, 13
    qx, qw = np.full((M, K), 2 ** 12, dtype=np.int64), np.full((K, N), 2 ** 12, dtype=np.int64)
    qw[0, 1] = 0
    X, Wt = _pack_values(oracle, qx, qw, a, w)
    f = oracle.bitmm2int(X, Wt, M, K, N, a, w, True)
    assert f[0, 0] == np.float32(-2.0 ** 31) and f[0, 1] == np.float32(2.0 ** 31 - 2.0 ** 24)
    for ob in (8, 24, 31):
        words = oracle.bitmm2bit(X, Wt, M, K, N, a, w, ob)
        vals = oracle.bit2val(words, ob, M, N, False, False)
        assert vals[0, 0] == 1 and vals[0, 1] == ((2 ** ob - 1) if ob < 31 else 2 ** 31 - 2 ** 24)
    _check_all_outputs(qgtc, oracle, torch, X, Wt, M, K, N, a, w, OBS)


@pytest.mark.parametrize("ob", [24, 25])
def test_float_compare_keeps_two_to_the_ob_plus_one(qgtc, oracle, ob):
    """c = 2^ob + 0 .. 5 (ob >= 24), compared as FLOATS (kernel.h:31-33): 2^24 + 1 rounds to 2^24 - not above the bound, kept, and
    converts back as 2^24 (low 24 bits 0) where an integer compare would clamp it to 2^24 - 1 (all ones); 2^24 + 2 is above the
    bound and becomes 2^24 - 1. At ob = 25 even the clamp value 2^25 - 1 is no float: `val = max_val - 1` stores 2^25, so EVERY sum
    from 2^25 up packs as zeros. 2^ob - 2^(ob/2) and a small sum beside them."""
    import torch
    a = w = 14
    half = ob // 2
    M, K, N = 8, 5, 8
    # column n of the right operand and row m of the left make c = sum_k qx[m,k] qw[k,n]
    qx, qw = np.zeros((M, K), dtype=np.int64), np.zeros((K, N), dtype=np.int64)
    qx[:, 0] = 2 ** half
    qw[0, :6] = 2 ** (ob - half)                    # 2^ob from the first k
    qx[:, 1] = 1
    qw[1, :6] = np.arange(6)                        # + 0 .. 5
    qw[0, 6] = 2 ** (ob - half) - 1                 # column 6: 2^ob - 2^half (below the bound, a float: kept as it is)
    qw[1, 7] = 4095                                 # column 7: small
    X, Wt = _pack_values(oracle, qx, qw, a, w)
    f = oracle.bitmm2int(X, Wt, M, K, N, a, w, True)
    assert [int(v) for v in f[0, :6]] == [int(np.float32(2 ** ob + i)) for i in range(6)]
    vals = oracle.bit2val(oracle.bitmm2bit(X, Wt, M, K, N, a, w, ob), ob, M, N, False, False)
    if ob == 24:
        assert list(vals[0, :6]) == [0, 0] + [2 ** 24 - 1] * 4    # 2^24 and 2^24 + 1 kept as 2^24; from 2^24 + 2 on clamped
    else:
        assert list(vals[0, :6]) == [0] * 6                       # float(2^25 - 1) is 2^25
    assert vals[0, 6] == 2 ** ob - 2 ** half and vals[0, 7] == 4095
    _check_all_outputs(qgtc, oracle, torch, X, Wt, M, K, N, a, w, (ob, ob - 1, 31))


@pytest.mark.parametrize("pa,pw", [(15, 16), (16, 16), (16, 17), (31, 0), (31, 1), (31, 31), (0, 31)])
def test_plane_pair_shifts_31_32_33(qgtc, oracle, pa, pw):
    """ONE plane set on each side: the product is popcount << (pa + pw); shifts of 32 and more contribute nothing (kernel.h:340 on PTX shl)."""
    import torch
    a, w = pa + 1, pw + 1
    M, K, N = 33, 200, 17
    rng = np.random.default_rng(pa * 100 + pw)
    X = np.zeros(a * P8(M) * S128(K) * 4, dtype=np.uint32)
    Wt = np.zeros(w * P128(N) * S128(K) * 4, dtype=np.uint32)
    X.reshape(a, -1)[pa] = _raw(rng, P8(M) * S128(K) * 4)
    Wt.reshape(w, -1)[pw] = _raw(rng, P128(N) * S128(K) * 4)
    f = oracle.bitmm2int(X, Wt, M, K, N, a, w, True)
    if pa + pw >= 32:
        assert (f == 0).all()
    else:
        assert (f != 0).any()
    _check_all_outputs(qgtc, oracle, torch, X, Wt, M, K, N, a, w, (8, 31, 32))


def test_matrix_core_routes_at_wide_output_widths(qgtc, oracle):
    """Few planes (the matrix-core kernels' domain) with ob = 24 .. 32: their float sums are exact integers below 2^24, so the
    re-quantisation is the identity there - but the route must still write all ob planes and honour the float rule."""
    import torch
    rng = np.random.default_rng(5)
    for (M, K, N, a, w) in ((100, 4096, 64, 1, 1), (70, 300, 33, 2, 2), (64, 5000, 40, 1, 8), (40, 200, 20, 4, 4), (33, 258, 16, 8, 8),
                            (200, 4096, 300, 1, 2)):
        X = _raw(rng, a * P8(M) * S128(K) * 4)
        Wt = _raw(rng, w * P128(N) * S128(K) * 4)
        _check_all_outputs(qgtc, oracle, torch, X, Wt, M, K, N, a, w, (23, 24, 32), batched=(N <= 64))
