"""Full-size checks at BASELINE.json's headline shape (4096 x 4096 x 64, 1-bit adjacency x w-bit
features; the 2_7c benchmark of the reference) and at the wide products the MFMA engine is for.

At these sizes the C oracle still finishes in seconds, so the first test is plain equality; the
others are size-independent properties of the path that do not involve the oracle at all:
  * the all-ones closed form of the reference's micro-benchmark inputs (2_7c_QGTC_GEMM_INT8.py
    feeds torch.ones): every output is min(K * (2^w - 1), 2^ob - 1);
  * plane linearity: the w-bit product is sum_p 2^p * (the 1-bit product with plane p);
  * a checksum of checksums: sum(C) = sum_k colsum(A)_k * rowsum(X)_k, computed with torch int64
    reductions that share no code with the kernels;
  * row-permutation equivariance, and decode(encode(x)) = quantise(x) at 4096 x 4096;
  * zero-skip on / off, both engines, grouped launch: the same words.
"""
import numpy as np
import pytest

from helpers import ENGINES, to_np_u32, use_engine

pytestmark = pytest.mark.gpu

M = K = 4096
N = 64


def _operands(torch, w, seed, density=0.5):
    """Quantised values on the device (int64) and their packed forms made by the device's own val2bit
    (val2bit has its own parity tests against the oracle)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    qa = (torch.rand((M, K), generator=g) < density).to(torch.float32).cuda()
    qx = torch.randint(0, 2 ** w, (K, N), generator=g).to(torch.float32).cuda()
    return qa, qx


@pytest.mark.parametrize("w", [1, 2, 4, 8])
@pytest.mark.parametrize("engine", ENGINES)
def test_headline_shape_equals_the_oracle(qgtc, oracle, w, engine):
    """Dense random operands at BASELINE.json's headline shape, every engine DIRECTLY against the oracle ("auto" -
    the default, what bench.py times - runs the FP4 narrow-operand kernel here, "mfma" the same, "popcount" the AND +
    v_bcnt kernels)."""
    import torch
    qgtc.set_engine(engine)     # (the autouse fixture puts the default back)
    qa, qx = _operands(torch, w, 10 + w)
    bit_A = qgtc.val2bit(qa, 1, False, False)
    bit_X = qgtc.val2bit(qx, w, True, False)
    A_o = oracle.val2bit(qa.cpu().numpy(), 1, False, False)
    X_o = oracle.val2bit(qx.cpu().numpy(), w, True, False)
    np.testing.assert_array_equal(to_np_u32(bit_A), A_o)
    np.testing.assert_array_equal(to_np_u32(bit_X), X_o)
    np.testing.assert_array_equal(to_np_u32(qgtc.bitMM2Bit(bit_A, bit_X, M, K, N, 1, w, w)),
                                  oracle.bitmm2bit(A_o, X_o, M, K, N, 1, w, w))
    np.testing.assert_array_equal(to_np_u32(qgtc.bitMM2Bit_col(bit_A, bit_X, M, K, N, 1, w, w)),
                                  oracle.bitmm2bit(A_o, X_o, M, K, N, 1, w, w, col=True))
    np.testing.assert_array_equal(qgtc.bitMM2Int(bit_A, bit_X, M, K, N, 1, w, True).cpu().numpy(),
                                  oracle.bitmm2int(A_o, X_o, M, K, N, 1, w, True))


@pytest.mark.parametrize("w", [1, 2, 4, 8])
def test_all_ones_closed_form(qgtc, w):
    """The reference's own benchmark inputs: torch.ones for both operands (2_7c_QGTC_GEMM_INT8.py:30-41).
    ones quantise to 1, so C = K everywhere; re-quantised to w bits that is 2^w - 1 wherever K > 2^w."""
    import torch
    A = torch.ones((M, K), device="cuda")
    X = torch.ones((K, N), device="cuda")
    bit_A, bit_X = qgtc.val2bit(A, 1, False, False), qgtc.val2bit(X, w, True, False)
    C = qgtc.bitMM2Int(bit_A, bit_X, M, K, N, 1, w, True)
    # quantiser (kernel.h:39-44): x > 2^b -> 2^b - 1, else round(x); 1.0 -> 1 for every b >= 1
    assert torch.equal(C, torch.full((M, N), float(K), device="cuda"))
    out = qgtc.bitMM2Bit(bit_A, bit_X, M, K, N, 1, w, w)
    back = qgtc.bit2val(out, w, M, N, False, False)
    assert torch.equal(back, torch.full((M, N), 2 ** w - 1, device="cuda", dtype=back.dtype))


@pytest.mark.parametrize("w", [2, 4, 8])
def test_plane_linearity(qgtc, w):
    """A x X = sum_p 2^p * (A x plane_p(X)): the shift-accumulate of QGTC_device.cu:478-490, checked
    against 1-bit products of the same kernel family at full size."""
    import torch
    qa, qx = _operands(torch, w, 20 + w)
    bit_A = qgtc.val2bit(qa, 1, False, False)
    full = qgtc.bitMM2Int(bit_A, qgtc.val2bit(qx, w, True, False), M, K, N, 1, w, True)
    acc = torch.zeros_like(full)
    xi = qx.to(torch.int64)
    for p in range(w):
        plane = ((xi >> p) & 1).to(torch.float32)
        acc += float(1 << p) * qgtc.bitMM2Int(bit_A, qgtc.val2bit(plane, 1, True, False), M, K, N, 1, 1, True)
    assert torch.equal(full, acc)


@pytest.mark.parametrize("w,density", [(1, 0.5), (4, 0.5), (8, 0.01)])
def test_checksum_of_checksums(qgtc, w, density):
    """sum_ij C_ij = sum_k (sum_i A_ik) (sum_j X_kj), and per row: C 1 = A (X 1) - torch int64
    reductions on the unpacked values, no shared code with the bit path."""
    import torch
    qa, qx = _operands(torch, w, 30 + w, density)
    C = qgtc.bitMM2Int(qgtc.val2bit(qa, 1, False, False), qgtc.val2bit(qx, w, True, False), M, K, N, 1, w, True)
    a64, x64 = qa.to(torch.int64), qx.to(torch.int64)
    total = int((a64.sum(0) * x64.sum(1)).sum().item())
    assert int(C.to(torch.int64).sum().item()) == total
    row = (qa.to(torch.float64) @ x64.sum(1).to(torch.float64)).to(torch.int64)
    assert torch.equal(C.to(torch.int64).sum(1), row)


def test_row_permutation_equivariance(qgtc):
    import torch
    w = 2
    qa, qx = _operands(torch, w, 41, density=0.03)
    perm = torch.randperm(M, generator=torch.Generator().manual_seed(1)).cuda()
    bit_X = qgtc.val2bit(qx, w, True, False)
    C = qgtc.bitMM2Int(qgtc.val2bit(qa, 1, False, False), bit_X, M, K, N, 1, w, True)
    Cp = qgtc.bitMM2Int(qgtc.val2bit(qa[perm].contiguous(), 1, False, False), bit_X, M, K, N, 1, w, True)
    assert torch.equal(Cp, C[perm])


@pytest.mark.parametrize("nbits,col", [(1, False), (3, False), (8, False), (2, True), (8, True)])
def test_encode_decode_roundtrip_4096(qgtc, nbits, col):
    """bit2val(val2bit(x)) keeps the low nbits of quantise(x) (kernel.h:39-44, :100) at 4096 x 4096."""
    import torch
    g = torch.Generator(device="cpu").manual_seed(50 + nbits)
    x = (torch.rand((M, K), generator=g) * (2 ** nbits + 2) - 1.0).cuda()
    hi = float(2 ** nbits)
    q = torch.where(x < 0, torch.ones_like(x), torch.where(x > hi, torch.full_like(x, hi - 1), torch.round(x)))
    want = q.to(torch.int64) & (2 ** nbits - 1)
    back = qgtc.bit2val(qgtc.val2bit(x, nbits, col, False), nbits, M, K, col, False)
    assert torch.equal(back.to(torch.int64), want)


@pytest.mark.parametrize("w", [1, 4])
def test_every_route_gives_the_same_words(qgtc, w):
    """Zero-tile skipping on / off, popcount / MFMA / auto engine and the grouped launch are routes
    to the same words at the headline shape (sparse adjacency so that skipping really skips)."""
    import torch
    qa, qx = _operands(torch, w, 60 + w, density=0.002)
    qa[:, 1024:3072] = 0                                   # whole k-quads of zeros
    bit_A, bit_X = qgtc.val2bit(qa, 1, False, False), qgtc.val2bit(qx, w, True, False)
    with use_engine(qgtc, "popcount"):
        ref = qgtc.bitMM2Bit(bit_A, bit_X, M, K, N, 1, w, w)
        qgtc.set_zero_skip(False)
        try:
            assert torch.equal(qgtc.bitMM2Bit(bit_A, bit_X, M, K, N, 1, w, w), ref)
        finally:
            qgtc.set_zero_skip(True)
    for eng in ("mfma", "auto"):
        qgtc.set_engine(eng)
        try:
            assert torch.equal(qgtc.bitMM2Bit(bit_A, bit_X, M, K, N, 1, w, w), ref)
        finally:
            qgtc.set_engine("auto")
    for eng in ENGINES:
        for zj in (False, True):
            with use_engine(qgtc, eng):
                bg = qgtc.BatchedGemm([bit_A, bit_A], [bit_X, bit_X], [(M, K, N), (M, K, N)], 1, w, w, 0, True, zj)
                bg.run()
            assert torch.equal(bg.outs[0].view(-1), ref.view(-1)) and torch.equal(bg.outs[1].view(-1), ref.view(-1))


@pytest.mark.parametrize("a,w,ob", [(1, 1, 1), (2, 2, 2), (1, 4, 4)])
def test_wide_product_both_engines_and_checksum(qgtc, a, w, ob):
    """4096 x 4096 x 1024 (the regime the MFMA engine is for): engines agree word for word, and the
    integer product satisfies the row checksum."""
    import torch
    NW = 1024
    g = torch.Generator(device="cpu").manual_seed(70 + a + w)
    qa = torch.randint(0, 2 ** a, (M, K), generator=g).to(torch.float32).cuda()
    qx = torch.randint(0, 2 ** w, (K, NW), generator=g).to(torch.float32).cuda()
    bit_A, bit_X = qgtc.val2bit(qa, a, False, False), qgtc.val2bit(qx, w, True, False)
    with use_engine(qgtc, "popcount"):
        pop = (qgtc.bitMM2Bit(bit_A, bit_X, M, K, NW, a, w, ob), qgtc.bitMM2Int(bit_A, bit_X, M, K, NW, a, w, True))
    for eng in ("mfma", "auto"):
        with use_engine(qgtc, eng):
            mf = (qgtc.bitMM2Bit(bit_A, bit_X, M, K, NW, a, w, ob), qgtc.bitMM2Int(bit_A, bit_X, M, K, NW, a, w, True))
        assert torch.equal(pop[0], mf[0]) and torch.equal(pop[1], mf[1]), eng
    row = (qa.to(torch.float64) @ qx.to(torch.float64).sum(1)).to(torch.int64)
    assert torch.equal(mf[1].to(torch.int64).sum(1), row)


@pytest.mark.parametrize("NW_,w", [(128, 1), (256, 2), (200, 8)])
def test_mid_width_products_all_engines_and_checksum(qgtc, NW_, w):
    """4096 x 4096 x {128, 256, 200}: the width range where `auto` uses the narrow-operand FP4 kernel with several
    column tiles per row tile. Engines agree word for word (rows bits, cols bits, float32) and the integer product
    satisfies the row checksum C 1 = A (X 1), computed with torch reductions that share no code with the kernels."""
    import torch
    g = torch.Generator(device="cpu").manual_seed(90 + NW_ + w)
    qa = (torch.rand((M, K), generator=g) < 0.3).to(torch.float32).cuda()
    qx = torch.randint(0, 2 ** w, (K, NW_), generator=g).to(torch.float32).cuda()
    bit_A, bit_X = qgtc.val2bit(qa, 1, False, False), qgtc.val2bit(qx, w, True, False)
    with use_engine(qgtc, "popcount"):
        ref = (qgtc.bitMM2Bit(bit_A, bit_X, M, K, NW_, 1, w, w), qgtc.bitMM2Bit_col(bit_A, bit_X, M, K, NW_, 1, w, w),
               qgtc.bitMM2Int(bit_A, bit_X, M, K, NW_, 1, w, True))
    for eng in ("mfma", "auto"):
        qgtc.set_engine(eng)
        try:
            got = (qgtc.bitMM2Bit(bit_A, bit_X, M, K, NW_, 1, w, w), qgtc.bitMM2Bit_col(bit_A, bit_X, M, K, NW_, 1, w, w),
                   qgtc.bitMM2Int(bit_A, bit_X, M, K, NW_, 1, w, True))
        finally:
            qgtc.set_engine("auto")
        for x, y in zip(got, ref):
            assert torch.equal(x, y), eng
    row = (qa.to(torch.float64) @ qx.to(torch.float64).sum(1)).to(torch.int64)
    assert torch.equal(ref[2].to(torch.int64).sum(1), row)


# BASELINE.json configs[1] = the reference's whole micro-benchmark grid (2_7c_QGTC_GEMM_INT8.py:6-19): M = K in
# {1024, 2048, 4096}, N in {16, 32, 64}, 1-bit adjacency x w-bit features, w = ob in {1, 2, 4, 8}. Every point selects
# its own kernel instantiation and grid on the default engine (k_bitmm_fp4_one<NA,NW,MODE,RF,CF>: 16 x 32 tiles at
# N <= 32, 32 x 32 at N = 64, 64 x 16 for more than four planes) - each is compared with the oracle at FULL size.
MICRO_SHAPES = [(mk, nn) for nn in (16, 32, 64) for mk in (1024, 2048, 4096)]


@pytest.mark.parametrize("w", [1, 2, 4, 8])
@pytest.mark.parametrize("mk,nn", MICRO_SHAPES)
@pytest.mark.parametrize("engine", ["auto", "popcount"])
def test_micro_bench_shapes_equal_the_oracle(qgtc, oracle, engine, mk, nn, w):
    """All nine 2_7c shapes x four widths, seeded random operands, `bitMM2Bit` words (what the benchmark launches) and
    the float32 accumulators against the C oracle, on the default engine and on the AND + popcount kernels. Also the
    benchmark's own all-ones inputs (2_7c:30-41) against their closed form through the same launch."""
    import torch
    qgtc.set_engine(engine)     # (the autouse fixture puts the default back)
    g = torch.Generator(device="cpu").manual_seed(1000 * w + mk + nn)
    qa = (torch.rand((mk, mk), generator=g) < 0.5).to(torch.float32)
    qx = torch.randint(0, 2 ** w, (mk, nn), generator=g).to(torch.float32)
    bit_A = qgtc.val2bit(qa.cuda(), 1, False, False)
    bit_X = qgtc.val2bit(qx.cuda(), w, True, False)
    A_o = oracle.val2bit(qa.numpy(), 1, False, False)
    X_o = oracle.val2bit(qx.numpy(), w, True, False)
    np.testing.assert_array_equal(to_np_u32(bit_A), A_o)
    np.testing.assert_array_equal(to_np_u32(bit_X), X_o)
    np.testing.assert_array_equal(to_np_u32(qgtc.bitMM2Bit(bit_A, bit_X, mk, mk, nn, 1, w, w)),
                                  oracle.bitmm2bit(A_o, X_o, mk, mk, nn, 1, w, w))
    np.testing.assert_array_equal(qgtc.bitMM2Int(bit_A, bit_X, mk, mk, nn, 1, w, True).cpu().numpy(),
                                  oracle.bitmm2int(A_o, X_o, mk, mk, nn, 1, w, True))
    # the benchmark's inputs: ones x ones -> C = K > 2^w everywhere -> every decoded element 2^w - 1
    ones_A = qgtc.val2bit(torch.ones((mk, mk), device="cuda"), 1, False, False)
    ones_X = qgtc.val2bit(torch.ones((mk, nn), device="cuda"), w, True, False)
    back = qgtc.bit2val(qgtc.bitMM2Bit(ones_A, ones_X, mk, mk, nn, 1, w, w), w, mk, nn, False, False)
    assert torch.equal(back, torch.full((mk, nn), 2 ** w - 1, device="cuda", dtype=back.dtype))
