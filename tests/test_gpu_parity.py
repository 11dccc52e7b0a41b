"""GPU parity tests: the HIP path (through the QGTC extension -> C-ABI -> kernels) against the CPU
oracle on identical seeded inputs. Bit-exact everywhere (integer / bit work); the float32 output of
bitMM2Int is an exact int->float conversion, so it is compared exactly too."""
import numpy as np
import pytest

from helpers import ENGINES, edge_floats, rand_q, to_dev, to_np_u32, use_engine
from qgtc_ppopp22_amd.shapes import P8, P128, S128, cols_shape, rows_shape

pytestmark = pytest.mark.gpu

PACK_SHAPES = [(1, 1), (3, 3), (8, 128), (9, 129), (37, 130), (130, 37), (64, 256), (257, 300),
               (1213, 128), (600, 50), (33, 1025)]


@pytest.mark.parametrize("H,W", PACK_SHAPES)
@pytest.mark.parametrize("nbits", [1, 2, 3, 4, 8])
def test_val2bit_rows(qgtc, oracle, H, W, nbits):
    import torch
    rng = np.random.default_rng(H * 1000 + W + nbits)
    x = edge_floats(rng, H, W, nbits)
    got = qgtc.val2bit(torch.from_numpy(x).cuda(), nbits, False, False)
    assert tuple(got.shape) == rows_shape(H, W, nbits) and got.dtype == torch.int32
    np.testing.assert_array_equal(to_np_u32(got), oracle.val2bit(x, nbits, False, False))


@pytest.mark.parametrize("H,W", PACK_SHAPES)
@pytest.mark.parametrize("nbits", [1, 2, 3, 4, 8])
@pytest.mark.parametrize("output_layer", [False, True])
def test_val2bit_cols(qgtc, oracle, H, W, nbits, output_layer):
    import torch
    rng = np.random.default_rng(H * 1000 + W + nbits + 17)
    x = edge_floats(rng, H, W, nbits)
    got = qgtc.val2bit(torch.from_numpy(x).cuda(), nbits, True, output_layer)
    assert tuple(got.shape) == cols_shape(H, W, nbits, output_layer)
    np.testing.assert_array_equal(to_np_u32(got), oracle.val2bit(x, nbits, True, output_layer))


@pytest.mark.parametrize("nbits", [9, 16, 17, 31, 32])
def test_val2bit_wide(qgtc, oracle, nbits):
    import torch
    rng = np.random.default_rng(nbits)
    x = (rng.uniform(0, 1, size=(40, 70)) * 2.0 ** nbits).astype(np.float32)
    x[0, :8] = [np.nan, -1, 2.0 ** nbits, 2.0 ** nbits * 2, 0.5, 1.5, 2.0 ** 31, 2.0 ** 32]
    for cm in (False, True):
        got = qgtc.val2bit(torch.from_numpy(x).cuda(), nbits, cm, False)
        np.testing.assert_array_equal(to_np_u32(got), oracle.val2bit(x, nbits, cm, False))


@pytest.mark.parametrize("H,W", [(3, 3), (32, 32), (37, 130), (130, 37), (257, 300)])
@pytest.mark.parametrize("nbits", [1, 3, 8])
def test_bit2val_roundtrip(qgtc, oracle, H, W, nbits):
    import torch
    rng = np.random.default_rng(5 + H + W + nbits)
    q = rand_q(rng, H, W, nbits)
    for cm, ol in ((False, False), (True, False), (True, True)):
        bits = oracle.pack(q, nbits, cm, ol)
        shape = cols_shape(H, W, nbits, ol) if cm else rows_shape(H, W, nbits)
        got = qgtc.bit2val(to_dev(torch, bits, shape), nbits, H, W, cm, ol)
        assert got.dtype == torch.int32 and tuple(got.shape) == (H, W)
        np.testing.assert_array_equal(got.cpu().numpy(), q)
        np.testing.assert_array_equal(got.cpu().numpy(), oracle.bit2val(bits, nbits, H, W, cm, ol))


MM_CASES = [
    # M, K, N, a, w, ob
    (3, 3, 3, 2, 2, 2), (8, 128, 8, 1, 1, 1), (9, 130, 17, 1, 2, 2), (33, 129, 31, 2, 3, 3),
    (32, 32, 32, 2, 2, 2), (100, 1000, 64, 1, 1, 1), (65, 257, 65, 1, 4, 4), (64, 4096, 64, 1, 1, 1),
    (129, 513, 100, 3, 2, 5), (40, 200, 10, 4, 4, 4), (300, 300, 128, 1, 8, 8), (70, 640, 33, 8, 8, 8),
    (1213, 1213, 128, 1, 2, 2), (1213, 128, 128, 2, 2, 2), (599, 599, 50, 1, 4, 4), (599, 64, 10, 4, 4, 4),
    (257, 9000, 40, 1, 1, 3), (31, 70, 200, 2, 1, 1), (16, 128, 130, 5, 7, 6),
    # more planes than the generic kernel stages at once (8 x 8 plane blocks, k iteration restarts)
    (20, 300, 40, 9, 12, 4), (33, 130, 20, 17, 3, 8), (8, 128, 8, 10, 10, 12),
    # single-wave, 2-3 wave and 4-7 wave workgroups of the fixed kernels
    (64, 256, 64, 1, 1, 1), (64, 384, 64, 1, 2, 2), (40, 700, 33, 2, 2, 2), (50, 900, 70, 4, 4, 3),
    # three / four left-hand planes: single launches on the row-block kernel (k_bitmm_fp4_rows_single; the per-batch 4 x 4-bit products
    # of the Batched-GIN chain, main_qgtc.py:132-138), one to eight waves per block
    (599, 50, 64, 4, 4, 4), (300, 1213, 33, 3, 7, 5), (1213, 1213, 64, 4, 8, 8), (33, 4000, 64, 4, 4, 2),
    (129, 250, 200, 3, 7, 5), (70, 100, 256, 4, 8, 8), (2100, 8192, 130, 4, 4, 4), (45, 4300, 16, 4, 8, 3),
    # five to eight left-hand planes (the b x b-bit X . W products of the drivers at --bit_width 5 .. 8) while float32 sums stay exact
    (1213, 128, 128, 8, 8, 8), (599, 50, 64, 5, 5, 5), (300, 1000, 100, 6, 6, 4), (129, 258, 200, 8, 8, 8), (77, 259, 40, 8, 8, 8), (64, 500, 10, 7, 3, 9),
]


def _operands(torch, oracle, rng, M, K, N, a, w, density=None):
    qx = rand_q(rng, M, K, a, density)
    qw = rand_q(rng, K, N, w)
    X = oracle.pack(qx, a, False)
    Wt = oracle.pack(qw, w, True)
    return (qx, qw, X, Wt, to_dev(torch, X, rows_shape(M, K, a)), to_dev(torch, Wt, cols_shape(K, N, w)))


@pytest.mark.parametrize("M,K,N,a,w,ob", MM_CASES)
@pytest.mark.parametrize("zero_skip", [True, False])
@pytest.mark.parametrize("engine", ENGINES)
def test_bitmm2bit(qgtc, oracle, M, K, N, a, w, ob, zero_skip, engine):
    """Every engine against the oracle directly ("auto" is the shipped default; plane counts an engine does not
    cover run on the AND + popcount kernels under any setting)."""
    import torch
    rng = np.random.default_rng(M + 7 * K + 13 * N + a + w)
    qx, qw, X, Wt, dX, dW = _operands(torch, oracle, rng, M, K, N, a, w)
    qgtc.set_zero_skip(zero_skip)
    try:
        with use_engine(qgtc, engine):
            got = qgtc.bitMM2Bit(dX, dW, M, K, N, a, w, ob)
    finally:
        qgtc.set_zero_skip(True)
    assert tuple(got.shape) == rows_shape(M, N, ob)
    np.testing.assert_array_equal(to_np_u32(got), oracle.bitmm2bit(X, Wt, M, K, N, a, w, ob))


@pytest.mark.parametrize("M,K,N,a,w,ob", MM_CASES)
@pytest.mark.parametrize("engine", ENGINES)
def test_bitmm2bit_col(qgtc, oracle, M, K, N, a, w, ob, engine):
    import torch
    rng = np.random.default_rng(M + 7 * K + 13 * N + a + w + 1)
    qx, qw, X, Wt, dX, dW = _operands(torch, oracle, rng, M, K, N, a, w)
    with use_engine(qgtc, engine):
        got = qgtc.bitMM2Bit_col(dX, dW, M, K, N, a, w, ob)
    assert tuple(got.shape) == cols_shape(M, N, ob)
    np.testing.assert_array_equal(to_np_u32(got), oracle.bitmm2bit(X, Wt, M, K, N, a, w, ob, col=True))


@pytest.mark.parametrize("M,K,N,a,w,ob", MM_CASES)
@pytest.mark.parametrize("pad_128", [True, False])
@pytest.mark.parametrize("engine", ENGINES)
def test_bitmm2int(qgtc, oracle, M, K, N, a, w, ob, pad_128, engine):
    qgtc.set_engine(engine)   # (plane counts above 8 stay on the popcount kernels either way)
    try:
        _bitmm2int_case(qgtc, oracle, M, K, N, a, w, ob, pad_128)
    finally:
        qgtc.set_engine("auto")


def _bitmm2int_case(qgtc, oracle, M, K, N, a, w, ob, pad_128):
    import torch
    rng = np.random.default_rng(M + 7 * K + 13 * N + a + w + 2)
    qx = rand_q(rng, M, K, a)
    qw = rand_q(rng, K, N, w)
    X = oracle.pack(qx, a, False)
    Wt = oracle.pack(qw, w, True, output_layer=not pad_128)
    dX = to_dev(torch, X, rows_shape(M, K, a))
    dW = to_dev(torch, Wt, cols_shape(K, N, w, not pad_128))
    got = qgtc.bitMM2Int(dX, dW, M, K, N, a, w, pad_128)
    assert got.dtype == torch.float32 and tuple(got.shape) == (M, N)
    ref = oracle.bitmm2int(X, Wt, M, K, N, a, w, pad_128)
    np.testing.assert_array_equal(got.cpu().numpy(), ref)
    # and against plain integer matmul of the quantised values
    np.testing.assert_array_equal(got.cpu().numpy(), (qx.astype(np.int64) @ qw.astype(np.int64)).astype(np.float32))


@pytest.mark.parametrize("density", [0.0, 0.001, 0.02])
def test_sparse_adjacency_zero_skip(qgtc, oracle, density):
    """Block-sparse left operand: results identical with and without zero-tile skipping."""
    import torch
    M = K = 1213
    N, a, w, ob = 128, 1, 2, 2
    rng = np.random.default_rng(int(density * 1e4) + 3)
    qx, qw, X, Wt, dX, dW = _operands(torch, oracle, rng, M, K, N, a, w, density=density)
    ref = oracle.bitmm2bit(X, Wt, M, K, N, a, w, ob)
    for zs in (True, False):
        qgtc.set_zero_skip(zs)
        try:
            np.testing.assert_array_equal(to_np_u32(qgtc.bitMM2Bit(dX, dW, M, K, N, a, w, ob)), ref)
            np.testing.assert_array_equal(qgtc.bitMM2Int(dX, dW, M, K, N, a, w, True).cpu().numpy(),
                                          oracle.bitmm2int(X, Wt, M, K, N, a, w, True))
        finally:
            qgtc.set_zero_skip(True)


def test_requant_boundaries(qgtc, oracle):
    """C == 2^ob is kept (packs as 0), C == 2^ob + 1 clamps to 2^ob - 1 (kernel.h:31-37,350)."""
    import torch
    for ob in (1, 2, 3):
        target = 2 ** ob
        for K in (target - 1, target, target + 1, target + 2):
            if K <= 0:
                continue
            M, N = 5, 6
            qx = np.ones((M, K), np.int32)
            qw = np.ones((K, N), np.int32)
            X, Wt = oracle.pack(qx, 1, False), oracle.pack(qw, 1, True)
            got = qgtc.bitMM2Bit(to_dev(torch, X, rows_shape(M, K, 1)), to_dev(torch, Wt, cols_shape(K, N, 1)),
                                 M, K, N, 1, 1, ob)
            np.testing.assert_array_equal(to_np_u32(got), oracle.bitmm2bit(X, Wt, M, K, N, 1, 1, ob))
            val = qgtc.bit2val(got, ob, M, N, False, False).cpu().numpy()
            expect = (K if K <= target else target - 1) & (target - 1)
            assert (val == expect).all()


def test_tile_counters(qgtc, oracle):
    import torch
    rng = np.random.default_rng(11)
    for (M, K, N, a, w, dens) in [(1024, 1024, 16, 1, 1, None), (1213, 1213, 128, 1, 2, 0.002),
                                  (100, 300, 20, 2, 3, 0.01), (64, 128, 8, 1, 1, 0.0)]:
        qx = rand_q(rng, M, K, a, dens)
        X = oracle.pack(qx, a, False)
        dX = to_dev(torch, X, rows_shape(M, K, a))
        assert tuple(qgtc.tile_counters(dX, M, K, N, a, w)) == oracle.tile_counters(X, M, K, N, a, w)


def test_counter_ops_are_cumulative_and_print(qgtc, oracle, capfd):
    import torch
    M = K = 64
    N = 16
    A = torch.ones((M, K)).cuda()
    Xf = torch.ones((K, N)).cuda()
    ba = qgtc.val2bit(A, 1, False, False)
    bx = qgtc.val2bit(Xf, 1, True, False)
    qgtc.reset_counters()
    per_call = (M // 8) * (N // 8) * 1
    o1 = qgtc.bitMM2Bit_base_cnt(ba, bx, M, K, N, 1, 1, 1)
    o2 = qgtc.bitMM2Bit_zerojump_cnt(ba, bx, M, K, N, 1, 1, 1)
    qgtc.bitMM2Bit_base_cnt(ba, bx, M, K, N, 1, 1, 1)
    out = capfd.readouterr().out
    assert f"counter_global: {per_call}\n" in out and f"counter: {per_call}\n" in out
    assert f"counter_global: {2 * per_call}\n" in out
    assert qgtc.get_counters() == (2 * per_call, per_call)
    ref = qgtc.bitMM2Bit(ba, bx, M, K, N, 1, 1, 1)
    assert torch.equal(o1, ref) and torch.equal(o2, ref)


def test_profile_line_format(qgtc, capfd):
    import re
    import torch
    M = K = 1024
    N = 16
    ba = qgtc.val2bit(torch.ones((M, K)).cuda(), 1, False, False)
    bx = qgtc.val2bit(torch.ones((K, N)).cuda(), 1, True, False)
    out_t = qgtc.bitMM2Bit_profile(ba, bx, M, K, N, 1, 1, 1)
    line = capfd.readouterr().out.strip().splitlines()[-1]
    assert re.fullmatch(r"X1_height 1024, X1_width: 1024, X2_width: 16, TFLOPs: \d+\.\d{3}", line), line
    assert torch.equal(out_t, qgtc.bitMM2Bit(ba, bx, M, K, N, 1, 1, 1))
    assert qgtc.last_profile_ms() > 0


def test_input_checks(qgtc):
    import torch
    with pytest.raises(RuntimeError, match="must be a CUDA tensor"):
        qgtc.val2bit(torch.ones(4, 4), 1, False, False)
    with pytest.raises(RuntimeError, match="must be contiguous"):
        qgtc.val2bit(torch.ones(8, 8).cuda().t(), 1, False, False)
    with pytest.raises(RuntimeError):
        qgtc.val2bit(torch.ones(4, 4).cuda(), 0, False, False)
    with pytest.raises(RuntimeError):
        qgtc.val2bit(torch.ones(4, 4).cuda(), 33, False, False)
    with pytest.raises(RuntimeError, match="int32"):
        qgtc.bitMM2Bit(torch.ones(8, 4).cuda(), torch.ones(8, 4).cuda(), 8, 8, 8, 1, 1, 1)
    with pytest.raises(TypeError):
        qgtc.bitMM2Bit(torch.ones(8, 4).cuda(), 8, 8, 8, 1, 1, 1)  # arity is enforced like the reference


def test_missized_operands_are_bounds_safe(qgtc, oracle):
    """The reference's literal Cluster-GCN chain feeds mis-laid / short operands; the
    HIP path must treat out-of-extent words as 0 exactly as the oracle does, and never fault."""
    import torch
    rng = np.random.default_rng(99)
    M = K = 333
    N, a, w, ob = 128, 1, 2, 2
    X = rng.integers(0, 2 ** 32, size=rows_shape(M, K, a), dtype=np.uint64).astype(np.uint32).reshape(-1)
    # right operand deliberately in the *rows* layout of a [K, N] result (what main_qgtc.py:148 passes)
    Wt = rng.integers(0, 2 ** 32, size=rows_shape(K, N, w), dtype=np.uint64).astype(np.uint32).reshape(-1)
    dX, dW = to_dev(torch, X, rows_shape(M, K, a)), to_dev(torch, Wt, rows_shape(K, N, w))
    np.testing.assert_array_equal(to_np_u32(qgtc.bitMM2Bit(dX, dW, M, K, N, a, w, ob)),
                                  oracle.bitmm2bit(X, Wt, M, K, N, a, w, ob))
    np.testing.assert_array_equal(qgtc.bitMM2Int(dX, dW, M, K, N, a, w, False).cpu().numpy(),
                                  oracle.bitmm2int(X, Wt, M, K, N, a, w, False))


def test_batched_matches_single(qgtc, oracle):
    import torch
    rng = np.random.default_rng(21)
    a, w, ob = 1, 2, 2
    dims = [(1213, 1213, 128), (1100, 1100, 128), (37, 37, 128), (640, 640, 128)]
    Xs, Ws, refs = [], [], []
    for (M, K, N) in dims:
        qx, qw, X, Wt, dX, dW = _operands(torch, oracle, rng, M, K, N, a, w, density=0.01)
        Xs.append(dX)
        Ws.append(dW)
        refs.append((X, Wt))
    for mode in (0, 1, 2):
        bg = qgtc.BatchedGemm(Xs, Ws, dims, a, w, ob, mode, True)
        bg.run()
        torch.cuda.synchronize()
        for i, (M, K, N) in enumerate(dims):
            X, Wt = refs[i]
            if mode == 2:
                np.testing.assert_array_equal(bg.outs[i].cpu().numpy(), oracle.bitmm2int(X, Wt, M, K, N, a, w, True))
            else:
                np.testing.assert_array_equal(to_np_u32(bg.outs[i]),
                                              oracle.bitmm2bit(X, Wt, M, K, N, a, w, ob, col=(mode == 1)))


@pytest.mark.parametrize("engine", ENGINES)
def test_random_shape_sweep(qgtc, oracle, engine):
    """Seeded random sweep over shapes, plane counts, output modes and both launch forms: catches
    what the hand-picked cases miss (tile edges, wave counts, plane blocking, zero rows). Run once per
    engine (plane counts above 8 stay on the popcount kernels under 'mfma' too)."""
    import torch
    qgtc.set_engine(engine)
    try:
        _random_shape_sweep(qgtc, oracle, 20260301 + ENGINES.index(engine))
    finally:
        qgtc.set_engine("auto")


def _random_shape_sweep(qgtc, oracle, seed):
    import torch
    rng = np.random.default_rng(seed)
    singles = {0: lambda *a: qgtc.bitMM2Bit(*a), 1: lambda *a: qgtc.bitMM2Bit_col(*a)}
    for case in range(70):
        M = int(rng.integers(1, 180))
        N = int(rng.integers(1, 150))
        K = int(rng.choice([rng.integers(1, 130), rng.integers(130, 1100), rng.integers(1100, 5000)]))
        a = int(rng.choice([1, 1, 1, 2, 2, 3, 4, 5, 8, 9]))
        w = int(rng.choice([1, 2, 2, 3, 4, 4, 7, 8, 10]))
        ob = int(rng.integers(1, 9))
        mode = int(rng.integers(0, 3))
        density = float(rng.choice([1.0, 0.3, 0.01]))
        qx = rand_q(rng, M, K, a, density)
        if case % 5 == 0:
            qx[rng.integers(0, M)] = 0                       # an all-zero row
        qw = rand_q(rng, K, N, w)
        X, Wt = oracle.pack(qx, a, False), oracle.pack(qw, w, True)
        dX, dW = to_dev(torch, X, rows_shape(M, K, a)), to_dev(torch, Wt, cols_shape(K, N, w))
        tag = f"case {case}: M={M} K={K} N={N} a={a} w={w} ob={ob} mode={mode} density={density}"
        if mode == 2:
            want = oracle.bitmm2int(X, Wt, M, K, N, a, w, True)
            got = qgtc.bitMM2Int(dX, dW, M, K, N, a, w, True).cpu().numpy()
        else:
            want = oracle.bitmm2bit(X, Wt, M, K, N, a, w, ob, col=(mode == 1))
            got = to_np_u32(singles[mode](dX, dW, M, K, N, a, w, ob))
        np.testing.assert_array_equal(got, want, err_msg=tag)
        # the grouped launch of the same product (with and without the occupancy bitmap)
        for zj in (False, True):
            bg = qgtc.BatchedGemm([dX], [dW], [(M, K, N)], a, w, ob, mode, True, zj)
            bg.run()
            got_b = bg.outs[0].cpu().numpy() if mode == 2 else to_np_u32(bg.outs[0])
            np.testing.assert_array_equal(got_b, want, err_msg=tag + f" (grouped, zero_jump={zj})")
