"""GPU parity of the paths that sit beside the bit-GEMM: adjacency packing from an edge list
(sampler.py:80-101 without the dense detour) and the int8 MFMA comparison GEMM
(cuBLASGemmEX/cublas_main.cu:123-172 analogue)."""
import numpy as np
import pytest

from helpers import edge_floats, rand_q, to_dev, to_np_u32, use_engine

pytestmark = pytest.mark.gpu


from oracle.qgtc_oracle import np_dense_adjacency, np_i8gemm, np_pack_edges, np_tile_occupancy


@pytest.mark.parametrize("H,W,nbits,edges,dup", [
    (37, 37, 1, 200, False), (130, 130, 1, 900, True), (1213, 1213, 1, 9000, True),
    (64, 200, 2, 3000, True), (9, 300, 3, 500, True), (33, 33, 1, 0, False),
])
def test_pack_edges_equals_val2bit_of_the_dense_adjacency(qgtc, oracle, H, W, nbits, edges, dup):
    import torch
    rng = np.random.default_rng(H * 31 + W + nbits)
    row = rng.integers(0, H, size=edges)
    col = rng.integers(0, W, size=edges)
    if dup and edges:   # force multiplicities 2, 3 and 5 on a few cells (a1: 2 -> plane 0 clear at b=1)
        extra_r = np.concatenate([row[:20]] * 1 + [row[20:30]] * 2 + [row[30:35]] * 4)
        extra_c = np.concatenate([col[:20]] * 1 + [col[20:30]] * 2 + [col[30:35]] * 4)
        row, col = np.concatenate([row, extra_r]), np.concatenate([col, extra_c])
    A = np_dense_adjacency(row, col, H, W)
    want = oracle.val2bit(A, nbits, False, False)
    np.testing.assert_array_equal(want, np_pack_edges(row, col, H, W, nbits))
    got = qgtc.pack_edges(torch.from_numpy(row).cuda(), torch.from_numpy(col).cuda(), H, W, nbits)
    assert tuple(got.shape) == (nbits * ((H + 7) // 8 * 8), (W + 127) // 128 * 4)
    np.testing.assert_array_equal(to_np_u32(got), want)
    # and it is word-for-word the device's own dense route
    dense = qgtc.val2bit(torch.from_numpy(A).cuda(), nbits, False, False)
    assert torch.equal(got, dense)


def test_pack_edges_rejects_bad_indices(qgtc):
    import torch
    with pytest.raises(RuntimeError):
        qgtc.pack_edges(torch.tensor([0, 40]).cuda(), torch.tensor([1, 2]).cuda(), 37, 37, 1)
    with pytest.raises(RuntimeError):
        qgtc.pack_edges(torch.tensor([0, 40]).cuda(), torch.tensor([1, 2]).cuda(), 37, 37, 2)
    # validate=False (no host round trip): out-of-range edges are skipped, the rest is packed
    got = qgtc.pack_edges(torch.tensor([0, 40, 3]).cuda(), torch.tensor([1, 2, -1]).cuda(), 37, 37, 1, False)
    want = qgtc.pack_edges(torch.tensor([0]).cuda(), torch.tensor([1]).cuda(), 37, 37, 1)
    assert torch.equal(got, want)
    with pytest.raises(RuntimeError):
        qgtc.pack_edges(torch.tensor([0, 1], dtype=torch.int32).cuda(), torch.tensor([1, 2], dtype=torch.int32).cuda(), 37, 37, 1)


@pytest.mark.parametrize("M,K,N", [(16, 64, 16), (17, 128, 70), (100, 1024, 64), (1024, 1024, 16),
                                   (333, 4096, 33), (4096, 4096, 64), (64, 16, 200), (50, 48, 10)])
def test_i8gemm_is_exact(qgtc, M, K, N):
    import torch
    rng = np.random.default_rng(M + K + N)
    A = rng.integers(-128, 128, size=(M, K), dtype=np.int64).astype(np.int8)
    Bt = rng.integers(-128, 128, size=(N, K), dtype=np.int64).astype(np.int8)
    want = np_i8gemm(A, Bt)   # exact integer reference, then the int32 -> float32 conversion
    got = qgtc.i8gemm(torch.from_numpy(A).cuda(), torch.from_numpy(Bt).cuda())
    assert got.dtype == torch.float32 and tuple(got.shape) == (M, N)
    np.testing.assert_array_equal(got.cpu().numpy(), want)


def test_i8gemm_identity_with_asymmetric_b(qgtc):
    """A = I picks rows of B: catches a transposed or permuted fragment map."""
    import torch
    K = N = 64
    A = np.eye(K, dtype=np.int8)
    B = (np.arange(K)[:, None] * 2 - np.arange(N)[None, :]).clip(-128, 127).astype(np.int8)   # B[k][n]
    got = qgtc.i8gemm(torch.from_numpy(A).cuda(), torch.from_numpy(np.ascontiguousarray(B.T)).cuda())
    np.testing.assert_array_equal(got.cpu().numpy(), B.astype(np.float32))


def test_i8gemm_profile_line_and_checks(qgtc, capfd):
    import torch
    A = torch.ones((1024, 1024), dtype=torch.int8).cuda()
    Bt = torch.ones((16, 1024), dtype=torch.int8).cuda()
    ms = qgtc.i8gemm_profile(A, Bt, 20, True)
    out = capfd.readouterr().out
    assert ms > 0 and out.startswith("M: 1024, K: 1024, N: 16, TFLOPS: ")
    with pytest.raises(RuntimeError):
        qgtc.i8gemm(torch.ones((8, 24), dtype=torch.int8).cuda(), torch.ones((8, 24), dtype=torch.int8).cuda())  # K % 16
    with pytest.raises(RuntimeError):
        qgtc.i8gemm(torch.ones((8, 32)).cuda(), torch.ones((8, 32)).cuda())


def test_sampler_edge_route_equals_dense_route(qgtc):
    """ClusterIter packs the adjacency from the edge list by default; the reference's dense route
    (sampler.py:80-101) must give the same words."""
    import random
    import torch
    from qgtc_ppopp22_amd import graph as G
    from qgtc_ppopp22_amd.sampler import ClusterIter

    g = G.make_graph("ppi", 1500)
    outs = []
    for dense in (False, True):
        random.seed(2)
        it = ClusterIter("ppi", g, 1500, 20, bit_width=2, device="cuda", qgtc=qgtc, batch_ids=[0, 1, 2],
                         dense_adjacency=dense)
        outs.append([ct.bit_A for ct in it.cTensor_li])
    for a, b in zip(*outs):
        assert torch.equal(a, b)


@pytest.mark.parametrize("M,K,a,density", [(100, 1000, 1, 0.002), (1213, 1213, 1, 0.0005), (70, 9000, 2, 0.0003),
                                           (33, 130, 3, 0.5), (64, 128, 1, 0.0)])
def test_tile_occupancy_bitmap(qgtc, oracle, M, K, a, density):
    import torch
    from helpers import rand_q, to_dev
    rng = np.random.default_rng(M + K + a)
    qx = rand_q(rng, M, K, a, density)
    X = oracle.pack(qx, a, False)
    dX = to_dev(torch, X, (a * ((M + 7) // 8 * 8), (K + 127) // 128 * 4))
    got = qgtc.tile_occupancy(dX, M, K, a).cpu().numpy().view(np.uint64)
    np.testing.assert_array_equal(got, np_tile_occupancy(X, M, K, a))


@pytest.mark.parametrize("a,w,ob", [(1, 2, 2), (1, 1, 1), (2, 2, 3), (3, 5, 4), (8, 8, 8)])
@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("engine", ["popcount", "mfma", "auto"])
def test_zero_jump_products_are_identical(qgtc, oracle, a, w, ob, mode, engine):
    """Grouped launch with the occupancy bitmap (tiles neither loaded nor multiplied) against the
    oracle: block-diagonal-dominant sparse left operands, including an all-zero one. Both engines;
    the matrix-core one jumps 128-row tiles when the bitmap is one word per row tile (K <= 8192)
    and visits every k-quad otherwise (the K = 9000 problem)."""
    import torch
    from helpers import rand_q, to_dev
    rng = np.random.default_rng(7 * a + w + mode)
    dims = [(1213, 1213, 128), (300, 9000, 40), (37, 37, 10), (640, 640, 128), (64, 256, 32)]
    Xs, Ws, refs = [], [], []
    for i, (M, K, N) in enumerate(dims):
        qx = rand_q(rng, M, K, a, 0.003)
        for blk in range(0, min(M, K), 64):          # dense diagonal blocks, empty elsewhere
            qx[blk:blk + 64, :] *= 0
            qx[blk:blk + 64, blk:blk + 64] = rng.integers(0, 2 ** a, size=qx[blk:blk + 64, blk:blk + 64].shape)
        if i == 4:
            qx[:] = 0
        qw = rand_q(rng, K, N, w)
        X, Wt = oracle.pack(qx, a, False), oracle.pack(qw, w, True)
        Xs.append(to_dev(torch, X, (a * ((M + 7) // 8 * 8), (K + 127) // 128 * 4)))
        Ws.append(to_dev(torch, Wt, (w * ((K + 127) // 128 * 4), (N + 127) // 128 * 128)))
        refs.append((X, Wt))
    for zj in (True, False):
        bg = qgtc.BatchedGemm(Xs, Ws, dims, a, w, ob, mode, True, zj)
        qgtc.set_engine(engine)
        try:
            bg.run()
        finally:
            qgtc.set_engine("auto")
        torch.cuda.synchronize()
        for i, (M, K, N) in enumerate(dims):
            X, Wt = refs[i]
            if mode == 2:
                np.testing.assert_array_equal(bg.outs[i].cpu().numpy(), oracle.bitmm2int(X, Wt, M, K, N, a, w, True))
            else:
                np.testing.assert_array_equal(to_np_u32(bg.outs[i]),
                                              oracle.bitmm2bit(X, Wt, M, K, N, a, w, ob, col=(mode == 1)))


def test_gcnconv_qnt_module_matches_integer_reference(qgtc, oracle):
    """qgtc_ppopp22_amd/conv.py (the working QGTC_conv.py): two quantised layers against integer
    NumPy on the quantised values, with dense and edge-list adjacency."""
    import torch
    from oracle.qgtc_oracle import np_quantize, np_requant
    from qgtc_ppopp22_amd.conv import GCNConv_Qnt

    torch.manual_seed(3)
    rng = np.random.default_rng(3)
    n, F, H, C, wb, ab = 150, 40, 24, 7, 2, 3
    row, col = rng.integers(0, n, size=900), rng.integers(0, n, size=900)
    A = np_dense_adjacency(row, col, n, n)
    X = (rng.standard_normal((n, F)) * 3).astype(np.float32)
    model = GCNConv_Qnt(F, H, C, w_bit=wb, act_bit=ab).cuda()
    out_dense = model(torch.from_numpy(A).cuda(), torch.from_numpy(X).cuda())
    out_edges = model((torch.from_numpy(row).cuda(), torch.from_numpy(col).cuda(), n), torch.from_numpy(X).cuda())
    assert torch.equal(out_dense, out_edges)

    low = lambda q, b: (q.astype(np.int64) & ((1 << b) - 1))          # only bits 0..b-1 are packed
    qA, qX = low(np_quantize(A, 1), 1), low(np_quantize(X, ab), ab)
    qWi = low(np_quantize(model.W_in.detach().cpu().numpy(), wb), wb)
    qWo = low(np_quantize(model.W_out.detach().cpu().numpy(), wb), wb)
    rq = lambda c: low(np_requant(c.astype(np.int32), ab), ab)
    h = rq(qA @ rq(qX @ qWi))
    want = (qA @ rq(h @ qWo)).astype(np.float32)
    np.testing.assert_array_equal(out_dense.cpu().numpy(), want)
    # an in-place parameter update must not leave stale packed weights behind
    with torch.no_grad():
        model.W_out.mul_(0.0).add_(1.0)
    want1 = (qA @ rq(h @ np.ones_like(qWo))).astype(np.float32)
    np.testing.assert_array_equal(model(torch.from_numpy(A).cuda(), torch.from_numpy(X).cuda()).cpu().numpy(), want1)


MFMA_CASES = [
    # M, K, N, a, w, ob
    (128, 128, 128, 1, 1, 1), (64, 256, 128, 1, 1, 1), (200, 1000, 150, 1, 2, 2), (129, 130, 257, 2, 2, 3),
    (1213, 1213, 128, 1, 2, 2), (300, 4096, 300, 1, 4, 4), (77, 513, 40, 3, 5, 6), (512, 640, 256, 7, 7, 8),
    (33, 33, 33, 1, 7, 2), (256, 9000, 128, 1, 1, 1),
    # 8-plane operands: plane 7 inverted on the way in, corrected with line sums in the epilogue
    (130, 300, 140, 1, 8, 8), (200, 520, 129, 8, 8, 8), (77, 129, 40, 8, 3, 4), (129, 1000, 257, 3, 8, 2),
    (64, 128, 64, 8, 8, 1),
]


@pytest.mark.parametrize("M,K,N,a,w,ob", MFMA_CASES)
def test_mfma_engine_matches_oracle_and_popcount(qgtc, oracle, M, K, N, a, w, ob):
    """The opt-in matrix-core engine (bit planes expanded to int8, v_mfma_i32_32x32x32_i8) must give
    the popcount engine's and the oracle's words, in all three output forms."""
    import torch
    from helpers import rand_q, to_dev
    from qgtc_ppopp22_amd.shapes import cols_shape, rows_shape
    rng = np.random.default_rng(M * 3 + K + N + a + w)
    qx, qw = rand_q(rng, M, K, a), rand_q(rng, K, N, w)
    X, Wt = oracle.pack(qx, a, False), oracle.pack(qw, w, True)
    dX, dW = to_dev(torch, X, rows_shape(M, K, a)), to_dev(torch, Wt, cols_shape(K, N, w))
    assert qgtc.get_engine() == "auto"      # the shipped default
    qgtc.set_engine("popcount")
    pop = (qgtc.bitMM2Bit(dX, dW, M, K, N, a, w, ob), qgtc.bitMM2Bit_col(dX, dW, M, K, N, a, w, ob),
           qgtc.bitMM2Int(dX, dW, M, K, N, a, w, True))
    qgtc.set_engine("mfma")
    try:
        mf = (qgtc.bitMM2Bit(dX, dW, M, K, N, a, w, ob), qgtc.bitMM2Bit_col(dX, dW, M, K, N, a, w, ob),
              qgtc.bitMM2Int(dX, dW, M, K, N, a, w, True))
    finally:
        qgtc.set_engine("auto")
    np.testing.assert_array_equal(to_np_u32(mf[0]), oracle.bitmm2bit(X, Wt, M, K, N, a, w, ob))
    np.testing.assert_array_equal(to_np_u32(mf[1]), oracle.bitmm2bit(X, Wt, M, K, N, a, w, ob, col=True))
    np.testing.assert_array_equal(mf[2].cpu().numpy(), oracle.bitmm2int(X, Wt, M, K, N, a, w, True))
    np.testing.assert_array_equal(mf[2].cpu().numpy(), (qx.astype(np.int64) @ qw.astype(np.int64)).astype(np.float32))
    for a_, b_ in zip(mf, pop):
        assert torch.equal(a_, b_)


def test_mfma_engine_falls_back_above_8_bits(qgtc, oracle):
    import torch
    from helpers import rand_q, to_dev
    from qgtc_ppopp22_amd.shapes import cols_shape, rows_shape
    rng = np.random.default_rng(8)
    M, K, N, a, w, ob = 130, 300, 140, 1, 9, 8
    qx, qw = rand_q(rng, M, K, a), rand_q(rng, K, N, w)
    X, Wt = oracle.pack(qx, a, False), oracle.pack(qw, w, True)
    dX, dW = to_dev(torch, X, rows_shape(M, K, a)), to_dev(torch, Wt, cols_shape(K, N, w))
    qgtc.set_engine("mfma")
    try:
        got = qgtc.bitMM2Bit(dX, dW, M, K, N, a, w, ob)
    finally:
        qgtc.set_engine("auto")
    np.testing.assert_array_equal(to_np_u32(got), oracle.bitmm2bit(X, Wt, M, K, N, a, w, ob))
    with pytest.raises(RuntimeError):
        qgtc.set_engine("tensor-core")


@pytest.mark.parametrize("M,K,N,a,w,ob", [(2048, 2048, 1024, 2, 2, 2), (2048, 2048, 64, 1, 1, 1)])
def test_auto_engine_same_words_either_way(qgtc, M, K, N, a, w, ob):
    """set_engine('auto') picks by a cost model (matrix cores for the wide product, popcount for the
    64-column one); whichever it picks, the words are those of the parity-tested popcount engine."""
    import torch
    g = torch.Generator(device="cpu").manual_seed(M + N)
    from qgtc_ppopp22_amd.shapes import cols_shape, rows_shape
    dX = torch.randint(-2**31, 2**31 - 1, rows_shape(M, K, a), dtype=torch.int32, generator=g).cuda()
    dW = torch.randint(-2**31, 2**31 - 1, cols_shape(K, N, w), dtype=torch.int32, generator=g).cuda()
    qgtc.set_engine("popcount")
    pop = (qgtc.bitMM2Bit(dX, dW, M, K, N, a, w, ob), qgtc.bitMM2Bit_col(dX, dW, M, K, N, a, w, ob),
           qgtc.bitMM2Int(dX, dW, M, K, N, a, w, True))
    qgtc.set_engine("auto")
    try:
        assert qgtc.get_engine() == "auto"
        au = (qgtc.bitMM2Bit(dX, dW, M, K, N, a, w, ob), qgtc.bitMM2Bit_col(dX, dW, M, K, N, a, w, ob),
              qgtc.bitMM2Int(dX, dW, M, K, N, a, w, True))
    finally:
        qgtc.set_engine("auto")
    for x, y in zip(au, pop):
        assert torch.equal(x, y)


def test_jump_decision_is_made_on_the_device(qgtc, oracle):
    """BatchedGemm(zero_jump=True) fills all bitmaps with one grouped launch and a one-workgroup kernel
    keeps or clears them (a quarter of the tiles occupied is the threshold); the host only learns the
    outcome when it asks. Counts equal the oracle's bitmaps; products are the same either way."""
    import torch
    from helpers import rand_q, to_dev
    rng = np.random.default_rng(77)
    a, w, ob = 1, 2, 2
    dims = [(300, 1213, 64), (1213, 1213, 128), (90, 200, 10)]
    for density, expect_jump in ((0.00003, True), (0.2, False)):
        Xs, Ws, want, set_bits, all_tiles = [], [], [], 0, 0
        for (M, K, N) in dims:
            qx, qw = rand_q(rng, M, K, a, density), rand_q(rng, K, N, w)
            X, Wt = oracle.pack(qx, a, False), oracle.pack(qw, w, True)
            Xs.append(to_dev(torch, X, (a * ((M + 7) // 8 * 8), (K + 127) // 128 * 4)))
            Ws.append(to_dev(torch, Wt, (w * ((K + 127) // 128 * 4), (N + 127) // 128 * 128)))
            want.append(oracle.bitmm2bit(X, Wt, M, K, N, a, w, ob))
            occ = np_tile_occupancy(X, M, K, a)
            set_bits += int(sum(bin(int(v)).count("1") for v in occ))
            all_tiles += ((M + 31) // 32) * ((K + 127) // 128)
        bg = qgtc.BatchedGemm(Xs, Ws, dims, a, w, ob, 0, True, True)
        bg.run()
        assert bg.zero_jump == expect_jump
        assert abs(bg.occupied_fraction - set_bits / all_tiles) < 1e-12
        for i in range(len(dims)):
            np.testing.assert_array_equal(to_np_u32(bg.outs[i]), want[i])
            np.testing.assert_array_equal(bg.occs[i].cpu().numpy().view(np.uint64),
                                          np_tile_occupancy(to_np_u32(Xs[i]), dims[i][0], dims[i][1], a))
        # a second stage reusing the bitmaps reaches the same decision
        bg2 = qgtc.BatchedGemm(Xs, Ws, dims, a, w, ob, 0, True, True, bg.occs)
        bg2.run()
        assert bg2.zero_jump == expect_jump
        for i in range(len(dims)):
            np.testing.assert_array_equal(to_np_u32(bg2.outs[i]), want[i])


def test_fp4_narrow_kernels_sweep(qgtc, oracle):
    """The FP4 matrix-core kernels for narrow right operands (k_bitmm_fp4_skinny: single launches, N <= 256;
    k_bitmm_fp4_rows: grouped launches) on shapes picked for their corners: K long enough for several
    super-steps per wave and ragged at every granularity (32 bits, 128 bits, 512 bits), 1..8 planes in base-4
    digits, all-zero row blocks (zero-tile skipping on and off), the float32-exactness bound (above it the
    other kernels must take over) - against the oracle, rows-layout bits and float32."""
    import torch
    from helpers import rand_q, to_dev
    from qgtc_ppopp22_amd.shapes import cols_shape, rows_shape
    rng = np.random.default_rng(424242)
    cases = [  # M, K, N, a, w, ob
        (100, 9000, 64, 1, 1, 1), (70, 20000, 33, 1, 2, 2), (33, 40000, 10, 2, 2, 3), (300, 4100, 200, 1, 3, 2),
        (50, 5000, 256, 1, 8, 8), (64, 2048, 17, 2, 8, 4), (129, 700, 129, 2, 5, 6), (40, 70000, 40, 1, 7, 8),
        (16, 512, 32, 1, 4, 4), (31, 130, 31, 2, 1, 1),
        (20, 70000, 20, 1, 8, 8),     # 70000 * 255 > 2^24: not exact in float32 -> the other kernels
    ]
    qgtc.set_engine("mfma")
    try:
        for (M, K, N, a, w, ob) in cases:
            qx, qw = rand_q(rng, M, K, a), rand_q(rng, K, N, w)
            qx[M // 3: M // 3 + 17] = 0                       # a block of all-zero rows
            X, Wt = oracle.pack(qx, a, False), oracle.pack(qw, w, True)
            dX, dW = to_dev(torch, X, rows_shape(M, K, a)), to_dev(torch, Wt, cols_shape(K, N, w))
            tag = f"M={M} K={K} N={N} a={a} w={w} ob={ob}"
            want_b = oracle.bitmm2bit(X, Wt, M, K, N, a, w, ob)
            want_f = oracle.bitmm2int(X, Wt, M, K, N, a, w, True)
            for zs in (True, False):
                qgtc.set_zero_skip(zs)
                try:
                    np.testing.assert_array_equal(to_np_u32(qgtc.bitMM2Bit(dX, dW, M, K, N, a, w, ob)), want_b, err_msg=tag)
                    np.testing.assert_array_equal(to_np_u32(qgtc.bitMM2Bit_col(dX, dW, M, K, N, a, w, ob)),
                                                  oracle.bitmm2bit(X, Wt, M, K, N, a, w, ob, col=True), err_msg=tag + " cols")
                    np.testing.assert_array_equal(qgtc.bitMM2Int(dX, dW, M, K, N, a, w, True).cpu().numpy(), want_f, err_msg=tag)
                finally:
                    qgtc.set_zero_skip(True)
            for mode in (0, 1, 2):                            # grouped: two copies of the problem
                for zj in (False, True):
                    bg = qgtc.BatchedGemm([dX, dX], [dW, dW], [(M, K, N)] * 2, a, w, ob, mode, True, zj)
                    bg.run()
                    for o in bg.outs:
                        if mode == 2:
                            np.testing.assert_array_equal(o.cpu().numpy(), want_f, err_msg=tag + f" grouped mode {mode}")
                        else:
                            np.testing.assert_array_equal(to_np_u32(o), oracle.bitmm2bit(X, Wt, M, K, N, a, w, ob, col=(mode == 1)),
                                                          err_msg=tag + f" grouped mode {mode} jump {zj}")
    finally:
        qgtc.set_engine("auto")


# ---------------------------------------------------------------------------------------------
# The FP4 matrix-core kernels accumulate in float32; the library only routes a product to them while
# K (2^a - 1)(2^w - 1) < 2^24 (launch_common.hip.h: skinny_ok / rows_ok / fp4_ok / rows_single_ok), i.e. while every partial sum
# is an exactly representable integer. That predicate is the whole safety case of those kernels (the reference
# accumulates in int32, kernel.h:292-341), so it is tested at its edge with worst-case operands.
# ---------------------------------------------------------------------------------------------
def _kmax(a, w):
    return (2 ** 24 - 1) // ((2 ** a - 1) * (2 ** w - 1))


def _k_edges(a, w):
    """The largest K the FP4 kernels are admitted for - the library counts whole k-quads, PAD128(K) (2^a - 1)(2^w - 1) < 2^24, since round 6
    (launch_common.hip.h::no_wrap: padding bits are multiplied like any others) - and one above; then the largest K whose sums are exact at
    all and one above (by then on the int32 kernels)."""
    km = _kmax(a, w)
    return sorted({max(km // 128 * 128, 1), km // 128 * 128 + 1, km, km + 1})


def _packed_pair(torch, oracle, qx, qw, a, w):
    from helpers import to_dev
    from qgtc_ppopp22_amd.shapes import cols_shape, rows_shape
    M, K = qx.shape
    N = qw.shape[1]
    X, Wt = oracle.pack(qx, a, False), oracle.pack(qw, w, True)
    return X, Wt, to_dev(torch, X, rows_shape(M, K, a)), to_dev(torch, Wt, cols_shape(K, N, w))


@pytest.mark.parametrize("a,w", [(1, 8), (2, 8), (2, 2), (1, 1), (2, 4), (4, 4), (4, 8), (5, 5), (6, 8), (8, 8)])   # (3-8 left planes: k_bitmm_fp4_rows_single)
@pytest.mark.parametrize("engine", ["auto", "mfma"])
def test_float32_exactness_bound_all_max_operands(qgtc, oracle, a, w, engine):
    """All-max X times all-max W at the LARGEST K the FP4 kernels are admitted for (every output = K (2^a-1)(2^w-1)
    <= 2^24 - 1, the largest float32 sum that can occur) and at K + 1 (the int32 kernels must take over): exact
    float32 output, exact packed output, and equal to the oracle."""
    import torch
    M, N = 20, 24
    for K in _k_edges(a, w):
        if a == 1 and w == 1:
            K = min(K, 3000000)     # 2^24 - 1 bits of K per row is only a size issue, not an exactness one
        qx = np.full((M, K), 2 ** a - 1, dtype=np.int32)
        qw = np.full((K, N), 2 ** w - 1, dtype=np.int32)
        X, Wt, dX, dW = _packed_pair(torch, oracle, qx, qw, a, w)
        want = K * (2 ** a - 1) * (2 ** w - 1)
        with use_engine(qgtc, engine):
            f = qgtc.bitMM2Int(dX, dW, M, K, N, a, w, True).cpu().numpy()
            b = qgtc.bitMM2Bit(dX, dW, M, K, N, a, w, 8)
        assert want < 2 ** 31
        np.testing.assert_array_equal(f, np.full((M, N), np.float32(want)), err_msg=f"K={K}")
        if float(np.float32(want)) == want:
            assert (f.astype(np.int64) == want).all()
        np.testing.assert_array_equal(f, oracle.bitmm2int(X, Wt, M, K, N, a, w, True))
        np.testing.assert_array_equal(to_np_u32(b), oracle.bitmm2bit(X, Wt, M, K, N, a, w, 8))


@pytest.mark.parametrize("a,w", [(1, 8), (2, 8), (2, 2), (4, 8), (6, 8), (8, 8)])
def test_float32_sums_do_not_truncate_small_addends(qgtc, oracle, a, w):
    """A pattern aimed at the INSIDE of the block-scaled MFMA: most of K drives every accumulator close to 2^24 with
    all-max operands, then the remaining k-quads contribute products of every magnitude down to single 1 x 1 pairs
    (lowest base-4 digits, scale 2 x 2) interleaved with all-max ones (highest digits, scale up to 2 x 128) in the
    SAME 128-element instruction. A datapath that aligned the addends of its 128-term sum to the largest one and
    dropped low bits would lose them; int32 accumulation (the reference, kernel.h:292-341) does not."""
    import torch
    rng = np.random.default_rng(1000 * a + w)
    M, N = 33, 40
    K = _kmax(a, w)
    ma, mw = 2 ** a - 1, 2 ** w - 1
    for tail in (512, 4096 + 77):              # ragged tails, several super-steps
        tail = min(tail, K // 2)               # (8 x 8 bits: K = 258)
        qx = np.full((M, K), ma, dtype=np.int32)
        qw = np.full((K, N), mw, dtype=np.int32)
        t0 = K - tail
        # the tail: alternating all-max and tiny values element by element, then pure 1 x 1 pairs, then zeros
        alt = (np.arange(tail) % 2 == 0)
        qx[:, t0:] = np.where(alt[None, :], ma, rng.integers(0, 2, size=(M, tail)))
        qw[t0:, :] = np.where(alt[:, None], mw, rng.integers(0, 2, size=(tail, N)))
        third = tail // 3
        qx[:, K - third:] = rng.integers(0, 2, size=(M, third))
        qw[K - third:, :] = (rng.random((third, N)) < 0.05).astype(np.int32)
        # keep every sum below 2^24: the head alone stays under the bound since the tail replaced all-max entries
        exact = (qx.astype(np.float64) @ qw.astype(np.float64)).astype(np.int64)     # BLAS; exact far below 2^53
        assert exact.max() < 2 ** 24 and exact.max() > 2 ** 23
        X, Wt, dX, dW = _packed_pair(torch, oracle, qx, qw, a, w)
        for engine in ("auto", "mfma"):
            with use_engine(qgtc, engine):
                f = qgtc.bitMM2Int(dX, dW, M, K, N, a, w, True).cpu().numpy()
                c = qgtc.bitMM2Bit_col(dX, dW, M, K, N, a, w, 3)
            np.testing.assert_array_equal(f.astype(np.int64), exact, err_msg=f"{engine} tail={tail}")
            np.testing.assert_array_equal(to_np_u32(c), oracle.bitmm2bit(X, Wt, M, K, N, a, w, 3, col=True))


@pytest.mark.parametrize("a,w", [(4, 8), (2, 8), (4, 4)])
def test_float32_exactness_bound_grouped(qgtc, oracle, a, w):
    """The same edge for the grouped FP4 kernel (k_bitmm_fp4_rows: a <= 8, w <= 8, N <= 256): all-max operands at the
    largest admitted K and one above (decided from max_K of the launch), all three output modes."""
    import torch
    M, N = 40, 33
    for K in _k_edges(a, w):
        rng = np.random.default_rng(K)
        qx = np.full((M, K), 2 ** a - 1, dtype=np.int32)
        qw = np.full((K, N), 2 ** w - 1, dtype=np.int32)
        qx2 = qx.copy()
        qx2[:, K // 2:] = rng.integers(0, 2, size=(M, K - K // 2))       # a second problem with small late addends
        X, Wt, dX, dW = _packed_pair(torch, oracle, qx, qw, a, w)
        X2, _, dX2, _ = _packed_pair(torch, oracle, qx2, qw, a, w)
        for engine in ("auto", "mfma"):
            for mode in (0, 1, 2):
                with use_engine(qgtc, engine):
                    bg = qgtc.BatchedGemm([dX, dX2], [dW], [(M, K, N)] * 2, a, w, 8, mode, True, False)
                    bg.run()
                for o, Xo in zip(bg.outs, (X, X2)):
                    if mode == 2:
                        np.testing.assert_array_equal(o.cpu().numpy(), oracle.bitmm2int(Xo, Wt, M, K, N, a, w, True), err_msg=f"K={K} {engine}")
                    else:
                        np.testing.assert_array_equal(to_np_u32(o), oracle.bitmm2bit(Xo, Wt, M, K, N, a, w, 8, col=(mode == 1)),
                                                      err_msg=f"K={K} {engine} mode {mode}")


# ---------------------------------------------------------------------------------------------
# the fused GNN layer (qgtc_gcn_layer_batched): X.W re-packed in the cols layout, then A.(XW), ONE call
# ---------------------------------------------------------------------------------------------
def _layer_batches(torch, oracle, rng, dims_nf, f_out, act, wb, a_bits=1):
    from helpers import rand_q, to_dev
    from qgtc_ppopp22_amd.shapes import cols_shape, rows_shape
    qw = rand_q(rng, dims_nf[0][1], f_out, wb)
    Wt = oracle.pack(qw, wb, True)
    dW = to_dev(torch, Wt, cols_shape(dims_nf[0][1], f_out, wb))
    As, Xs, want = [], [], []
    for (n, f_in) in dims_nf:
        qa = rand_q(rng, n, n, a_bits, 0.01)
        for blk in range(0, n, 64):                       # block-diagonal-dominant, like a cluster batch
            qa[blk:blk + 64, blk:blk + 64] = rng.integers(0, 2 ** a_bits, size=qa[blk:blk + 64, blk:blk + 64].shape)
        qx = rand_q(rng, n, f_in, act)
        A, X = oracle.pack(qa, a_bits, False), oracle.pack(qx, act, False)
        As.append(to_dev(torch, A, rows_shape(n, n, a_bits)))
        Xs.append(to_dev(torch, X, rows_shape(n, f_in, act)))
        T = oracle.bitmm2bit(X, Wt, n, f_in, f_out, act, wb, act, col=True)
        want.append((T, oracle.bitmm2bit(A, T, n, n, f_out, a_bits, act, act), oracle.bitmm2int(A, T, n, n, f_out, a_bits, act, True)))
    return As, Xs, dW, want


@pytest.mark.parametrize("f_in,f_out,act,wb", [(128, 128, 2, 2), (50, 64, 4, 4), (128, 10, 2, 2), (64, 200, 1, 1), (40, 33, 3, 5),
                                               (128, 128, 8, 8), (64, 10, 9, 2)])
@pytest.mark.parametrize("engine", ["auto", "mfma", "popcount"])
@pytest.mark.parametrize("zero_jump", [False, True])
def test_fused_layer_matches_oracle(qgtc, oracle, f_in, f_out, act, wb, engine, zero_jump):
    """FusedLayer (one call per GNN layer for a group of cluster batches: two grouped launches) against the oracle's two
    products, both output forms, batches of ragged sizes (incl. fewer rows than a tile), run several times, on every engine."""
    import torch
    rng = np.random.default_rng(f_in * 7 + f_out + act + wb)
    dims_nf = [(1213, f_in), (640, f_in), (37, f_in), (129, f_in), (300, f_in)]
    As, Xs, dW, want = _layer_batches(torch, oracle, rng, dims_nf, f_out, act, wb)
    d1 = [(n, f, f_out) for n, f in dims_nf]
    d2 = [(n, n, f_out) for n, _ in dims_nf]
    with use_engine(qgtc, engine):
        for mode2 in (0, 2):
            s1 = qgtc.BatchedGemm(Xs, [dW], d1, act, wb, act, 1, True)
            s2 = qgtc.BatchedGemm(As, s1.outs, d2, 1, act, act, mode2, True, zero_jump)
            layer = qgtc.FusedLayer(s1, s2)
            for rep in range(3):
                layer.run()
                torch.cuda.synchronize()
                for i in range(len(dims_nf)):
                    np.testing.assert_array_equal(to_np_u32(s1.outs[i]), want[i][0], err_msg=f"T of batch {i}, run {rep}")
                    if mode2 == 2:
                        np.testing.assert_array_equal(layer.outs[i].cpu().numpy(), want[i][2], err_msg=f"float out of batch {i}, run {rep}")
                    else:
                        np.testing.assert_array_equal(to_np_u32(layer.outs[i]), want[i][1], err_msg=f"bits out of batch {i}, run {rep}")
                if rep == 0:
                    for o in list(s1.outs) + list(layer.outs):      # the next run must rewrite everything
                        o.fill_(-1 if o.dtype == torch.int32 else 7.0)


def test_fused_layer_rejects_plans_that_do_not_chain(qgtc, oracle):
    import torch
    rng = np.random.default_rng(5)
    As, Xs, dW, _ = _layer_batches(torch, oracle, rng, [(100, 64), (70, 64)], 32, 2, 2)
    s1 = qgtc.BatchedGemm(Xs, [dW], [(100, 64, 32), (70, 64, 32)], 2, 2, 2, 1, True)
    s1_rows = qgtc.BatchedGemm(Xs, [dW], [(100, 64, 32), (70, 64, 32)], 2, 2, 2, 0, True)
    s2 = qgtc.BatchedGemm(As, s1.outs, [(100, 100, 32), (70, 70, 32)], 1, 2, 2, 0, True)
    with pytest.raises(RuntimeError, match="cols-layout"):
        qgtc.FusedLayer(s1_rows, s2)
    s2_other = qgtc.BatchedGemm(As, s1_rows.outs, [(100, 100, 32), (70, 70, 32)], 1, 2, 2, 0, True)
    with pytest.raises(RuntimeError, match="must be stage 1's output"):
        qgtc.FusedLayer(s1, s2_other)
    s2_bits = qgtc.BatchedGemm(As, s1.outs, [(100, 100, 32), (70, 70, 32)], 1, 3, 2, 0, True)
    with pytest.raises(RuntimeError, match="output bits"):
        qgtc.FusedLayer(s1, s2_bits)


@pytest.mark.parametrize("n,f_in,f_out,act,wb", [(300, 64, 64, 2, 2), (1213, 128, 128, 2, 2), (599, 50, 10, 4, 4), (40, 33, 200, 3, 3)])
def test_gcn_layer_single_subgraph(qgtc, oracle, n, f_in, f_out, act, wb):
    """QGTC.gcn_layer (what conv.py's Aggregation_Qnt calls): one subgraph, one call, both output forms."""
    import torch
    rng = np.random.default_rng(n + f_in + f_out)
    As, Xs, dW, want = _layer_batches(torch, oracle, rng, [(n, f_in)], f_out, act, wb)
    for engine in ("auto", "popcount"):
        with use_engine(qgtc, engine):
            bits = qgtc.gcn_layer(As[0], Xs[0], dW, n, f_in, f_out, 1, act, wb, False)
            flt = qgtc.gcn_layer(As[0], Xs[0], dW, n, f_in, f_out, 1, act, wb, True)
            np.testing.assert_array_equal(to_np_u32(bits), want[0][1])
            np.testing.assert_array_equal(flt.cpu().numpy(), want[0][2])



def test_small_k_cols_layout_stages(qgtc, oracle):
    """Grouped launches with K <= 128 and cols-layout output (the X . W stages of the epochs: k_bitmm_fp4_xw_rows at 2 / 4 bits,
    k_bitmm_fp4_rows otherwise) against the oracle: ragged rows (not a multiple of 32 / 128), K from 1 to 128, N from 1 column to several 128-line groups,
    1..8 planes in base-4 digits, 1..10 output planes, all-zero row blocks, problems of different sizes in one launch
    (the waves past a smaller problem's last column write its padding lines)."""
    import torch
    from helpers import rand_q, to_dev
    from qgtc_ppopp22_amd.shapes import cols_shape, rows_shape
    rng = np.random.default_rng(31337)
    for (a, w, ob, K, N) in [(2, 2, 2, 128, 128), (4, 4, 4, 50, 64), (2, 2, 2, 128, 10), (1, 1, 1, 1, 1), (3, 5, 10, 77, 130),
                             (4, 8, 8, 128, 33), (1, 8, 3, 100, 300), (2, 1, 9, 64, 96)]:
        dims = [(1213, K, N), (37, K, N), (640, K, max(1, N // 2)), (129, K, N), (32, K, N)]
        Xs, Ws, want = [], [], []
        for (M, K_, N_) in dims:
            qx, qw = rand_q(rng, M, K_, a), rand_q(rng, K_, N_, w)
            qx[M // 2: M // 2 + 40] = 0
            X, Wt = oracle.pack(qx, a, False), oracle.pack(qw, w, True)
            Xs.append(to_dev(torch, X, rows_shape(M, K_, a)))
            Ws.append(to_dev(torch, Wt, cols_shape(K_, N_, w)))
            want.append(oracle.bitmm2bit(X, Wt, M, K_, N_, a, w, ob, col=True))
        for engine in ("auto", "mfma"):
            with use_engine(qgtc, engine):
                bg = qgtc.BatchedGemm(Xs, Ws, dims, a, w, ob, 1, True)
                for o in bg.outs:
                    o.fill_(-1)
                bg.run()
            for i, o in enumerate(bg.outs):
                np.testing.assert_array_equal(to_np_u32(o), want[i], err_msg=f"{engine} a={a} w={w} ob={ob} problem {dims[i]}")


@pytest.mark.parametrize("a,w,ob", [(1, 2, 2), (1, 4, 4), (4, 8, 8), (2, 2, 10), (1, 1, 1), (3, 5, 3)])
def test_row_block_kernel_for_sparse_left_operands(qgtc, oracle, a, w, ob):
    """k_bitmm_fp4_rows (grouped launches with occupancy bitmaps or narrow outputs: the A . (XW) stages) against the
    oracle: K up to 64 k-quads (8192: the bitmap is one word per 32-row tile), odd and even numbers of occupied k-quads
    per row block (k-quads are multiplied in pairs), an all-zero problem, N from 1 to 256 columns, rows-layout bits and
    float32, with and without the bitmap, problems of different sizes in one launch."""
    import torch
    from helpers import rand_q, to_dev
    from qgtc_ppopp22_amd.shapes import cols_shape, rows_shape
    rng = np.random.default_rng(1000 * a + 10 * w + ob)
    for N in (10, 64, 128, 200, 256):
        dims = [(1213, 1213, N), (70, 8192, N), (333, 5000, max(1, N - 7)), (40, 300, N), (129, 129, N)]
        if 8192 * (2 ** a - 1) * (2 ** w - 1) >= 2 ** 24:
            dims[1], dims[2] = (70, 900, N), (333, 1000, max(1, N - 7))     # keep inside the float32 exactness bound
        Xs, Ws, refs = [], [], []
        for i, (M, K, N_) in enumerate(dims):
            qx = rand_q(rng, M, K, a, 0.0004)
            for blk in range(0, min(M, K), 64):          # dense diagonal blocks, a few stray bits elsewhere
                qx[blk:blk + 64, blk:blk + 64] = rng.integers(0, 2 ** a, size=qx[blk:blk + 64, blk:blk + 64].shape)
            if i == 3:
                qx[:] = 0
            qw = rand_q(rng, K, N_, w)
            X, Wt = oracle.pack(qx, a, False), oracle.pack(qw, w, True)
            Xs.append(to_dev(torch, X, rows_shape(M, K, a)))
            Ws.append(to_dev(torch, Wt, cols_shape(K, N_, w)))
            refs.append((X, Wt))
        for mode in (0, 2):
            for zj in (True, False):
                with use_engine(qgtc, "auto"):
                    bg = qgtc.BatchedGemm(Xs, Ws, dims, a, w, ob, mode, True, zj)
                    for o in bg.outs:
                        o.fill_(-1 if mode == 0 else 7.0)
                    bg.run()
                for i, (M, K, N_) in enumerate(dims):
                    X, Wt = refs[i]
                    if mode == 2:
                        np.testing.assert_array_equal(bg.outs[i].cpu().numpy(), oracle.bitmm2int(X, Wt, M, K, N_, a, w, True),
                                                      err_msg=f"N={N} problem {i} float, bitmap {zj}")
                    else:
                        np.testing.assert_array_equal(to_np_u32(bg.outs[i]), oracle.bitmm2bit(X, Wt, M, K, N_, a, w, ob),
                                                      err_msg=f"N={N} problem {i} bits, bitmap {zj}")


@pytest.mark.gpu
@pytest.mark.parametrize("rf", ["", "2", "4", "42"])
@pytest.mark.parametrize("a,w", [(1, 1), (1, 2), (2, 1), (2, 2), (1, 4), (4, 1), (2, 4), (4, 2), (1, 8), (2, 8), (8, 1), (8, 2)])
def test_wide_operand_kernel_equals_the_oracle(qgtc, oracle, monkeypatch, rf, a, w):
    """k_bitmm_fp4_wide (one-, two- and four-plane operands, N > 256: packed words staged by LDS-DMA, expanded in
    registers, base-4 digits for four planes) against the oracle in all three output forms: ragged M and N, K with 1..8 k-quads in the last group, fewer lines than
    a tile, several output widths (one / two planes have their own epilogue), the three tile shapes forced in turn."""
    import torch
    from helpers import rand_q, to_dev
    from qgtc_ppopp22_amd.shapes import cols_shape, rows_shape
    if rf:
        monkeypatch.setenv("QGTC_WIDE_RF", rf)
    rng = np.random.default_rng(77 + 10 * a + w)
    shapes = ((129, 1024, 257), (300, 896, 513), (8, 3968, 264), (77, 128, 1000), (513, 2176, 1030), (1000, 1152, 300), (64, 8320, 520),
              (1, 5, 300), (3, 4097, 258))
    for (M, K, N) in (shapes if not rf else shapes[1:4]):
        qx, qw = rand_q(rng, M, K, a), rand_q(rng, K, N, w)
        X, Wt = oracle.pack(qx, a, False), oracle.pack(qw, w, True)
        bX, bW = to_dev(torch, X, rows_shape(M, K, a)), to_dev(torch, Wt, cols_shape(K, N, w))
        for eng in ("mfma", "auto"):
            with use_engine(qgtc, eng):
                for ob in (1, 2, 3, 10):
                    np.testing.assert_array_equal(to_np_u32(qgtc.bitMM2Bit(bX, bW, M, K, N, a, w, ob)), oracle.bitmm2bit(X, Wt, M, K, N, a, w, ob),
                                                  err_msg=f"{M}x{K}x{N} rows ob={ob} {eng}")
                    np.testing.assert_array_equal(to_np_u32(qgtc.bitMM2Bit_col(bX, bW, M, K, N, a, w, ob)), oracle.bitmm2bit(X, Wt, M, K, N, a, w, ob, col=True),
                                                  err_msg=f"{M}x{K}x{N} cols ob={ob} {eng}")
                np.testing.assert_array_equal(qgtc.bitMM2Int(bX, bW, M, K, N, a, w, True).cpu().numpy(), oracle.bitmm2int(X, Wt, M, K, N, a, w, True),
                                              err_msg=f"{M}x{K}x{N} float {eng}")


@pytest.mark.gpu
@pytest.mark.parametrize("engine", ["auto", "mfma"])
def test_wide_kernel_at_the_float32_exactness_bound(qgtc, engine):
    """The wide-operand kernel at the edge of its admission predicate (K (2^a-1)(2^w-1) < 2^24): all-max 2 x 2-bit
    operands at K = 1 864 135, where every sum is 2^24 - 1, and at K + 1, where the int32 kernels must take over
    (sum 2^24 + 8). The packed operands are built word by word (the dense K x N array would be 2 GB) and the expected
    values are known in closed form."""
    import torch
    from qgtc_ppopp22_amd.shapes import cols_shape, rows_shape
    M, N, a, w = 8, 257, 2, 2
    for K in (_kmax(a, w), _kmax(a, w) + 1):
        line = torch.zeros(rows_shape(M, K, a)[-1], dtype=torch.int64)
        line[:K // 32] = 0xFFFFFFFF
        if K % 32:
            line[K // 32] = 0xFFFFFFFF ^ (0xFFFFFFFF >> (K % 32))      # element i at bit 31 - (i & 31)
        line = line.to(torch.uint32).view(torch.int32)
        bX = torch.zeros(rows_shape(M, K, a), dtype=torch.int32)
        bW = torch.zeros(cols_shape(K, N, w), dtype=torch.int32)
        bX.view(a, -1, line.numel())[:, :M] = line
        bW.view(w, -1, line.numel())[:, :N] = line
        bX, bW = bX.cuda(), bW.cuda()
        want = K * 9
        with use_engine(qgtc, engine):
            f = qgtc.bitMM2Int(bX, bW, M, K, N, a, w, True)
            b = qgtc.bitMM2Bit(bX, bW, M, K, N, a, w, 8)
        assert float(np.float32(want)) == want
        assert bool((f == float(want)).all()), f"K={K}: {f.unique()[:4].tolist()} instead of {want}"
        words = to_np_u32(b).reshape(8, -1, 12)                        # [ob][PAD8(M)][STEP128(N) * 4]; requantised to 255
        expect = np.zeros_like(words)
        expect[:, :M, :8] = 0xFFFFFFFF
        expect[:, :M, 8] = 0x80000000                                  # column 256
        np.testing.assert_array_equal(words, expect, err_msg=f"K={K}")


@pytest.mark.gpu
@pytest.mark.parametrize("a,w", [(1, 8), (1, 5), (2, 7)])
def test_headline_kernel_on_tall_tiles(qgtc, oracle, a, w):
    """k_bitmm_fp4_one on 64 x 16 tiles (more than four right-hand planes, N a multiple of 32): ragged M and K,
    rows-layout bits and float32, against the oracle."""
    import torch
    from helpers import rand_q, to_dev
    from qgtc_ppopp22_amd.shapes import cols_shape, rows_shape
    rng = np.random.default_rng(500 + 10 * a + w)
    for (M, K, N) in ((2050, 640, 64), (4100, 1000, 32), (2300, 4096, 96), (515, 300, 64)):
        qx, qw = rand_q(rng, M, K, a), rand_q(rng, K, N, w)
        X, Wt = oracle.pack(qx, a, False), oracle.pack(qw, w, True)
        bX, bW = to_dev(torch, X, rows_shape(M, K, a)), to_dev(torch, Wt, cols_shape(K, N, w))
        for eng in ("auto", "mfma"):
            with use_engine(qgtc, eng):
                for ob in (1, w):
                    np.testing.assert_array_equal(to_np_u32(qgtc.bitMM2Bit(bX, bW, M, K, N, a, w, ob)), oracle.bitmm2bit(X, Wt, M, K, N, a, w, ob),
                                                  err_msg=f"{M}x{K}x{N} rows ob={ob} {eng}")
                np.testing.assert_array_equal(qgtc.bitMM2Int(bX, bW, M, K, N, a, w, True).cpu().numpy(), oracle.bitmm2int(X, Wt, M, K, N, a, w, True),
                                              err_msg=f"{M}x{K}x{N} float {eng}")


# ---------------------------------------------------------------------------------------------
# qgtc_gcn_chain_batched: an aggregation stage with the NEXT layer's X.W stage in the same call
# ---------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("f1,f2,act,wb", [(128, 128, 2, 2), (128, 10, 2, 2), (64, 64, 4, 4), (50, 64, 4, 4), (100, 77, 2, 1),
                                          (128, 128, 1, 1), (200, 64, 2, 2), (64, 200, 2, 2), (128, 128, 3, 2)])
@pytest.mark.parametrize("engine", ["auto", "popcount"])
@pytest.mark.parametrize("zero_jump", [False, True])
def test_chained_pair_matches_oracle(qgtc, oracle, f1, f2, act, wb, engine, zero_jump):
    """ChainedPair (out = requant(A . T), then T' = requant(out . W') in the cols layout) against the oracle's two
    products: ragged batches (fewer rows than a block, row counts that leave padding words in T'), widths inside and
    outside the row-block kernels' range (more than 128 columns, the popcount engine), with and without occupancy bitmaps, run
    three times over poisoned outputs."""
    import torch
    from helpers import rand_q, to_dev
    from qgtc_ppopp22_amd.shapes import cols_shape, rows_shape
    rng = np.random.default_rng(31 * f1 + f2 + act + wb)
    ns = [1213, 640, 37, 129, 300, 1025]
    qw = rand_q(rng, f1, f2, wb)
    W2 = oracle.pack(qw, wb, True)
    dW2 = to_dev(torch, W2, cols_shape(f1, f2, wb))
    As, Ts, want = [], [], []
    for n in ns:
        qa = rand_q(rng, n, n, 1, 0.01)
        for blk in range(0, n, 64):
            qa[blk:blk + 64, blk:blk + 64] = rng.integers(0, 2, size=qa[blk:blk + 64, blk:blk + 64].shape)
        qt = rand_q(rng, n, f1, act)
        A, T = oracle.pack(qa, 1, False), oracle.pack(qt, act, True)
        As.append(to_dev(torch, A, rows_shape(n, n, 1)))
        Ts.append(to_dev(torch, T, cols_shape(n, f1, act)))
        out = oracle.bitmm2bit(A, T, n, n, f1, 1, act, act)
        want.append((out, oracle.bitmm2bit(out, W2, n, f1, f2, act, wb, act, col=True), oracle.bitmm2int(out, W2, n, f1, f2, act, wb, True)))
    with use_engine(qgtc, engine):
        for mode2 in (1, 2):      # T' as cols-layout bits / the output layer's float32
            sa = qgtc.BatchedGemm(As, Ts, [(n, n, f1) for n in ns], 1, act, act, 0, True, zero_jump)
            sx = qgtc.BatchedGemm(sa.outs, [dW2], [(n, f1, f2) for n in ns], act, wb, act, mode2, True)
            for rep in range(3):      # the last run with QGTC_CHAIN_DISCARD (a hint: T' is the same either way)
                pair = qgtc.ChainedPair(sa, sx, rep == 2)
                for o in list(sa.outs) + list(sx.outs):
                    o.fill_(-1 if o.dtype == torch.int32 else 7.0)
                pair.run()
                torch.cuda.synchronize()
                for i, n in enumerate(ns):
                    if rep < 2:
                        np.testing.assert_array_equal(to_np_u32(sa.outs[i]), want[i][0], err_msg=f"aggregate of batch {i} (n = {n}), run {rep}")
                    if mode2 == 1:
                        np.testing.assert_array_equal(to_np_u32(pair.outs[i]), want[i][1], err_msg=f"T' of batch {i} (n = {n}), run {rep}")
                    else:
                        np.testing.assert_array_equal(pair.outs[i].cpu().numpy(), want[i][2], err_msg=f"float T' of batch {i} (n = {n}), run {rep}")


@pytest.mark.gpu
def test_chained_pair_rejects_plans_that_do_not_chain(qgtc, oracle):
    import torch
    from helpers import rand_q, to_dev
    from qgtc_ppopp22_amd.shapes import cols_shape, rows_shape
    rng = np.random.default_rng(8)
    n, f = 100, 64
    A = to_dev(torch, oracle.pack(rand_q(rng, n, n, 1), 1, False), rows_shape(n, n, 1))
    T = to_dev(torch, oracle.pack(rand_q(rng, n, f, 2), 2, True), cols_shape(n, f, 2))
    W = to_dev(torch, oracle.pack(rand_q(rng, f, f, 2), 2, True), cols_shape(f, f, 2))
    sa = qgtc.BatchedGemm([A], [T], [(n, n, f)], 1, 2, 2, 0, True)
    other = qgtc.BatchedGemm([A], [T], [(n, n, f)], 1, 2, 2, 0, True)
    with pytest.raises(RuntimeError):      # the second stage does not read the first one's output
        qgtc.ChainedPair(sa, qgtc.BatchedGemm(other.outs, [W], [(n, f, f)], 2, 2, 2, 1, True))
    with pytest.raises(RuntimeError):      # rows-layout output where the cols layout is needed
        qgtc.ChainedPair(sa, qgtc.BatchedGemm(sa.outs, [W], [(n, f, f)], 2, 2, 2, 0, True))
    with pytest.raises(RuntimeError):      # plane counts that do not chain
        qgtc.ChainedPair(sa, qgtc.BatchedGemm(sa.outs, [W], [(n, f, f)], 1, 2, 2, 1, True))


@pytest.mark.parametrize("nbits", [1, 2, 4, 8, 11])
def test_val2bit_many_equals_val2bit(qgtc, oracle, nbits):
    """One launch for several matrices (the weights an epoch packs inside its clock): every output word for word the
    single call's and the oracle's - cols layout with and without output_layer, rows layout, ragged shapes, edge values."""
    import torch
    rng = np.random.default_rng(nbits)
    shapes = [(128, 128), (50, 64), (64, 10), (64, 10), (9, 300), (129, 1)]
    col = [True, True, True, True, False, False]
    outl = [False, False, True, False, False, False]
    xs = [edge_floats(rng, h, w, nbits) for h, w in shapes]
    got = qgtc.val2bit_many([torch.from_numpy(x).cuda() for x in xs], nbits, col, outl)
    for x, c, o, g in zip(xs, col, outl, got):
        one = qgtc.val2bit(torch.from_numpy(x).cuda(), nbits, c, o)
        assert one.shape == g.shape and torch.equal(one, g)
        np.testing.assert_array_equal(to_np_u32(g), oracle.val2bit(x, nbits, c, o))
    with pytest.raises(RuntimeError):
        qgtc.val2bit_many([torch.zeros((4, 4), device="cuda")] * 9, nbits, [True] * 9, [False] * 9)


def test_layer_plan_survives_engine_changes_between_runs(qgtc, oracle):
    """One FusedLayer plan run under a sequence of engine settings: the route is re-decided on every call and every route
    gives the same float32 outputs."""
    import torch
    rng = np.random.default_rng(77)
    act, wb, f_in, f_out = 2, 2, 64, 64
    ns = [150, 333, 40]
    Wt = oracle.pack(rand_q(rng, f_in, f_out, wb), wb, True)
    dW = to_dev(torch, Wt, (-1,))
    As, Xs, want = [], [], []
    for n in ns:
        A, X = oracle.pack((rng.random((n, n)) < 0.03).astype(np.int32), 1, False), oracle.pack(rand_q(rng, n, f_in, act), act, False)
        As.append(to_dev(torch, A, (-1,)))
        Xs.append(to_dev(torch, X, (-1,)))
        T = oracle.bitmm2bit(X, Wt, n, f_in, f_out, act, wb, act, col=True)
        want.append(oracle.bitmm2int(A, T, n, n, f_out, 1, act, True))
    g1 = qgtc.BatchedGemm(Xs, [dW], [(n, f_in, f_out) for n in ns], act, wb, act, 1, False, False)
    g2 = qgtc.BatchedGemm(As, g1.outs, [(n, n, f_out) for n in ns], 1, act, 1, 2, True, False)
    layer = qgtc.FusedLayer(g1, g2)
    for eng in ("auto", "popcount", "auto", "mfma", "popcount", "auto"):
        qgtc.set_engine(eng)
        for o in g2.outs:
            o.fill_(-5.0)
        layer.run()
        torch.cuda.synchronize()
        for o, w in zip(g2.outs, want):
            np.testing.assert_array_equal(o.cpu().numpy(), w, err_msg=eng)
    qgtc.set_engine("auto")


def test_lean_and_checked_entry_points_agree(qgtc, oracle):
    """The four per-batch operators are METH_FASTCALL functions (qgtc_torch.cpp, namespace lean) that skip pybind11's casters,
    the dispatcher and the device guard for the plain good case; the pybind11 functions stay reachable as QGTC.checked_*. Same
    tensors (dtype, shape, words) from both on every engine, the reference's call forms included (7-argument bitMM2Int, defaulted
    trailing bools, keywords - which the lean entries hand to the checked ones), and the same exceptions for bad calls."""
    import torch
    rng = np.random.default_rng(12)
    for engine in ("auto", "popcount", "mfma"):
        with use_engine(qgtc, engine):
            for (M, K, N, a, w, ob) in ((300, 300, 64, 1, 2, 2), (37, 130, 10, 2, 2, 2), (1213, 128, 128, 2, 2, 2), (64, 4100, 33, 1, 1, 3), (9, 9, 9, 3, 5, 4)):
                x = torch.from_numpy(rng.uniform(-1, 2 ** a + 1, size=(M, K)).astype(np.float32)).cuda()
                y = torch.from_numpy(rng.uniform(-1, 2 ** w + 1, size=(K, N)).astype(np.float32)).cuda()
                for args in ((a, False, False), (a, True, False), (a, True, True), (a,)):
                    p, q = qgtc.val2bit(x, *args), qgtc.checked_val2bit(x, *args)
                    assert p.dtype == q.dtype and p.shape == q.shape and torch.equal(p, q)
                bx, bw = qgtc.val2bit(x, a, False, False), qgtc.val2bit(y, w, True, False)
                np.testing.assert_array_equal(to_np_u32(bx), oracle.val2bit(x.cpu().numpy(), a, False, False))
                for lean, checked in ((qgtc.bitMM2Bit, qgtc.checked_bitMM2Bit), (qgtc.bitMM2Bit_col, qgtc.checked_bitMM2Bit_col)):
                    p, q = lean(bx, bw, M, K, N, a, w, ob), checked(bx, bw, M, K, N, a, w, ob)
                    assert p.dtype == torch.int32 and p.shape == q.shape and torch.equal(p, q)
                for tail in ((), (False,), (True,)):
                    p, q = qgtc.bitMM2Int(bx, bw, M, K, N, a, w, *tail), qgtc.checked_bitMM2Int(bx, bw, M, K, N, a, w, *tail)
                    assert p.dtype == torch.float32 and tuple(p.shape) == (M, N) and torch.equal(p, q)
                assert torch.equal(qgtc.bitMM2Int(bx, bw, M, K, N, a, w, pad_128=True), qgtc.checked_bitMM2Int(bx, bw, M, K, N, a, w, True))
                assert torch.equal(qgtc.val2bit(x, nbits=a, col_major=True), qgtc.checked_val2bit(x, a, True, False))
    # bad calls raise the pybind11 functions' exceptions
    bx = qgtc.val2bit(torch.ones((8, 8)).cuda(), 1, False, False)
    with pytest.raises(RuntimeError, match="CUDA tensor"):
        qgtc.bitMM2Bit(bx.cpu(), bx, 8, 8, 8, 1, 1, 1)
    with pytest.raises(RuntimeError, match="int32"):
        qgtc.bitMM2Bit(bx.float(), bx, 8, 8, 8, 1, 1, 1)
    with pytest.raises(RuntimeError, match="contiguous"):
        qgtc.bitMM2Bit(bx.t(), bx, 8, 8, 8, 1, 1, 1)
    with pytest.raises(RuntimeError, match="bad dimensions"):
        qgtc.bitMM2Bit(bx, bx, 0, 8, 8, 1, 1, 1)
    with pytest.raises(RuntimeError):
        qgtc.bitMM2Bit(bx, bx, 8, 8, 8, 1, 1, 40)          # output_bit beyond 32
    with pytest.raises(TypeError):
        qgtc.bitMM2Bit(bx, bx, 8, 8, 8, 1, 1)              # the reference's binding has no default here either
    with pytest.raises(TypeError):
        qgtc.bitMM2Bit(bx, bx, 8.5, 8, 8, 1, 1, 1)
    with pytest.raises(RuntimeError, match="float32"):
        qgtc.val2bit(bx, 1, False, False)
