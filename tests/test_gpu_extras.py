"""GPU parity of the paths that sit beside the bit-GEMM: adjacency packing from an edge list
(sampler.py:80-101 without the dense detour) and the int8 MFMA comparison GEMM
(cuBLASGemmEX/cublas_main.cu:123-172 analogue)."""
import numpy as np
import pytest

from helpers import to_np_u32

pytestmark = pytest.mark.gpu


def _dense_from_edges(row, col, H, W):
    A = np.zeros((H, W), dtype=np.float32)
    np.add.at(A, (row, col), 1.0)      # duplicates sum, like to_dense() in sampler.py:87-89
    return A


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
    A = _dense_from_edges(row, col, H, W)
    want = oracle.val2bit(A, nbits, False, False)
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
        qgtc.pack_edges(torch.tensor([0, 1], dtype=torch.int32).cuda(), torch.tensor([1, 2], dtype=torch.int32).cuda(), 37, 37, 1)


@pytest.mark.parametrize("M,K,N", [(16, 64, 16), (17, 128, 70), (100, 1024, 64), (1024, 1024, 16),
                                   (333, 4096, 33), (4096, 4096, 64), (64, 16, 200), (50, 48, 10)])
def test_i8gemm_is_exact(qgtc, M, K, N):
    import torch
    rng = np.random.default_rng(M + K + N)
    A = rng.integers(-128, 128, size=(M, K), dtype=np.int64).astype(np.int8)
    Bt = rng.integers(-128, 128, size=(N, K), dtype=np.int64).astype(np.int8)
    # asymmetric, exact integer reference (int64 accumulate, then the int32 -> float32 conversion)
    want = (A.astype(np.int64) @ Bt.astype(np.int64).T).astype(np.int32).astype(np.float32)
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
