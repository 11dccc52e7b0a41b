"""CPU checks of the oracle restatements that sit beside the bit-GEMM (edge-list packing, tile
occupancy, the INT8 comparison GEMM): closed-form known answers and agreement with the C oracle."""
import numpy as np
import pytest

from oracle.qgtc_oracle import (P8, S128, np_dense_adjacency, np_i8gemm, np_pack_edges, np_tile_occupancy)


def test_pack_edges_known_answers(oracle):
    # multiplicities 1, 2, 3 at b = 1: Quantize_val maps 1 -> 1, 2 (== 2^b) -> 2 -> plane 0 clear,
    # 3 (> 2^b) -> 2^b - 1 = 1 (kernel.h:39-44): the doubled edge disappears
    row = [0, 1, 1, 2, 2, 2]
    col = [5, 33, 33, 64, 64, 64]
    words = np_pack_edges(row, col, 3, 130, 1).reshape(P8(3), S128(130) * 4)
    assert words[0, 0] == 1 << (31 - 5)
    assert words[1, 1] == 0
    assert words[2, 2] == 1 << 31
    assert int(np.count_nonzero(words)) == 2
    # and the same through the C oracle's val2bit of the dense matrix
    np.testing.assert_array_equal(oracle.val2bit(np_dense_adjacency(row, col, 3, 130), 1, False, False), words.reshape(-1))


@pytest.mark.parametrize("H,W,b,edges", [(37, 37, 1, 300), (64, 200, 2, 2500), (9, 300, 3, 700)])
def test_pack_edges_matches_c_oracle(oracle, H, W, b, edges):
    rng = np.random.default_rng(H + W + b)
    row, col = rng.integers(0, H, size=edges), rng.integers(0, W, size=edges)
    row, col = np.concatenate([row, row[:50], row[:20], row[:20]]), np.concatenate([col, col[:50], col[:20], col[:20]])
    np.testing.assert_array_equal(np_pack_edges(row, col, H, W, b),
                                  oracle.val2bit(np_dense_adjacency(row, col, H, W), b, False, False))


def test_tile_occupancy_known_answers(oracle):
    M, K = 70, 9000                                  # 3 row tiles, 71 k-quads -> 2 words per tile
    q = np.zeros((M, K), dtype=np.int32)
    q[0, 0] = 1                                      # tile 0, k-quad 0
    q[33, 128 * 64 + 5] = 1                          # tile 1, k-quad 64 -> word 1 bit 0
    q[69, 8999] = 1                                  # tile 2, k-quad 70 -> word 1 bit 6
    occ = np_tile_occupancy(oracle.pack(q, 1, False), M, K, 1).reshape(3, 2)
    assert occ.tolist() == [[1, 0], [0, 1], [0, 1 << 6]]
    # a second plane contributes too
    q2 = np.zeros((M, K), dtype=np.int32)
    q2[40, 300] = 2                                  # plane 1 only: tile 1, k-quad 2
    occ2 = np_tile_occupancy(oracle.pack(q2, 2, False), M, K, 2).reshape(3, 2)
    assert occ2.tolist() == [[0, 0], [1 << 2, 0], [0, 0]]


def test_tile_occupancy_consistent_with_tile_counters(oracle):
    """The reference counts non-zero 8 x 128-bit tiles (kernel.h:574-592 as intended); a 32-row tile
    is occupied exactly when one of its four 8-row tiles is."""
    rng = np.random.default_rng(5)
    M, K = 96, 1024
    q = (rng.random((M, K)) < 0.0008).astype(np.int32)
    X = oracle.pack(q, 1, False)
    occ = np_tile_occupancy(X, M, K, 1)
    fine = np_tile_occupancy(X, M, K, 1, tile_rows=8).reshape(-1, 1)
    coarse = np.array([np.bitwise_or.reduce(fine[4 * t:4 * t + 4, 0]) for t in range(M // 32)], dtype=np.uint64)
    np.testing.assert_array_equal(occ, coarse)
    total, nonzero = oracle.tile_counters(X, M, K, 8, 1, 1)   # N = 8: one 8-column tile, one W plane
    assert nonzero == sum(bin(int(v)).count("1") for v in fine[:, 0])


def test_i8gemm_known_answers():
    A = np.eye(16, 32, dtype=np.int8) * 3
    Bt = (np.arange(5)[:, None] - np.arange(32)[None, :]).astype(np.int8)       # Bt[n][k] = n - k
    C = np_i8gemm(A, Bt)
    assert C.dtype == np.float32 and C.shape == (16, 5)
    np.testing.assert_array_equal(C, 3.0 * (np.arange(5)[None, :] - np.arange(16)[:, None]))
    # extreme values: -128 * -128 * K stays exact
    C2 = np_i8gemm(np.full((2, 4096), -128, np.int8), np.full((3, 4096), -128, np.int8))
    assert (C2 == 128.0 * 128.0 * 4096).all()
