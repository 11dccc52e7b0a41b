"""GPU parity in the throughput-bound half of the reference's adjacency-size study (5_9_adjmatrix_size.py:15-18: M = K = 2^7 .. 2^15,
N = 16 .. 1024, 1 bit; QGTC_module/logs/profile_new.log:26 records m = k = 32768, n = 1024): the one regime where the packed adjacency
(128 MiB at 32768^2) does not fit any cache. Operands are packed on the device (QGTC.val2bit), the packed words pulled to the host and
multiplied by the C oracle; every output form, the default engine (k_bitmm_fp4_stream up to 256 columns, k_bitmm_fp4_wide beyond) and
the AND + popcount kernels.

Where the full oracle product is too slow for a test (N = 1024: 1.7e10 word pairs) 128 aligned blocks of 32 rows (4096 rows) are checked
word for word - whole words of both packed layouts - and ALL rows through the int64 row checksum C 1 = A (X 1) of the float32 output."""
import numpy as np
import pytest

from helpers import kernel_behind, to_np_u32, use_engine
from qgtc_ppopp22_amd.shapes import P8, P128, S128

pytestmark = pytest.mark.gpu

ENGINES2 = ("auto", "popcount")


def _pack_on_device(qgtc, torch, M, K, N, kind, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    if kind == "ones":        # 5_9_adjmatrix_size.py's inputs
        A, X = torch.ones(M, K, device="cuda"), torch.ones(K, N, device="cuda")
    elif kind == "sparse":    # a few edges a row: sums on both sides of the clamp
        A = (torch.rand(M, K, device="cuda", generator=g) < 3.0 / K).float()
        X = (torch.rand(K, N, device="cuda", generator=g) < 0.5).float()
    else:
        A = (torch.rand(M, K, device="cuda", generator=g) < 0.5).float()
        X = (torch.rand(K, N, device="cuda", generator=g) < 0.5).float()
    bA, bX = qgtc.val2bit(A, 1, False, False), qgtc.val2bit(X, 1, True, False)
    rowsum_x = X.double().sum(dim=1)
    want_rowsum = (A.double() @ rowsum_x).cpu().numpy()      # C 1 = A (X 1), exact in float64 (< 2^53)
    del A, X
    return bA, bX, want_rowsum


def _rows_of(words, M, row_words, planes, rows):
    """The packed rows `rows` of a rows-layout tensor [planes][PAD8(M)][row_words] as [planes][len(rows)][row_words]."""
    return words.reshape(planes, P8(M), row_words)[:, rows, :]


@pytest.mark.parametrize("M,N,kind", [(32768, 16, "random"), (32768, 64, "random"), (32768, 64, "ones"), (32768, 64, "sparse"),
                                      (16384, 256, "random"), (16384, 256, "sparse"), (8192, 200, "random"), (16384, 33, "sparse")])
def test_big_square_products_match_the_oracle(qgtc, oracle, M, N, kind):
    """Full oracle product (seconds on the host's cores): M = K, every output form, both engines."""
    import torch
    K = M
    bA, bX, want_rowsum = _pack_on_device(qgtc, torch, M, K, N, kind, seed=M + N)
    hA, hX = to_np_u32(bA), to_np_u32(bX)
    want_f = oracle.bitmm2int(hA, hX, M, K, N, 1, 1, True)
    np.testing.assert_array_equal(want_f.astype(np.float64).sum(axis=1), want_rowsum)      # the oracle against plain float64 arithmetic
    obs = (1, 3) if kind == "sparse" else (1,)
    want_bits = {ob: (oracle.bitmm2bit(hA, hX, M, K, N, 1, 1, ob), oracle.bitmm2bit(hA, hX, M, K, N, 1, 1, ob, col=True)) for ob in obs}
    for eng in ENGINES2:
        with use_engine(qgtc, eng):
            np.testing.assert_array_equal(qgtc.bitMM2Int(bA, bX, M, K, N, 1, 1, True).cpu().numpy(), want_f, err_msg=f"float {eng}")
            for ob in obs:
                np.testing.assert_array_equal(to_np_u32(qgtc.bitMM2Bit(bA, bX, M, K, N, 1, 1, ob)), want_bits[ob][0], err_msg=f"rows ob={ob} {eng}")
                np.testing.assert_array_equal(to_np_u32(qgtc.bitMM2Bit_col(bA, bX, M, K, N, 1, 1, ob)), want_bits[ob][1], err_msg=f"cols ob={ob} {eng}")
    if kind == "ones":     # the script's closed form: every sum is K, re-quantised to 2^ob - 1 = 1
        back = qgtc.bit2val(qgtc.bitMM2Bit(bA, bX, M, K, N, 1, 1, 1), 1, M, N, False, False)
        assert bool((back == 1).all().item())


@pytest.mark.parametrize("kind", ["random", "sparse"])
def test_32768_squared_times_1024_sampled_rows_and_row_checksums(qgtc, oracle, kind):
    """profile_new.log:26's shape. 128 aligned 32-row blocks word for word in all three outputs, every row by its checksum."""
    import torch
    M = K = 32768
    N = 1024
    bA, bX, want_rowsum = _pack_on_device(qgtc, torch, M, K, N, kind, seed=7)
    hA, hX = to_np_u32(bA), to_np_u32(bX)
    rng = np.random.default_rng(11)
    blocks = np.sort(rng.choice(M // 32, size=128, replace=False))
    blocks[0], blocks[-1] = 0, M // 32 - 1
    rows = (blocks[:, None] * 32 + np.arange(32)[None, :]).reshape(-1)
    Ms = rows.size
    kw = S128(K) * 4
    subA = np.ascontiguousarray(hA.reshape(P8(M), kw)[rows]).reshape(-1)              # a rows-layout operand of the sampled rows (Ms % 8 == 0)
    want_f = oracle.bitmm2int(subA, hX, Ms, K, N, 1, 1, True)
    obs = (1, 3) if kind == "sparse" else (1,)
    rw = S128(N) * 4
    for eng in ENGINES2:
        with use_engine(qgtc, eng):
            got_f = qgtc.bitMM2Int(bA, bX, M, K, N, 1, 1, True)
            np.testing.assert_array_equal(got_f.double().sum(dim=1).cpu().numpy(), want_rowsum, err_msg=f"row checksums {eng}")
            np.testing.assert_array_equal(got_f[torch.from_numpy(rows).cuda()].cpu().numpy(), want_f, err_msg=f"float rows {eng}")
            for ob in obs:
                got_r = to_np_u32(qgtc.bitMM2Bit(bA, bX, M, K, N, 1, 1, ob))
                want_r = oracle.bitmm2bit(subA, hX, Ms, K, N, 1, 1, ob).reshape(ob, P8(Ms), rw)
                np.testing.assert_array_equal(_rows_of(got_r, M, rw, ob, rows), want_r[:, :Ms], err_msg=f"rows layout ob={ob} {eng}")
                # cols layout [ob][PAD128(N)][S128(M) * 4]: word m / 32 of a line = one sampled block
                got_c = to_np_u32(qgtc.bitMM2Bit_col(bA, bX, M, K, N, 1, 1, ob)).reshape(ob, P128(N), S128(M) * 4)
                want_c = oracle.bitmm2bit(subA, hX, Ms, K, N, 1, 1, ob, col=True).reshape(ob, P128(N), S128(Ms) * 4)
                np.testing.assert_array_equal(got_c[:, :, blocks], want_c[:, :, :blocks.size], err_msg=f"cols layout ob={ob} {eng}")
                # (outside the samples: every row's re-quantised values against the float output's, on the device)
                back = qgtc.bit2val(qgtc.bitMM2Bit(bA, bX, M, K, N, 1, 1, ob), ob, M, N, False, False)
                lim = float(2 ** ob)
                want_back = torch.where(got_f > lim, torch.full_like(got_f, lim - 1.0), got_f).to(torch.int32) & (2 ** ob - 1)
                assert torch.equal(back, want_back), f"decoded bits vs float output, ob={ob} {eng}"


def test_long_k_kernel_on_ragged_shapes(qgtc, oracle):
    """k_bitmm_fp4_stream's edges against the oracle (engine "mfma": the kernel wherever it applies; "auto" keeps small M on
    k_bitmm_fp4_skinny): K one bit past / short of its 128-byte blocks and 256- / 512-byte groups, rows and columns off every tile size,
    one to four column tiles, float output with N % 4 != 0."""
    import torch
    rng = np.random.default_rng(3)
    qgtc.set_engine("mfma")
    for (M, K, N) in ((70, 4097, 1), (129, 6143, 17), (200, 8191, 64), (64, 4224, 65), (333, 5000, 130), (1000, 12289, 200), (65, 16385, 256),
                      (31, 20000, 33), (16400, 8200, 48)):
        assert kernel_behind(M, K, N, engine="mfma") == "k_bitmm_fp4_stream"
        for density in (0.5, 0.002, 0.0):
            qx = (rng.random((M, K)) < density).astype(np.int32)
            qw = (rng.random((K, N)) < 0.5).astype(np.int32)
            X, Wt = oracle.pack(qx, 1, False), oracle.pack(qw, 1, True)
            dX = torch.from_numpy(X.view(np.int32).reshape(P8(M), S128(K) * 4)).cuda()
            dW = torch.from_numpy(Wt.view(np.int32).reshape(S128(K) * 4, P128(N))).cuda()
            np.testing.assert_array_equal(qgtc.bitMM2Int(dX, dW, M, K, N, 1, 1, True).cpu().numpy(), oracle.bitmm2int(X, Wt, M, K, N, 1, 1, True))
            for ob in (1, 2, 13, 24):
                np.testing.assert_array_equal(to_np_u32(qgtc.bitMM2Bit(dX, dW, M, K, N, 1, 1, ob)), oracle.bitmm2bit(X, Wt, M, K, N, 1, 1, ob))
                np.testing.assert_array_equal(to_np_u32(qgtc.bitMM2Bit_col(dX, dW, M, K, N, 1, 1, ob)),
                                              oracle.bitmm2bit(X, Wt, M, K, N, 1, 1, ob, col=True))
        # PAD8 weights (bitMM2Int pad_128 = False): fewer lines in W than a 32-line fragment reads
        Wt8 = oracle.pack(qw, 1, True, True)
        dW8 = torch.from_numpy(Wt8.view(np.int32).reshape(S128(K) * 4, P8(N))).cuda()
        np.testing.assert_array_equal(qgtc.bitMM2Int(dX, dW8, M, K, N, 1, 1, False).cpu().numpy(), oracle.bitmm2int(X, Wt8, M, K, N, 1, 1, False))
    qgtc.set_engine("auto")


def test_long_k_kernel_random_shape_sweep(qgtc, oracle):
    """Forty seeded random shapes (M 1 .. 3000, K 4097 .. 20000, N 1 .. 256) on k_bitmm_fp4_stream (engine "mfma") against the oracle: the rows-
    layout bits at a random ob, the float output, and the 128-row tiles (QGTC_STREAM_RF=4: what big launches run) on the shapes with more than
    32 columns. Left operands: random density, with whole all-zero row blocks and an all-zero span of K (the zero-step skip on real boundaries)."""
    import os
    import torch
    rng = np.random.default_rng(2022)
    qgtc.set_engine("mfma")
    try:
        for case in range(40):
            M, K, N = int(rng.integers(1, 3001)), int(rng.integers(4097, 20001)), int(rng.integers(1, 257))
            ob = int(rng.integers(1, 25))
            assert kernel_behind(M, K, N, engine="mfma") == "k_bitmm_fp4_stream"
            qx = (rng.random((M, K)) < rng.choice([0.5, 0.05, 0.001])).astype(np.int32)
            if M > 200:
                qx[100:100 + int(rng.integers(1, M - 100))] = 0           # whole tiles of zero rows
            k0 = int(rng.integers(0, K - 1024))
            qx[:, k0:k0 + int(rng.integers(256, 1024))] = 0              # a span of K that is zero for every row
            qw = (rng.random((K, N)) < 0.5).astype(np.int32)
            X, Wt = oracle.pack(qx, 1, False), oracle.pack(qw, 1, True)
            dX = torch.from_numpy(X.view(np.int32).reshape(P8(M), S128(K) * 4)).cuda()
            dW = torch.from_numpy(Wt.view(np.int32).reshape(S128(K) * 4, P128(N))).cuda()
            want_f, want_b = oracle.bitmm2int(X, Wt, M, K, N, 1, 1, True), oracle.bitmm2bit(X, Wt, M, K, N, 1, 1, ob)
            for rf in (None, "4") if N > 32 else (None,):
                if rf:
                    os.environ["QGTC_STREAM_RF"] = rf
                try:
                    np.testing.assert_array_equal(qgtc.bitMM2Int(dX, dW, M, K, N, 1, 1, True).cpu().numpy(), want_f, err_msg=f"case {case} {M}x{K}x{N} rf={rf}")
                    np.testing.assert_array_equal(to_np_u32(qgtc.bitMM2Bit(dX, dW, M, K, N, 1, 1, ob)), want_b, err_msg=f"case {case} {M}x{K}x{N} ob={ob} rf={rf}")
                finally:
                    os.environ.pop("QGTC_STREAM_RF", None)
    finally:
        qgtc.set_engine("auto")
