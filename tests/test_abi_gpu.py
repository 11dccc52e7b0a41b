"""The C-ABI of libqgtc_hip.so driven WITHOUT the PyTorch extension: ctypes + raw device pointers, the way
INTEGRATION.md section 3 shows a non-torch host (cgo / JNI / plain C) would call it. torch is used only as the
device allocator here; every compute call goes through `extern "C"` entry points with plain pointers and sizes, and
the results are compared with the oracle."""
import ctypes

import numpy as np
import pytest

from helpers import rand_q

pytestmark = pytest.mark.gpu

u32p, f32p, vp = ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_float), ctypes.c_void_p


@pytest.fixture(scope="module")
def lib():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    import qgtc_ppopp22_amd

    L = ctypes.CDLL(qgtc_ppopp22_amd.lib_path())
    L.qgtc_rows_words.restype = ctypes.c_size_t
    L.qgtc_rows_words.argtypes = [ctypes.c_int] * 3
    L.qgtc_cols_words.restype = ctypes.c_size_t
    L.qgtc_cols_words.argtypes = [ctypes.c_int] * 4
    L.qgtc_val2bit.argtypes = [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, ctypes.c_size_t, vp]
    L.qgtc_bitmm2bit.argtypes = [vp, ctypes.c_size_t, vp, ctypes.c_size_t] + [ctypes.c_int] * 6 + [vp, ctypes.c_size_t, ctypes.c_uint, vp]
    L.qgtc_bitmm2int.argtypes = [vp, ctypes.c_size_t, vp, ctypes.c_size_t] + [ctypes.c_int] * 6 + [vp, ctypes.c_size_t, ctypes.c_uint, vp]
    L.qgtc_bit2val.argtypes = [vp, ctypes.c_size_t] + [ctypes.c_int] * 5 + [vp, vp]
    L.qgtc_strerror.restype = ctypes.c_char_p
    return L


def _dev(torch, n, dtype):
    return torch.empty(int(n), dtype=dtype, device="cuda")


@pytest.mark.parametrize("M,K,N,a,w,ob", [(300, 300, 64, 1, 2, 2), (129, 1000, 40, 2, 2, 3), (64, 4096, 64, 1, 1, 1),
                                          (77, 500, 130, 3, 5, 4)])
@pytest.mark.parametrize("flags", [0x0, 0x10, 0x8, 0x2], ids=["popcount", "auto", "mfma", "no-zero-skip"])
def test_c_abi_with_raw_device_pointers(lib, oracle, M, K, N, a, w, ob, flags):
    import torch
    rng = np.random.default_rng(M + K + N + flags)
    qx, qw = rand_q(rng, M, K, a).astype(np.float32), rand_q(rng, K, N, w).astype(np.float32)
    dx, dw = torch.from_numpy(qx).cuda(), torch.from_numpy(qw).cuda()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    # val2bit: float -> packed planes, straight into caller-owned buffers
    xw_, ww_ = lib.qgtc_rows_words(M, K, a), lib.qgtc_cols_words(K, N, w, 0)
    bx, bw = _dev(torch, xw_, torch.int32), _dev(torch, ww_, torch.int32)
    assert lib.qgtc_val2bit(dx.data_ptr(), M, K, a, 0, 0, bx.data_ptr(), xw_, st) == 0
    assert lib.qgtc_val2bit(dw.data_ptr(), K, N, w, 1, 0, bw.data_ptr(), ww_, st) == 0
    X_o, W_o = oracle.val2bit(qx, a, False, False), oracle.val2bit(qw, w, True, False)
    np.testing.assert_array_equal(bx.cpu().numpy().view(np.uint32), X_o)
    np.testing.assert_array_equal(bw.cpu().numpy().view(np.uint32), W_o)
    # bitMM2Bit (rows layout), bitMM2Bit_col (QGTC_OUT_COLS = 0x1), bitMM2Int
    ow = lib.qgtc_rows_words(M, N, ob)
    out = _dev(torch, ow, torch.int32)
    rc = lib.qgtc_bitmm2bit(bx.data_ptr(), xw_, bw.data_ptr(), ww_, M, K, N, a, w, ob, out.data_ptr(), ow, flags, st)
    assert rc == 0, lib.qgtc_strerror(rc)
    np.testing.assert_array_equal(out.cpu().numpy().view(np.uint32), oracle.bitmm2bit(X_o, W_o, M, K, N, a, w, ob))
    oc = lib.qgtc_cols_words(M, N, ob, 0)
    outc = _dev(torch, oc, torch.int32)
    assert lib.qgtc_bitmm2bit(bx.data_ptr(), xw_, bw.data_ptr(), ww_, M, K, N, a, w, ob, outc.data_ptr(), oc, flags | 0x1, st) == 0
    np.testing.assert_array_equal(outc.cpu().numpy().view(np.uint32), oracle.bitmm2bit(X_o, W_o, M, K, N, a, w, ob, col=True))
    outf = _dev(torch, M * N, torch.float32)
    assert lib.qgtc_bitmm2int(bx.data_ptr(), xw_, bw.data_ptr(), ww_, M, K, N, a, w, 1, outf.data_ptr(), M * N, flags, st) == 0
    np.testing.assert_array_equal(outf.cpu().numpy().reshape(M, N), oracle.bitmm2int(X_o, W_o, M, K, N, a, w, True))
    # bit2val of the packed result closes the loop
    dec = _dev(torch, M * N, torch.int32)
    assert lib.qgtc_bit2val(out.data_ptr(), ow, ob, M, N, 0, 0, dec.data_ptr(), st) == 0
    np.testing.assert_array_equal(dec.cpu().numpy().reshape(M, N), oracle.bit2val(oracle.bitmm2bit(X_o, W_o, M, K, N, a, w, ob), ob, M, N, False, False))
    # an undersized output buffer is an error code, not a fault
    assert lib.qgtc_bitmm2bit(bx.data_ptr(), xw_, bw.data_ptr(), ww_, M, K, N, a, w, ob, out.data_ptr(), ow - 1, flags, st) == 2


class QgtcProblem(ctypes.Structure):
    """include/qgtc.h: struct qgtc_problem (the layout a cgo / JNI / plain-C host would declare)."""
    _fields_ = [("X", ctypes.c_void_p), ("W", ctypes.c_void_p), ("out", ctypes.c_void_p), ("x_words", ctypes.c_uint64),
                ("w_words", ctypes.c_uint64), ("M", ctypes.c_int32), ("K", ctypes.c_int32), ("N", ctypes.c_int32),
                ("w_lines", ctypes.c_int32), ("occ_words", ctypes.c_int32), ("occ", ctypes.c_void_p)]


@pytest.mark.parametrize("flags", [0x0, 0x10, 0x10 | 0x20], ids=["popcount", "auto", "auto-one-launch"])
def test_layer_entry_with_raw_descriptors(lib, oracle, flags):
    """qgtc_gcn_layer_batched through ctypes: descriptors written by the HOST into a device buffer (struct layout of
    include/qgtc.h), raw device pointers, arrival counters with QGTC_ARRIVAL_STRIDE - against the oracle's two products."""
    import torch
    assert ctypes.sizeof(QgtcProblem) == 72   # three pointers, two u64, five i32 (+4 padding), one pointer
    lib.qgtc_gcn_layer_batched.argtypes = [vp, vp, ctypes.c_int] + [ctypes.c_int] * 4 + [ctypes.c_int] * 6 + [vp, ctypes.c_uint32, ctypes.c_uint, vp]
    rng = np.random.default_rng(17 + flags)
    act, wb, f_in, f_out = 2, 2, 64, 96
    batches = [(150, f_in), (333, f_in), (40, f_in)]
    qw = rand_q(rng, f_in, f_out, wb)
    Wt = oracle.pack(qw, wb, True)
    dW = torch.from_numpy(Wt.view(np.int32)).cuda()
    keep, s1, s2, want = [dW], [], [], []
    P128 = lambda x: (x + 127) // 128 * 128   # noqa: E731
    for (n, f) in batches:
        qa = (rng.random((n, n)) < 0.02).astype(np.int32)
        qx = rand_q(rng, n, f, act)
        A, X = oracle.pack(qa, 1, False), oracle.pack(qx, act, False)
        dA, dX = torch.from_numpy(A.view(np.int32)).cuda(), torch.from_numpy(X.view(np.int32)).cuda()
        T = torch.empty(int(lib.qgtc_cols_words(n, f_out, act, 0)), dtype=torch.int32, device="cuda")
        out = torch.empty(n * f_out, dtype=torch.float32, device="cuda")
        keep += [dA, dX, T, out]
        s1.append(QgtcProblem(dX.data_ptr(), dW.data_ptr(), T.data_ptr(), dX.numel(), dW.numel(), n, f, f_out, P128(f_out), 0, None))
        s2.append(QgtcProblem(dA.data_ptr(), T.data_ptr(), out.data_ptr(), dA.numel(), T.numel(), n, n, f_out, P128(f_out), 0, None))
        T_o = oracle.bitmm2bit(X, Wt, n, f, f_out, act, wb, act, col=True)
        want.append((T, T_o, out, oracle.bitmm2int(A, T_o, n, n, f_out, 1, act, True)))
    count = len(batches)
    host = (QgtcProblem * (2 * count))(*(s1 + s2))
    descs = torch.frombuffer(bytearray(bytes(host)), dtype=torch.uint8).cuda()
    arrival = torch.zeros(count * 64, dtype=torch.int32, device="cuda")       # QGTC_ARRIVAL_STRIDE = 64
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for epoch in (1, 2):
        rc = lib.qgtc_gcn_layer_batched(descs.data_ptr(), descs.data_ptr() + 72 * count, count, max(n for n, _ in batches), f_in,
                                        max(n for n, _ in batches), f_out, act, wb, act, 1, 1, 2, arrival.data_ptr(), epoch, flags, st)
        assert rc == 0, lib.qgtc_strerror(rc)
        torch.cuda.synchronize()
        for (T, T_o, out, out_o) in want:
            np.testing.assert_array_equal(T.cpu().numpy().view(np.uint32), T_o)
            np.testing.assert_array_equal(out.cpu().numpy().reshape(out_o.shape), out_o)
            out.fill_(-3.0)
    # bad arguments are error codes
    assert lib.qgtc_gcn_layer_batched(None, descs.data_ptr(), count, 10, 10, 10, 10, 2, 2, 2, 1, 1, 2, arrival.data_ptr(), 1, flags, st) == 1
    assert lib.qgtc_gcn_layer_batched(descs.data_ptr(), descs.data_ptr(), count, 10, 10, 10, 10, 2, 2, 2, 1, 1, 1, arrival.data_ptr(), 1, flags, st) == 1


@pytest.mark.gpu
@pytest.mark.parametrize("flags", [0x10, 0x10 | 0x4, 0])
def test_chain_entry_with_raw_descriptors(lib, oracle, flags):
    """qgtc_gcn_chain_batched through ctypes (host-written descriptors, raw device pointers): the aggregation stage and the
    next layer's X.W stage against the oracle's two products; one launch on the matrix cores (0x10 = QGTC_ENGINE_AUTO, with
    and without QGTC_ZERO_JUMP - the descriptors carry no bitmaps, so every k-quad is visited) and as the two grouped
    launches of the popcount engine (flags 0)."""
    import torch
    lib.qgtc_gcn_chain_batched.argtypes = [vp, vp, ctypes.c_int] + [ctypes.c_int] * 4 + [ctypes.c_int] * 6 + [ctypes.c_uint, vp]
    rng = np.random.default_rng(23 + flags)
    act, wb, f1, f2 = 2, 2, 128, 96
    ns = [150, 333, 40]
    W2 = oracle.pack(rand_q(rng, f1, f2, wb), wb, True)
    dW2 = torch.from_numpy(W2.view(np.int32)).cuda()
    keep, sa, sx, want = [dW2], [], [], []
    P128 = lambda x: (x + 127) // 128 * 128   # noqa: E731
    for n in ns:
        qa = (rng.random((n, n)) < 0.03).astype(np.int32)
        A, T = oracle.pack(qa, 1, False), oracle.pack(rand_q(rng, n, f1, act), act, True)
        dA, dT = torch.from_numpy(A.view(np.int32)).cuda(), torch.from_numpy(T.view(np.int32)).cuda()
        out = torch.full((int(lib.qgtc_rows_words(n, f1, act)),), -1, dtype=torch.int32, device="cuda")
        T2 = torch.full((int(lib.qgtc_cols_words(n, f2, act, 0)),), -1, dtype=torch.int32, device="cuda")
        keep += [dA, dT, out, T2]
        sa.append(QgtcProblem(dA.data_ptr(), dT.data_ptr(), out.data_ptr(), dA.numel(), dT.numel(), n, n, f1, P128(f1), 0, None))
        sx.append(QgtcProblem(out.data_ptr(), dW2.data_ptr(), T2.data_ptr(), out.numel(), dW2.numel(), n, f1, f2, P128(f2), 0, None))
        out_o = oracle.bitmm2bit(A, T, n, n, f1, 1, act, act)
        want.append((out, out_o, T2, oracle.bitmm2bit(out_o, W2, n, f1, f2, act, wb, act, col=True)))
    count = len(ns)
    host = (QgtcProblem * (2 * count))(*(sa + sx))
    descs = torch.frombuffer(bytearray(bytes(host)), dtype=torch.uint8).cuda()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    rc = lib.qgtc_gcn_chain_batched(descs.data_ptr(), descs.data_ptr() + 72 * count, count, max(ns), max(ns), f1, f2, 1, act, act, wb, act, 1, flags, st)
    assert rc == 0, lib.qgtc_strerror(rc)
    torch.cuda.synchronize()
    for (out, out_o, T2, T2_o) in want:
        np.testing.assert_array_equal(out.cpu().numpy().view(np.uint32).reshape(out_o.shape), out_o)
        np.testing.assert_array_equal(T2.cpu().numpy().view(np.uint32).reshape(T2_o.shape), T2_o)
    assert lib.qgtc_gcn_chain_batched(None, descs.data_ptr(), count, 10, 10, 10, 10, 1, 2, 2, 2, 2, 1, flags, st) == 1
    assert lib.qgtc_gcn_chain_batched(descs.data_ptr(), descs.data_ptr(), count, 10, 10, 10, 10, 1, 2, 2, 2, 0, 1, flags, st) == 1
    assert lib.qgtc_gcn_chain_batched(descs.data_ptr(), descs.data_ptr(), count, 10, 10, 10, 10, 1, 2, 2, 2, 2, 0, flags, st) == 1
    # T' in the chain's own format (QGTC_CHAIN_CODES_OUT 0x100 / _IN 0x80) exists in the one-launch kernels only: a call that could
    # not keep the format is refused, not run with the flags dropped (its neighbour would misread the buffer)
    assert lib.qgtc_gcn_chain_batched(descs.data_ptr(), descs.data_ptr() + 72 * count, count, max(ns), max(ns), f1, f2, 1, 2, 2, 2, 2, 2, flags | 0x100, st) == 1   # (float32 out)
    assert lib.qgtc_gcn_chain_batched(descs.data_ptr(), descs.data_ptr() + 72 * count, count, max(ns), max(ns), f1, f2, 1, 1, 2, 2, 2, 1, flags | 0x80, st) == 1    # (one-plane T)
    lib.qgtc_bitmm_batched.argtypes = [vp, ctypes.c_int] + [ctypes.c_int] * 3 + [ctypes.c_int] * 4 + [ctypes.c_uint, vp]
    assert lib.qgtc_bitmm_batched(descs.data_ptr(), count, max(ns), max(ns), 512, 1, 2, 2, 0, flags | 0x80, st) == 1    # (512 columns: not the row-block kernel)
    assert lib.qgtc_bitmm_batched(descs.data_ptr(), count, max(ns), 256, f2, 2, 2, 2, 1, flags | 0x100, st) == 1        # (K = 256: not the X.W row-block kernel)
    assert lib.qgtc_gcn_chain_batched(descs.data_ptr(), descs.data_ptr() + 72 * count, count, max(ns), max(ns), 256, f2, 1, 4, 4, 4, 4, 1, flags | 0x80, st) == 1
