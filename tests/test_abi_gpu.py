"""The C-ABI of libqgtc_hip.so driven WITHOUT the PyTorch extension: ctypes + raw device pointers, the way
INTEGRATION.md section 3 shows a non-torch host (cgo / JNI / plain C) would call it. torch is used only as the
device allocator here; every compute call goes through `extern "C"` entry points with plain pointers and sizes, and
the results are compared with the oracle."""
import ctypes

import numpy as np
import pytest

from helpers import QgtcBatch, QgtcExpandJob, QgtcOperand, QgtcPackJob, QgtcProblem, QgtcStage, rand_q

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




@pytest.mark.parametrize("flags", [0x0, 0x10, 0x8], ids=["popcount", "auto", "mfma"])
def test_layer_entry_with_raw_descriptors(lib, oracle, flags):
    """qgtc_gcn_layer_batched through ctypes: descriptors written by the HOST into a device buffer (struct layout of
    include/qgtc.h), raw device pointers - against the oracle's two products."""
    import torch
    assert ctypes.sizeof(QgtcProblem) == 72   # three pointers, two u64, five i32 (+4 padding), one pointer
    lib.qgtc_gcn_layer_batched.argtypes = [vp, vp, ctypes.c_int] + [ctypes.c_int] * 4 + [ctypes.c_int] * 6 + [ctypes.c_uint, vp]
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
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for epoch in (1, 2):
        rc = lib.qgtc_gcn_layer_batched(descs.data_ptr(), descs.data_ptr() + 72 * count, count, max(n for n, _ in batches), f_in,
                                        max(n for n, _ in batches), f_out, act, wb, act, 1, 1, 2, flags, st)
        assert rc == 0, lib.qgtc_strerror(rc)
        torch.cuda.synchronize()
        for (T, T_o, out, out_o) in want:
            np.testing.assert_array_equal(T.cpu().numpy().view(np.uint32), T_o)
            np.testing.assert_array_equal(out.cpu().numpy().reshape(out_o.shape), out_o)
            out.fill_(-3.0)
    # bad arguments are error codes
    assert lib.qgtc_gcn_layer_batched(None, descs.data_ptr(), count, 10, 10, 10, 10, 2, 2, 2, 1, 1, 2, flags, st) == 1
    assert lib.qgtc_gcn_layer_batched(descs.data_ptr(), descs.data_ptr(), count, 10, 10, 10, 10, 2, 2, 2, 1, 1, 1, flags, st) == 1


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










SRC_A, SRC_X, SRC_XR, SRC_WEIGHT, SRC_STAGE, DIM_NODES = 0, 1, 2, 16, 32, -1


@pytest.mark.parametrize("count", [1, 75, 256, 257, 1025, 2500])
def test_epoch_plan_fill_offsets_for_many_batch_counts(lib, count):
    """qgtc_epoch_plan_fill's descriptors against qgtc_epoch_pool_layout for batch counts on both sides of its 256-batch pass (one
    pass keeps the table line and the handed-out pointers in registers, more than one re-reads them): every output where the host
    layout puts it, every chained operand the pointer of the stage it names. No product is launched (the operands are never read)."""
    import torch
    lib.qgtc_epoch_pool_layout.restype = ctypes.c_size_t
    lib.qgtc_epoch_pool_layout.argtypes = [vp, ctypes.c_int, vp, ctypes.c_int, vp]
    lib.qgtc_epoch_plan_fill.argtypes = [vp, ctypes.c_int, vp, ctypes.c_int, vp, ctypes.c_int, vp, ctypes.c_size_t, vp, vp]
    rng = np.random.default_rng(count)
    ns = [int(v) for v in rng.integers(1, 300, size=count)]
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    base = torch.zeros(1024, dtype=torch.int32, device="cuda")       # one real allocation: every operand points into it
    op = lambda k: QgtcOperand(base.data_ptr() + 16 * k, 4)          # noqa: E731
    hb = [QgtcBatch(op(1), op(2), op(3), op(4), op(5), None, n, 0) for n in ns]
    batches = torch.frombuffer(bytearray(bytes((QgtcBatch * count)(*hb))), dtype=torch.uint8).cuda()
    F, H, C, b = 48, 64, 10, 2
    SRC_XC, SRC_AT = 3, 4
    recipes = [(SRC_XR, SRC_WEIGHT + 0, F, H, b, b, b, 1, 0, 0, 1), (SRC_AT, SRC_STAGE + 0, DIM_NODES, H, 1, b, b, 0, 0, 0, 0),
               (SRC_STAGE + 1, SRC_WEIGHT + 1, H, C, b, b, b, 1, 0, 0, 0), (SRC_A, SRC_STAGE + 2, DIM_NODES, C, 1, b, 1, 2, 1, 0, 0),
               (SRC_A, SRC_XC, DIM_NODES, F, 1, b, b, 0, 0, 0, 0)]
    S = len(recipes)
    stages = (QgtcStage * S)(*[QgtcStage(*r) for r in recipes])
    weights = (QgtcOperand * 2)(op(6), op(7))
    nodes = (ctypes.c_int32 * count)(*ns)
    offs = (ctypes.c_uint64 * (S * count))()
    pool_words = lib.qgtc_epoch_pool_layout(ctypes.addressof(nodes), count, ctypes.addressof(stages), S, ctypes.addressof(offs))
    pool = torch.empty(pool_words, dtype=torch.int32, device="cuda")
    descs = torch.zeros(S * count * 72, dtype=torch.uint8, device="cuda")
    rc = lib.qgtc_epoch_plan_fill(batches.data_ptr(), count, ctypes.addressof(stages), S, ctypes.addressof(weights), 2, pool.data_ptr(), pool_words,
                                  descs.data_ptr(), st)
    assert rc == 0, lib.qgtc_strerror(rc)
    got = np.frombuffer(descs.cpu().numpy().tobytes(), dtype=np.uint64).reshape(S, count, 9)    # X, W, out, x_words, w_words, (M, K), (N, w_lines), (occ_words, pad), occ
    O = np.array(list(offs), dtype=np.uint64).reshape(S, count)
    np.testing.assert_array_equal(got[:, :, 2], np.uint64(pool.data_ptr()) + np.uint64(4) * O)
    np.testing.assert_array_equal(got[1, :, 1], got[0, :, 2])                                    # A . T1: T1 is stage 0's output
    np.testing.assert_array_equal(got[2, :, 0], got[1, :, 2])
    np.testing.assert_array_equal(got[3, :, 1], got[2, :, 2])
    a = np.uint64(base.data_ptr())
    assert (got[0, :, 0] == a + np.uint64(48)).all() and (got[1, :, 0] == a + np.uint64(80)).all() and (got[3, :, 0] == a + np.uint64(16)).all()   # XR, AT, A
    assert (got[4, :, 1] == a + np.uint64(64)).all() and (got[0, :, 1] == a + np.uint64(96)).all()                                       # XC, weight 0
    M = (got[:, :, 5] & np.uint64(0xffffffff)).astype(np.int64)
    K = (got[:, :, 5] >> np.uint64(32)).astype(np.int64)
    assert (M == np.array(ns)[None, :]).all() and (K[1] == np.array(ns)).all() and (K[0] == F).all() and (K[2] == H).all()
    assert (np.diff(np.concatenate([O.reshape(-1), [pool_words]]).astype(np.int64)) >= 0).all()   # (stage, batch) order, nothing overlaps
    lib.qgtc_last_batched_violation.argtypes = [ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), vp]
    assert lib.qgtc_last_batched_violation(None, None, st) == 0


def test_epoch_plan_filled_on_the_device_with_raw_pointers(lib, oracle):
    """qgtc_val2bit_batched + qgtc_epoch_pool_layout + qgtc_epoch_plan_fill + the grouped entries through ctypes: the three
    weights packed in one launch, the descriptors of a layout-correct two-layer GCN slice (X.W1 -> A.T1 -> .W2 -> A.T2 as
    float32) filled by ONE launch from the per-batch table, QGTC_CHECK_DESCRIPTORS on every launch - against the oracle."""
    import torch
    assert ctypes.sizeof(QgtcBatch) == 96 and ctypes.sizeof(QgtcStage) == 44 and ctypes.sizeof(QgtcPackJob) == 48
    lib.qgtc_val2bit_batched.argtypes = [vp, ctypes.c_int, vp]
    lib.qgtc_epoch_pool_layout.restype = ctypes.c_size_t
    lib.qgtc_epoch_pool_layout.argtypes = [vp, ctypes.c_int, vp, ctypes.c_int, vp]
    lib.qgtc_epoch_plan_fill.argtypes = [vp, ctypes.c_int, vp, ctypes.c_int, vp, ctypes.c_int, vp, ctypes.c_size_t, vp, vp]
    lib.qgtc_bitmm_batched.argtypes = [vp, ctypes.c_int] + [ctypes.c_int] * 3 + [ctypes.c_int] * 4 + [ctypes.c_uint, vp]
    lib.qgtc_last_batched_violation.argtypes = [ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), vp]
    rng = np.random.default_rng(41)
    b, F, H, C = 2, 48, 64, 10
    ns = [150, 333, 40, 97]
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    # weights: float matrices -> packed, one launch
    Wf = [rng.uniform(-1, 5, size=s).astype(np.float32) for s in ((F, H), (H, C))]
    dWf = [torch.from_numpy(w).cuda() for w in Wf]
    words = [int(lib.qgtc_cols_words(F, H, b, 0)), int(lib.qgtc_cols_words(H, C, b, 0))]
    dW = [torch.full((w,), -1, dtype=torch.int32, device="cuda") for w in words]
    jobs = (QgtcPackJob * 2)(QgtcPackJob(dWf[0].data_ptr(), dW[0].data_ptr(), words[0], F, H, b, 1, 0, 0),
                             QgtcPackJob(dWf[1].data_ptr(), dW[1].data_ptr(), words[1], H, C, b, 1, 0, 0))
    assert lib.qgtc_val2bit_batched(ctypes.addressof(jobs), 2, st) == 0
    W_o = [oracle.val2bit(Wf[0], b, True, False), oracle.val2bit(Wf[1], b, True, False)]
    for d, o in zip(dW, W_o):
        np.testing.assert_array_equal(d.cpu().numpy().view(np.uint32), o)
    assert lib.qgtc_val2bit_batched(ctypes.addressof(jobs), 9, st) == 1            # more jobs than QGTC_MAX_PACK_JOBS
    jobs[1].out_words = 3
    assert lib.qgtc_val2bit_batched(ctypes.addressof(jobs), 2, st) == 2            # an undersized output is an error code
    # the data loader's table
    keep, hb, ref = [], [], []
    for n in ns:
        qa = (rng.random((n, n)) < 0.03).astype(np.int32)
        qx = rand_q(rng, n, F, b)
        A, Xr = oracle.pack(qa, 1, False), oracle.pack(qx, b, False)
        dA, dXr = torch.from_numpy(A.view(np.int32)).cuda(), torch.from_numpy(Xr.view(np.int32)).cuda()
        keep += [dA, dXr]
        hb.append(QgtcBatch(QgtcOperand(dA.data_ptr(), dA.numel()), QgtcOperand(None, 0), QgtcOperand(dXr.data_ptr(), dXr.numel()), QgtcOperand(None, 0), QgtcOperand(None, 0), None, n, 0))
        t1 = oracle.bitmm2bit(Xr, W_o[0], n, F, H, b, b, b, col=True)
        h1 = oracle.bitmm2bit(A, t1, n, n, H, 1, b, b)
        t2 = oracle.bitmm2bit(h1, W_o[1], n, H, C, b, b, b, col=True)
        ref.append((t1, h1, t2, oracle.bitmm2int(A, t2, n, n, C, 1, b, True)))
    count = len(ns)
    batches = torch.frombuffer(bytearray(bytes((QgtcBatch * count)(*hb))), dtype=torch.uint8).cuda()
    stages = (QgtcStage * 4)(QgtcStage(SRC_XR, SRC_WEIGHT + 0, F, H, b, b, b, 1, 0, 0, 0), QgtcStage(SRC_A, SRC_STAGE + 0, DIM_NODES, H, 1, b, b, 0, 0, 0, 0),
                             QgtcStage(SRC_STAGE + 1, SRC_WEIGHT + 1, H, C, b, b, b, 1, 0, 0, 0), QgtcStage(SRC_A, SRC_STAGE + 2, DIM_NODES, C, 1, b, 1, 2, 1, 0, 0))
    weights = (QgtcOperand * 2)(QgtcOperand(dW[0].data_ptr(), dW[0].numel()), QgtcOperand(dW[1].data_ptr(), dW[1].numel()))
    nodes = (ctypes.c_int32 * count)(*ns)
    offs = (ctypes.c_uint64 * (4 * count))()
    pool_words = lib.qgtc_epoch_pool_layout(ctypes.addressof(nodes), count, ctypes.addressof(stages), 4, ctypes.addressof(offs))
    want_words = sum(lib.qgtc_cols_words(n, H, b, 0) + lib.qgtc_rows_words(n, H, b) + lib.qgtc_cols_words(n, C, b, 0) + (n * C + 3) // 4 * 4 for n in ns)
    assert pool_words == want_words and offs[0] == 0 and all(o % 4 == 0 for o in offs)
    pool = torch.full((pool_words,), -1, dtype=torch.int32, device="cuda")
    descs = torch.zeros(4 * count * 72, dtype=torch.uint8, device="cuda")
    rc = lib.qgtc_epoch_plan_fill(batches.data_ptr(), count, ctypes.addressof(stages), 4, ctypes.addressof(weights), 2, pool.data_ptr(), pool_words,
                                  descs.data_ptr(), st)
    assert rc == 0, lib.qgtc_strerror(rc)
    got = (QgtcProblem * (4 * count)).from_buffer_copy(descs.cpu().numpy().tobytes())
    for s in range(4):
        for i, n in enumerate(ns):
            d = got[s * count + i]
            assert d.out == pool.data_ptr() + 4 * offs[s * count + i] and d.M == n and d.N == (H if s < 2 else C)
            assert d.K == (n if s in (1, 3) else (F if s == 0 else H)) and d.occ is None
    assert got[count].W == got[0].out and got[2 * count].X == got[count].out and got[3 * count].W == got[2 * count].out
    CHECK, AUTO = 0x200, 0x10
    mx = max(ns)
    for s, (K, N, b1, b2, ob, mode) in enumerate(((F, H, b, b, b, 1), (mx, H, 1, b, b, 0), (H, C, b, b, b, 1), (mx, C, 1, b, 1, 2))):
        rc = lib.qgtc_bitmm_batched(descs.data_ptr() + 72 * count * s, count, mx, K, N, b1, b2, ob, mode, AUTO | CHECK, st)
        assert rc == 0, lib.qgtc_strerror(rc)
    prob, field = ctypes.c_int(7), ctypes.c_int(7)
    assert lib.qgtc_last_batched_violation(ctypes.byref(prob), ctypes.byref(field), st) == 0 and prob.value == -1 and field.value == 0
    P = pool.cpu().numpy().view(np.uint32)
    for i, n in enumerate(ns):
        for s in range(3):
            o = int(offs[s * count + i])
            np.testing.assert_array_equal(P[o:o + ref[i][s].size], ref[i][s].reshape(-1), err_msg=f"stage {s} batch {i}")
        o = int(offs[3 * count + i])
        np.testing.assert_array_equal(P[o:o + n * C].view(np.float32).reshape(n, C), ref[i][3])
    # ---- detected, not documented: a descriptor above the stated maxima sets the record (field 1 = M, 2 = K, 3 = N)
    assert lib.qgtc_bitmm_batched(descs.data_ptr() + 72 * count, count, 200, mx, H, 1, b, b, 0, AUTO | CHECK, st) == 0     # batch 1 has 333 rows
    assert lib.qgtc_last_batched_violation(ctypes.byref(prob), ctypes.byref(field), st) == 1 and (prob.value, field.value) == (1, 1)
    assert lib.qgtc_last_batched_violation(ctypes.byref(prob), ctypes.byref(field), st) == 0                                 # read once, then cleared
    assert lib.qgtc_bitmm_batched(descs.data_ptr(), count, mx, F, H - 1, b, b, b, 1, AUTO | CHECK, st) == 0                # N above max_N in every problem
    assert lib.qgtc_last_batched_violation(ctypes.byref(prob), ctypes.byref(field), st) == 1 and (prob.value, field.value) == (0, 3)
    lib.qgtc_gcn_chain_batched.argtypes = [vp, vp, ctypes.c_int] + [ctypes.c_int] * 4 + [ctypes.c_int] * 6 + [ctypes.c_uint, vp]
    # a "chain" whose second stage does not read the first one's output (stage 1 then stage 3)
    assert lib.qgtc_gcn_chain_batched(descs.data_ptr() + 72 * count, descs.data_ptr() + 72 * count * 3, count, mx, mx, H, C, 1, b, b, b, b, 2, CHECK, st) == 0
    assert lib.qgtc_last_batched_violation(ctypes.byref(prob), ctypes.byref(field), st) == 1 and field.value in (2, 5)
    # without the flag nothing is recorded
    assert lib.qgtc_bitmm_batched(descs.data_ptr() + 72 * count, count, 200, mx, H, 1, b, b, 0, AUTO, st) == 0
    assert lib.qgtc_last_batched_violation(None, None, st) == 0
    # a pool smaller than the layout says is found by the fill kernel itself
    assert lib.qgtc_epoch_plan_fill(batches.data_ptr(), count, ctypes.addressof(stages), 4, ctypes.addressof(weights), 2, pool.data_ptr(), pool_words - 8,
                                    descs.data_ptr(), st) == 0
    assert lib.qgtc_last_batched_violation(ctypes.byref(prob), ctypes.byref(field), st) == 1 and field.value == 4
    torch.cuda.synchronize()
    over = (QgtcProblem * (4 * count)).from_buffer_copy(descs.cpu().numpy().tobytes())
    last = over[4 * count - 1]                                   # the output that passes the pool's end: M = 0 (every grouped kernel skips it),
    assert last.M == 0 and pool.data_ptr() <= last.out <= pool.data_ptr() + 4 * (pool_words - 8) and last.out % 16 == 0   # a pointer inside the pool, aligned
    assert all(over[i].M == ns[i % count] for i in range(4 * count - 1))
    # a batch without nodes in the device table: M = 0 descriptors and QGTC_VIOL_M for that batch, the others are planned as before
    hb0 = list(hb)
    hb0[2] = QgtcBatch(hb[2].A, hb[2].X, hb[2].XR, hb[2].XC, hb[2].AT, None, 0, 0)
    batches0 = torch.frombuffer(bytearray(bytes((QgtcBatch * count)(*hb0))), dtype=torch.uint8).cuda()
    assert lib.qgtc_epoch_plan_fill(batches0.data_ptr(), count, ctypes.addressof(stages), 4, ctypes.addressof(weights), 2, pool.data_ptr(), pool_words,
                                    descs.data_ptr(), st) == 0
    assert lib.qgtc_last_batched_violation(ctypes.byref(prob), ctypes.byref(field), st) == 1 and (prob.value, field.value) == (2, 1)
    zero_n = (QgtcProblem * (4 * count)).from_buffer_copy(descs.cpu().numpy().tobytes())
    assert all(zero_n[s * count + 2].M == 0 for s in range(4)) and zero_n[0].M == ns[0] and zero_n[3].M == ns[3]
    # bad recipes are error codes
    stages[1].left = SRC_STAGE + 3
    assert lib.qgtc_epoch_plan_fill(batches.data_ptr(), count, ctypes.addressof(stages), 4, ctypes.addressof(weights), 2, pool.data_ptr(), pool_words,
                                    descs.data_ptr(), st) == 1
    torch.cuda.synchronize()




def _chain_covered(b, F, H, C):
    """include/qgtc.h (ABI 11): 1 .. 4 bits with up to 256 columns, 5 .. 8 bits with up to 128 and an exact X . W (PAD128(F) (2^b - 1)^2 < 2^24)."""
    return max(H, C) <= (256 if b <= 4 else 128) and F <= 8192 and (F + 127) // 128 * 128 * ((1 << b) - 1) ** 2 < (1 << 24)


@pytest.mark.parametrize("M,K,F,H,C,bitmaps", [(333, 333, 48, 128, 10, True), (150, 150, 128, 64, 128, False), (300, 150, 32, 100, 33, True), (200, 200, 602, 128, 41, True), (70, 70, 3703, 33, 7, False), (120, 120, 300, 70, 90, True),
                                               (150, 300, 100, 33, 70, True), (1213, 1213, 128, 128, 128, True), (40, 40, 7, 5, 3, False),
                                               # ABI 11: 129 .. 256 hidden units / classes (main_qgtc.py:31: --n-hidden is free)
                                               (333, 333, 48, 256, 10, True), (150, 300, 128, 200, 256, False), (300, 150, 300, 40, 130, True), (1213, 1213, 128, 256, 256, True),
                                               (200, 200, 64, 160, 16, True)])
@pytest.mark.parametrize("b", [2, 1, 3, 4, 5, 6, 7, 8])     # one width per chain (main_qgtc.py's --bit_width; 2_7c_QGTC_GEMM_INT8.py:15 sweeps 1 .. 8); the BASELINE epoch: 2
def test_chain_entries_with_raw_descriptors(lib, oracle, M, K, F, H, C, bitmaps, b):
    """qgtc_expand_weights + qgtc_chain_transform + qgtc_chain_aggregate through ctypes: T = requant(X . W1) in the chain's
    private format, T' = requant(requant(A . T) . W2) (out_mode 1), then float32(A2 . T') (out_mode 0) and
    float32(requant(A . T) . W2) (out_mode 2) against the oracle's public operators. Ragged sizes, widths that are not
    multiples of 32, non-square adjacencies (the diagonal k-quad does not exist for every row group), with and without
    occupancy bitmaps, three batches per launch."""
    import torch
    if not _chain_covered(b, F, H, C):
        pytest.skip("outside the chain entries' range at this width (callers fall back to the grouped GEMMs)")
    lib.qgtc_weight_codes_words.restype = lib.qgtc_chain_words.restype = lib.qgtc_occupancy_words.restype = ctypes.c_size_t
    lib.qgtc_expand_weights.argtypes = [vp, ctypes.c_int, vp]
    lib.qgtc_chain_transform.argtypes = [vp, ctypes.c_int] + [ctypes.c_int] * 5 + [vp, ctypes.c_uint, vp]
    lib.qgtc_chain_aggregate.argtypes = [vp, vp, ctypes.c_int] + [ctypes.c_int] * 8 + [vp, ctypes.c_uint, vp]
    lib.qgtc_tile_occupancy.argtypes = [vp, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, ctypes.c_size_t, vp]
    rng = np.random.default_rng(M + 3 * K + F)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    P128 = lambda x: (x + 127) // 128 * 128   # noqa: E731
    W1, W2 = oracle.pack(rand_q(rng, F, H, b), b, True), oracle.pack(rand_q(rng, H, C, b), b, True)
    dW1, dW2 = torch.from_numpy(W1.view(np.int32)).cuda(), torch.from_numpy(W2.view(np.int32)).cuda()
    c1 = torch.full((int(lib.qgtc_weight_codes_words(F, H, b, 0)),), -1, dtype=torch.int32, device="cuda")   # (a table per k-quad of F)
    c2 = torch.full((int(lib.qgtc_weight_codes_words(H, C, b, 1)),), -1, dtype=torch.int32, device="cuda")
    jobs = (QgtcExpandJob * 2)(QgtcExpandJob(dW1.data_ptr(), c1.data_ptr(), dW1.numel(), F, H, b, P128(H), 0, c1.numel()),
                               QgtcExpandJob(dW2.data_ptr(), c2.data_ptr(), dW2.numel(), H, C, b, P128(C), 1, c2.numel()))
    assert lib.qgtc_expand_weights(ctypes.addressof(jobs), 2, st) == 0
    count = 3
    keep, s_x, s_a, s_xw, s_a2, s_f, want = [dW1, dW2, c1, c2], [], [], [], [], [], []
    for i in range(count):
        k = max(1, K - 17 * i)          # ragged batches
        m = max(1, M - 29 * i)
        qx = rand_q(rng, k, F, b)
        qa = (rng.random((m, k)) < 0.04).astype(np.int32)
        qa[:, (k // 2):(k // 2 + k // 4)] = 0                      # whole k-quads of zeros for the bitmaps to name
        qa2 = (rng.random((m, m)) < 0.05).astype(np.int32)
        X, A, A2 = oracle.pack(qx, b, False), oracle.pack(qa, 1, False), oracle.pack(qa2, 1, False)
        dX, dA, dA2 = (torch.from_numpy(t.view(np.int32)).cuda() for t in (X, A, A2))
        T = torch.full((int(lib.qgtc_chain_words(k, H, b)),), -1, dtype=torch.int32, device="cuda")
        T2 = torch.full((int(lib.qgtc_chain_words(m, C, b)),), -1, dtype=torch.int32, device="cuda")
        out0 = torch.full((m * C,), -7.0, dtype=torch.float32, device="cuda")
        out2 = torch.full((m * C,), -7.0, dtype=torch.float32, device="cuda")
        occ = occ2 = None
        if bitmaps:
            occ = torch.empty(int(lib.qgtc_occupancy_words(m, k)), dtype=torch.int64, device="cuda")
            occ2 = torch.empty(int(lib.qgtc_occupancy_words(m, m)), dtype=torch.int64, device="cuda")
            assert lib.qgtc_tile_occupancy(dA.data_ptr(), dA.numel(), m, k, 1, occ.data_ptr(), occ.numel(), st) == 0
            assert lib.qgtc_tile_occupancy(dA2.data_ptr(), dA2.numel(), m, m, 1, occ2.data_ptr(), occ2.numel(), st) == 0
        keep += [dX, dA, dA2, T, T2, out0, out2, occ, occ2]
        ow = lambda kk: (((kk + 127) // 128) + 63) // 64      # noqa: E731
        s_x.append(QgtcProblem(dX.data_ptr(), dW1.data_ptr(), T.data_ptr(), dX.numel(), dW1.numel(), k, F, H, P128(H), 0, None))
        s_a.append(QgtcProblem(dA.data_ptr(), T.data_ptr(), None, dA.numel(), T.numel(), m, k, H, P128(H), ow(k) if bitmaps else 0, occ.data_ptr() if bitmaps else None))
        s_xw.append(QgtcProblem(None, dW2.data_ptr(), T2.data_ptr(), 0, dW2.numel(), m, H, C, P128(C), 0, None))
        s_f.append(QgtcProblem(None, dW2.data_ptr(), out2.data_ptr(), 0, dW2.numel(), m, H, C, P128(C), 0, None))
        s_a2.append(QgtcProblem(dA2.data_ptr(), T2.data_ptr(), out0.data_ptr(), dA2.numel(), T2.numel(), m, m, C, P128(C), ow(m) if bitmaps else 0, occ2.data_ptr() if bitmaps else None))
        t_o = oracle.bitmm2bit(X, W1, k, F, H, b, b, b, col=True)
        h_o = oracle.bitmm2bit(A, t_o, m, k, H, 1, b, b)
        t2_o = oracle.bitmm2bit(h_o, W2, m, H, C, b, b, b, col=True)
        want.append((out0, oracle.bitmm2int(A2, t2_o, m, m, C, 1, b, True), out2, oracle.bitmm2int(h_o, W2, m, H, C, b, b, True)))
    host = (QgtcProblem * (5 * count))(*(s_x + s_a + s_xw + s_f + s_a2))
    descs = torch.frombuffer(bytearray(bytes(host)), dtype=torch.uint8).cuda()
    d = lambda i: descs.data_ptr() + 72 * count * i       # noqa: E731
    CHECK = 0x200
    for rep in range(2):     # (a second pass over the same buffers: nothing depends on what a launch left behind)
        rc = lib.qgtc_chain_transform(d(0), count, K, F, H, b, b, c1.data_ptr(), CHECK, st)
        assert rc == 0, lib.qgtc_strerror(rc)
        rc = lib.qgtc_chain_aggregate(d(1), d(2), count, M, K, H, C, b, b, b, 1, c2.data_ptr(), CHECK, st)
        assert rc == 0, lib.qgtc_strerror(rc)
        rc = lib.qgtc_chain_aggregate(d(1), d(3), count, M, K, H, C, b, b, b, 2, c2.data_ptr(), 0, st)
        assert rc == 0, lib.qgtc_strerror(rc)
        rc = lib.qgtc_chain_aggregate(d(4), None, count, M, M, C, 0, b, 0, 0, 0, None, CHECK, st)
        assert rc == 0, lib.qgtc_strerror(rc)
        torch.cuda.synchronize()
        for i, (o0, w0, o2, w2) in enumerate(want):
            np.testing.assert_array_equal(o2.cpu().numpy().reshape(w2.shape), w2, err_msg=f"out_mode 2, batch {i}")
            np.testing.assert_array_equal(o0.cpu().numpy().reshape(w0.shape), w0, err_msg=f"out_mode 0 after out_mode 1, batch {i}")
            o0.fill_(-7.0)
            o2.fill_(-7.0)
    lib.qgtc_last_batched_violation.argtypes = [ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), vp]
    assert lib.qgtc_last_batched_violation(None, None, st) == 0
    # outside the entries' range: error codes (callers fall back to qgtc_gcn_chain_batched)
    assert lib.qgtc_chain_transform(d(0), count, K, 8193, H, b, b, c1.data_ptr(), 0, st) == 1         # K > 8192
    assert lib.qgtc_chain_transform(d(0), count, K, F, H, b, 9, c1.data_ptr(), 0, st) == 1            # 9-bit T
    assert lib.qgtc_chain_transform(d(0), count, K, F, H, 3, 2, c1.data_ptr(), 0, st) == 1            # three planes of X into a one-digit chain
    assert lib.qgtc_chain_transform(d(0), count, K, 300, 128, 8, 8, c1.data_ptr(), 0, st) == 1        # 8 x 8 bits over 300 features: float32 sums inexact
    assert lib.qgtc_chain_transform(d(0), count, K, 128, 129, 8, 8, c1.data_ptr(), 0, st) == 1        # 5 .. 8 bits: at most 128 columns
    assert lib.qgtc_chain_aggregate(d(1), d(2), count, M, K, H, C, b, 5 - b, 5 - b, 1, c2.data_ptr(), 0, st) == 1   # T and the aggregate in different format classes
    assert lib.qgtc_chain_aggregate(d(1), d(2), count, M, K, 257, C, b, b, b, 1, c2.data_ptr(), 0, st) == 1
    assert lib.qgtc_chain_aggregate(d(1), d(2), count, M, K, 129, C, 8, 8, 8, 1, c2.data_ptr(), 0, st) == 1
    assert lib.qgtc_chain_aggregate(d(1), None, count, M, K, H, C, b, b, b, 1, c2.data_ptr(), 0, st) == 1
    assert lib.qgtc_expand_weights(ctypes.addressof(jobs), 9, st) == 1
    # ADVICE r4 / ABI 11: the capacity of `codes` travels with the job - a table sized for ONE k-quad (what ABI 10's two-argument size
    # helper returned) is QGTC_ESIZE for a K of several, never a device write past the buffer
    one_kq = int(lib.qgtc_weight_codes_words(128, H, b, 0))
    assert int(lib.qgtc_weight_codes_words(F, H, b, 0)) == one_kq * ((F + 127) // 128) and int(lib.qgtc_weight_codes_words(min(F, 128), H, b, 1)) == one_kq
    small = (QgtcExpandJob * 1)(QgtcExpandJob(dW1.data_ptr(), c1.data_ptr(), dW1.numel(), F, H, b, P128(H), 0, one_kq - 1))
    assert lib.qgtc_expand_weights(ctypes.addressof(small), 1, st) == 2, "QGTC_ESIZE"
    if F > 128:
        small[0].codes_words = one_kq
        assert lib.qgtc_expand_weights(ctypes.addressof(small), 1, st) == 2
        # a descriptor whose K names more k-quads than the tables hold: the kernel's loop bound is the HOST's K (no read past w_codes),
        # and QGTC_CHECK_DESCRIPTORS reports the mismatch
        rc = lib.qgtc_chain_transform(d(0), count, K, 128, H, b, b, c1.data_ptr(), 0, st)
        assert rc == 0, lib.qgtc_strerror(rc)
        torch.cuda.synchronize()
        assert lib.qgtc_chain_transform(d(0), count, K, 128, H, b, b, c1.data_ptr(), CHECK, st) == 0     # (the check is asynchronous: its record says)
        prob, field = ctypes.c_int(-2), ctypes.c_int(-2)
        assert lib.qgtc_last_batched_violation(ctypes.byref(prob), ctypes.byref(field), st) == 1      # QGTC_EINVAL: a violation is on record
        assert prob.value == 0 and field.value == 2, (prob.value, field.value)     # QGTC_VIOL_K of problem 0


@pytest.mark.parametrize("M,F,H,C,bitmaps", [(599, 50, 64, 10, True), (333, 64, 33, 64, False), (40, 7, 5, 3, True), (333, 100, 128, 70, True),
                                             (300, 200, 256, 130, True), (150, 29, 160, 10, False)])
@pytest.mark.parametrize("b", [4, 8, 5, 3])
def test_chain_entries_at_four_bits(lib, oracle, M, F, H, C, bitmaps, b):
    """The Batched-GIN shape of the chain entries (main_qgtc.py:131-138 with every right operand in the cols layout): X arrives in
    the PUBLIC cols layout and is converted once (qgtc_chain_from_cols, the data loader's step), then
    T1 = requant(requant(A . X) . W1) (out_mode 1), out = float32(requant(A . T1) . W2) (out_mode 2) - 4-bit values, two base-4
    digits a nibble, 4-plane weights (W2 in val2bit's output_layer form: PAD8 lines) - against the oracle. ABI 11: the same at 5 .. 8 bits
    (X converted into two arrays of the 4-bit form) and with up to 256 columns at 3 / 4 bits."""
    import torch
    if max(F, H, C) > (256 if b <= 4 else 128):
        pytest.skip("outside the chain entries' range at this width")
    lib.qgtc_weight_codes_words.restype = lib.qgtc_chain_words.restype = lib.qgtc_occupancy_words.restype = ctypes.c_size_t
    lib.qgtc_expand_weights.argtypes = [vp, ctypes.c_int, vp]
    lib.qgtc_chain_from_cols.argtypes = [vp, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, ctypes.c_size_t, vp]
    lib.qgtc_chain_aggregate.argtypes = [vp, vp, ctypes.c_int] + [ctypes.c_int] * 8 + [vp, ctypes.c_uint, vp]
    lib.qgtc_tile_occupancy.argtypes = [vp, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, ctypes.c_size_t, vp]
    rng = np.random.default_rng(M + F)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    P128, P8 = (lambda x: (x + 127) // 128 * 128), (lambda x: (x + 7) // 8 * 8)   # noqa: E731
    W1, W2 = oracle.pack(rand_q(rng, F, H, b), b, True), oracle.pack(rand_q(rng, H, C, b), b, True, True)
    dW1, dW2 = torch.from_numpy(W1.view(np.int32)).cuda(), torch.from_numpy(W2.view(np.int32)).cuda()
    c1 = torch.full((int(lib.qgtc_weight_codes_words(F, H, b, 1)),), -1, dtype=torch.int32, device="cuda")
    c2 = torch.full((int(lib.qgtc_weight_codes_words(H, C, b, 1)),), -1, dtype=torch.int32, device="cuda")
    jobs = (QgtcExpandJob * 2)(QgtcExpandJob(dW1.data_ptr(), c1.data_ptr(), dW1.numel(), F, H, b, P128(H), 1, c1.numel()),
                               QgtcExpandJob(dW2.data_ptr(), c2.data_ptr(), dW2.numel(), H, C, b, P8(C), 1, c2.numel()))
    assert lib.qgtc_expand_weights(ctypes.addressof(jobs), 2, st) == 0
    count = 3
    keep, s0, s1, s2, s3, want = [], [], [], [], [], []
    for i in range(count):
        m = max(1, M - 23 * i)
        qx = rand_q(rng, m, F, b)
        qa = (rng.random((m, m)) < 0.05).astype(np.int32)
        X, A = oracle.pack(qx, b, True), oracle.pack(qa, 1, False)
        dX, dA = torch.from_numpy(X.view(np.int32)).cuda(), torch.from_numpy(A.view(np.int32)).cuda()
        XC = torch.full((int(lib.qgtc_chain_words(m, F, b)),), -1, dtype=torch.int32, device="cuda")
        assert lib.qgtc_chain_from_cols(dX.data_ptr(), dX.numel(), m, F, b, XC.data_ptr(), XC.numel(), st) == 0
        T1 = torch.full((int(lib.qgtc_chain_words(m, H, b)),), -1, dtype=torch.int32, device="cuda")
        out = torch.full((m * C,), -7.0, dtype=torch.float32, device="cuda")
        occ = None
        if bitmaps:
            occ = torch.empty(int(lib.qgtc_occupancy_words(m, m)), dtype=torch.int64, device="cuda")
            assert lib.qgtc_tile_occupancy(dA.data_ptr(), dA.numel(), m, m, 1, occ.data_ptr(), occ.numel(), st) == 0
        keep += [dX, dA, XC, T1, out, occ]
        ow = (((m + 127) // 128) + 63) // 64
        oc = (ow if bitmaps else 0, occ.data_ptr() if bitmaps else None)
        s0.append(QgtcProblem(dA.data_ptr(), XC.data_ptr(), None, dA.numel(), XC.numel(), m, m, F, P128(F), *oc))
        s1.append(QgtcProblem(None, dW1.data_ptr(), T1.data_ptr(), 0, dW1.numel(), m, F, H, P128(H), 0, None))
        s2.append(QgtcProblem(dA.data_ptr(), T1.data_ptr(), None, dA.numel(), T1.numel(), m, m, H, P128(H), *oc))
        s3.append(QgtcProblem(None, dW2.data_ptr(), out.data_ptr(), 0, dW2.numel(), m, H, C, P8(C), 0, None))
        ax = oracle.bitmm2bit(A, X, m, m, F, 1, b, b)
        t1 = oracle.bitmm2bit(ax, W1, m, F, H, b, b, b, col=True)
        a1 = oracle.bitmm2bit(A, t1, m, m, H, 1, b, b)
        want.append((out, oracle.bitmm2int(a1, W2, m, H, C, b, b, False)))
    host = (QgtcProblem * (4 * count))(*(s0 + s1 + s2 + s3))
    descs = torch.frombuffer(bytearray(bytes(host)), dtype=torch.uint8).cuda()
    d = lambda i: descs.data_ptr() + 72 * count * i       # noqa: E731
    for rep in range(2):
        rc = lib.qgtc_chain_aggregate(d(0), d(1), count, M, M, F, H, b, b, b, 1, c1.data_ptr(), 0x200, st)
        assert rc == 0, lib.qgtc_strerror(rc)
        rc = lib.qgtc_chain_aggregate(d(2), d(3), count, M, M, H, C, b, b, b, 2, c2.data_ptr(), 0x200, st)
        assert rc == 0, lib.qgtc_strerror(rc)
        torch.cuda.synchronize()
        for i, (o, w) in enumerate(want):
            np.testing.assert_array_equal(o.cpu().numpy().reshape(w.shape), w, err_msg=f"batch {i}")
            o.fill_(-7.0)
    lib.qgtc_last_batched_violation.argtypes = [ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), vp]
    assert lib.qgtc_last_batched_violation(None, None, st) == 0
    # the second product's descriptors are checked too: a stage_xw entry for another batch size is a chaining violation (the entry only
    # takes `out` and N from it, so the product itself is unaffected)
    bad = (QgtcProblem * (4 * count))(*(s0 + s1 + s2 + s3))
    bad[3 * count + 1].M += 1
    bad_descs = torch.frombuffer(bytearray(bytes(bad)), dtype=torch.uint8).cuda()
    assert lib.qgtc_chain_aggregate(bad_descs.data_ptr() + 72 * count * 2, bad_descs.data_ptr() + 72 * count * 3, count, M, M, H, C, b, b, b, 2, c2.data_ptr(), 0x200, st) == 0
    prob, field = ctypes.c_int(-1), ctypes.c_int(0)
    assert lib.qgtc_last_batched_violation(ctypes.byref(prob), ctypes.byref(field), st) == 1 and (prob.value, field.value) == (1, 5)
    # ... and the descriptors' N must EQUAL the width the host states (the stores are sized from it): stating H + 1 for these descriptors is recorded
    if H + 1 <= 64 and b == 4:
        assert lib.qgtc_chain_aggregate(d(0), d(1), count, M, M, F, H + 1, b, b, b, 1, c1.data_ptr(), 0x200, st) == 0
        assert lib.qgtc_last_batched_violation(ctypes.byref(prob), ctypes.byref(field), st) == 1 and field.value == 3
    assert lib.qgtc_chain_aggregate(d(0), d(1), count, M, M, 257, H, b, b, b, 1, c1.data_ptr(), 0, st) == 1    # N <= 256
    assert lib.qgtc_chain_aggregate(d(0), d(1), count, M, M, 129, H, 8, 8, 8, 1, c1.data_ptr(), 0, st) == 1    # 5 .. 8 bits: N <= 128
    assert lib.qgtc_chain_from_cols(dX.data_ptr(), dX.numel(), 10, 10, 9, XC.data_ptr(), XC.numel(), st) == 1   # at most eight planes
    assert lib.qgtc_chain_from_cols(dX.data_ptr(), dX.numel(), M, F, b, XC.data_ptr(), 3, st) == 2


@pytest.mark.parametrize("seed", range(150))
def test_chain_entries_random_sweep(lib, oracle, seed):
    """Random shapes through the chain entries against the oracle: node counts around the 32 / 128 boundaries (and below 32), widths
    around the 32-column blocks, 2-bit (N <= 128) and 4-bit (N <= 64) chains, adjacency density from empty to dense, K != M, pooled
    and absent bitmaps, the adjacency in the rows layout (even seeds) or as tiles (odd seeds: qgtc_adj_tiles_from_rows, checked word
    for word, + QGTC_CHAIN_ADJ_TILES) - X.W (or the converted X) -> aggregation + transform -> float32 aggregation.
    Seeds 80 .. 149 (ABI 11): chains of 5 .. 8 bits (four base-4 digits a value, N <= 128, X . W within the float32-exact range) and
    1 .. 4-bit chains with 129 .. 256 columns on at least one side."""
    import torch
    rng = np.random.default_rng(1000 + seed)
    lib.qgtc_weight_codes_words.restype = lib.qgtc_chain_words.restype = lib.qgtc_occupancy_words.restype = ctypes.c_size_t
    lib.qgtc_expand_weights.argtypes = [vp, ctypes.c_int, vp]
    lib.qgtc_chain_from_cols.argtypes = [vp, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, ctypes.c_size_t, vp]
    lib.qgtc_chain_transform.argtypes = [vp, ctypes.c_int] + [ctypes.c_int] * 5 + [vp, ctypes.c_uint, vp]
    lib.qgtc_chain_aggregate.argtypes = [vp, vp, ctypes.c_int] + [ctypes.c_int] * 8 + [vp, ctypes.c_uint, vp]
    lib.qgtc_tile_occupancy.argtypes = [vp, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, ctypes.c_size_t, vp]
    lib.qgtc_adj_tiles_words.restype = ctypes.c_size_t
    lib.qgtc_adj_tiles_from_rows.argtypes = [vp, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, vp, ctypes.c_size_t, vp]
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    P128 = lambda x: (x + 127) // 128 * 128   # noqa: E731
    tiles = seed % 2 == 1
    wide = seed >= 80 and seed % 2 == 0          # 1 .. 4 bits with more than 128 columns somewhere
    many = seed >= 80 and not wide               # 5 .. 8 bits
    b = int(rng.choice([5, 6, 7, 8, 8])) if many else int(rng.choice([1, 2, 2, 3, 4, 4]))      # one width per chain
    use_xw = bool(rng.integers(0, 2))            # T from qgtc_chain_transform (X . W1), or a data-loader operand converted from the cols layout
    wmax = 128
    pick = lambda hi: int(rng.choice([1, 7, 31, 32, 33, 63, 64, 65, 96, 127, 128, int(rng.integers(1, hi + 1))]))   # noqa: E731
    F, H, C = min(pick(wmax), wmax), min(pick(wmax), wmax), min(pick(wmax), wmax)
    if wide:
        big = lambda: int(rng.choice([129, 160, 161, 192, 255, 256, int(rng.integers(129, 257))]))   # noqa: E731
        which = int(rng.integers(0, 3))
        H, C = (big() if which != 1 else H), (big() if which != 0 else C)
    if use_xw and rng.integers(0, 3) == 0:
        F = int(rng.choice([129, 256, 300, 602, int(rng.integers(129, 900))]))   # more than one k-quad of features
    if use_xw:
        # the X . W product's float32 sums stay exact - counted over WHOLE k-quads (PAD128(F): launch_common.hip.h::no_wrap)
        F = min(F, ((1 << 24) - 1) // ((1 << b) - 1) ** 2 // 128 * 128)
    count = int(rng.integers(1, 5))
    bitmaps = bool(rng.integers(0, 2))
    density = float(rng.choice([0.0, 0.01, 0.05, 0.5, 1.0]))
    W1, W2 = oracle.pack(rand_q(rng, F, H, b), b, True), oracle.pack(rand_q(rng, H, C, b), b, True)
    dW1, dW2 = torch.from_numpy(W1.view(np.int32)).cuda(), torch.from_numpy(W2.view(np.int32)).cuda()
    c1 = torch.full((int(lib.qgtc_weight_codes_words(F, H, b, 0)),), -1, dtype=torch.int32, device="cuda")
    c2 = torch.full((int(lib.qgtc_weight_codes_words(H, C, b, 1)),), -1, dtype=torch.int32, device="cuda")
    jobs = (QgtcExpandJob * 2)(QgtcExpandJob(dW1.data_ptr(), c1.data_ptr(), dW1.numel(), min(F, 128) if not use_xw else F, H, b, P128(H), 0 if use_xw else 1, c1.numel()),
                               QgtcExpandJob(dW2.data_ptr(), c2.data_ptr(), dW2.numel(), H, C, b, P128(C), 1, c2.numel()))
    assert lib.qgtc_expand_weights(ctypes.addressof(jobs), 2, st) == 0
    keep, sx, sa, sw, sf, want, ms, ks = [], [], [], [], [], [], [], []
    for i in range(count):
        m = int(rng.choice([1, 31, 32, 33, 127, 128, 129, 200, 257, int(rng.integers(1, 700))]))
        k = m if rng.integers(0, 2) else int(rng.integers(1, 700))          # K != M half the time
        ms.append(m)
        ks.append(k)
        qa = (rng.random((m, k)) < density).astype(np.int32)
        A = oracle.pack(qa, 1, False)
        dA = torch.from_numpy(A.view(np.int32)).cuda()
        T = torch.full((int(lib.qgtc_chain_words(k, H, b)),), -1, dtype=torch.int32, device="cuda")     # requant(X . W1), or the converted X'
        if use_xw:    # T = requant(X . W1): X [k, F] rows layout
            X = oracle.pack(rand_q(rng, k, F, b), b, False)
            dX = torch.from_numpy(X.view(np.int32)).cuda()
            sx.append(QgtcProblem(dX.data_ptr(), dW1.data_ptr(), T.data_ptr(), dX.numel(), dW1.numel(), k, F, H, P128(H), 0, None))
            t_o = oracle.bitmm2bit(X, W1, k, F, H, b, b, b, col=True)
        else:         # T = a data-loader operand [k, H] converted from the public cols layout
            t_o = oracle.pack(rand_q(rng, k, H, b), b, True)
            dX = torch.from_numpy(t_o.view(np.int32)).cuda()
            assert lib.qgtc_chain_from_cols(dX.data_ptr(), dX.numel(), k, H, b, T.data_ptr(), T.numel(), st) == 0
        T2 = torch.full((int(lib.qgtc_chain_words(m, C, b)),), -1, dtype=torch.int32, device="cuda")
        out = torch.full((m * C,), -7.0, dtype=torch.float32, device="cuda")
        occ = None
        if bitmaps:
            occ = torch.empty(int(lib.qgtc_occupancy_words(m, k)), dtype=torch.int64, device="cuda")
            assert lib.qgtc_tile_occupancy(dA.data_ptr(), dA.numel(), m, k, 1, occ.data_ptr(), occ.numel(), st) == 0
        keep += [dA, dX, T, T2, out, occ]
        ow = (((k + 127) // 128) + 63) // 64
        if tiles:   # [32-row block][k-quad][32 rows][4 words], zeros past the packed rows
            kq, rbs = (k + 127) // 128, (m + 31) // 32
            assert lib.qgtc_adj_tiles_words(m, k) == rbs * kq * 128
            dT = torch.full((rbs * kq * 128,), -1, dtype=torch.int32, device="cuda")
            assert lib.qgtc_adj_tiles_from_rows(dA.data_ptr(), dA.numel(), m, k, dT.data_ptr(), dT.numel() - 1, st) == 2      # undersized
            assert lib.qgtc_adj_tiles_from_rows(dA.data_ptr(), dA.numel(), m, k, dT.data_ptr(), dT.numel(), st) == 0
            rows = np.zeros((rbs * 32, kq, 4), dtype=np.uint32)
            rows[:A.size // (kq * 4)] = A.reshape(-1, kq, 4)
            np.testing.assert_array_equal(dT.cpu().numpy().view(np.uint32).reshape(rbs, kq, 32, 4), rows.reshape(rbs, 32, kq, 4).transpose(0, 2, 1, 3))
            keep.append(dT)
            dA = dT
        sa.append(QgtcProblem(dA.data_ptr(), T.data_ptr(), None, dA.numel(), T.numel(), m, k, H, P128(H), ow if bitmaps else 0, occ.data_ptr() if bitmaps else None))
        sf.append(QgtcProblem(None, dW2.data_ptr(), out.data_ptr(), 0, dW2.numel(), m, H, C, P128(C), 0, None))
        h_o = oracle.bitmm2bit(A, t_o, m, k, H, 1, b, b)
        want.append((out, oracle.bitmm2int(h_o, W2, m, H, C, b, b, True)))
    host = (QgtcProblem * (3 * count))(*((sx if use_xw else sa) + sa + sf))
    descs = torch.frombuffer(bytearray(bytes(host)), dtype=torch.uint8).cuda()
    d = lambda i: descs.data_ptr() + 72 * count * i       # noqa: E731
    if use_xw:
        rc = lib.qgtc_chain_transform(d(0), count, max(ks), F, H, b, b, c1.data_ptr(), 0x200, st)
        assert rc == 0, lib.qgtc_strerror(rc)
    rc = lib.qgtc_chain_aggregate(d(1), d(2), count, max(ms), max(ks), H, C, b, b, b, 2, c2.data_ptr(), 0x200 | (0x400 if tiles else 0), st)
    assert rc == 0, lib.qgtc_strerror(rc)
    torch.cuda.synchronize()
    for i, (o, w) in enumerate(want):
        np.testing.assert_array_equal(o.cpu().numpy().reshape(w.shape), w, err_msg=f"seed {seed} b={b} xw={use_xw} F={F} H={H} C={C} m={ms[i]} k={ks[i]} density={density} bitmaps={bitmaps}")
    lib.qgtc_last_batched_violation.argtypes = [ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), vp]
    assert lib.qgtc_last_batched_violation(None, None, st) == 0


@pytest.mark.parametrize("t_bits,act_bits,H", [(8, 6, 64), (5, 8, 128), (7, 5, 33), (1, 2, 200), (4, 3, 256), (3, 4, 64), (2, 1, 100)])
def test_chain_aggregate_with_different_widths_inside_a_format_class(lib, oracle, t_bits, act_bits, H):
    """qgtc_chain_aggregate takes T's width (t_bits) and the aggregate's (act_bits = the planes of W') separately; they only have to
    share a format class (1 / 2 bits, 3 / 4 bits, 5 .. 8 bits). T from the public cols layout (qgtc_chain_from_cols), then
    out = float32(requant(A . T) . W') (out_mode 2) against the oracle - the 5 .. 8-bit kernels skip zero top digits per operand."""
    import torch
    lib.qgtc_weight_codes_words.restype = lib.qgtc_chain_words.restype = lib.qgtc_occupancy_words.restype = ctypes.c_size_t
    lib.qgtc_expand_weights.argtypes = [vp, ctypes.c_int, vp]
    lib.qgtc_chain_from_cols.argtypes = [vp, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, ctypes.c_size_t, vp]
    lib.qgtc_chain_aggregate.argtypes = [vp, vp, ctypes.c_int] + [ctypes.c_int] * 8 + [vp, ctypes.c_uint, vp]
    rng = np.random.default_rng(100 * t_bits + act_bits)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    P128 = lambda x: (x + 127) // 128 * 128   # noqa: E731
    m, k, C = 333, 270, 40
    W2 = oracle.pack(rand_q(rng, H, C, act_bits), act_bits, True)
    dW2 = torch.from_numpy(W2.view(np.int32)).cuda()
    c2 = torch.full((int(lib.qgtc_weight_codes_words(H, C, act_bits, 1)),), -1, dtype=torch.int32, device="cuda")
    jobs = (QgtcExpandJob * 1)(QgtcExpandJob(dW2.data_ptr(), c2.data_ptr(), dW2.numel(), H, C, act_bits, P128(C), 1, c2.numel()))
    assert lib.qgtc_expand_weights(ctypes.addressof(jobs), 1, st) == 0
    qa = (rng.random((m, k)) < 0.05).astype(np.int32)
    A = oracle.pack(qa, 1, False)
    t_o = oracle.pack(rand_q(rng, k, H, t_bits), t_bits, True)
    dA, dT0 = torch.from_numpy(A.view(np.int32)).cuda(), torch.from_numpy(t_o.view(np.int32)).cuda()
    T = torch.full((int(lib.qgtc_chain_words(k, H, t_bits)),), -1, dtype=torch.int32, device="cuda")
    assert lib.qgtc_chain_from_cols(dT0.data_ptr(), dT0.numel(), k, H, t_bits, T.data_ptr(), T.numel(), st) == 0
    out = torch.full((m * C,), -7.0, dtype=torch.float32, device="cuda")
    host = (QgtcProblem * 2)(QgtcProblem(dA.data_ptr(), T.data_ptr(), None, dA.numel(), T.numel(), m, k, H, P128(H), 0, None),
                             QgtcProblem(None, dW2.data_ptr(), out.data_ptr(), 0, dW2.numel(), m, H, C, P128(C), 0, None))
    descs = torch.frombuffer(bytearray(bytes(host)), dtype=torch.uint8).cuda()
    rc = lib.qgtc_chain_aggregate(descs.data_ptr(), descs.data_ptr() + 72, 1, m, k, H, C, t_bits, act_bits, act_bits, 2, c2.data_ptr(), 0x200, st)
    assert rc == 0, lib.qgtc_strerror(rc)
    torch.cuda.synchronize()
    h_o = oracle.bitmm2bit(A, t_o, m, k, H, 1, t_bits, act_bits)
    np.testing.assert_array_equal(out.cpu().numpy().reshape(m, C), oracle.bitmm2int(h_o, W2, m, H, C, act_bits, act_bits, True))


@pytest.mark.parametrize("b", [1, 2, 3, 4, 5, 6, 7, 8])
def test_chain_aggregate_requantises_at_the_clamp_boundary(lib, oracle, b):
    """kernel.h:31-37,350 on the chain entries: a sum c > 2^b becomes 2^b - 1, c == 2^b SURVIVES and packs as 0 (only the low b bits are
    kept), c < 2^b is itself. Sums of exactly 2^b - 1, 2^b, 2^b + 1 and 3 x 2^b are built from two / three adjacency bits on known T
    values and read back through an identity W' (out_mode 2): float32(requant(A . T) . I) - against the oracle and the closed form.
    8 bits is the case a saturating byte conversion gets wrong (256 -> 255 instead of 0)."""
    import torch
    lib.qgtc_weight_codes_words.restype = lib.qgtc_chain_words.restype = ctypes.c_size_t
    lib.qgtc_expand_weights.argtypes = [vp, ctypes.c_int, vp]
    lib.qgtc_chain_from_cols.argtypes = [vp, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, ctypes.c_size_t, vp]
    lib.qgtc_chain_aggregate.argtypes = [vp, vp, ctypes.c_int] + [ctypes.c_int] * 8 + [vp, ctypes.c_uint, vp]
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    P128 = lambda x: (x + 127) // 128 * 128   # noqa: E731
    m = k = 70
    H = C = 40
    top = (1 << b) - 1
    qt = np.zeros((k, H), dtype=np.int32)
    qa = np.zeros((m, k), dtype=np.int32)
    # column c of T: rows 0, 1, 2 hold values whose partial sums hit the boundary cases on rows 0 .. 3 of A
    qt[0, :] = top                    # row 0 of A = {0}            -> 2^b - 1
    qt[1, :] = 1                      # row 1 of A = {0, 1}         -> 2^b      (packs as 0)
    qt[2, :] = 1                      # row 2 of A = {0, 1, 2}      -> 2^b + 1  (clamps to 2^b - 1)
    qt[3:6, :] = top                  # row 3 of A = {0, 3, 4, 5}   -> 4 (2^b - 1) > 2^b for b >= 2 (b = 1: 4 > 2 as well)
    qa[0, 0] = 1
    qa[1, :2] = 1
    qa[2, :3] = 1
    qa[3, [0, 3, 4, 5]] = 1
    qw = np.eye(H, C, dtype=np.int32)
    A, T0, W2 = oracle.pack(qa, 1, False), oracle.pack(qt, b, True), oracle.pack(qw, b, True)
    dA, dT0, dW2 = (torch.from_numpy(t.view(np.int32)).cuda() for t in (A, T0, W2))
    c2 = torch.full((int(lib.qgtc_weight_codes_words(H, C, b, 1)),), -1, dtype=torch.int32, device="cuda")
    jobs = (QgtcExpandJob * 1)(QgtcExpandJob(dW2.data_ptr(), c2.data_ptr(), dW2.numel(), H, C, b, P128(C), 1, c2.numel()))
    assert lib.qgtc_expand_weights(ctypes.addressof(jobs), 1, st) == 0
    T = torch.full((int(lib.qgtc_chain_words(k, H, b)),), -1, dtype=torch.int32, device="cuda")
    assert lib.qgtc_chain_from_cols(dT0.data_ptr(), dT0.numel(), k, H, b, T.data_ptr(), T.numel(), st) == 0
    out = torch.full((m * C,), -7.0, dtype=torch.float32, device="cuda")
    host = (QgtcProblem * 2)(QgtcProblem(dA.data_ptr(), T.data_ptr(), None, dA.numel(), T.numel(), m, k, H, P128(H), 0, None),
                             QgtcProblem(None, dW2.data_ptr(), out.data_ptr(), 0, dW2.numel(), m, H, C, P128(C), 0, None))
    descs = torch.frombuffer(bytearray(bytes(host)), dtype=torch.uint8).cuda()
    rc = lib.qgtc_chain_aggregate(descs.data_ptr(), descs.data_ptr() + 72, 1, m, k, H, C, b, b, b, 2, c2.data_ptr(), 0x200, st)
    assert rc == 0, lib.qgtc_strerror(rc)
    torch.cuda.synchronize()
    got = out.cpu().numpy().reshape(m, C)
    h_o = oracle.bitmm2bit(A, T0, m, k, H, 1, b, b)
    np.testing.assert_array_equal(got, oracle.bitmm2int(h_o, W2, m, H, C, b, b, True))
    want_rows = [top, 0, top, top] if b > 1 else [1, 0, 1, 1]      # (b = 1: 2^b - 1 = 1; rows 2, 3 exceed 2 and clamp to 1)
    for r, wv in enumerate(want_rows):
        assert (got[r] == wv).all(), (b, r, got[r][:4], wv)
    assert (got[4:] == 0).all()
