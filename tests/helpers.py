"""Shared input generators for the parity tests, and the ctypes mirrors of include/qgtc.h's structs (what a cgo / JNI / plain-C host would
declare) for the tests that drive the C-ABI with raw device pointers."""
import contextlib
import ctypes

import numpy as np

# every engine of the library; "auto" is what an unmodified caller gets (qgtc_torch.cpp: g_engine)
ENGINES = ("popcount", "mfma", "auto")
DEFAULT_ENGINE = "auto"


@contextlib.contextmanager
def use_engine(qgtc, name):
    """Run a block on one engine and put the previous one back."""
    prev = qgtc.get_engine()
    qgtc.set_engine(name)
    try:
        yield
    finally:
        qgtc.set_engine(prev)


def edge_floats(rng, H, W, nbits):
    """Random floats around the quantiser's range plus every edge case of kernel.h:39-44,68."""
    hi = float(2 ** nbits)
    x = rng.uniform(-1.0, hi + 1.5, size=(H, W)).astype(np.float32)
    specials = [np.nan, -0.0, 0.0, 0.5, 1.5, 2.5, 0.51, hi, hi - 0.5, hi + 1e-3, -1e-3, 100.0 * hi,
                np.inf, -np.inf, hi - 1.0, 3.5]
    flat = x.reshape(-1)
    for i, s in enumerate(specials):
        if i < flat.size:
            flat[(i * 7919) % flat.size] = s
    return x


def rand_q(rng, H, W, nbits, density=None):
    q = rng.integers(0, 2 ** nbits, size=(H, W), dtype=np.int64)
    if density is not None:
        q = q * (rng.random((H, W)) < density)
    return q.astype(np.int32)


def to_dev(torch, arr_u32, shape, device="cuda"):
    """flat uint32 numpy -> int32 torch tensor with the reference's reported shape."""
    return torch.from_numpy(np.ascontiguousarray(arr_u32).view(np.int32).reshape(shape)).to(device)


def to_np_u32(t):
    return t.detach().cpu().numpy().view(np.uint32).reshape(-1)


# ---------------------------------------------------------------------------------------------
# oracle-side restatement of the driver's six-operator chains (qgtc_ppopp22_amd/driver.py), used to
# check whole cluster batches
# ---------------------------------------------------------------------------------------------
def oracle_batch_inputs(oracle, graph, par_li, cid, psize, batch_size, bits):
    from qgtc_ppopp22_amd import graph as G

    nodes = G.batch_nodes(par_li, cid, psize, batch_size)
    row, col = G.induced_edges(graph, nodes)
    n = nodes.size
    A = np.zeros((n, n), dtype=np.float32)
    np.add.at(A, (row, col), 1.0)
    X = graph.feat[nodes]
    return {"n": n, "F": X.shape[1], "A": A, "X": X,
            "bit_A": oracle.val2bit(A, 1, False, False),
            "bit_X": oracle.val2bit(X, bits, True, False),
            "bit_X_rows": oracle.val2bit(X, bits, False, False)}


def oracle_weights(oracle, feat, hidden, classes, bits):
    one = lambda h, w: np.ones((h, w), dtype=np.float32)  # noqa: E731
    return {"W1": oracle.val2bit(one(feat, hidden), bits, True, False),
            "W2": oracle.val2bit(one(hidden, hidden), bits, True, False),
            "W3": oracle.val2bit(one(hidden, classes), bits, True, True),
            "W3h": oracle.val2bit(one(hidden, classes), bits, True, False),
            "hidden": hidden, "classes": classes}


def oracle_chain(oracle, bi, W, b, chain, gin):
    """Returns the list of the six per-operator results (packed uint32 / float32)."""
    n, F, H, C = bi["n"], bi["F"], W["hidden"], W["classes"]
    mm, mmc, mi = oracle.bitmm2bit, (lambda *a: oracle.bitmm2bit(*a, col=True)), oracle.bitmm2int
    A, Xc, Xr = bi["bit_A"], bi["bit_X"], bi["bit_X_rows"]
    if chain == "reference" and not gin:      # main_qgtc.py:147-154
        t0 = mm(Xc, W["W1"], n, F, H, b, b, b)
        t1 = mm(A, t0, n, n, H, 1, b, b)
        t2 = mm(t1, W["W2"], n, H, H, b, b, b)
        t3 = mm(A, t2, n, n, H, 1, b, b)
        t4 = mm(t3, W["W3"], n, H, C, b, b, b)
        return [t0, t1, t2, t3, t4, mi(A, t4, n, n, H, 1, b, False)]
    if chain == "reference":                  # main_qgtc.py:131-138
        t0 = mm(A, Xc, n, n, F, 1, b, b)
        t1 = mm(t0, W["W1"], n, F, H, b, b, b)
        t2 = mm(A, t1, n, n, H, 1, b, b)
        t3 = mm(t2, W["W2"], n, H, H, b, b, b)
        t4 = mm(A, t3, n, n, H, 1, b, b)
        return [t0, t1, t2, t3, t4, mi(t4, W["W3"], n, H, C, b, b, False)]
    if not gin:
        xw = mmc(Xr, W["W1"], n, F, H, b, b, b)
        h1 = mm(A, xw, n, n, H, 1, b, b)
        hw = mmc(h1, W["W2"], n, H, H, b, b, b)
        h2 = mm(A, hw, n, n, H, 1, b, b)
        hw3 = mmc(h2, W["W3h"], n, H, C, b, b, b)
        return [xw, h1, hw, h2, hw3, mi(A, hw3, n, n, C, 1, b, True)]
    ax = mm(A, Xc, n, n, F, 1, b, b)
    h1 = mmc(ax, W["W1"], n, F, H, b, b, b)
    a1 = mm(A, h1, n, n, H, 1, b, b)
    h2 = mmc(a1, W["W2"], n, H, H, b, b, b)
    a2 = mm(A, h2, n, n, H, 1, b, b)
    return [ax, h1, a1, h2, a2, mi(a2, W["W3"], n, H, C, b, b, False)]


def integer_gcn_reference(A, X, H, C, b, oracle):
    """What the layout-correct GCN chain means in plain integer arithmetic:
    clamp is kernel.h:31-37's rule, all-ones weights."""
    from oracle.qgtc_oracle import np_quantize, np_requant

    mask = (1 << b) - 1
    qa = np_quantize(A, 1).astype(np.int64) & 1
    qx = np_quantize(X, b).astype(np.int64) & mask
    one = lambda h, w: np.ones((h, w), dtype=np.int64)  # noqa: E731
    cl = lambda c: np_requant(c.astype(np.int32), b).astype(np.int64) & mask  # noqa: E731
    h = cl(qa @ cl(qx @ one(X.shape[1], H)))
    h = cl(qa @ cl(h @ one(H, H)))
    return (qa @ cl(h @ one(H, C))).astype(np.float32)


# ---------------------------------------------------------------------------------------------
# include/qgtc.h through ctypes
# ---------------------------------------------------------------------------------------------
class QgtcProblem(ctypes.Structure):
    """struct qgtc_problem"""
    _fields_ = [("X", ctypes.c_void_p), ("W", ctypes.c_void_p), ("out", ctypes.c_void_p), ("x_words", ctypes.c_uint64),
                ("w_words", ctypes.c_uint64), ("M", ctypes.c_int32), ("K", ctypes.c_int32), ("N", ctypes.c_int32),
                ("w_lines", ctypes.c_int32), ("occ_words", ctypes.c_int32), ("occ", ctypes.c_void_p)]


class QgtcOperand(ctypes.Structure):
    _fields_ = [("ptr", ctypes.c_void_p), ("words", ctypes.c_uint64)]


class QgtcBatch(ctypes.Structure):
    """struct qgtc_batch - what the data loader knows of one cluster batch"""
    _fields_ = [("A", QgtcOperand), ("X", QgtcOperand), ("XR", QgtcOperand), ("XC", QgtcOperand), ("AT", QgtcOperand), ("occ", ctypes.c_void_p),
                ("n", ctypes.c_int32), ("occ_words", ctypes.c_int32)]


class QgtcStage(ctypes.Structure):
    _fields_ = [(k, ctypes.c_int32) for k in ("left", "right", "K", "N", "bit1", "bit2", "ob", "mode", "pad128", "use_occ", "fmt")]


class QgtcPackJob(ctypes.Structure):
    _fields_ = [("x", ctypes.c_void_p), ("out", ctypes.c_void_p), ("out_words", ctypes.c_uint64), ("H", ctypes.c_int32), ("W", ctypes.c_int32),
                ("nbits", ctypes.c_int32), ("col_major", ctypes.c_int32), ("output_layer", ctypes.c_int32), ("reserved", ctypes.c_int32)]


class QgtcExpandJob(ctypes.Structure):
    _fields_ = [("W", ctypes.c_void_p), ("codes", ctypes.c_void_p), ("w_words", ctypes.c_uint64), ("K", ctypes.c_int32), ("N", ctypes.c_int32),
                ("nbits", ctypes.c_int32), ("w_lines", ctypes.c_int32), ("order", ctypes.c_int32), ("codes_words", ctypes.c_uint32)]


class QgtcLoaderBatch(ctypes.Structure):
    """struct qgtc_loader_batch (88 bytes)"""
    _fields_ = [("edge_off", ctypes.c_uint64), ("n_edges", ctypes.c_uint64), ("feat_row", ctypes.c_uint64), ("n", ctypes.c_int32),
                ("reserved", ctypes.c_int32)] + [(k, ctypes.c_void_p) for k in ("A", "scratch", "AT", "occ", "X", "XR", "XC")]


ENGINE_FLAGS = {"popcount": 0x0, "mfma": 0x8, "auto": 0x10}


def c_library():
    """libqgtc_hip.so as ctypes sees it (no GPU needed to load it or to call its host-only entries)."""
    import qgtc_ppopp22_amd

    L = ctypes.CDLL(qgtc_ppopp22_amd.lib_path())
    L.qgtc_bitmm_route.restype = L.qgtc_bitmm_batched_route.restype = ctypes.c_char_p
    for f in (L.qgtc_rows_words, L.qgtc_cols_words, L.qgtc_adj_tiles_words, L.qgtc_chain_words, L.qgtc_weight_codes_words, L.qgtc_occupancy_words,
              L.qgtc_load_work_words):
        f.restype = ctypes.c_size_t
    return L


def kernel_behind(M, K, N, a=1, w=1, ob=1, mode=0, engine="auto"):
    """The kernel family a single launch of this shape takes (qgtc_bitmm_route: the rule functions the launchers switch on)."""
    return c_library().qgtc_bitmm_route(M, K, N, a, w, ob, mode, ENGINE_FLAGS[engine]).decode()
