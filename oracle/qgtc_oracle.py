"""Python face of the CPU oracle (TEST INFRASTRUCTURE ONLY — see oracle/qgtc_oracle.h).

Two independent restatements of the reference semantics live here:

* ``Oracle``  — ctypes binding of ``oracle/libqgtc_oracle.so`` (plain C, qgtc_oracle.c);
* ``np_*``    — closed-form NumPy versions, written separately so that
                the C code can be cross-checked against a second implementation on CPU.

Only tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module. PARITY STATUS: pinned only by the unitest.py-derived known answers and the warp-level
emulation (oracle/warp_emulation.py); otherwise "parity unpinned" — the reference records no
expected outputs and cannot be compiled or imported in this image.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = [os.path.join(_HERE, "qgtc_oracle.c"), os.path.join(_HERE, "qgtc_oracle.h")]


def _stale(lib: str) -> bool:
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    return any(os.path.getmtime(s) > t for s in _SRC)


def build(native: bool = False, out_dir: str | None = None, force: bool = False) -> str:
    """Compile the C oracle with gcc (recipe: oracle/Makefile). Returns the .so path."""
    out_dir = out_dir or _HERE
    name = "libqgtc_oracle_native.so" if native else "libqgtc_oracle.so"
    lib = os.path.join(out_dir, name)
    if force or _stale(lib):
        target = "native" if native else "all"
        subprocess.run(["make", "-s", "-C", _HERE, target, f"OUT={os.path.abspath(out_dir)}"],
                       check=True)
    return lib


def S8(x):
    return (x + 7) >> 3


def S128(x):
    return (x + 127) >> 7


def P8(x):
    return S8(x) << 3


def P128(x):
    return S128(x) << 7


_u32p = ctypes.POINTER(ctypes.c_uint32)
_i32p = ctypes.POINTER(ctypes.c_int32)
_f32p = ctypes.POINTER(ctypes.c_float)


def _p(a, t):
    return a.ctypes.data_as(t)


class Oracle:
    """ctypes wrapper; all arrays are NumPy, packed tensors are flat uint32."""

    def __init__(self, native: bool = False, out_dir: str | None = None):
        path = build(native=native, out_dir=out_dir)
        self.path = path
        L = ctypes.CDLL(path)
        self.L = L
        sz, i, u64p = ctypes.c_size_t, ctypes.c_int, ctypes.POINTER(ctypes.c_uint64)
        L.qo_rows_words.restype = sz
        L.qo_rows_words.argtypes = [i, i, i]
        L.qo_cols_words.restype = sz
        L.qo_cols_words.argtypes = [i, i, i, i]
        L.qo_quantize.argtypes = [_f32p, sz, i, _i32p]
        L.qo_pack_rows.argtypes = [_i32p, i, i, i, _u32p]
        L.qo_pack_cols.argtypes = [_i32p, i, i, i, i, _u32p]
        L.qo_val2bit.argtypes = [_f32p, i, i, i, i, i, _u32p]
        L.qo_bit2val.argtypes = [_u32p, i, i, i, i, i, _i32p]
        L.qo_acc.argtypes = [_u32p, sz, _u32p, sz, i, i, i, i, i, i, _i32p]
        L.qo_requant.restype = ctypes.c_int32
        L.qo_requant.argtypes = [ctypes.c_int32, i]
        for f in (L.qo_bitmm2bit, L.qo_bitmm2bit_col):
            f.argtypes = [_u32p, sz, _u32p, sz, i, i, i, i, i, i, _u32p]
        L.qo_bitmm2int.argtypes = [_u32p, sz, _u32p, sz, i, i, i, i, i, i, _f32p]
        L.qo_tile_counters.argtypes = [_u32p, sz, i, i, i, i, i, u64p, u64p]
        L.qo_num_threads.restype = i
        # A checker's products are small and many: on a 256-thread host an OpenMP team of every hardware thread spends its time in the
        # region barriers (a 32-bit Batched-GIN chain on the tiny test graph: 7 s with the 256-thread default this library gets when it
        # is loaded BEFORE torch - which would have capped the process at the physical cores - against 0.4 s). bench.py's cpu_baseline
        # probes its own thread count.
        if not os.environ.get("OMP_NUM_THREADS") and self.num_threads() > 32:
            self.set_num_threads(32)

    # -- sizes ---------------------------------------------------------------------------
    def rows_words(self, H, W, b):
        return int(self.L.qo_rows_words(H, W, b))

    def cols_words(self, H, W, b, output_layer=False):
        return int(self.L.qo_cols_words(H, W, b, int(output_layer)))

    def num_threads(self):
        return int(self.L.qo_num_threads())

    def set_num_threads(self, n):
        self.L.qo_set_num_threads(int(n))

    # -- ops -----------------------------------------------------------------------------
    def quantize(self, x, nbits):
        x = np.ascontiguousarray(x, dtype=np.float32)
        q = np.empty(x.shape, dtype=np.int32)
        self.L.qo_quantize(_p(x, _f32p), x.size, nbits, _p(q, _i32p))
        return q

    def val2bit(self, x, nbits, col_major=False, output_layer=False):
        x = np.ascontiguousarray(x, dtype=np.float32)
        H, W = x.shape
        n = self.cols_words(H, W, nbits, output_layer) if col_major else self.rows_words(H, W, nbits)
        out = np.empty(n, dtype=np.uint32)
        self.L.qo_val2bit(_p(x, _f32p), H, W, nbits, int(col_major), int(output_layer),
                          _p(out, _u32p))
        return out

    def pack(self, q, nbits, col_major=False, output_layer=False):
        q = np.ascontiguousarray(q, dtype=np.int32)
        H, W = q.shape
        if col_major:
            out = np.empty(self.cols_words(H, W, nbits, output_layer), dtype=np.uint32)
            self.L.qo_pack_cols(_p(q, _i32p), H, W, nbits, int(output_layer), _p(out, _u32p))
        else:
            out = np.empty(self.rows_words(H, W, nbits), dtype=np.uint32)
            self.L.qo_pack_rows(_p(q, _i32p), H, W, nbits, _p(out, _u32p))
        return out

    def bit2val(self, bits, nbits, H, W, col_major=False, output_layer=False):
        bits = np.ascontiguousarray(bits).view(np.uint32).reshape(-1)
        need = self.cols_words(H, W, nbits, output_layer) if col_major else self.rows_words(H, W, nbits)
        assert bits.size >= need, (bits.size, need)
        out = np.empty((H, W), dtype=np.int32)
        self.L.qo_bit2val(_p(bits, _u32p), nbits, H, W, int(col_major), int(output_layer),
                          _p(out, _i32p))
        return out

    def acc(self, X, Wt, M, K, N, a, w, w_lines=None):
        X = np.ascontiguousarray(X).view(np.uint32).reshape(-1)
        Wt = np.ascontiguousarray(Wt).view(np.uint32).reshape(-1)
        out = np.empty((M, N), dtype=np.int32)
        self.L.qo_acc(_p(X, _u32p), X.size, _p(Wt, _u32p), Wt.size, M, K, N, a, w,
                      P128(N) if w_lines is None else w_lines, _p(out, _i32p))
        return out

    def requant(self, c, ob):
        return int(self.L.qo_requant(int(c), ob))

    def bitmm2bit(self, X, Wt, M, K, N, a, w, ob, col=False):
        X = np.ascontiguousarray(X).view(np.uint32).reshape(-1)
        Wt = np.ascontiguousarray(Wt).view(np.uint32).reshape(-1)
        n = self.cols_words(M, N, ob) if col else self.rows_words(M, N, ob)
        out = np.empty(n, dtype=np.uint32)
        f = self.L.qo_bitmm2bit_col if col else self.L.qo_bitmm2bit
        f(_p(X, _u32p), X.size, _p(Wt, _u32p), Wt.size, M, K, N, a, w, ob, _p(out, _u32p))
        return out

    def bitmm2int(self, X, Wt, M, K, N, a, w, pad_128=False):
        X = np.ascontiguousarray(X).view(np.uint32).reshape(-1)
        Wt = np.ascontiguousarray(Wt).view(np.uint32).reshape(-1)
        out = np.empty((M, N), dtype=np.float32)
        self.L.qo_bitmm2int(_p(X, _u32p), X.size, _p(Wt, _u32p), Wt.size, M, K, N, a, w,
                            int(pad_128), _p(out, _f32p))
        return out

    def tile_counters(self, X, M, K, N, a, w):
        X = np.ascontiguousarray(X).view(np.uint32).reshape(-1)
        t, z = ctypes.c_uint64(0), ctypes.c_uint64(0)
        self.L.qo_tile_counters(_p(X, _u32p), X.size, M, K, N, a, w, ctypes.byref(t),
                                ctypes.byref(z))
        return int(t.value), int(z.value)


# ======================================================================================
# Independent NumPy restatement (closed forms). Deliberately shares no
# code with the C oracle so that the two can check each other.
# ======================================================================================
def np_quantize(x, nbits):
    """kernel.h:39-44 clip + :68 __float2int_rn (round-half-even, NaN -> 0)."""
    x = np.asarray(x, dtype=np.float32)
    ub = np.float32(2.0 ** nbits)
    y = np.where(x < 0, np.float32(1.0), np.where(x > ub, ub - np.float32(1.0), x))
    y = np.where(np.isnan(y), np.float32(0.0), y)
    r = np.rint(y.astype(np.float64))  # np.rint is round-half-to-even
    return (r.astype(np.int64) & 0xFFFFFFFF).astype(np.uint32).view(np.int32)


def _pack_lines(bitsmat):
    """bitsmat: uint8 [L, n] of 0/1 -> uint32 [L, ceil(n/32)], element i at bit 31-(i&31)."""
    L, n = bitsmat.shape
    nw = (n + 31) // 32
    padded = np.zeros((L, nw * 32), dtype=np.uint8)
    padded[:, :n] = bitsmat
    # np.packbits is MSB-first within bytes; 4 big-endian bytes -> one word
    by = np.packbits(padded, axis=1).reshape(L, nw, 4).astype(np.uint32)
    return (by[:, :, 0] << 24) | (by[:, :, 1] << 16) | (by[:, :, 2] << 8) | by[:, :, 3]


def np_pack_rows(q, nbits):
    """kernel.h:204-242 -> uint32 [nbits, P8(H), S128(W)*4]."""
    q = np.asarray(q).astype(np.int64) & 0xFFFFFFFF
    H, W = q.shape
    out = np.zeros((nbits, P8(H), S128(W) * 4), dtype=np.uint32)
    for p in range(nbits):
        words = _pack_lines(((q >> p) & 1).astype(np.uint8))
        out[p, :H, :words.shape[1]] = words
    return out


def np_pack_cols(q, nbits, output_layer=False):
    """kernel.h:75-106 -> uint32 [nbits, P128(W) (P8(W) if output_layer), S128(H)*4]."""
    q = np.asarray(q).astype(np.int64) & 0xFFFFFFFF
    H, W = q.shape
    lines = P8(W) if output_layer else P128(W)
    out = np.zeros((nbits, lines, S128(H) * 4), dtype=np.uint32)
    for p in range(nbits):
        words = _pack_lines(((q.T >> p) & 1).astype(np.uint8))
        out[p, :W, :words.shape[1]] = words
    return out


def _unpack_lines(words, n):
    L = words.shape[0]
    by = np.stack([(words >> 24) & 0xFF, (words >> 16) & 0xFF, (words >> 8) & 0xFF, words & 0xFF],
                  axis=-1).astype(np.uint8).reshape(L, -1)
    return np.unpackbits(by, axis=1)[:, :n]


def np_unpack_rows(bits, nbits, H, W):
    b = np.asarray(bits).view(np.uint32).reshape(nbits, P8(H), S128(W) * 4)
    v = np.zeros((H, W), dtype=np.int64)
    for p in range(nbits):
        v += _unpack_lines(b[p, :H], W).astype(np.int64) << p
    return (v & 0xFFFFFFFF).astype(np.uint32).view(np.int32)


def np_unpack_cols(bits, nbits, H, W, output_layer=False):
    lines = P8(W) if output_layer else P128(W)
    b = np.asarray(bits).view(np.uint32).reshape(nbits, lines, S128(H) * 4)
    v = np.zeros((H, W), dtype=np.int64)
    for p in range(nbits):
        v += _unpack_lines(b[p, :W], H).T.astype(np.int64) << p
    return (v & 0xFFFFFFFF).astype(np.uint32).view(np.int32)


def np_acc_from_values(QX, QW, a, w):
    """Integer product of the quantised matrices restricted to their low a / w bits — what
    kernel.h:292-341 computes for well-formed operands (wrap to int32)."""
    QX = (np.asarray(QX).astype(np.int64) & 0xFFFFFFFF) & ((1 << a) - 1)
    QW = (np.asarray(QW).astype(np.int64) & 0xFFFFFFFF) & ((1 << w) - 1)
    c = QX @ QW
    return (c & 0xFFFFFFFF).astype(np.uint32).view(np.int32)


def np_requant(c, ob):
    """kernel.h:31-37 as called at :350 (float compare, keep c == 2^ob, negatives -> 1)."""
    c = np.asarray(c, dtype=np.int32)
    val = c.astype(np.float32)
    maxv = np.float32(2.0 ** ob)
    val = np.where(val > maxv, maxv - np.float32(1.0), val)
    val = np.where(val < 0, np.float32(1.0), val)
    return np.trunc(val.astype(np.float64)).astype(np.int64).clip(max=2 ** 31 - 1).astype(np.int32)


# ======================================================================================
# Restatements of the paths beside the bit-GEMM (test infrastructure, like everything here)
# ======================================================================================
def np_dense_adjacency(row, col, H, W):
    """The dense float matrix the reference builds from a batch's edge list:
    torch.sparse.FloatTensor(i, v, [n, n]).to_dense() with v = 1 sums duplicate edges
    (sampler.py:80-89)."""
    A = np.zeros((H, W), dtype=np.float32)
    np.add.at(A, (np.asarray(row, dtype=np.int64), np.asarray(col, dtype=np.int64)), 1.0)
    return A


def np_pack_edges(row, col, H, W, nbits):
    """What QGTC.val2bit(dense adjacency, nbits, False, False) yields (sampler.py:98/101), computed
    from the edge list: cell value = multiplicity, quantised by Quantize_val (kernel.h:39-44,49-71),
    packed in the rows layout (kernel.h:204-242)."""
    return np_pack_rows(np_quantize(np_dense_adjacency(row, col, H, W), nbits), nbits).reshape(-1)


def np_tile_occupancy(X_words, M, K, a, tile_rows=32):
    """Occupancy bitmap of a rows-layout operand with `a` planes: bit q of word [tile][q // 64] is
    set when rows tile_rows*tile .. +tile_rows-1, packed words 4q .. 4q+3 hold a set bit in any
    plane. This is the build's own zero-tile unit (32 rows x 128 bits); the reference's counter
    kernel tests 8 x 128-bit tiles (kernel.h:574-592) - same idea, same 128-bit k step."""
    kq = S128(K)
    planes = np.ascontiguousarray(X_words).view(np.uint32).reshape(a, P8(M), kq * 4)
    tiles, ow = (M + tile_rows - 1) // tile_rows, (kq + 63) // 64
    occ = np.zeros((tiles, ow), dtype=np.uint64)
    for t in range(tiles):
        blk = planes[:, tile_rows * t:min(tile_rows * (t + 1), M), :].reshape(a, -1, kq, 4)
        for q in np.nonzero((blk != 0).any(axis=(0, 1, 3)))[0]:
            occ[t, q // 64] |= np.uint64(1) << np.uint64(q % 64)
    return occ.reshape(-1)


def np_i8gemm(A, Bt):
    """The INT8 comparison GEMM (cuBLASGemmEX/cublas_main.cu:123-172: CUDA_R_8I operands,
    CUDA_R_32F result): C[M,N] = A[M,K] x B[K,N] with B given as Bt[N,K]; exact integer sum,
    then the int32 -> float32 conversion the device epilogue performs."""
    acc = np.asarray(A, dtype=np.int64) @ np.asarray(Bt, dtype=np.int64).T
    return acc.astype(np.int32).astype(np.float32)
