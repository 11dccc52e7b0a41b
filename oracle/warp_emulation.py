"""Warp-level emulation of the reference kernels' data movement (TEST INFRASTRUCTURE ONLY).

The closed forms in qgtc_oracle.c / qgtc_oracle.py say *where each bit ends up*. This module
re-derives that independently by playing through what a 32-lane CUDA warp does in the reference
kernels — ballot, bit-reverse, lane-0 word stores, the 8x8x128 AND+popcount tile product and the
byte-granular epilogue stores — using the reference's own index arithmetic (cited per function),
so that the closed forms can be checked against the mechanism they summarise.

It is an emulator of observable behaviour (32 lanes, 32 warps per block, one block), written
for small shapes; it is not used by the product path.
"""
from __future__ import annotations

import numpy as np

from .qgtc_oracle import P8, P128, S8, S128

WARPS = 32  # config.h:4 warpPerBlock


def _ballot_brev(preds):
    """__brev(__ballot_sync(full, pred)): lane l's predicate lands in bit 31-l."""
    word = 0
    for lane, p in enumerate(preds):
        if p:
            word |= 1 << (31 - lane)
    return word


def emu_pack_rows(q, nbits):
    """QGTC_layer_input, kernel.h:204-242. Block = 8 rows x 128 cols; warp (lx=warp>>2,
    ly=warp&3) owns row bx*8+lx, columns by*128+ly*32+lane; lane 0 stores the word."""
    H, W = q.shape
    gdx, gdy = S8(H), S128(W)
    plane = P8(H) * S128(W) * 4                      # :214
    out = np.zeros(nbits * plane, dtype=np.uint32)
    for bid in range(gdx * gdy):
        bx, by = bid // gdy, bid % gdy               # :222-223
        for warp in range(WARPS):
            lx, ly = warp >> 2, warp & 3             # :217-218
            for p in range(nbits):
                preds = []
                for lane in range(32):
                    r, c = bx * 8 + lx, by * 128 + ly * 32 + lane
                    f0 = ((int(q[r, c]) >> p) & 1) if (c < W and r < H) else 0   # :228-229
                    preds.append(f0 > 0)
                out[p * plane + (bx * 8 + lx) * gdy * 4 + by * 4 + ly] = _ballot_brev(preds)  # :237
    return out


def emu_pack_cols(q, nbits):
    """PackFcWeight128, kernel.h:75-106. Block = 128 rows x 8 cols; warp (lx=warp&3,
    ly=warp>>2) owns column by*8+ly, rows bx*128+lx*32+lane."""
    H, W = q.shape
    gdx, gdy = S128(H), S8(W)
    plane = S128(H) * P128(W) * 4                    # :88
    out = np.zeros(nbits * plane, dtype=np.uint32)
    for bid in range(gdx * gdy):
        bx, by = bid % gdx, bid // gdx               # :92-93
        for warp in range(WARPS):
            lx, ly = warp & 3, warp >> 2             # :85-86
            for p in range(nbits):
                preds = []
                for lane in range(32):
                    r, c = bx * 128 + lx * 32 + lane, by * 8 + ly
                    f0 = float((int(q[r, c]) >> p) & 1) if (r < H and c < W) else -1.0  # :96-97
                    preds.append(f0 > 0)
                out[p * plane + (by * 8 + ly) * gdx * 4 + bx * 4 + lx] = _ballot_brev(preds)  # :101
    return out


def _popc128(words_a, words_b):
    return sum(bin(int(x) & int(y)).count("1") for x, y in zip(words_a, words_b))


def _requant_ref(c, ob):
    """quantize(), kernel.h:31-37 with max_val=1<<ob, min_val=0, evaluated in float32."""
    val = np.float32(np.int32(c))
    max_val, min_val = 1 << ob, 0
    if val > np.float32(max_val):
        val = np.float32(max_val - 1)
    if val < np.float32(min_val):
        val = np.float32(min_val + 1)
    ans = np.float32(np.float32(val - np.float32(min_val)) * np.float32(1 << ob)) / np.float32(max_val - min_val)
    return int(ans)


def emu_bitmm2bit(X, Wt, M, K, N, a, w, ob):
    """QGTC_layer_hidden, kernel.h:245-391, one warp per 8x8 output tile.

    X, Wt: flat uint32 arrays in the rows / cols layouts. Returns the flat packed output
    {ob*PAD8(M), STEP128(N)*4} (QGTC_device.cu:223) as uint32."""
    gdx, gdy, gdk, gdm = S8(M), S8(N), S128(K), S128(N)      # :271-274
    act_off = P8(M) * gdk * 4                                # :265
    w_off = gdk * P128(N) * 4                                # :266
    opt_off = P8(M) * gdm * 4                                # :267
    out_bytes = np.zeros(ob * opt_off * 4, dtype=np.uint8)   # Cb view of bit_X_out (:360)
    for bid in range(gdx * gdy):
        bx, by = bid // gdy, bid % gdy                       # :288-289
        c = np.zeros((8, 8), dtype=np.int64)
        for bit in range(a * w):
            b_act, b_w = bit % a, bit // a                   # :293-294
            tmp = np.zeros((8, 8), dtype=np.int64)
            for i in range(gdk):
                # load_matrix_sync(a_frag, bit_X + b_act*act_off + bx*8*gdk*4 + i*4, ldm=gdk*128 bits)
                # row r of the 8x128-bit tile starts r*gdk*4 words further on (:306-307)
                for r in range(8):
                    xa = b_act * act_off + (bx * 8 + r) * gdk * 4 + i * 4
                    for n in range(8):
                        wa = b_w * w_off + (by * 8 + n) * gdk * 4 + i * 4
                        tmp[r, n] += _popc128(X[xa:xa + 4], Wt[wa:wa + 4])   # :308 bmmaBitOpAND
            c += tmp << (b_act + b_w)                        # :340
        c = (c & 0xFFFFFFFF).astype(np.uint32).view(np.int32).astype(np.int64)
        Cs = [_requant_ref(v, ob) for v in c.reshape(-1)]    # :350, :354 row-major 8x8
        for p in range(ob):
            v0, v1 = [], []
            for lane in range(32):
                gy, gx = lane % 8, lane // 8                 # :363-364
                v0_in = (by * 8 + gy) < N and (bx * 8 + gx) < M        # :367
                v1_in = (by * 8 + gy) < N and (bx * 8 + gx + 4) < M    # :368
                v0.append(v0_in and ((Cs[lane] >> p) & 1) > 0)         # :371
                v1.append(v1_in and ((Cs[32 + lane] >> p) & 1) > 0)    # :372
            p0, p1 = _ballot_brev(v0), _ballot_brev(v1)      # :377-378
            # union{int; uin8[4]} on a little-endian machine: elements[k] = bits 8k..8k+7
            e0 = [(p0 >> (8 * k)) & 0xFF for k in range(4)]
            e1 = [(p1 >> (8 * k)) & 0xFF for k in range(4)]
            base = p * opt_off * 4
            for lane in range(4):                            # :384-388
                out_bytes[base + (bx * 8 + lane) * gdm * 16 + (by ^ 3)] = e0[3 - lane]
                out_bytes[base + (bx * 8 + 4 + lane) * gdm * 16 + (by ^ 3)] = e1[3 - lane]
    return out_bytes.view("<u4").copy()


def emu_bitmm2int(X, Wt, M, K, N, a, w, pad_128):
    """QGTC_layer_output_PAD8 / _PAD128, kernel.h:816-932 / :938-1054 -> float32 [M,N]."""
    gdx, gdy, gdk = S8(M), S8(N), S128(K)
    act_off = P8(M) * gdk * 4                                        # :835
    w_off = gdk * (P128(N) if pad_128 else P8(N)) * 4                # :958 / :836
    out = np.zeros((M, N), dtype=np.float32)
    for bid in range(gdx * gdy):
        bx, by = bid // gdy, bid % gdy
        c = np.zeros((8, 8), dtype=np.int64)
        for bit in range(a * w):
            b_act, b_w = bit % a, bit // a
            tmp = np.zeros((8, 8), dtype=np.int64)
            for i in range(gdk):
                for r in range(8):
                    xa = b_act * act_off + (bx * 8 + r) * gdk * 4 + i * 4
                    for n in range(8):
                        wa = b_w * w_off + (by * 8 + n) * gdk * 4 + i * 4
                        tmp[r, n] += _popc128(X[xa:xa + 4], Wt[wa:wa + 4])
            c += tmp << (b_act + b_w)                                # :900
        c = (c & 0xFFFFFFFF).astype(np.uint32).view(np.int32)
        for j in range(8):                                           # :917-929
            for lane in range(8):
                if bx * 8 + j < M and by * 8 + lane < N:
                    out[bx * 8 + j, by * 8 + lane] = np.float32(c[j, lane])
    return out
