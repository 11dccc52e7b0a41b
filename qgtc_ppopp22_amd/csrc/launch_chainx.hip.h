// launch_chainx.hip.h — part of libqgtc_hip.so (qgtc_chainx.hip): launchers of bitmm_fp4_rbx.hip.h.
#pragma once

// column blocks a kernel is instantiated for: 1, 2, 4, 8 (three run as four, five to seven as eight: lines past N are zeros, columns
// past N are not stored)
static inline int rbx_blocks(int N) {
    const int b = (N + 31) / 32;
    return b <= 2 ? b : (b <= 4 ? 4 : 8);
}

int qgtc_launch_rbx_xw(const qgtc_problem *prs, int count, int max_M, int K, int N, int a, int ob, const uint32_t *w_codes, hipStream_t st) {
    const int per = getenv_flag("QGTC_NO_XCD") ? 0 : 1;
    const dim3 grid(step128(max_M), count), block(256);
    const int gx = static_cast<int>(grid.x), gy = static_cast<int>(grid.y);
    const u32x4 *wc = reinterpret_cast<const u32x4 *>(w_codes);
    const int ncb = rbx_blocks(N), kq = step128(K), tb = weight_table_blocks(N);
    const int nd = ob <= 2 ? 1 : (ob <= 4 ? 2 : 4);   // digits of W (a chain has one width: W has the planes of T)
    if (ob < 1 || ob > 8 || a < 1 || a > 2 * nd) return QGTC_EINVAL;
#define QGTC_RBX_XW(NA_, ND_, NCB_) hipLaunchKernelGGL((k_rbx_xw<NA_, ND_, NCB_>), grid, block, 0, st, prs, wc, per, a, gx, gy, kq, ob, tb)
    if (nd == 4) {
        switch (ncb) {
            case 1: QGTC_RBX_XW(8, 4, 1); break;
            case 2: QGTC_RBX_XW(8, 4, 2); break;
            case 4: QGTC_RBX_XW(8, 4, 4); break;
            default: return QGTC_EINVAL;   // (5 .. 8 bits: N <= 128)
        }
    } else {
        if (ncb != 8) return QGTC_EINVAL;  // (1 .. 4 bits at N <= 128: k_rbw_xw)
        if (nd == 1) QGTC_RBX_XW(2, 1, 8);
        else QGTC_RBX_XW(4, 2, 8);
    }
#undef QGTC_RBX_XW
    HIP_TRY(hipGetLastError());
    return QGTC_OK;
}

int qgtc_launch_rbx_chain(const qgtc_problem *p1, const qgtc_problem *p2, int count, int max_M, int N1, int N2, int t_bits, int act_bits,
                          int mode2, const uint32_t *w2_codes, bool a_tiles, hipStream_t st) {
    const int per = getenv_flag("QGTC_NO_XCD") ? 0 : 1, tiles = a_tiles ? 1 : 0;
    const dim3 grid(step128(max_M), count), block(256);
    const int gx = static_cast<int>(grid.x), gy = static_cast<int>(grid.y);
    const u32x4 *wc = reinterpret_cast<const u32x4 *>(w2_codes);
    const int c1 = rbx_blocks(N1), c2 = mode2 == 0 ? 1 : rbx_blocks(N2);
    const int bits = mode2 == 0 ? t_bits : act_bits;          // (the format class of T and the aggregate's width agree: rbx_chain_ok)
    const int nd = bits <= 2 ? 1 : (bits <= 4 ? 2 : 4);
    if (mode2 < 0 || mode2 > 2 || (nd == 4 && (c1 > 4 || c2 > 4))) return QGTC_EINVAL;
#define QGTC_RBX_GO(ND_, MODE2_, C1_, C2_) hipLaunchKernelGGL((k_rbx_chain<ND_, MODE2_, C1_, C2_>), grid, block, 0, st, p1, p2, wc, per, tiles, gx, gy, bits)
#define QGTC_RBX_C2(ND_, MODE2_, C1_)                      \
    switch (c2) {                                          \
        case 1: QGTC_RBX_GO(ND_, MODE2_, C1_, 1); break;   \
        case 2: QGTC_RBX_GO(ND_, MODE2_, C1_, 2); break;   \
        default: QGTC_RBX_GO(ND_, MODE2_, C1_, 4); break;  \
    }
#define QGTC_RBX_NARROW(MODE2_)   /* 5 .. 8 bits: up to four column blocks either side */ \
    switch (c1) {                                          \
        case 1: QGTC_RBX_C2(4, MODE2_, 1) break;           \
        case 2: QGTC_RBX_C2(4, MODE2_, 2) break;           \
        default: QGTC_RBX_C2(4, MODE2_, 4) break;          \
    }
#define QGTC_RBX_WIDE(ND_, MODE2_)   /* 1 .. 4 bits with eight column blocks on at least one side */ \
    if (c1 == 8) {                                         \
        switch (c2) {                                      \
            case 1: QGTC_RBX_GO(ND_, MODE2_, 8, 1); break; \
            case 2: QGTC_RBX_GO(ND_, MODE2_, 8, 2); break; \
            case 4: QGTC_RBX_GO(ND_, MODE2_, 8, 4); break; \
            default: QGTC_RBX_GO(ND_, MODE2_, 8, 8); break; \
        }                                                  \
    } else if (c2 == 8) {                                  \
        switch (c1) {                                      \
            case 1: QGTC_RBX_GO(ND_, MODE2_, 1, 8); break; \
            case 2: QGTC_RBX_GO(ND_, MODE2_, 2, 8); break; \
            default: QGTC_RBX_GO(ND_, MODE2_, 4, 8); break; \
        }                                                  \
    } else {                                               \
        return QGTC_EINVAL;                                \
    }
    if (mode2 == 0) {
        if (nd == 4) {
            switch (c1) {
                case 1: QGTC_RBX_GO(4, 0, 1, 1); break;
                case 2: QGTC_RBX_GO(4, 0, 2, 1); break;
                default: QGTC_RBX_GO(4, 0, 4, 1); break;
            }
        } else if (c1 == 8) {
            if (nd == 1) QGTC_RBX_GO(1, 0, 8, 1);
            else QGTC_RBX_GO(2, 0, 8, 1);
        } else {
            return QGTC_EINVAL;
        }
    } else if (mode2 == 1) {
        if (nd == 4) { QGTC_RBX_NARROW(1) } else if (nd == 1) { QGTC_RBX_WIDE(1, 1) } else { QGTC_RBX_WIDE(2, 1) }
    } else {
        if (nd == 4) { QGTC_RBX_NARROW(2) } else if (nd == 1) { QGTC_RBX_WIDE(1, 2) } else { QGTC_RBX_WIDE(2, 2) }
    }
#undef QGTC_RBX_WIDE
#undef QGTC_RBX_NARROW
#undef QGTC_RBX_C2
#undef QGTC_RBX_GO
    HIP_TRY(hipGetLastError());
    return QGTC_OK;
}
