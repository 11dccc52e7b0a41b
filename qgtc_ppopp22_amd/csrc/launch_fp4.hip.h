// launch_fp4.hip.h — part of libqgtc_hip.so (qgtc_fp4.hip): launchers of the FP4 narrow-operand kernels.
#pragma once

int qgtc_launch_skinny(const qgtc_problem &pr, int a, int w, int ob, int mode, bool zero_skip, hipStream_t st) {
    MMShape sh = base_shape(a, w, ob, mode);
    sh.nowrap = 1;
    const int zs = zero_skip ? 1 : 0;
    const bool wide = pr.N > 32 || mode == 1;   // 32 x 32 tiles (two column tiles at N = 64), else 16 x 32
    const dim3 grid(wide ? (pr.M + 31) / 32 : (pr.M + 15) / 16, (pr.N + 31) / 32);
#define QGTC_SK_LAUNCH(NA_, NW_)                                                                                   \
    if (!done && a <= NA_ && w <= NW_) {                                                                           \
        done = true;                                                                                               \
        if (mode == 1) {                                                                                           \
            hipLaunchKernelGGL((k_bitmm_fp4_skinny<NA_, NW_, 1, 2, 2>), grid, dim3(64 * SK_WAVES), 0, st, pr, sh, zs);            \
        } else if (mode == 2) {                                                                                    \
            if (wide) hipLaunchKernelGGL((k_bitmm_fp4_skinny<NA_, NW_, 2, 2, 2>), grid, dim3(64 * SK_WAVES), 0, st, pr, sh, zs);  \
            else hipLaunchKernelGGL((k_bitmm_fp4_skinny<NA_, NW_, 2, 1, 2>), grid, dim3(64 * SK_WAVES), 0, st, pr, sh, zs);       \
        } else {                                                                                                   \
            if (wide) hipLaunchKernelGGL((k_bitmm_fp4_skinny<NA_, NW_, 0, 2, 2>), grid, dim3(64 * SK_WAVES), 0, st, pr, sh, zs);  \
            else hipLaunchKernelGGL((k_bitmm_fp4_skinny<NA_, NW_, 0, 1, 2>), grid, dim3(64 * SK_WAVES), 0, st, pr, sh, zs);       \
        }                                                                                                          \
    }
    bool done = false;
    QGTC_SK_LAUNCH(1, 1) QGTC_SK_LAUNCH(1, 2) QGTC_SK_LAUNCH(1, 4) QGTC_SK_LAUNCH(1, 8)
    QGTC_SK_LAUNCH(2, 1) QGTC_SK_LAUNCH(2, 2) QGTC_SK_LAUNCH(2, 4) QGTC_SK_LAUNCH(2, 8)
#undef QGTC_SK_LAUNCH
    HIP_TRY(hipGetLastError());
    return QGTC_OK;
}

int qgtc_launch_fp4_wave(const qgtc_problem *prs, int count, int max_M, int max_N, int a, int w, int ob, int mode,
                    bool zero_skip, hipStream_t st) {
    MMShape sh = base_shape(a, w, ob, mode);
    sh.nowrap = 1;
    const int zs = zero_skip ? 1 : 0;
    const dim3 grid(((max_M + 31) / 32) * ((max_N + 31) / 32), count);   // 32 x 32 outputs per wave
#define QGTC_FW_LAUNCH(NA_, NW_)                                                                                 \
    if (!done && a <= NA_ && w <= NW_) {                                                                         \
        done = true;                                                                                             \
        if (mode == 2) hipLaunchKernelGGL((k_bitmm_fp4_wave<NA_, NW_, 2, 2, 2>), grid, dim3(64), 0, st, prs, sh, zs);      \
        else if (mode == 1) hipLaunchKernelGGL((k_bitmm_fp4_wave<NA_, NW_, 1, 2, 2>), grid, dim3(64), 0, st, prs, sh, zs); \
        else hipLaunchKernelGGL((k_bitmm_fp4_wave<NA_, NW_, 0, 2, 2>), grid, dim3(64), 0, st, prs, sh, zs);                \
    }
    bool done = false;
    QGTC_FW_LAUNCH(1, 1) QGTC_FW_LAUNCH(1, 2) QGTC_FW_LAUNCH(1, 4) QGTC_FW_LAUNCH(1, 8)
    QGTC_FW_LAUNCH(2, 1) QGTC_FW_LAUNCH(2, 2) QGTC_FW_LAUNCH(2, 4) QGTC_FW_LAUNCH(2, 8)
    QGTC_FW_LAUNCH(4, 4) QGTC_FW_LAUNCH(4, 8)   // ppi's 4 x 4-bit X.W stages
#undef QGTC_FW_LAUNCH
    HIP_TRY(hipGetLastError());
    return QGTC_OK;
}

