// launch_mfma.hip.h — part of libqgtc_hip.so (qgtc_mfma.hip): launchers of the 128 x 128-tile matrix-core engine.
#pragma once

#define QGTC_MF_FOR_ALL(F)                                                                   \
    F(1, 4, false) F(2, 4, false) F(4, 4, false) F(8, 4, false) F(1, 8, false) F(2, 8, false) \
    F(4, 8, false) F(8, 8, false) F(1, 4, true) F(2, 4, true) F(1, 8, true) F(2, 8, true)

int qgtc_launch_mfma(const qgtc_problem &pr, int a, int w, int ob, int mode, hipStream_t st) {
    MMShape sh = base_shape(a, w, ob, mode);
    sh.nowrap = no_wrap(pr.K, a, w);
    const int tiles_m = (pr.M + MF_T - 1) / MF_T, tiles_n = (pr.N + MF_T - 1) / MF_T;
    const int maxp = a > w ? a : w;
    static PerDeviceOnce attr;
    const int arc = attr.run([]() -> int {
#define QGTC_MF_ATTR(P, E, F4) \
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_bitmm_mfma<P, E, F4>), hipFuncAttributeMaxDynamicSharedMemorySize, mf_lds_bytes(F4)));
        QGTC_MF_FOR_ALL(QGTC_MF_ATTR)
#undef QGTC_MF_ATTR
        return QGTC_OK;
    });
    if (arc != QGTC_OK) return arc;
    const dim3 grid(tiles_m * tiles_n);
    // two workgroups per CU overlap each other from 512 tiles on; below that one 12-wave workgroup per CU
    const bool wide = tiles_m * tiles_n < 512;
    const bool fp4 = fp4_ok(pr.K, a, w);
#define QGTC_MF_LAUNCH(P, F4)                                                                               \
    if (wide) hipLaunchKernelGGL((k_bitmm_mfma<P, 8, F4>), grid, dim3(768), mf_lds_bytes(F4), st, pr, sh, tiles_n);    \
    else hipLaunchKernelGGL((k_bitmm_mfma<P, 4, F4>), grid, dim3(512), mf_lds_bytes(F4), st, pr, sh, tiles_n);
    if (fp4 && maxp <= 1) { QGTC_MF_LAUNCH(1, true) }
    else if (fp4) { QGTC_MF_LAUNCH(2, true) }
    else if (maxp <= 1) { QGTC_MF_LAUNCH(1, false) }
    else if (maxp <= 2) { QGTC_MF_LAUNCH(2, false) }
    else if (maxp <= 4) { QGTC_MF_LAUNCH(4, false) }
    else { QGTC_MF_LAUNCH(8, false) }
#undef QGTC_MF_LAUNCH
    HIP_TRY(hipGetLastError());
    return QGTC_OK;
}

// grouped launch on the matrix cores: one workgroup per 128 x 128 tile of every problem
int qgtc_launch_mfma_batched(const qgtc_problem *prs, int count, int max_M, int max_K, int max_N, int a, int w,
                        int ob, int mode, hipStream_t st) {
    MMShape sh = base_shape(a, w, ob, mode);
    sh.nowrap = no_wrap(max_K, a, w);
    const int tiles = ((max_M + MF_T - 1) / MF_T) * ((max_N + MF_T - 1) / MF_T);
    const int maxp = a > w ? a : w;
    static PerDeviceOnce attr;
    const int arc = attr.run([]() -> int {
#define QGTC_MF_ATTR(P, E, F4) \
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_bitmm_mfma_batched<P, E, F4>), hipFuncAttributeMaxDynamicSharedMemorySize, mf_lds_bytes(F4)));
        QGTC_MF_FOR_ALL(QGTC_MF_ATTR)
#undef QGTC_MF_ATTR
        return QGTC_OK;
    });
    if (arc != QGTC_OK) return arc;
    const dim3 grid(tiles, count);
    const bool wide = static_cast<long>(tiles) * count < 512;
    const bool fp4 = fp4_ok(max_K, a, w);
#define QGTC_MF_LAUNCH(P, F4)                                                                              \
    if (wide) hipLaunchKernelGGL((k_bitmm_mfma_batched<P, 8, F4>), grid, dim3(768), mf_lds_bytes(F4), st, prs, sh);   \
    else hipLaunchKernelGGL((k_bitmm_mfma_batched<P, 4, F4>), grid, dim3(512), mf_lds_bytes(F4), st, prs, sh);
    if (fp4 && maxp <= 1) { QGTC_MF_LAUNCH(1, true) }
    else if (fp4) { QGTC_MF_LAUNCH(2, true) }
    else if (maxp <= 1) { QGTC_MF_LAUNCH(1, false) }
    else if (maxp <= 2) { QGTC_MF_LAUNCH(2, false) }
    else if (maxp <= 4) { QGTC_MF_LAUNCH(4, false) }
    else { QGTC_MF_LAUNCH(8, false) }
#undef QGTC_MF_LAUNCH
    HIP_TRY(hipGetLastError());
    return QGTC_OK;
}
