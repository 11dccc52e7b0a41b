// launch_wide.hip.h — part of libqgtc_hip.so (qgtc_wide.hip): launcher of the wide-operand FP4 kernel.
#pragma once

#define QGTC_WD_FOR_PLANES(F, NL, NR, GB) \
    F(NL, NR, 0, 4, 4, GB) F(NL, NR, 2, 4, 4, GB) F(NL, NR, 0, 2, 4, GB) F(NL, NR, 2, 2, 4, GB) F(NL, NR, 0, 4, 2, GB) F(NL, NR, 2, 4, 2, GB)
#define QGTC_WD_FOR_ALL(F)                                                                                                   \
    QGTC_WD_FOR_PLANES(F, 1, 1, 128) QGTC_WD_FOR_PLANES(F, 1, 2, 64) QGTC_WD_FOR_PLANES(F, 2, 1, 64) QGTC_WD_FOR_PLANES(F, 2, 2, 64) \
    QGTC_WD_FOR_PLANES(F, 1, 4, 64) QGTC_WD_FOR_PLANES(F, 2, 4, 64) F(4, 1, 0, 2, 4, 64) F(4, 1, 2, 2, 4, 64) F(4, 2, 0, 2, 4, 64) F(4, 2, 2, 2, 4, 64) \
    F(1, 8, 0, 4, 2, 64) F(1, 8, 2, 4, 2, 64) F(2, 8, 0, 4, 2, 64) F(2, 8, 2, 4, 2, 64) F(8, 1, 0, 2, 4, 64) F(8, 1, 2, 2, 4, 64) F(8, 2, 0, 2, 4, 64) F(8, 2, 2, 2, 4, 64)

// mode 0 rows-layout bits, 1 cols-layout bits (the operands change places: a cols-layout word runs along M), 2 float32
int qgtc_launch_wide(const qgtc_problem &pr, int a, int w, int ob, int mode, hipStream_t st) {
    const bool swap = mode == 1;
    const uint32_t *Lp = swap ? pr.W : pr.X, *Rp = swap ? pr.X : pr.W;
    const uint32_t l_bytes = static_cast<uint32_t>((swap ? pr.w_words : pr.x_words) * 4u), r_bytes = static_cast<uint32_t>((swap ? pr.x_words : pr.w_words) * 4u);
    const int Lc = swap ? pr.N : pr.M, Rc = swap ? pr.M : pr.N;
    const int x_lines = pad8(pr.M);
    const int l_lines = swap ? pr.w_lines : x_lines, r_lines = swap ? x_lines : pr.w_lines;
    const int out_lines = mode == 2 ? pr.M : swap ? pad128(pr.N) : pad8(pr.M);
    const size_t out_bytes = mode == 2 ? static_cast<size_t>(pr.M) * pr.N * 4u : static_cast<size_t>(ob) * out_lines * step128(Rc) * 16u;
    static PerDeviceOnce attr;
    const int arc = attr.run([]() -> int {
#define QGTC_WD_ATTR(NL, NR, MD, RF, CF, GB) \
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_bitmm_fp4_wide<NL, NR, MD, RF, CF, GB>), hipFuncAttributeMaxDynamicSharedMemorySize, wd_lds_bytes(NL, NR, RF, CF, GB)));
        QGTC_WD_FOR_ALL(QGTC_WD_ATTR)
#undef QGTC_WD_ATTR
        return QGTC_OK;
    });
    if (arc != QGTC_OK) return arc;
    const int nl = swap ? w : a, nr = swap ? a : w;   // planes of the left / right operand (wide_planes_ok)
    const int cover = mode == 2 ? Lc : out_lines;   // (the padding lines of the bit layouts are written as zeros)
    int rf = 4, cf = 4;
    wide_plan(cover, Rc, pr.K, nl, nr, &rf, &cf);
    if (const char *e = std::getenv("QGTC_WIDE_RF")) {   // (tuning only: 4 = 4 x 4 fragments per wave, 2 = 2 x 4, 42 = 4 x 2)
        const int v = std::atoi(e);
        rf = v == 2 ? 2 : 4;
        cf = v == 42 ? 2 : 4;
    }
    if (nl >= 4) rf = 2, cf = 4;   // (four or eight left-hand planes: only on 2 x 4 fragments per wave - registers, LDS)
    if (nr == 8) rf = 4, cf = 2;   // (eight right-hand planes: only on 4 x 2 - the stage of a wider tile does not fit the LDS)
    const int nt_r = (Rc + wd_tr(cf) - 1) / wd_tr(cf);
    const int nt_l = (cover + wd_tl(rf) - 1) / wd_tl(rf);
    const uint32_t cfg = static_cast<uint32_t>(ob) | static_cast<uint32_t>(nt_r) << 8;
    const dim3 grid(static_cast<unsigned>(nt_l * nt_r)), block(64 * WD_WAVES);
    const int md = mode == 2 ? 2 : 0;
    bool launched = false;
#define QGTC_WD_LAUNCH(NL, NR, MD, RF, CF, GB)                                                                                              \
    if (!launched && nl == NL && nr == NR && md == MD && rf == RF && cf == CF) {                                                                        \
        hipLaunchKernelGGL((k_bitmm_fp4_wide<NL, NR, MD, RF, CF, GB>), grid, block, wd_lds_bytes(NL, NR, RF, CF, GB), st, Lp, Rp, pr.out, l_bytes, \
                           r_bytes, static_cast<uint32_t>(out_bytes), Lc, Rc, pr.K, l_lines, r_lines, out_lines, cfg);                      \
        launched = true;                                                                                                                    \
    }
    QGTC_WD_FOR_ALL(QGTC_WD_LAUNCH)
#undef QGTC_WD_LAUNCH
    if (!launched) return QGTC_EINVAL;
    HIP_TRY(hipGetLastError());
    return QGTC_OK;
}
