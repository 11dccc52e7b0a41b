// launch_stream.hip.h — part of libqgtc_hip.so (qgtc_stream.hip): launcher of the long-K FP4 kernel.
#pragma once

#define QGTC_ST_FOR_ALL(F) F(0, 2, 1) F(0, 2, 2) F(0, 4, 2) F(1, 2, 1) F(1, 2, 2) F(1, 4, 2) F(2, 2, 1) F(2, 2, 2) F(2, 4, 2)

// mode 0 rows-layout bits, 1 cols-layout bits, 2 float32. One-plane operands (stream_ok).
int qgtc_launch_stream(const qgtc_problem &pr, int ob, int mode, bool zero_skip, hipStream_t st) {
    static PerDeviceOnce attr;
    const int arc = attr.run([]() -> int {
#define QGTC_ST_ATTR(MD, RF_, CF_) \
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_bitmm_fp4_stream<MD, RF_, CF_>), hipFuncAttributeMaxDynamicSharedMemorySize, st_lds_bytes(RF_, CF_)));
        QGTC_ST_FOR_ALL(QGTC_ST_ATTR)
#undef QGTC_ST_ATTR
        return QGTC_OK;
    });
    if (arc != QGTC_OK) return arc;
    const int tiles_n = (pr.N + ST_COLS - 1) / ST_COLS;
    const int cf = (tiles_n > 1 || pr.N > 32) ? 2 : 1;   // right-hand fragments of 32 lines
    // 128-row tiles (4 x 2 fragments a multiplying wave: 3.75 VALU operations per MFMA, W fetched once per 128 rows) once they fill the chip;
    // 64-row tiles below that and for up to 32 columns, where the launch follows HBM and two workgroups a CU (72 KB of LDS each) cover each
    // other's first and last groups (tools/stream_route_sweep.sh: 65536 x 16384 x 32 26.5 against 29.4 us; at 64 columns 71 against 60)
    int rf = (cf == 2 && static_cast<long long>((pr.M + 127) / 128) * tiles_n >= 256) ? 4 : 2;
    if (const char *e = std::getenv("QGTC_STREAM_RF")) rf = (std::atoi(e) == 4 && cf == 2) ? 4 : 2;   // (tuning only)
    const int tiles_m = (pr.M + 32 * rf - 1) / (32 * rf);
    const int n_wg = tiles_m * tiles_n;
    const size_t out_bytes = mode == 2 ? static_cast<size_t>(pr.M) * pr.N * 4u
                                       : static_cast<size_t>(ob) * (mode == 1 ? pad128(pr.N) : pad8(pr.M)) * step128(mode == 1 ? pr.M : pr.N) * 16u;
    uint32_t cfg = static_cast<uint32_t>(mode == 2 ? 1 : ob) | (zero_skip ? 1u : 0u) << 8 | static_cast<uint32_t>(tiles_n) << 16;
#ifdef QGTC_STREAM_TUNE
    cfg |= (getenv_flag("ABL_X") ? 1u << 9 : 0u) | (getenv_flag("ABL_NODMA") ? 1u << 10 : 0u);
#endif
    const uint32_t xb = static_cast<uint32_t>(pr.x_words * 4u), wb = static_cast<uint32_t>(pr.w_words * 4u);
    const dim3 grid(static_cast<unsigned>(n_wg)), block(64 * ST_WAVES);
    bool launched = false;
#define QGTC_ST_LAUNCH(MD, RF_, CF_)                                                                                                                      \
    if (!launched && mode == MD && rf == RF_ && cf == CF_) {                                                                                              \
        QGTC_LAUNCH((k_bitmm_fp4_stream<MD, RF_, CF_>), grid, block, static_cast<unsigned>(st_lds_bytes(RF_, CF_)), st, pr.X, pr.W, pr.out, xb, wb,       \
                    static_cast<uint32_t>(out_bytes), pr.M, pr.K, pr.N, pr.w_lines, cfg, n_wg);                                                           \
        launched = true;                                                                                                                                  \
    }
    QGTC_ST_FOR_ALL(QGTC_ST_LAUNCH)
#undef QGTC_ST_LAUNCH
    if (!launched) return QGTC_EINVAL;
    HIP_TRY(launch_status());
    return QGTC_OK;
}
