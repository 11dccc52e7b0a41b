// launch.hip.h — part of libqgtc_hip.so (included by qgtc_hip.hip, one translation unit).
// Host-side launch plumbing: split-K plan, kernel selection, launchers of both engines.
#pragma once

namespace {

// ------------------------------------------------------------------------------------------
// host-side launch plumbing
// ------------------------------------------------------------------------------------------
struct Plan {
    int waves;  // waves per workgroup (in-workgroup split-K factor)
    MMShape sh;
    size_t lds;
};

// Split K over the waves of a workgroup: `per` k-quads each. Split-K only buys parallelism: every
// extra wave repeats the prologue and adds a slab to the reduction, so a launch with many tiles
// (grouped cluster batches, wide N) runs one or two waves per tile and a launch with few tiles
// (the 4096 x 4096 x 64 micro-benchmark: 256 tiles) runs eight.
constexpr long kTargetWaves = 256 * 4 * 4;  // 4 waves on every SIMD of the chip
inline void plan_split(int K, int planes, int qw, long total_tiles, Plan *pl) {
    const int kq = step128(K);
    long want = (kTargetWaves + total_tiles - 1) / (total_tiles > 0 ? total_tiles : 1);
    if (want < 1) want = 1;
    if (want > MAX_WAVES) want = MAX_WAVES;
    const int per = (kq + static_cast<int>(want) - 1) / static_cast<int>(want);
    pl->sh.per = per;
    pl->waves = (kq + per - 1) / per;
    pl->sh.waves = pl->waves;
    // every wave's staging region, then every wave's partial-sum slab (separate, so that a wave
    // can store its slab while others are still multiplying); single-wave workgroups need no slab
    pl->lds = pl->waves * (region_bytes(planes, qw) + (pl->waves > 1 ? SLAB_BYTES : 0));
}

inline MMShape base_shape(int a, int w, int ob, int mode) {
    MMShape sh{};
    sh.a = a;
    sh.w = w;
    sh.ob = ob;
    sh.mode = mode;
    sh.ab = a;
    sh.wb = w;
    sh.maxv = std::ldexp(1.0f, ob);
    sh.maxm1 = sh.maxv - 1.0f;
    sh.nowrap = 0;
    return sh;
}

// no int32 accumulator of a product with this K can wrap (then requantisation needs no sign test)
inline int no_wrap(int K, int a, int w) {
    if (a > 16 || w > 16) return 0;
    return static_cast<double>(K) * ((1u << a) - 1u) * ((1u << w) - 1u) < 2147483648.0;
}

template <int QW, int NA, int NW, bool ZS>
int launch_single(const qgtc_problem &pr, const Plan &pl, hipStream_t st) {
    const int tiles_m = (pr.M + TM - 1) / TM, tiles_n = (pr.N + TN - 1) / TN;
    static bool attr_set = false;
    if (!attr_set) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_bitmm<QW, NA, NW, ZS>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        attr_set = true;
    }
    MMShape sh = pl.sh;
    sh.inv_tiles_n = tiles_n > 1 ? static_cast<uint32_t>((1ull << 32) / tiles_n) : 0xffffffffu;
    hipLaunchKernelGGL((k_bitmm<QW, NA, NW, ZS>), dim3(tiles_m * tiles_n), dim3(64 * pl.waves), pl.lds,
                       st, pr, sh, tiles_m, tiles_n);
    HIP_TRY(hipGetLastError());
    return QGTC_OK;
}

template <int QW, int NA, int NW, bool ZS, bool OCC>
int launch_batched(const qgtc_problem *prs, int count, int max_M, int max_N, const Plan &pl,
                   hipStream_t st) {
    const int tiles = ((max_M + TM - 1) / TM) * ((max_N + TN - 1) / TN);
    static bool attr_set = false;
    if (!attr_set) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_bitmm_batched<QW, NA, NW, ZS, OCC>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        attr_set = true;
    }
    hipLaunchKernelGGL((k_bitmm_batched<QW, NA, NW, ZS, OCC>), dim3(tiles, count), dim3(64 * pl.waves),
                       pl.lds, st, prs, pl.sh);
    HIP_TRY(hipGetLastError());
    return QGTC_OK;
}

// Kernel selection: the plane combinations the reference's drivers and benchmarks use get a
// kernel with compile-time plane loops and per-shift accumulators; everything else (any a, w in
// 1..32) runs the generic kernel, which blocks the planes 8 x 8 at a time.
template <bool ZS, typename F>
int with_kernel(int a, int w, int K, int ob, int mode, long total_tiles, Plan *pl, F &&go) {
    pl->sh = base_shape(a, w, ob, mode);
    pl->sh.nowrap = no_wrap(K, a, w);
#define QGTC_FIXED(QW_, NA_, NW_)                              \
    if (a == NA_ && w == NW_) {                                \
        plan_split(K, NA_ + NW_, QW_, total_tiles, pl);        \
        return go(std::integral_constant<int, QW_>{}, std::integral_constant<int, NA_>{}, \
                  std::integral_constant<int, NW_>{});         \
    }
    QGTC_FIXED(4, 1, 1)
    QGTC_FIXED(4, 1, 2)
    QGTC_FIXED(2, 1, 4)
    QGTC_FIXED(1, 1, 8)
    QGTC_FIXED(4, 2, 2)
    QGTC_FIXED(2, 4, 4)
#undef QGTC_FIXED
    pl->sh.ab = a < 8 ? a : 8;
    pl->sh.wb = w < 8 ? w : 8;
    plan_split(K, pl->sh.ab + pl->sh.wb, 1, total_tiles, pl);
    return go(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{},
              std::integral_constant<int, 0>{});
}

template <bool ZS>
int dispatch_single(const qgtc_problem &pr, int K, int a, int w, int ob, int mode, hipStream_t st) {
    Plan pl;
    const long tiles = static_cast<long>((pr.M + TM - 1) / TM) * ((pr.N + TN - 1) / TN);
    return with_kernel<ZS>(a, w, K, ob, mode, tiles, &pl, [&](auto qw, auto na, auto nw) {
        return launch_single<decltype(qw)::value, decltype(na)::value, decltype(nw)::value, ZS>(pr, pl, st);
    });
}

template <bool ZS, bool OCC>
int dispatch_batched(const qgtc_problem *prs, int count, int max_M, int max_N, int K_hint, int a,
                     int w, int ob, int mode, hipStream_t st) {
    Plan pl;
    const long tiles = static_cast<long>(count) * ((max_M + TM - 1) / TM) * ((max_N + TN - 1) / TN);
    return with_kernel<ZS>(a, w, K_hint, ob, mode, tiles, &pl, [&](auto qw, auto na, auto nw) {
        return launch_batched<decltype(qw)::value, decltype(na)::value, decltype(nw)::value, ZS, OCC>(
            prs, count, max_M, max_N, pl, st);
    });
}

// the MFMA engine handles up to 8 planes per operand (8: offset by 128, corrected in the epilogue)
inline bool mfma_ok(int a, int w) { return a >= 1 && a <= 8 && w >= 1 && w <= 8; }

// the FP4 form of the matrix-core engine: 2-bit values at most and float32 sums that stay exact
inline bool fp4_ok(int K, int a, int w) {
    return a <= 2 && w <= 2 && static_cast<double>(K) * ((1 << a) - 1) * ((1 << w) - 1) < 16777216.0;
}

// QGTC_ENGINE_AUTO: pick the engine by a two-line cost model fitted to the round-1 measurements
// (DESIGN.md section 5.4b): popcount runs at ~0.95e15 bit-ops/s plus ~3 us of launch and tail; the
// matrix-core engine pays ~6 us fixed and ~0.46 us per k-quad and 128 x 128 tile round (a quarter
// more per extra plane to expand), rounds = tiles / 256 CUs. MFMA only when it wins by 10 %.
inline bool auto_prefers_mfma(int M, int K, int N, int a, int w) {
    if (!mfma_ok(a, w)) return false;
    const double tiles = static_cast<double>((M + MF_T - 1) / MF_T) * ((N + MF_T - 1) / MF_T);
    const double rounds = tiles <= 256.0 ? 1.0 : tiles / 256.0 * 0.9;
    const int maxp = a > w ? a : w;
    // per k-quad and round: 0.32 us in the FP4 form (2-bit values at most), else 0.46 us plus 15 % per extra plane
    const double per_kq = fp4_ok(K, a, w) ? 0.32 * (1.0 + 0.25 * (maxp - 1)) : 0.46 * (1.0 + 0.15 * (maxp - 1));
    const double t_mfma = 6.0 + per_kq * step128(K) * rounds;
    const double t_pop = 3.0 + 2.0 * M * static_cast<double>(K) * N * a * w / 0.95e15 * 1e6;
    return t_mfma < 0.9 * t_pop;
}

#define QGTC_MF_FOR_ALL(F)                                                                   \
    F(1, 4, false) F(2, 4, false) F(4, 4, false) F(8, 4, false) F(1, 8, false) F(2, 8, false) \
    F(4, 8, false) F(8, 8, false) F(1, 4, true) F(2, 4, true) F(1, 8, true) F(2, 8, true)

int launch_mfma(const qgtc_problem &pr, int a, int w, int ob, int mode, hipStream_t st) {
    MMShape sh = base_shape(a, w, ob, mode);
    sh.nowrap = no_wrap(pr.K, a, w);
    const int tiles_m = (pr.M + MF_T - 1) / MF_T, tiles_n = (pr.N + MF_T - 1) / MF_T;
    const int maxp = a > w ? a : w;
    static bool attr_set = false;
    if (!attr_set) {
#define QGTC_MF_ATTR(P, E, F4) \
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_bitmm_mfma<P, E, F4>), hipFuncAttributeMaxDynamicSharedMemorySize, mf_lds_bytes(F4)));
        QGTC_MF_FOR_ALL(QGTC_MF_ATTR)
#undef QGTC_MF_ATTR
        attr_set = true;
    }
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

// QGTC_ENGINE_AUTO for grouped launches (cluster batches: many small products, every workgroup short-
// lived). Measured on the ogbn-arxiv- and ppi-sized epochs: the matrix-core
// engine wins when its 128-wide tile is mostly full (N = 128: X.W 13 us against 24, A.(XW) 22
// against 27) or when four or more plane pairs share one expansion at N >= 48 (ppi: 4 x 4 bits at
// N = 64 14 against 22, 1 x 4 bits at N = 50 18.6 against 20.8); narrow outputs (N = 10 classes)
// and one- or two-pair products at N <= 64 stay on the popcount kernels.
inline bool auto_prefers_mfma_batched(int max_M, int max_N, int a, int w) {
    if (!mfma_ok(a, w) || max_M < MF_T) return false;
    return max_N >= 96 || (a * w >= 4 && max_N >= 48);
}

// grouped launch on the matrix cores: one workgroup per 128 x 128 tile of every problem
int launch_mfma_batched(const qgtc_problem *prs, int count, int max_M, int max_K, int max_N, int a, int w,
                        int ob, int mode, hipStream_t st) {
    MMShape sh = base_shape(a, w, ob, mode);
    sh.nowrap = no_wrap(max_K, a, w);
    const int tiles = ((max_M + MF_T - 1) / MF_T) * ((max_N + MF_T - 1) / MF_T);
    const int maxp = a > w ? a : w;
    static bool attr_set = false;
    if (!attr_set) {
#define QGTC_MF_ATTR(P, E, F4) \
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_bitmm_mfma_batched<P, E, F4>), hipFuncAttributeMaxDynamicSharedMemorySize, mf_lds_bytes(F4)));
        QGTC_MF_FOR_ALL(QGTC_MF_ATTR)
#undef QGTC_MF_ATTR
        attr_set = true;
    }
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

// narrow right operands: the LDS-free FP4 kernel (bitmm_fp4_skinny.hip.h). Measured at 4096 x 4096 x N: ahead of
// both other kernels up to N = 256 (N = 128: 5.4 us against 7.0 popcount / 15.5 128-tile MFMA at 1 bit, 16.6 against
// 40 / 46 at 8 bits), level with the 128-tile kernel at N = 512, behind it at N = 1024
// (plane capacities 1 / 2 for X and 1 / 2 / 4 / 8 for W are instantiated; float32 sums must stay exact)
inline bool skinny_ok(int K, int N, int a, int w) {
    return N <= 256 && a <= 2 && w <= 8 &&
           static_cast<double>(K) * ((1 << a) - 1) * ((1 << w) - 1) < 16777216.0;
}
// QGTC_ENGINE_AUTO: measured against the popcount kernels on the reference's micro-benchmark shapes
// (1024 / 2048 / 4096 square, N = 16 / 32 / 64, 1- and 2-bit): ahead on all of them (4096 x 4096 x 64:
// 4.1 us against 4.8 at 1 bit, 5.8 against 7.0 at 2 bits); tiny problems stay where they were.
inline bool auto_prefers_skinny(int M, int K, int N, int a, int w) {
    (void)N; (void)a; (void)w;
    return M >= 512 && K >= 512;
}

int launch_skinny(const qgtc_problem &pr, int a, int w, int ob, int mode, bool zero_skip, hipStream_t st) {
    MMShape sh = base_shape(a, w, ob, mode);
    sh.nowrap = 1;
    const int zs = zero_skip ? 1 : 0;
    const bool wide = pr.N > 32;   // 32 x 32 tiles (two column tiles at N = 64), else 16 x 32
    const dim3 grid(wide ? (pr.M + 31) / 32 : (pr.M + 15) / 16, (pr.N + 31) / 32);
#define QGTC_SK_LAUNCH(NA_, NW_)                                                                                   \
    if (!done && a <= NA_ && w <= NW_) {                                                                           \
        done = true;                                                                                               \
        if (mode == 2) {                                                                                           \
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

// grouped launches on the matrix cores, one wave per 32 x 32 tile (bitmm_fp4_wave.hip.h): for NARROW outputs.
// Measured on the epochs (tools/epoch_stages.py): the class-count stages (N = 10) 7.1 / 10.9 us against 13.8 / 12.6
// for the popcount kernels, ppi's 1 x 4-bit A-stages at N = 50 14.3 against 16.3 for the 128-tile kernel; at
// N = 128 the 128-tile kernel is ahead (12 / 18.5 us against 18 / 25), and 64 x 64 outputs per wave are worse
// still (24 / 41 us: 2850 waves do not fill the chip).
inline bool fp4_wave_ok(int K, int N, int a, int w) {
    return N <= 64 && a <= 2 && w <= 8 && static_cast<double>(K) * ((1 << a) - 1) * ((1 << w) - 1) < 16777216.0;
}

int launch_fp4_wave(const qgtc_problem *prs, int count, int max_M, int max_N, int a, int w, int ob, int mode,
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
#undef QGTC_FW_LAUNCH
    HIP_TRY(hipGetLastError());
    return QGTC_OK;
}

int check_mm_args(const uint32_t *X, const uint32_t *W, const void *out, int M, int K, int N,
                  int a, int w) {
    if (!X || !W || !out) return QGTC_EINVAL;
    if (M <= 0 || K <= 0 || N <= 0) return QGTC_EINVAL;
    if (!bits_ok(a) || !bits_ok(w)) return QGTC_EINVAL;
    if (!aligned16(X) || !aligned16(W)) return QGTC_EALIGN;
    if (step128(K) * 4 >= (1 << 24)) return QGTC_EINVAL;  // 24-bit row-stride multiplies in the kernel
    return QGTC_OK;
}

// in-kernel byte offsets inside one operand are 32-bit
inline bool words_ok(size_t x_words, size_t w_words) {
    return x_words < (1ull << 30) && w_words < (1ull << 30);  // < 4 GiB per packed operand
}

int grid_for(size_t work_items, int per_block) {
    size_t blocks = (work_items + per_block - 1) / per_block;
    if (blocks < 1) blocks = 1;
    if (blocks > 2048) blocks = 2048;  // 256 CUs x 8 resident blocks, grid-stride the rest
    return static_cast<int>(blocks);
}

}  // namespace
