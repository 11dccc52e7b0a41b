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

template <int QW, int NA, int NW, bool ZS>
int launch_single(const qgtc_problem &pr, const Plan &pl, hipStream_t st) {
    const int tiles_m = (pr.M + TM - 1) / TM, tiles_n = (pr.N + TN - 1) / TN;
    static PerDeviceOnce attr;
    const int arc = attr.run([]() -> int {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_bitmm<QW, NA, NW, ZS>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        return QGTC_OK;
    });
    if (arc != QGTC_OK) return arc;
    MMShape sh = pl.sh;
    sh.inv_tiles_n = tiles_n > 1 ? static_cast<uint32_t>((1ull << 32) / tiles_n) : 0xffffffffu;
    QGTC_LAUNCH((k_bitmm<QW, NA, NW, ZS>), dim3(tiles_m * tiles_n), dim3(64 * pl.waves), pl.lds,
                       st, pr, sh, tiles_m, tiles_n);
    HIP_TRY(launch_status());
    return QGTC_OK;
}

template <int QW, int NA, int NW, bool ZS, bool OCC>
int launch_batched(const qgtc_problem *prs, int count, int max_M, int max_N, const Plan &pl,
                   hipStream_t st) {
    const int tiles = ((max_M + TM - 1) / TM) * ((max_N + TN - 1) / TN);
    static PerDeviceOnce attr;
    const int arc = attr.run([]() -> int {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_bitmm_batched<QW, NA, NW, ZS, OCC>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        return QGTC_OK;
    });
    if (arc != QGTC_OK) return arc;
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

// more than eight LEFT-hand planes on a single launch (bitmm_planes.hip.h: census of the set planes, then only those): the b x b-bit X . W
// products of a --bit_width 9 .. 32 epoch - 1213 x 128 x 16 at 32 x 32 planes 27 -> 11-13 us. Measured level or behind the generic kernel
// where the left operand has few planes (the 1 x 32-plane aggregations: 31 against 29 us) and on grouped launches (2850 tiles at 78 KB
// of LDS each: 1.1 against 0.55 ms an epoch): those stay on k_bitmm<1, 0, 0>.
inline bool planes_route(int a, int w, bool grouped) { (void)w; return a > 8 && !grouped && !getenv_flag("QGTC_NO_PLANES"); }
template <bool BATCHED>
int launch_planes(const qgtc_problem *prs, const qgtc_problem &pr1, int count, int max_M, int max_N, int a, int w, int ob, int mode, hipStream_t st) {
    static PerDeviceOnce attr;
    const int arc = attr.run([]() -> int {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_bitmm_planes<BATCHED>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(sizeof(PlanesLds))));
        return QGTC_OK;
    });
    if (arc != QGTC_OK) return arc;
    const MMShape sh = base_shape(a, w, ob, mode);
    const int tiles = ((max_M + 31) / 32) * ((max_N + 31) / 32);
    hipLaunchKernelGGL((k_bitmm_planes<BATCHED>), dim3(tiles, BATCHED ? count : 1), dim3(PL_THREADS), sizeof(PlanesLds), st, prs, pr1, sh);
    HIP_TRY(hipGetLastError());
    return QGTC_OK;
}

template <bool ZS>
int dispatch_single(const qgtc_problem &pr, int K, int a, int w, int ob, int mode, hipStream_t st) {
    if (ZS && planes_route(a, w, false)) return launch_planes<false>(nullptr, pr, 1, pr.M, pr.N, a, w, ob, mode, st);
    Plan pl;
    const long tiles = static_cast<long>((pr.M + TM - 1) / TM) * ((pr.N + TN - 1) / TN);
    return with_kernel<ZS>(a, w, K, ob, mode, tiles, &pl, [&](auto qw, auto na, auto nw) {
        return launch_single<decltype(qw)::value, decltype(na)::value, decltype(nw)::value, ZS>(pr, pl, st);
    });
}

template <bool ZS, bool OCC>
int dispatch_batched(const qgtc_problem *prs, int count, int max_M, int max_N, int K_hint, int a,
                     int w, int ob, int mode, hipStream_t st) {
    if (ZS && planes_route(a, w, true)) return launch_planes<true>(prs, qgtc_problem{}, count, max_M, max_N, a, w, ob, mode, st);
    Plan pl;
    const long tiles = static_cast<long>(count) * ((max_M + TM - 1) / TM) * ((max_N + TN - 1) / TN);
    return with_kernel<ZS>(a, w, K_hint, ob, mode, tiles, &pl, [&](auto qw, auto na, auto nw) {
        return launch_batched<decltype(qw)::value, decltype(na)::value, decltype(nw)::value, ZS, OCC>(
            prs, count, max_M, max_N, pl, st);
    });
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

int grid_for(size_t work_items, int per_block, size_t cap = 2048) {
    size_t blocks = (work_items + per_block - 1) / per_block;
    if (blocks < 1) blocks = 1;
    if (blocks > cap) blocks = cap;  // default: 256 CUs x 8 resident blocks, grid-stride the rest
    return static_cast<int>(blocks);
}

}  // namespace
