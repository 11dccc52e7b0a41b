// launch_fp4.hip.h — part of libqgtc_hip.so (qgtc_fp4.hip): launchers of the FP4 narrow-operand kernels.
#pragma once

// bytes of the result (32-bit offsets inside the kernel)
static size_t one_out_bytes(const qgtc_problem &pr, int ob, int mode) {
    if (mode == 2) return static_cast<size_t>(pr.M) * pr.N * 4u;
    if (mode == 1) return static_cast<size_t>(ob) * step128(pr.M) * 4u * pad128(pr.N) * 4u;
    return static_cast<size_t>(ob) * pad8(pr.M) * step128(pr.N) * 4u * 4u;
}
static bool one_ok(const qgtc_problem &pr, int ob, int mode) {
    return pr.K <= ONE_MAX_K && (mode == 2 || ob <= 23) && pr.M < (1 << 24) && one_out_bytes(pr, ob, mode) < (1ull << 32);
}

// K <= 4096: one super-step per wave, the latency-trimmed kernel (bitmm_fp4_one.hip.h)
static int launch_one(const qgtc_problem &pr, int a, int w, int ob, int mode, bool zero_skip, hipStream_t st) {
    const bool wide = pr.N > 32 || mode == 1;   // 32 x 32 tiles (two column tiles at N = 64), else 16 x 32
    // more than four right-hand planes: 64 x 16 tiles - a workgroup then fetches and expands a quarter of W instead of
    // half of it (4096 x 4096 x 64 at 8 bits: 32 -> 16 MB out of L2, 424 -> 272 VALU operations per wave: 7.9 -> 6.4 us; no
    // change at 4 bits; measured ahead from 512 x 512 x 64 up, at N = 32 only from M = 4096). Every output
    // word must be covered by two column tiles: N a multiple of 32.
    const bool tall = w > 4 && mode != 1 && pr.N % 32 == 0 && ((pr.N >= 64 && pr.M >= 512) || pr.M >= 4096) && !getenv_flag("QGTC_NO_TALL");
    const dim3 grid(tall ? (pr.M + 63) / 64 : wide ? (pr.M + 31) / 32 : (pr.M + 15) / 16, tall ? pr.N / 16 : (pr.N + 31) / 32), block(64 * ONE_WAVES);
    const uint32_t cfg = static_cast<uint32_t>(a) | static_cast<uint32_t>(w) << 8 | static_cast<uint32_t>(mode == 2 ? 1 : ob) << 16 |
                         (zero_skip ? 1u : 0u) << 24
#ifdef QGTC_ABL
                         | (getenv_flag("ABL_X") ? 1u << 28 : 0u) | (getenv_flag("ABL_W") ? 1u << 29 : 0u)
#endif
#ifdef QGTC_STAGGER_EXP
                         | (getenv_flag("STAGGER_SLEEP") ? 1u << 30 : 0u) | (getenv_flag("STAGGER_PRIO") ? 1u << 31 : 0u) | (getenv_flag("STAGGER_SLEEP2") ? 1u << 27 : 0u)
#endif
        ;
    const uint32_t xb = static_cast<uint32_t>(pr.x_words * 4u), wb = static_cast<uint32_t>(pr.w_words * 4u);
    const uint32_t ob_ = static_cast<uint32_t>(one_out_bytes(pr, ob, mode));
#define QGTC_ONE_GO(NA_, NW_, MODE_, RF_)                                                                          \
    QGTC_LAUNCH((k_bitmm_fp4_one<NA_, NW_, MODE_, RF_, 2>), grid, block, 0, st, pr.X, pr.W, pr.out, xb, wb, ob_, pr.M, \
                       pr.K, pr.N, pr.w_lines, cfg)
#define QGTC_ONE_TALL(NA_, NW_)                                                                                               \
    if (!done && tall && a <= NA_ && w <= NW_) {                                                                               \
        done = true;                                                                                                           \
        if (mode == 2) QGTC_LAUNCH((k_bitmm_fp4_one<NA_, NW_, 2, 4, 1>), grid, block, 0, st, pr.X, pr.W, pr.out, xb, wb, ob_, pr.M, pr.K, pr.N, pr.w_lines, cfg); \
        else QGTC_LAUNCH((k_bitmm_fp4_one<NA_, NW_, 0, 4, 1>), grid, block, 0, st, pr.X, pr.W, pr.out, xb, wb, ob_, pr.M, pr.K, pr.N, pr.w_lines, cfg);           \
    }
#define QGTC_ONE_LAUNCH(NA_, NW_)                                       \
    if (!done && a <= NA_ && w <= NW_) {                                \
        done = true;                                                    \
        if (mode == 1) { QGTC_ONE_GO(NA_, NW_, 1, 2); }                 \
        else if (mode == 2) { if (wide) { QGTC_ONE_GO(NA_, NW_, 2, 2); } else { QGTC_ONE_GO(NA_, NW_, 2, 1); } } \
        else { if (wide) { QGTC_ONE_GO(NA_, NW_, 0, 2); } else { QGTC_ONE_GO(NA_, NW_, 0, 1); } }                \
    }
    bool done = false;
    QGTC_ONE_TALL(1, 8) QGTC_ONE_TALL(2, 8)
    QGTC_ONE_LAUNCH(1, 1) QGTC_ONE_LAUNCH(1, 2) QGTC_ONE_LAUNCH(1, 4) QGTC_ONE_LAUNCH(1, 8)
    QGTC_ONE_LAUNCH(2, 1) QGTC_ONE_LAUNCH(2, 2) QGTC_ONE_LAUNCH(2, 4) QGTC_ONE_LAUNCH(2, 8)
#undef QGTC_ONE_LAUNCH
#undef QGTC_ONE_TALL
#undef QGTC_ONE_GO
    HIP_TRY(launch_status());
    return QGTC_OK;
}

// (for qgtc_bitmm_route: which of the two narrow-operand kernels qgtc_launch_skinny picks)
bool qgtc_skinny_is_one(const qgtc_problem &pr, int ob, int mode) { return one_ok(pr, ob, mode) && !getenv_flag("QGTC_NO_ONE"); }

int qgtc_launch_skinny(const qgtc_problem &pr, int a, int w, int ob, int mode, bool zero_skip, hipStream_t st) {
    if (one_ok(pr, ob, mode) && !getenv_flag("QGTC_NO_ONE")) return launch_one(pr, a, w, ob, mode, zero_skip, st);
    MMShape sh = base_shape(a, w, ob, mode);
    sh.nowrap = 1;
    const int zs = zero_skip ? 1 : 0;
    const bool wide = pr.N > 32 || mode == 1;   // 32 x 32 tiles (two column tiles at N = 64), else 16 x 32
    const dim3 grid(wide ? (pr.M + 31) / 32 : (pr.M + 15) / 16, (pr.N + 31) / 32);
#define QGTC_SK_LAUNCH(NA_, NW_)                                                                                   \
    if (!done && a <= NA_ && w <= NW_) {                                                                           \
        done = true;                                                                                               \
        if (mode == 1) {                                                                                           \
            QGTC_LAUNCH((k_bitmm_fp4_skinny<NA_, NW_, 1, 2, 2>), grid, dim3(64 * SK_WAVES), 0, st, pr, sh, zs);            \
        } else if (mode == 2) {                                                                                    \
            if (wide) QGTC_LAUNCH((k_bitmm_fp4_skinny<NA_, NW_, 2, 2, 2>), grid, dim3(64 * SK_WAVES), 0, st, pr, sh, zs);  \
            else QGTC_LAUNCH((k_bitmm_fp4_skinny<NA_, NW_, 2, 1, 2>), grid, dim3(64 * SK_WAVES), 0, st, pr, sh, zs);       \
        } else {                                                                                                   \
            if (wide) QGTC_LAUNCH((k_bitmm_fp4_skinny<NA_, NW_, 0, 2, 2>), grid, dim3(64 * SK_WAVES), 0, st, pr, sh, zs);  \
            else QGTC_LAUNCH((k_bitmm_fp4_skinny<NA_, NW_, 0, 1, 2>), grid, dim3(64 * SK_WAVES), 0, st, pr, sh, zs);       \
        }                                                                                                          \
    }
    bool done = false;
    QGTC_SK_LAUNCH(1, 1) QGTC_SK_LAUNCH(1, 2) QGTC_SK_LAUNCH(1, 4) QGTC_SK_LAUNCH(1, 8)
    QGTC_SK_LAUNCH(2, 1) QGTC_SK_LAUNCH(2, 2) QGTC_SK_LAUNCH(2, 4) QGTC_SK_LAUNCH(2, 8)
#undef QGTC_SK_LAUNCH
    HIP_TRY(launch_status());
    return QGTC_OK;
}

// grouped "X . W" stages by row blocks (bitmm_fp4_rows.hip.h: k_bitmm_fp4_xw_rows): K <= 128, N <= 128, the plane counts of
// the two epochs
int qgtc_launch_xw_rows(const qgtc_problem *prs, int count, int max_M, int a, int w, int ob, hipStream_t st) {
    MMShape sh = base_shape(a, w, ob, 1);
    sh.nowrap = 1;
    sh.per = getenv_flag("QGTC_NO_XCD") ? 0 : 1;
    const dim3 grid(step128(max_M) * 4, count), block(64 * 4);
    if (a <= 2 && w <= 2 && ob == 2) hipLaunchKernelGGL((k_bitmm_fp4_xw_rows<2, 2, 2>), grid, block, 0, st, prs, sh);
    else if (a <= 4 && w <= 4 && ob == 4) hipLaunchKernelGGL((k_bitmm_fp4_xw_rows<4, 4, 4>), grid, block, 0, st, prs, sh);
    else return QGTC_EINVAL;
    HIP_TRY(hipGetLastError());
    return QGTC_OK;
}

// grouped "A . (XW)" stages: one workgroup per 32-row block of a batch, only the occupied k-quads (bitmm_fp4_rows.hip.h); and grouped
// cols-layout stages no narrower kernel takes (mode 1: five to eight left-hand planes - the X . W stages at --bit_width 5 .. 8 -, or
// more than 128 bits of K with more than 64 columns): one workgroup per WORD of a line
int qgtc_launch_rows(const qgtc_problem *prs, int count, int max_M, int max_N, int a, int w, int ob, int mode, hipStream_t st) {
    MMShape sh = base_shape(a, w, ob, mode);
    sh.nowrap = 1;
    sh.per = getenv_flag("QGTC_NO_XCD") ? 0 : 1;   // (here: row blocks of a batch on one XCD)
    // 32-column blocks: per 32 columns / per word of a packed row / per 32 lines of the cols layout, the padding ones included
    const int blocks = mode == 2 ? (max_N + 31) / 32 : mode == 1 ? pad128(max_N) / 32 : step128(max_N) * 4;
    // (two column blocks per wave - half the waves, one round of them on the chip instead of 1.4 - measured no faster:
    // 11.1 against 10.6 us on the ogbn-arxiv-sized A-stages; kept as a tuning switch)
    const bool two = blocks >= 2 && mode != 1 && a <= 4 && getenv_flag("QGTC_ROWS_CB2");
    const int waves = two ? (blocks + 1) / 2 : blocks;
    if (waves > 8) return QGTC_EINVAL;
    const dim3 grid(mode == 1 ? step128(max_M) * 4 : (max_M + 31) / 32, count), block(64 * waves);
#define QGTC_RW_GO(NA_, NW_, MODE_, OB_)                                                                                   \
    do {                                                                                                                    \
        if (two) hipLaunchKernelGGL((k_bitmm_fp4_rows<NA_, NW_, MODE_, OB_, 2>), grid, block, 0, st, prs, sh);               \
        else hipLaunchKernelGGL((k_bitmm_fp4_rows<NA_, NW_, MODE_, OB_, 1>), grid, block, 0, st, prs, sh);                   \
    } while (0)
#define QGTC_RW_GO1(NA_, NW_, MODE_, OB_) hipLaunchKernelGGL((k_bitmm_fp4_rows<NA_, NW_, MODE_, OB_, 1>), grid, block, 0, st, prs, sh)
#define QGTC_RW_LAUNCH(NA_, NW_)                                         \
    if (!done && a <= NA_ && w <= NW_) {                                 \
        done = true;                                                     \
        if (mode == 2) QGTC_RW_GO(NA_, NW_, 2, 0);                       \
        else if (mode == 1 && ob == NW_) QGTC_RW_GO1(NA_, NW_, 1, NW_);  \
        else if (mode == 1) QGTC_RW_GO1(NA_, NW_, 1, 0);                 \
        else if (ob == 1) QGTC_RW_GO(NA_, NW_, 0, 1);                    \
        else if (ob == 2) QGTC_RW_GO(NA_, NW_, 0, 2);                    \
        else if (ob == 4) QGTC_RW_GO(NA_, NW_, 0, 4);                    \
        else if (ob == 8) QGTC_RW_GO(NA_, NW_, 0, 8);                    \
        else QGTC_RW_GO(NA_, NW_, 0, 0);                                 \
    }
    bool done = false;
    QGTC_RW_LAUNCH(1, 1) QGTC_RW_LAUNCH(1, 2) QGTC_RW_LAUNCH(1, 4) QGTC_RW_LAUNCH(1, 8) QGTC_RW_LAUNCH(2, 2) QGTC_RW_LAUNCH(4, 4) QGTC_RW_LAUNCH(4, 8)
    if (!done && a <= 8 && w <= 8) {   // five to eight left-hand planes
        done = true;
        if (mode == 2) QGTC_RW_GO1(8, 8, 2, 0);
        else if (mode == 1 && ob == 8) QGTC_RW_GO1(8, 8, 1, 8);
        else if (mode == 1) QGTC_RW_GO1(8, 8, 1, 0);
        else if (ob == 8) QGTC_RW_GO1(8, 8, 0, 8);
        else QGTC_RW_GO1(8, 8, 0, 0);
    }
#undef QGTC_RW_LAUNCH
#undef QGTC_RW_GO1
#undef QGTC_RW_GO
    if (!done) return QGTC_EINVAL;
    HIP_TRY(hipGetLastError());
    return QGTC_OK;
}


// ONE problem on the row-block kernel (by value: no descriptor in device memory): 3-8 left-hand planes, any of the three outputs
int qgtc_launch_rows_single(const qgtc_problem &pr, int a, int w, int ob, int mode, hipStream_t st) {
    MMShape sh = base_shape(a, w, ob, mode);
    sh.nowrap = 1;
    sh.per = 0;
    // a wave per 32 columns (float32) / per word of a packed row (rows layout) / per 32 lines, the padding ones included (cols layout)
    const int waves = mode == 2 ? (pr.N + 31) / 32 : mode == 1 ? pad128(pr.N) / 32 : step128(pr.N) * 4;
    if (waves > 8 || mode < 0 || mode > 2) return QGTC_EINVAL;
    const dim3 grid(mode == 1 ? step128(pr.M) * 4 : (pr.M + 31) / 32), block(64 * waves);   // (cols layout: a workgroup per word of a line)
#define QGTC_RW1_GO(NA_, NW_, MODE_, OB_) QGTC_LAUNCH((k_bitmm_fp4_rows_single<NA_, NW_, MODE_, OB_>), grid, block, 0, st, pr, sh)
#define QGTC_RW1_LAUNCH(NA_, NW_, OBS_)                                  \
    if (!done && a <= NA_ && w <= NW_) {                                 \
        done = true;                                                     \
        if (mode == 2) QGTC_RW1_GO(NA_, NW_, 2, 0);                      \
        else if (mode == 1 && ob == OBS_) QGTC_RW1_GO(NA_, NW_, 1, OBS_); \
        else if (mode == 1) QGTC_RW1_GO(NA_, NW_, 1, 0);                 \
        else if (ob == OBS_) QGTC_RW1_GO(NA_, NW_, 0, OBS_);             \
        else QGTC_RW1_GO(NA_, NW_, 0, 0);                                \
    }
    bool done = false;
    QGTC_RW1_LAUNCH(4, 4, 4) QGTC_RW1_LAUNCH(4, 8, 4) QGTC_RW1_LAUNCH(8, 8, 8)   // (b x b-bit X . W of the drivers, b = 3 .. 8)
#undef QGTC_RW1_LAUNCH
#undef QGTC_RW1_GO
    if (!done) return QGTC_EINVAL;
    HIP_TRY(launch_status());
    return QGTC_OK;
}

// ---- row block per wave (bitmm_fp4_rbw.hip.h)
int qgtc_launch_expand_weights(const qgtc_expand_job *jobs, int n_jobs, hipStream_t st) {
    ExpandJobs ej{};
    int most = 0, most_kq = 1;
    for (int i = 0; i < n_jobs; i++) {
        const qgtc_expand_job &j = jobs[i];
        const int ms = weight_table_slices(j.K, j.order);
        ej.job[i] = ExpandJob{j.W, j.codes, j.w_words, j.K, j.N, j.w_lines, j.nbits, j.order, weight_table_blocks(j.N), ms};   // (columns past N: zero codes)
        most = std::max(most, weight_table_blocks(j.N) * ms);
        if (j.order == 0) most_kq = std::max(most_kq, step128(j.K));   // (a table per k-quad of K)
    }
    hipLaunchKernelGGL(k_expand_weights, dim3(most, n_jobs, most_kq), dim3(64), 0, st, ej);
    HIP_TRY(hipGetLastError());
    return QGTC_OK;
}

int qgtc_launch_cols_to_chain(const uint32_t *cols, size_t words, int H, int W, int nbits, uint32_t *chain, hipStream_t st) {
    const size_t total = static_cast<size_t>(step128(H)) * 4u * pad128(W);   // (threads: one per 16 bytes of ONE array; a thread writes both arrays of a 5 .. 8-bit X)
    hipLaunchKernelGGL(k_cols_to_chain, dim3(static_cast<unsigned>(std::min<size_t>((total + 255) / 256, 4096))), dim3(256), 0, st, cols,
                       static_cast<unsigned long long>(words), H, W, nbits, chain);
    HIP_TRY(hipGetLastError());
    return QGTC_OK;
}

int qgtc_launch_cols_to_chain_batched(const qgtc_loader_batch *batches, int count, int max_n, int W, int nbits, hipStream_t st) {
    const size_t total = static_cast<size_t>(step128(max_n)) * 4u * pad128(W);
    hipLaunchKernelGGL(k_cols_to_chain_batched, dim3(static_cast<unsigned>(std::min<size_t>((total + 255) / 256, 4096)), count), dim3(256), 0, st,
                       batches, W, nbits);
    HIP_TRY(hipGetLastError());
    return QGTC_OK;
}

int qgtc_launch_rbw_xw(const qgtc_problem *prs, int count, int max_M, int K, int N, int a, int ob, const uint32_t *w_codes, hipStream_t st) {
    const int per = getenv_flag("QGTC_NO_XCD") ? 0 : 1;   // (the workgroups of a batch on one XCD)
    const dim3 grid(step128(max_M), count), block(256);
    const int gx = static_cast<int>(grid.x), gy = static_cast<int>(grid.y);
    const u32x4 *wc = reinterpret_cast<const u32x4 *>(w_codes);
    const int ncb = (N + 31) / 32, kq = step128(K);   // (the k-quads of the WEIGHT TABLES: launch-uniform, preloaded into an SGPR)
    if (ob < 1 || ob > 4 || a < 1 || a > (ob > 2 ? 4 : 2) || ncb < 1 || ncb > 4) return QGTC_EINVAL;
    // column blocks 1 / 2 / 4 (three run as four: the lines past N are zeros); the 2-bit chain of the BASELINE epoch also has its own three
#define QGTC_RBWX_GO(NA_, OB_, NCB_) QGTC_LAUNCH((k_rbw_xw<NA_, OB_, NCB_>), grid, block, 0, st, prs, wc, per, a, gx, gy, kq)
#define QGTC_RBWX_OB(NA_, OB_)                                            \
    if (ncb == 1) QGTC_RBWX_GO(NA_, OB_, 1);                              \
    else if (ncb == 2) QGTC_RBWX_GO(NA_, OB_, 2);                         \
    else if (ncb == 3 && OB_ == 2) QGTC_RBWX_GO(NA_, OB_, (OB_ == 2 ? 3 : 4)); \
    else QGTC_RBWX_GO(NA_, OB_, 4)
    switch (ob) {
        case 1: QGTC_RBWX_OB(2, 1); break;
        case 2: QGTC_RBWX_OB(2, 2); break;
        case 3: QGTC_RBWX_OB(4, 3); break;
        default: QGTC_RBWX_OB(4, 4); break;
    }
#undef QGTC_RBWX_OB
#undef QGTC_RBWX_GO
    HIP_TRY(launch_status());
    return QGTC_OK;
}

int qgtc_launch_rows_to_tiles(const uint32_t *rows, size_t words, int M, int K, uint32_t *tiles, hipStream_t st) {
    const size_t total = static_cast<size_t>((M + 31) / 32) * step128(K) * 32u;
    hipLaunchKernelGGL(k_rows_to_tiles, dim3(static_cast<unsigned>(std::min<size_t>((total + 255) / 256, 4096))), dim3(256), 0, st, rows,
                       static_cast<unsigned long long>(words), M, K, tiles);
    HIP_TRY(hipGetLastError());
    return QGTC_OK;
}

int qgtc_launch_rbw_chain(const qgtc_problem *p1, const qgtc_problem *p2, int count, int max_M, int N1, int N2, int t_bits, int act_bits,
                          int out_bits, int mode2, const uint32_t *w2_codes, bool a_tiles, hipStream_t st) {
    // One kernel family per WIDTH of the chain: OB = the planes of the re-quantised aggregate (= of W': a chain has one width), OB2 =
    // those of T'. T's own planes only select the format class - 1- and 2-bit T are the same codes (one base-4 digit a nibble), 3- and
    // 4-bit T two digits a nibble - which rbw_chain_ok has checked against the aggregate's.
    const int per = getenv_flag("QGTC_NO_XCD") ? 0 : 1, tiles = a_tiles ? 1 : 0;   // (row blocks of a batch on one XCD)
    const dim3 grid(step128(max_M), count), block(256);
    const int gx = static_cast<int>(grid.x), gy = static_cast<int>(grid.y);
    const u32x4 *wc = reinterpret_cast<const u32x4 *>(w2_codes);
    const int c1 = (N1 + 31) / 32, c2 = mode2 == 0 ? 1 : (N2 + 31) / 32;
    if (c1 < 1 || c1 > 4 || c2 < 1 || c2 > 4 || mode2 < 0 || mode2 > 2) return QGTC_EINVAL;
    const int ob = mode2 == 0 ? t_bits : act_bits;   // (the float32 aggregation re-quantises nothing: the format class of T is all it needs)
    (void)out_bits;                                  // (== act_bits: rbw_chain_ok)
#define QGTC_RBW_GO(OB_, MODE2_, C1_, C2_) QGTC_LAUNCH((k_rbw_chain<OB_, OB_, MODE2_, C1_, C2_>), grid, block, 0, st, p1, p2, wc, per, tiles, gx, gy)
    // column blocks 1 / 2 / 4 (three run as four: lines past N are zeros, columns past N are not stored); the 2-bit family - the
    // BASELINE epoch's - keeps its own three
#define QGTC_RBW_C2(OB_, MODE2_, C1_)                                    \
    switch (c2) {                                                        \
        case 1: QGTC_RBW_GO(OB_, MODE2_, C1_, 1); break;                 \
        case 2: QGTC_RBW_GO(OB_, MODE2_, C1_, 2); break;                 \
        case 3: QGTC_RBW_GO(OB_, MODE2_, C1_, (OB_ == 2 ? 3 : 4)); break; \
        default: QGTC_RBW_GO(OB_, MODE2_, C1_, 4); break;                \
    }
#define QGTC_RBW_C1(OB_, MODE2_)                                         \
    switch (c1) {                                                        \
        case 1: QGTC_RBW_C2(OB_, MODE2_, 1) break;                       \
        case 2: QGTC_RBW_C2(OB_, MODE2_, 2) break;                       \
        case 3: QGTC_RBW_C2(OB_, MODE2_, (OB_ == 2 ? 3 : 4)) break;      \
        default: QGTC_RBW_C2(OB_, MODE2_, 4) break;                      \
    }
#define QGTC_RBW_M0(OB_)                                                 \
    switch (c1) {                                                        \
        case 1: QGTC_RBW_GO(OB_, 0, 1, 1); break;                        \
        case 2: QGTC_RBW_GO(OB_, 0, 2, 1); break;                        \
        case 3: QGTC_RBW_GO(OB_, 0, (OB_ == 2 ? 3 : 4), 1); break;       \
        default: QGTC_RBW_GO(OB_, 0, 4, 1); break;                       \
    }
#define QGTC_RBW_M12(OB_)                                                \
    if (mode2 == 1) {                                                    \
        QGTC_RBW_C1(OB_, 1)                                              \
    } else {                                                             \
        QGTC_RBW_C1(OB_, 2)                                              \
    }
    if (mode2 == 0) {   // (the float32 aggregation re-quantises nothing: one kernel per format class of T)
        if (ob <= 2) { QGTC_RBW_M0(2) } else { QGTC_RBW_M0(4) }
    } else {
        switch (ob) {
            case 1: QGTC_RBW_M12(1) break;
            case 2: QGTC_RBW_M12(2) break;
            case 3: QGTC_RBW_M12(3) break;
            default: QGTC_RBW_M12(4) break;
        }
    }
#undef QGTC_RBW_M12
#undef QGTC_RBW_M0
#undef QGTC_RBW_C1
#undef QGTC_RBW_C2
#undef QGTC_RBW_GO
    HIP_TRY(launch_status());
    return QGTC_OK;
}
