// bitmm_fp4_skinny.hip.h — part of libqgtc_hip.so (included by qgtc_hip.hip, one translation unit).
// The bit-GEMM on the matrix cores for NARROW right operands (the reference's benchmark shapes: N <= 64;
// used up to N = 256): no LDS staging, no barrier in the main loop.
#pragma once

namespace {

// ------------------------------------------------------------------------------------------
// v_mfma_scale_f32_16x16x128_f8f6f4 multiplies 16 lines x 128 elements of K per instruction and lane l
// supplies, for both operands, 32 elements (16 bytes of E2M1 codes) of line l & 15: exactly the expansion of
// ONE packed 32-bit word. Which 128 elements of K an instruction covers is irrelevant as long as X and W agree
// (the sum over k commutes), so lane (line, g = l >> 4) simply loads the 16 bytes of k-quad 4S + g of its X rows
// and of its W lines - 64 contiguous bytes per line and load instruction - and instruction t = 0..3 of
// the super-step takes word t of every lane's 16 bytes, expanded in registers (expand_word_fp4: a shift and an
// AND per plane and dword; the codes 0..3 mean 0, 0.5, 1, 1.5 and the E8M0 scale 2 on both operands makes the
// float32 sum the integer product, see bitmm_mfma.hip.h). Nothing is shared between lanes: no LDS staging and
// no barrier until the end.
//
// A workgroup owns a 32 x 32 (N > 32) or 16 x 32 tile for the whole K; its eight waves split K (wave v takes
// the super-steps of 512 bits v, v+8, ..), keep up to two super-steps of packed words in flight and are summed
// through LDS once (float32 adds of exact integers). 4096 x 4096 x 64: 256 workgroups, one super-step = 16 MFMAs
// per wave, two waves per SIMD.
// All-zero (tile rows) x 512-bit X tiles are skipped with one ballot.
// All three output forms; cols-layout bits (mode 1) always on 32 x 32 tiles (a word is 32 rows of a column).
//
// More than two planes: an operand is taken two planes at a time - base-4 "digits", each again the codes 0..3 -
// and digit d is multiplied with the E8M0 scale 2 * 4^d, so sum_d 4^d (X . W_d) accumulates in the same float32
// registers, exactly while K (2^a - 1)(2^w - 1) < 2^24 (skinny_ok). NA, NW are plane CAPACITIES (planes beyond
// sh.a / sh.w are not loaded); one MFMA per pair of digits instead of one AND + popcount pass per pair of planes.
// ------------------------------------------------------------------------------------------
constexpr int SK_WAVES = 8;   // waves per workgroup = in-workgroup split-K factor

// RF x CF fragments of 16 lines: the workgroup's tile is 16 RF rows x 16 CF columns. 2 x 2 (32 x 32) when N > 32:
// every W line is then expanded (and fetched from L2) by M / 32 workgroups instead of M / 16, which is what
// matters from four planes up; 1 x 2 (16 x 32) for N <= 32 keeps 256 workgroups at M = 4096.
template <int NA, int NW, int MODE, int RF, int CF>
__global__ __launch_bounds__(64 * SK_WAVES) void k_bitmm_fp4_skinny(qgtc_problem pr, MMShape sh, int zero_skip) {
    static_assert(NA >= 1 && NA <= 8 && NW >= 1 && NW <= 8, "plane capacities");
    constexpr int TR = 16 * RF, TC = 16 * CF, PITCH = TC + 4;   // tile; floats between the rows of a partial tile in LDS
    constexpr int SK_DEPTH = RF * NA + CF * NW <= 10 ? 2 : 1;    // super-steps (4 k-quads) of packed words in flight per wave
    constexpr int NDA = (NA + 1) / 2, NDW = (NW + 1) / 2;        // base-4 digits
    constexpr int QPR = TC / 4, ETHREADS = TR * QPR;             // epilogue: quads per row, threads
    static_assert(ETHREADS <= 64 * SK_WAVES && (TC == 32 || TC == 64), "tile");
    __shared__ __attribute__((aligned(16))) float part[SK_WAVES][TR][PITCH];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef QGTC_STAMPS
    unsigned long long st_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define SK_STAMP(i) st_[i] = __builtin_amdgcn_s_memtime()
#else
#define SK_STAMP(i) do { } while (0)
#endif
    SK_STAMP(0);
    const int li = lane & 15, g = lane >> 4;       // line within the 16-line fragment, k-quad of the super-step
    const int M = pr.M, K = pr.K, N = pr.N;
    const int m0 = blockIdx.x * TR, n0 = blockIdx.y * TC;
    const int kq = step128(K);
    const uint32_t kw = static_cast<uint32_t>(kq) * 4u;
    const uint32_t x_plane = static_cast<uint32_t>(pad8(M)) * kw, w_plane = static_cast<uint32_t>(pr.w_lines) * kw;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint32_t *>(pr.X), 0, static_cast<int>(static_cast<uint32_t>(pr.x_words) * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint32_t *>(pr.W), 0, static_cast<int>(static_cast<uint32_t>(pr.w_words) * 4u), 0x00020000);
    // byte offsets of the lane's lines; lines outside the matrix read as zero (offset 0xffffffff)
    uint32_t x_off[RF], w_off[CF];
    bool x_ok[RF], w_ok[CF];
#pragma unroll
    for (int i = 0; i < RF; i++) {
        x_ok[i] = m0 + 16 * i + li < M;
        x_off[i] = static_cast<uint32_t>(m0 + 16 * i + li) * kw * 4u;
    }
#pragma unroll
    for (int j = 0; j < CF; j++) {
        w_ok[j] = n0 + 16 * j + li < N;
        w_off[j] = static_cast<uint32_t>(n0 + 16 * j + li) * kw * 4u;
    }
    const int nss = (kq + 3) / 4;                                        // super-steps of four k-quads
    const int ns = nss > wv ? (nss - wv + SK_WAVES - 1) / SK_WAVES : 0;  // this wave's: wv, wv + SK_WAVES, ..

    u32x4 xr[SK_DEPTH][RF][NA], wr[SK_DEPTH][CF][NW];
    auto issue = [&](int s, u32x4 (&xd)[RF][NA], u32x4 (&wd)[CF][NW]) {  // unconditional: exact vmcnt waits
        const int q = 4 * (wv + SK_WAVES * s) + g;     // the lane's k-quad of the super-step
        const bool in = s < ns && q < kq;
        const uint32_t ko = static_cast<uint32_t>(q) * 16u;
#pragma unroll
        for (int i = 0; i < RF; i++)
#pragma unroll
            for (int p = 0; p < NA; p++)
                xd[i][p] = __builtin_amdgcn_raw_buffer_load_b128(rx, (in && x_ok[i] && p < sh.a) ? static_cast<uint32_t>(p) * x_plane * 4u + x_off[i] + ko : 0xffffffffu, 0, 0);
#pragma unroll
        for (int j = 0; j < CF; j++)
#pragma unroll
            for (int p = 0; p < NW; p++)
                wd[j][p] = __builtin_amdgcn_raw_buffer_load_b128(rw, (in && w_ok[j] && p < sh.w) ? static_cast<uint32_t>(p) * w_plane * 4u + w_off[j] + ko : 0xffffffffu, 0, 0);
    };
#pragma unroll
    for (int d = 0; d < SK_DEPTH; d++) issue(d, xr[d], wr[d]);
    SK_STAMP(1);

    f32x4 acc[RF][CF];
#pragma unroll
    for (int i = 0; i < RF; i++)
#pragma unroll
        for (int j = 0; j < CF; j++) acc[i][j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

    for (int s0 = 0; s0 < ns; s0 += SK_DEPTH) {
#pragma unroll
        for (int d = 0; d < SK_DEPTH; d++) {
            if (s0 + d >= ns) break;
            uint32_t any = 0u;
#pragma unroll
            for (int i = 0; i < RF; i++)
#pragma unroll
                for (int p = 0; p < NA; p++) any |= (xr[d][i][p].x | xr[d][i][p].y) | (xr[d][i][p].z | xr[d][i][p].w);
            const bool live = !zero_skip || __ballot(any != 0u) != 0ull;   // wave-uniform
            if (s0 + d == 0) SK_STAMP(2);
            if (live) {
#pragma unroll
                for (int t = 0; t < 4; t++) {   // word t of every lane's 16 bytes
                    i32x8 a8[RF][NDA];
#pragma unroll
                    for (int i = 0; i < RF; i++)
#pragma unroll
                        for (int da = 0; da < NDA; da++) {   // digit da of X: planes 2 da, 2 da + 1
                            uint32_t xw[2], xe[4];
                            xw[0] = xr[d][i][2 * da][t];
                            xw[1] = 2 * da + 1 < NA ? xr[d][i][(2 * da + 1) % NA][t] : 0u;
                            expand_word_fp4<2>(xw, 2, xe);
                            a8[i][da] = i32x8{static_cast<int>(xe[0]), static_cast<int>(xe[1]), static_cast<int>(xe[2]), static_cast<int>(xe[3]), 0, 0, 0, 0};
                        }
#pragma unroll
                    for (int j = 0; j < CF; j++)
#pragma unroll
                        for (int dw = 0; dw < NDW; dw++) {
                            uint32_t ww[2], we[4];
                            ww[0] = wr[d][j][2 * dw][t];
                            ww[1] = 2 * dw + 1 < NW ? wr[d][j][(2 * dw + 1) % NW][t] : 0u;
                            expand_word_fp4<2>(ww, 2, we);
                            const i32x8 b8 = {static_cast<int>(we[0]), static_cast<int>(we[1]), static_cast<int>(we[2]), static_cast<int>(we[3]), 0, 0, 0, 0};
#pragma unroll
                            for (int i = 0; i < RF; i++)
#pragma unroll
                                for (int da = 0; da < NDA; da++)
                                    // cbsz = blgp = 4: E2M1 operands; E8M0 scales 2 * 4^digit: the code v of digit d counts as v 4^d
                                    acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8[i][da], b8, acc[i][j], 4, 4, 0, 128 + 2 * da, 0, 128 + 2 * dw);
                        }
                }
            }
            issue(s0 + d + SK_DEPTH, xr[d], wr[d]);
        }
    }

    SK_STAMP(3);
    // ---- reduce the waves' partial tiles. MFMA 16 x 16 C/D layout: col = lane & 15, row = 4 (lane >> 4) + reg
    if (MODE == 1) {
        // cols layout: a word is 32 ROWS of a column - the tile (32 x 32, always) is kept transposed in LDS, so that
        // an epilogue thread reads four consecutive rows of its column with one 16-byte read
        static_assert(MODE != 1 || (RF == 2 && CF == 2), "cols-layout words need whole 32-row tiles");
#pragma unroll
        for (int i = 0; i < RF; i++)
#pragma unroll
            for (int j = 0; j < CF; j++) *reinterpret_cast<f32x4 *>(&part[wv][16 * j + li][16 * i + 4 * g]) = acc[i][j];
        __syncthreads();
        if (tid >= 256) return;
        const int col = tid >> 3, rq = tid & 7;   // column of the tile, rows 4 rq .. 4 rq + 3
        f32x4 sum1 = *reinterpret_cast<const f32x4 *>(&part[0][col][4 * rq]);
#pragma unroll
        for (int v = 1; v < SK_WAVES; v++) {
            const f32x4 t = *reinterpret_cast<const f32x4 *>(&part[v][col][4 * rq]);
            sum1 += t;
        }
        const int maxi1 = 1 << (sh.ob & 31);
        const bool int_rq1 = sh.ob <= 23;
        uint32_t q1[4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const int c = static_cast<int>(sum1[e]);   // exact: the sums are integers below 2^24
            const int r = int_rq1 ? (c > maxi1 ? maxi1 - 1 : c) : requant(c, sh.maxv, sh.maxm1);
            q1[e] = (m0 + 4 * rq + e < M && n0 + col < N) ? static_cast<uint32_t>(r) : 0u;
        }
        // cols layout [ob][PAD128(N)][STEP128(M)*4] (intended semantics of kernel.h:651-810): word (n, m / 32)
        const int lines = pad128(N), line_words = step128(M) * 4;
        const size_t oplane1 = static_cast<size_t>(lines) * line_words;
        const int word1 = m0 >> 5, n1 = n0 + col;
        uint32_t *out1 = static_cast<uint32_t *>(pr.out);
        uint32_t *dst1 = out1 + static_cast<size_t>(n1) * line_words + word1;
        for (int p = 0; p < sh.ob; p++, dst1 += oplane1) {
            const uint32_t nib = (((q1[0] >> p) & 1u) << 3) | (((q1[1] >> p) & 1u) << 2) | (((q1[2] >> p) & 1u) << 1) | ((q1[3] >> p) & 1u);
            const uint32_t wrd = or_reduce8(nib << (28u - 4u * static_cast<uint32_t>(rq)));   // row 4 rq + e at bit 31 - 4 rq - e
            if (rq == 0 && n1 < lines && word1 < line_words) dst1[0] = wrd;
        }
        // zero what no tile computes: words past the last row tile, lines past the last column tile
        // (the grid from the shape, as qgtc_launch_skinny sizes it - gridDim is a hidden kernel argument: bitmm_fp4_one.hip.h)
        const bool last_m = static_cast<int>(blockIdx.x) == (M + TR - 1) / TR - 1, last_n = static_cast<int>(blockIdx.y) == (N + 31) / 32 - 1;
        const int w_core1 = min(line_words, word1 + 1);
        if (last_m && w_core1 < line_words) {
            for (int e = tid; e < sh.ob * 32; e += 256) {
                const int line = n0 + (e & 31), p = e >> 5;
                if (line < lines)
                    for (int wi = w_core1; wi < line_words; wi++) out1[p * oplane1 + static_cast<size_t>(line) * line_words + wi] = 0u;
            }
        }
        if (last_n && n0 + 32 < lines) {
            const int nl = lines - (n0 + 32), w_end = last_m ? line_words : w_core1;
            for (int e = tid; e < sh.ob * nl; e += 256) {
                const int line = n0 + 32 + e % nl, p = e / nl;
                for (int wi = word1; wi < w_end; wi++) out1[p * oplane1 + static_cast<size_t>(line) * line_words + wi] = 0u;
            }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < RF; i++)
#pragma unroll
        for (int j = 0; j < CF; j++)
#pragma unroll
            for (int r = 0; r < 4; r++) part[wv][16 * i + 4 * g + r][16 * j + li] = acc[i][j][r];
    __syncthreads();
    SK_STAMP(4);
    // thread t < ETHREADS: row t / QPR, four consecutive columns 4 (t % QPR) ..; the other waves are done
    if (tid >= ETHREADS) return;
    const int row = tid / QPR, quad = tid % QPR;
    f32x4 sum = *reinterpret_cast<const f32x4 *>(&part[0][row][4 * quad]);
#pragma unroll
    for (int v = 1; v < SK_WAVES; v++) {
        const f32x4 t = *reinterpret_cast<const f32x4 *>(&part[v][row][4 * quad]);
        sum += t;
    }
    const int m = m0 + row, n = n0 + 4 * quad;
    if (MODE == 2) {  // float32 [M,N] (reference kernel.h:915-930)
        if (m < M) {
            float *dst = static_cast<float *>(pr.out) + static_cast<size_t>(m) * N + n;
            if (n + 3 < N && (N & 3) == 0) {
                *reinterpret_cast<f32x4 *>(dst) = sum;
            } else {
#pragma unroll
                for (int e = 0; e < 4; e++)
                    if (n + e < N) dst[e] = sum[e];
            }
        }
        return;
    }
    // rows layout [ob][PAD8(M)][STEP128(N)*4] (reference kernel.h:357-389): word (m, n / 32)
    const int maxi = 1 << (sh.ob & 31);
    const bool int_rq = sh.ob <= 23;  // float(c) > 2^ob  <=>  c > 2^ob for every int c >= 0
    uint32_t q[4];
#pragma unroll
    for (int e = 0; e < 4; e++) {
        const int c = static_cast<int>(sum[e]);   // exact: the sums are integers below 2^24
        const int r = int_rq ? (c > maxi ? maxi - 1 : c) : requant(c, sh.maxv, sh.maxm1);
        q[e] = (m < M && n + e < N) ? static_cast<uint32_t>(r) : 0u;
    }
    const int rows_pad = pad8(M), row_words = step128(N) * 4;
    const size_t oplane = static_cast<size_t>(rows_pad) * row_words;
    const bool store = (quad & 7) == 0 && m < rows_pad;
    const uint32_t sh_n = 28u - 4u * static_cast<uint32_t>(quad & 7);
    const int word = (n0 >> 5) + (quad >> 3);
    // the last column tile also zeroes the row words past it (the kernels write every word of the output)
    const int extra = (static_cast<int>(blockIdx.y) == (N + 31) / 32 - 1 && quad == QPR - 8) ? row_words - word - 1 : 0;
    uint32_t *dst = static_cast<uint32_t *>(pr.out) + static_cast<size_t>(m) * row_words + word;
    for (int p = 0; p < sh.ob; p++, dst += oplane) {
        const uint32_t nib = (((q[0] >> p) & 1u) << 3) | (((q[1] >> p) & 1u) << 2) | (((q[2] >> p) & 1u) << 1) | ((q[3] >> p) & 1u);
        const uint32_t wrd = or_reduce8(nib << sh_n);
        if (store) {
            if (word < row_words) dst[0] = wrd;
            for (int x = 1; x <= extra; x++) dst[x] = 0u;
        }
    }
#ifdef QGTC_STAMPS
    SK_STAMP(5);
    if (tid == 0 && blockIdx.x < 1024)
        for (int i = 0; i < 8; i++) g_stamps[blockIdx.x * 16 + i] = st_[i];
#endif
#undef SK_STAMP
}

}  // namespace
