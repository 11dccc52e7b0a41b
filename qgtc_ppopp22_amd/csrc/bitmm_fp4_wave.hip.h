// bitmm_fp4_wave.hip.h — part of libqgtc_hip.so (included by qgtc_hip.hip, one translation unit).
// The bit-GEMM on the matrix cores for GROUPED launches over cluster batches with narrow outputs: one wave per tile.
#pragma once

namespace {

// ------------------------------------------------------------------------------------------
// Cluster batches are many small products (n ~ 600 .. 1200 rows, K = 128 or n, N = 10 .. 128): per tile there are
// one to three super-steps of 512 bits of K, too few to split among waves, and every workgroup-level
// mechanism (staging buffers, barriers, cross-wave reductions) is pure overhead. Here a workgroup IS one wave:
// the 16 x 16 x 128 FP4 MFMA scheme of bitmm_fp4_skinny.hip.h (every lane expands the packed words of its own
// lines straight into its fragment registers, K order free, base-4 digits for more than two planes) on a
// 32 x 32 tile, up to two super-steps of packed words in flight, and an epilogue straight from the accumulators:
// MFMA 16 x 16 C/D layout: lane (li = l & 15, g = l >> 4) holds column 16 j + li and rows 16 i + 4 g + r (r = 0..3).
//   rows layout : the 32 columns of a row live in the 16 lanes of one DPP row (two per lane): one 16-lane OR;
//   cols layout : the 32 rows of a column live in the 4 lanes li, li+16, li+32, li+48 (eight per lane): two
//                 half-wave swaps (v_permlane16_swap, v_permlane32_swap);
//   float32     : 16 consecutive columns per store.
// With an occupancy bitmap (qgtc_tile_occupancy: one bit per 32-row tile and k-quad) super-steps whose four
// k-quads are all empty are neither loaded nor multiplied. Needs a <= 4, w <= 8, float32 sums exact
// (K (2^a - 1)(2^w - 1) < 2^24).
// ------------------------------------------------------------------------------------------
// RF x CF fragments of 16 lines per wave (2 x 2 = 32 x 32 outputs is what is launched; 4 x 4 was measured: with
// 2850 waves for an epoch stage the chip is not filled and every wave's long serial chain is exposed).
template <int NA, int NW, int MODE, int RF, int CF, bool PUB = false, bool WCOH = false>
__device__ __forceinline__ void fw_tile(const qgtc_problem &pr, const MMShape &sh, int zero_skip, int tile) {
    static_assert(RF % 2 == 0 && CF % 2 == 0, "whole 32-bit output words");
    constexpr int TR = 16 * RF, TC = 16 * CF;
    constexpr int DEPTH = (RF * NA + CF * NW) <= 12 ? 2 : 1;
    constexpr int NDA = (NA + 1) / 2, NDW = (NW + 1) / 2;  // base-4 digits
    const int M = pr.M, K = pr.K, N = pr.N;
    const int tiles_m = (M + TR - 1) / TR, tiles_n = (N + TC - 1) / TC;
    if (tile >= tiles_m * tiles_n) return;
    const int tm = tile / tiles_n, tn = tile % tiles_n;
    const int lane = threadIdx.x, li = lane & 15, g = lane >> 4;
    const int m0 = tm * TR, n0 = tn * TC;
    const int kq = step128(K);
    const uint32_t kw = static_cast<uint32_t>(kq) * 4u;
    const uint32_t x_plane = static_cast<uint32_t>(pad8(M)) * kw, w_plane = static_cast<uint32_t>(pr.w_lines) * kw;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint32_t *>(pr.X), 0, static_cast<int>(static_cast<uint32_t>(pr.x_words) * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint32_t *>(pr.W), 0, static_cast<int>(static_cast<uint32_t>(pr.w_words) * 4u), 0x00020000);
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
    // the super-steps (four k-quads) this tile visits: all of them, or those with an occupied k-quad
    const int nss = (kq + 3) >> 2;
    const bool jump = pr.occ != nullptr && pr.occ_words == 1;   // K <= 8192
    unsigned long long live = 0ull;  // bit S: super-step S has an occupied k-quad
    if (jump) {
        unsigned long long o = 0ull;   // the bitmap has one word per 32-row tile
        const int rt_last = ((M + 31) >> 5) - 1;
#pragma unroll
        for (int k = 0; k < TR / 32; k++) o |= pr.occ[min(tm * (TR / 32) + k, rt_last)];
        // OR the four k-quad bits of every nibble into its lowest bit, then compact is not needed: test per step
        live = (o | (o >> 1) | (o >> 2) | (o >> 3)) & 0x1111111111111111ull;
    }
    int s_next = 0;
    auto next_s = [&]() -> int {   // next super-step to load (nss when exhausted)
        while (s_next < nss && jump && !((live >> (4 * s_next)) & 1ull)) s_next++;
        return s_next < nss ? s_next++ : nss;
    };

    u32x4 xr[DEPTH][RF][NA], wr[DEPTH][CF][NW];
    auto issue = [&](u32x4 (&xd)[RF][NA], u32x4 (&wd)[CF][NW]) -> bool {   // unconditional loads: exact vmcnt waits
        const int S = next_s();
        const int q = 4 * S + g;
        const bool in = S < nss && q < kq;
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
                wd[j][p] = __builtin_amdgcn_raw_buffer_load_b128(rw, (in && w_ok[j] && p < sh.w) ? static_cast<uint32_t>(p) * w_plane * 4u + w_off[j] + ko : 0xffffffffu, 0, WCOH ? AUX_SC1 : 0);
        return S < nss;
    };
    bool have[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; d++) have[d] = issue(xr[d], wr[d]);

    f32x4 acc[RF][CF];
#pragma unroll
    for (int i = 0; i < RF; i++)
#pragma unroll
        for (int j = 0; j < CF; j++) acc[i][j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

    while (have[0]) {   // the slots are consumed and refilled in order: slot 0 empty = the stream is exhausted
#pragma unroll
        for (int d = 0; d < DEPTH; d++) {
            if (!have[d]) break;
            uint32_t any = 0u;
#pragma unroll
            for (int i = 0; i < RF; i++)
#pragma unroll
                for (int p = 0; p < NA; p++) any |= (xr[d][i][p].x | xr[d][i][p].y) | (xr[d][i][p].z | xr[d][i][p].w);
            if (!zero_skip || __ballot(any != 0u) != 0ull) {   // wave-uniform
#pragma unroll
                for (int t = 0; t < 4; t++) {   // word t of every lane's 16 bytes
                    i32x8 a8[RF][NDA];
#pragma unroll
                    for (int i = 0; i < RF; i++)
#pragma unroll
                        for (int da = 0; da < NDA; da++) {
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
                                    acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8[i][da], b8, acc[i][j], 4, 4, 0, 128 + 2 * da, 0, 128 + 2 * dw);
                        }
                }
            }
            have[d] = issue(xr[d], wr[d]);
        }
    }

    // ---- epilogue from the accumulators: element (row 16 i + 4 g + r, column 16 j + li) in acc[i][j][r]
    if (MODE == 2) {  // float32 [M,N] (reference kernel.h:915-930)
        float *outf = static_cast<float *>(pr.out);
#pragma unroll
        for (int i = 0; i < RF; i++)
#pragma unroll
            for (int j = 0; j < CF; j++) {
                const int col = n0 + 16 * j + li;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int row = m0 + 16 * i + 4 * g + r;
                    if (row < M && col < N) outf[static_cast<size_t>(row) * N + col] = acc[i][j][r];
                }
            }
        return;
    }
    const int maxi = 1 << (sh.ob & 31);
    const bool int_rq = sh.ob <= 23;  // float(c) > 2^ob  <=>  c > 2^ob for every int c >= 0
    uint32_t q[RF][CF][4];
#pragma unroll
    for (int i = 0; i < RF; i++)
#pragma unroll
        for (int j = 0; j < CF; j++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int c = static_cast<int>(acc[i][j][r]);   // exact: the sums are integers below 2^24
                const int v = int_rq ? (c > maxi ? maxi - 1 : c) : requant(c, sh.maxv, sh.maxm1);
                q[i][j][r] = (m0 + 16 * i + 4 * g + r < M && n0 + 16 * j + li < N) ? static_cast<uint32_t>(v) : 0u;
            }
    uint32_t *out = static_cast<uint32_t *>(pr.out);
    if (MODE == 0) {  // rows layout [ob][PAD8(M)][STEP128(N)*4] (reference kernel.h:357-389): word (m, n / 32)
        const int rows_pad = pad8(M), row_words = step128(N) * 4;
        const size_t oplane = static_cast<size_t>(rows_pad) * row_words;
        const int word0 = n0 >> 5;
        // the last column tile zeroes the words past it
        const int extra = tn == tiles_n - 1 ? row_words - word0 - CF / 2 : 0;
#pragma unroll
        for (int i = 0; i < RF; i++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int m = m0 + 16 * i + 4 * g + r;
                uint32_t *dst = out + static_cast<size_t>(m) * row_words + word0;
                for (int p = 0; p < sh.ob; p++, dst += oplane) {
#pragma unroll
                    for (int jw = 0; jw < CF / 2; jw++) {
                        // column 32 jw + 16 jj + li at bit 31 - 16 jj - li of word jw
                        const uint32_t x = (((q[i][2 * jw][r] >> p) & 1u) << 16 | ((q[i][2 * jw + 1][r] >> p) & 1u)) << (15 - li);
                        const uint32_t wrd = or_reduce_group<2>(x);   // the 16 lanes of the DPP row
                        if (li == 0 && m < rows_pad && word0 + jw < row_words) st_word<PUB>(dst + jw, wrd);
                    }
                    if (li == 0 && m < rows_pad)
                        for (int e = 0; e < extra; e++) st_word<PUB>(dst + CF / 2 + e, 0u);
                }
            }
    } else {  // cols layout [ob][PAD128(N)][STEP128(M)*4] (intended semantics of kernel.h:651-810): word (n, m / 32)
        const int lines = pad128(N), line_words = step128(M) * 4;
        const size_t oplane = static_cast<size_t>(lines) * line_words;
        const int word0 = m0 >> 5;
#pragma unroll
        for (int j = 0; j < CF; j++) {
            const int n = n0 + 16 * j + li;
            uint32_t *dst = out + static_cast<size_t>(n) * line_words + word0;
            for (int p = 0; p < sh.ob; p++, dst += oplane) {
#pragma unroll
                for (int iw = 0; iw < RF / 2; iw++) {
                    // row 32 iw + 16 ii + 4 g + r at bit 31 - 16 ii - 4 g - r of word iw
                    uint32_t x = 0u;
#pragma unroll
                    for (int ii = 0; ii < 2; ii++)
#pragma unroll
                        for (int r = 0; r < 4; r++) x |= ((q[2 * iw + ii][j][r] >> p) & 1u) << (31 - 16 * ii - r);
                    x >>= 4 * g;
                    const auto s16 = __builtin_amdgcn_permlane16_swap(x, x, false, false);   // rows of 16 lanes: (0,1), (2,3)
                    x = s16[0] | s16[1];
                    x = or_with_partner_half(x);                                            // halves of the wave
                    if (g == 0 && n < lines && word0 + iw < line_words) st_word<PUB>(dst + iw, x);
                }
            }
        }
        // zero what no tile computes: words past the last row tile, lines past the last column tile
        const bool last_m = tm == tiles_m - 1, last_n = tn == tiles_n - 1;
        const int w_core1 = min(line_words, word0 + TR / 32);
        if (last_m && w_core1 < line_words) {
            for (int e = lane; e < sh.ob * TC; e += 64) {
                const int line = n0 + e % TC, p = e / TC;
                if (line < lines)
                    for (int wi = w_core1; wi < line_words; wi++) st_word<PUB>(out + p * oplane + static_cast<size_t>(line) * line_words + wi, 0u);
            }
        }
        if (last_n && n0 + TC < lines) {
            const int nl = lines - (n0 + TC), w_end = last_m ? line_words : w_core1;
            for (int e = lane; e < sh.ob * nl; e += 64) {
                const int line = n0 + TC + e % nl, p = e / nl;
                for (int wi = word0; wi < w_end; wi++) st_word<PUB>(out + p * oplane + static_cast<size_t>(line) * line_words + wi, 0u);
            }
        }
    }
}

template <int NA, int NW, int MODE, int RF, int CF>
__global__ __launch_bounds__(64) void k_bitmm_fp4_wave(const qgtc_problem *__restrict__ prs, MMShape sh, int zero_skip) {
    const qgtc_problem pr = prs[blockIdx.y];
    fw_tile<NA, NW, MODE, RF, CF>(pr, sh, zero_skip, static_cast<int>(blockIdx.x));
}

}  // namespace
