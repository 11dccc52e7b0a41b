// bitmm_fp4_skinny.hip.h — part of libqgtc_hip.so (included by qgtc_hip.hip, one translation unit).
// The bit-GEMM on the matrix cores for NARROW right operands (the reference's benchmark shapes: N <= 64)
// with 1- or 2-bit operands: no LDS staging, no barrier in the main loop.
#pragma once

namespace {

// ------------------------------------------------------------------------------------------
// v_mfma_scale_f32_16x16x128_f8f6f4 multiplies 16 lines x 128 elements of K per instruction and lane l
// supplies, for both operands, 32 elements (16 bytes of E2M1 codes) of line l & 15: exactly the expansion of
// ONE packed 32-bit word. Which 128 elements of K an instruction covers is irrelevant as long as X and W agree
// (the sum over k commutes), so lane (line, g = l >> 4) simply loads the 16 bytes of k-quad 4S + g of its X row
// and of its four W lines - 64 contiguous bytes per line and load instruction - and instruction t = 0..3 of
// the super-step takes word t of every lane's 16 bytes, expanded in registers (expand_word_fp4: a shift and an
// AND per plane and dword; the codes 0..3 mean 0, 0.5, 1, 1.5 and the E8M0 scale 2 on both operands makes the
// float32 sum the integer product, see bitmm_mfma.hip.h). Nothing is shared between lanes: no LDS staging and
// no barrier until the end.
//
// A workgroup owns 16 rows x 64 columns for the whole K; its eight waves split K (wave v takes the super-steps
// of 512 bits v, v+8, ..), keep two super-steps of packed words in flight and are summed through LDS once
// (float32 adds of exact integers). 4096 x 4096 x 64: 256 workgroups, one super-step = 16 MFMAs per wave, two waves per SIMD.
// All-zero 16-row x 512-bit X tiles are skipped with one ballot.
// Rows-layout bits (mode 0) and float32 (mode 2) only: a cols-layout word spans 32 rows, i.e. two workgroups.
// Needs a, w <= 2 and K (2^a - 1)(2^w - 1) < 2^24 (fp4_ok).
// ------------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int SK_ROWS = 16, SK_COLS = 64, SK_DEPTH = 2;   // tile, super-steps (4 k-quads) of packed words in flight per wave
constexpr int SK_WAVES = 8;                                // waves per workgroup = in-workgroup split-K factor
constexpr int SK_PITCH = 68;                              // floats between the rows of a wave's partial tile in LDS

template <int NA, int NW, int MODE, bool ZS>
__global__ __launch_bounds__(64 * SK_WAVES) void k_bitmm_fp4_skinny(qgtc_problem pr, MMShape sh) {
    static_assert(NA >= 1 && NA <= 2 && NW >= 1 && NW <= 2, "FP4 codes hold 2-bit values at most");
    __shared__ __attribute__((aligned(16))) float part[SK_WAVES][SK_ROWS][SK_PITCH];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef QGTC_STAMPS
    unsigned long long st_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define SK_STAMP(i) st_[i] = __builtin_amdgcn_s_memtime()
#else
#define SK_STAMP(i) do { } while (0)
#endif
    SK_STAMP(0);
    const int li = lane & 15, g = lane >> 4;       // line within the 16-line fragment, word of the k-quad
    const int M = pr.M, K = pr.K, N = pr.N;
    const int m0 = blockIdx.x * SK_ROWS, n0 = blockIdx.y * SK_COLS;
    const int kq = step128(K);
    const uint32_t kw = static_cast<uint32_t>(kq) * 4u;
    const uint32_t x_plane = static_cast<uint32_t>(pad8(M)) * kw, w_plane = static_cast<uint32_t>(pr.w_lines) * kw;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint32_t *>(pr.X), 0, static_cast<int>(static_cast<uint32_t>(pr.x_words) * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint32_t *>(pr.W), 0, static_cast<int>(static_cast<uint32_t>(pr.w_words) * 4u), 0x00020000);
    // byte offsets of the lane's lines; lines outside the matrix read as zero (offset 0xffffffff)
    const bool x_ok = m0 + li < M;
    const uint32_t x_off = static_cast<uint32_t>(m0 + li) * kw * 4u;
    uint32_t w_off[4];
    bool w_ok[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        w_ok[j] = n0 + 16 * j + li < N;
        w_off[j] = static_cast<uint32_t>(n0 + 16 * j + li) * kw * 4u;
    }
    const int nss = (kq + 3) / 4;                       // super-steps of four k-quads
    const int ns = nss > wv ? (nss - wv + SK_WAVES - 1) / SK_WAVES : 0;   // this wave's: wv, wv + SK_WAVES, ..

    u32x4 xr[SK_DEPTH][NA], wr[SK_DEPTH][4][NW];
    auto issue = [&](int s, u32x4 (&xd)[NA], u32x4 (&wd)[4][NW]) {  // unconditional: exact vmcnt waits
        const int q = 4 * (wv + SK_WAVES * s) + g;     // the lane's k-quad of the super-step
        const bool in = s < ns && q < kq;
        const uint32_t ko = static_cast<uint32_t>(q) * 16u;
#pragma unroll
        for (int p = 0; p < NA; p++)
            xd[p] = __builtin_amdgcn_raw_buffer_load_b128(rx, (in && x_ok) ? static_cast<uint32_t>(p) * x_plane * 4u + x_off + ko : 0xffffffffu, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int p = 0; p < NW; p++)
                wd[j][p] = __builtin_amdgcn_raw_buffer_load_b128(rw, (in && w_ok[j]) ? static_cast<uint32_t>(p) * w_plane * 4u + w_off[j] + ko : 0xffffffffu, 0, 0);
    };
#pragma unroll
    for (int d = 0; d < SK_DEPTH; d++) issue(d, xr[d], wr[d]);
    SK_STAMP(1);

    f32x4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; j++) acc[j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

    for (int s0 = 0; s0 < ns; s0 += SK_DEPTH) {
#pragma unroll
        for (int d = 0; d < SK_DEPTH; d++) {
            if (s0 + d >= ns) break;
            uint32_t any = (xr[d][0].x | xr[d][0].y) | (xr[d][0].z | xr[d][0].w);
            if (NA > 1) any |= (xr[d][1].x | xr[d][1].y) | (xr[d][1].z | xr[d][1].w);
            const bool live = !ZS || __ballot(any != 0u) != 0ull;   // wave-uniform
            if (s0 + d == 0) SK_STAMP(2);
            if (live) {
#pragma unroll
                for (int t = 0; t < 4; t++) {   // word t of every lane's 16 bytes
                    uint32_t xw[NA], xe[4];
#pragma unroll
                    for (int p = 0; p < NA; p++) xw[p] = xr[d][p][t];
                    expand_word_fp4<NA>(xw, NA, xe);
                    const i32x8 a8 = {static_cast<int>(xe[0]), static_cast<int>(xe[1]), static_cast<int>(xe[2]), static_cast<int>(xe[3]), 0, 0, 0, 0};
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        uint32_t ww[NW], we[4];
#pragma unroll
                        for (int p = 0; p < NW; p++) ww[p] = wr[d][j][p][t];
                        expand_word_fp4<NW>(ww, NW, we);
                        const i32x8 b8 = {static_cast<int>(we[0]), static_cast<int>(we[1]), static_cast<int>(we[2]), static_cast<int>(we[3]), 0, 0, 0, 0};
                        // cbsz = blgp = 4: E2M1 operands; E8M0 scale 128 = x2 on each: the code v counts as v
                        acc[j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8, b8, acc[j], 4, 4, 0, 128, 0, 128);
                    }
                }
            }
            issue(s0 + d + SK_DEPTH, xr[d], wr[d]);
        }
    }

    SK_STAMP(3);
    // ---- reduce the four waves' partial tiles. MFMA 16 x 16 C/D layout: col = lane & 15, row = 4 (lane >> 4) + reg
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
        for (int r = 0; r < 4; r++) part[wv][4 * g + r][16 * j + li] = acc[j][r];
    __syncthreads();
    SK_STAMP(4);
    // thread t < 256: row t >> 4, four consecutive columns 4 (t & 15) ..; the upper waves are done
    if (tid >= 256) return;
    const int row = tid >> 4, quad = tid & 15;
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
    const int extra = (blockIdx.y == gridDim.y - 1 && quad == 8) ? row_words - word - 1 : 0;
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
