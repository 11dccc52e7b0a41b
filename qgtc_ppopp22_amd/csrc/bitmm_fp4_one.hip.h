// bitmm_fp4_one.hip.h — part of libqgtc_hip.so (qgtc_fp4.hip).
// The bit-GEMM on the matrix cores for narrow right operands when K <= 4096 (every shape of the reference's
// micro-benchmark, 2_7c_QGTC_GEMM_INT8.py: M = K in 1024 / 2048 / 4096, N <= 64): the latency-trimmed form of
// bitmm_fp4_skinny.hip.h. Same arithmetic (E2M1 codes of the bit planes, v_mfma_scale_f32_16x16x128_f8f6f4, float32
// sums of exact integers, base-4 digits for more than two planes), same words out.
#pragma once

namespace {

// ------------------------------------------------------------------------------------------
// The 4096 x 4096 x 64 launch is ONE exposed latency chain per workgroup: kernel arguments ->
// addresses -> one round trip to memory -> 16 MFMAs -> cross-wave reduction -> epilogue. This kernel shortens every
// link that is ours to shorten:
//   * scalar kernel arguments (13 dwords) instead of two structs by value: with -mllvm -amdgpu-kernarg-preload-count
//     they arrive in SGPRs with the wave, no s_load round trip ahead of the first address;
//   * a wave owns exactly ONE super-step (k-quads 4 v .. 4 v + 3): no loop, no second set of (empty) loads, the
//     global loads are issued before anything else is computed;
//   * the MFMAs are issued with the operands swapped for the rows layout / float32 (a lane then owns four consecutive
//     COLUMNS of one row; for the cols layout it owns four consecutive rows of one column), partial tiles go to LDS in
//     lane order (4 linear ds_write_b128 per wave) and reducer wave f sums fragment f with 8 linear ds_read_b128;
//   * a fragment is 16 lines x 16 elements = one HALF of the 32-bit output words of its 16 lines: four lanes OR their
//     nibbles with two half-wave swaps (v_permlane16_swap, v_permlane32_swap) and 16 lanes store 16 bits each - the two
//     fragments of a word store its two halves, nothing is combined across waves;
//   * the waves that are not reducers zero the padding words meanwhile.
// One-plane operands are expanded with one AND per dword: MFMA s of the super-step takes the bits s, s + 4, s + 8 ..
// of all four words IN PLACE (nibble code 1 << s = 0.5, 1, 2) and the E8M0 scale 2^(1 - s) makes every product 1
// again (the fourth uses one shift: code 8 is the sign bit). 5 instead of 7 VALU operations per packed word.
// ------------------------------------------------------------------------------------------
constexpr int ONE_WAVES = 8;
constexpr int ONE_MAX_K = 4 * 128 * ONE_WAVES;   // 4096: one super-step per wave

// the four MFMA operand registers (dwords t = 0..3, one per packed word) for MFMA s = 0..3, and the E8M0 scale
// that makes a set bit count as 1 (one plane) / a 2-bit digit v count as v 4^digit
template <int NP>
__device__ __forceinline__ void one_expand(const u32x4 (&pl)[NP], int digit, uint32_t (&ops)[4][4], int (&scale)[4]) {
    if constexpr (NP == 1) {
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const uint32_t wd = pl[0][t];
            ops[0][t] = wd & 0x11111111u;          // code 1 = 0.5
            ops[1][t] = wd & 0x22222222u;          // code 2 = 1.0
            ops[2][t] = wd & 0x44444444u;          // code 4 = 2.0
            ops[3][t] = (wd >> 3) & 0x11111111u;   // (code 8 would be -0)
        }
        scale[0] = 128; scale[1] = 127; scale[2] = 126; scale[3] = 128;
    } else {   // the 2-bit code of (plane 2d, plane 2d + 1), built for the even and the odd bit positions of a word at once
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const uint32_t w0 = pl[2 * digit][t], w1 = 2 * digit + 1 < NP ? pl[(2 * digit + 1) % NP][t] : 0u;
            uint32_t ev = (w0 & 0x55555555u) | ((w1 & 0x55555555u) << 1), od = ((w0 >> 1) & 0x55555555u) | (w1 & 0xaaaaaaaau);
            // (opaque from here: hipcc otherwise re-derives ops[0] / ops[1] from the planes with masks folded in - two ANDs and a
            // three-input op each beside the v_bfi that already made ev / od: 14 instructions a word where 10 do)
            asm("" : "+v"(ev), "+v"(od));
            ops[0][t] = ev & 0x33333333u;
            ops[1][t] = od & 0x33333333u;
            ops[2][t] = (ev >> 2) & 0x33333333u;
            ops[3][t] = (od >> 2) & 0x33333333u;
        }
#pragma unroll
        for (int s = 0; s < 4; s++) scale[s] = 128 + 2 * digit;
    }
}

template <int NA, int NW, int MODE, int RF, int CF>
__global__ __launch_bounds__(64 * ONE_WAVES) void k_bitmm_fp4_one(
    const uint32_t *__restrict__ Xp, const uint32_t *__restrict__ Wp, void *__restrict__ outp, uint32_t x_bytes,
    uint32_t w_bytes, uint32_t out_bytes, int M, int K, int N, int w_lines,
    uint32_t cfg /* a | w << 8 | ob << 16 | zero_skip << 24; host: ob <= 23, every byte count < 2^32, M < 2^24 */) {
    static_assert(NA >= 1 && NA <= 8 && NW >= 1 && NW <= 8, "plane capacities");
    static_assert(MODE != 1 || RF == 2, "cols-layout words need whole 32-row tiles");
    constexpr int NF = RF * CF;                               // fragments = reducer waves
    constexpr int NDA = (NA + 1) / 2, NDW = (NW + 1) / 2;    // base-4 digits
    constexpr bool SWAP = MODE != 1;                          // lane = row, registers = 4 consecutive columns
    __shared__ __attribute__((aligned(16))) f32x4 part[ONE_WAVES][NF][64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, g = lane >> 4;
    const int kq = (K + 127) >> 7;
    const int m0 = blockIdx.x * (16 * RF), n0 = blockIdx.y * (16 * CF);
    const int a = cfg & 255u, w = (cfg >> 8) & 255u, ob = (cfg >> 16) & 255u;
    const bool zero_skip = ((cfg >> 24) & 1u) != 0u;
    const int nwv = min(ONE_WAVES, (kq + 3) >> 2);            // waves that have a super-step (host: kq <= 32)
    const bool worker = 4 * wv < kq;

#ifdef QGTC_STAMPS
    unsigned long long st_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define ONE_STAMP(i) st_[i] = __builtin_amdgcn_s_memtime()
#else
#define ONE_STAMP(i) do { } while (0)
#endif
    ONE_STAMP(0);
    f32x4 acc[RF][CF];
    u32x4 xr[RF][NA], wr[CF][NW];
    {
        // ---- the loads first: everything else runs under their latency (unconditional: a wave without a super-step
        // loads zeros from offset 0xffffffff, which costs nothing and keeps the registers free of merge copies)
        const uint32_t row_bytes = static_cast<uint32_t>(kq) * 16u;   // <= 512
        const int q = 4 * wv + g;                              // the lane's k-quad
        const uint32_t ko = static_cast<uint32_t>(q) * 16u;
        const bool q_ok = q < kq;
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(Xp), 0, static_cast<int>(x_bytes), 0x00020000);
        const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(Wp), 0, static_cast<int>(w_bytes), 0x00020000);
        const uint32_t x_plane = static_cast<uint32_t>(pad8(M)) * row_bytes, w_plane = static_cast<uint32_t>(w_lines) * row_bytes;
#pragma unroll
        for (int i = 0; i < RF; i++) {
            const int m = m0 + 16 * i + li;
            uint32_t off = __umul24(static_cast<uint32_t>(m), row_bytes) + ko;   // full-rate 24 x 24 -> 32 multiply
#ifdef QGTC_ABL   // timing-only build: every wave of the chip reads the same 1 KB of X
            if (cfg & (1u << 28)) off = static_cast<uint32_t>(lane) * 16u;
#endif
#pragma unroll
            for (int p = 0; p < NA; p++)   // lines / planes / k-quads that do not exist read as zero (offset 0xffffffff)
                xr[i][p] = __builtin_amdgcn_raw_buffer_load_b128(rx, (q_ok && m < M && p < a) ? off + static_cast<uint32_t>(p) * x_plane : 0xffffffffu, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < CF; j++) {
            const int n = n0 + 16 * j + li;
            uint32_t off = __umul24(static_cast<uint32_t>(n), row_bytes) + ko;
#ifdef QGTC_ABL
            if (cfg & (1u << 29)) off = static_cast<uint32_t>(lane) * 16u;
#endif
#pragma unroll
            for (int p = 0; p < NW; p++)
                wr[j][p] = __builtin_amdgcn_raw_buffer_load_b128(rw, (q_ok && n < N && p < w) ? off + static_cast<uint32_t>(p) * w_plane : 0xffffffffu, 0, 0);
        }
    }
    if (worker) {
        ONE_STAMP(1);
        uint32_t any = 0u;
#pragma unroll
        for (int i = 0; i < RF; i++)
#pragma unroll
            for (int p = 0; p < NA; p++) any |= (xr[i][p].x | xr[i][p].y) | (xr[i][p].z | xr[i][p].w);
        const bool mult = !zero_skip || __ballot(any != 0u) != 0ull;   // wave-uniform: an all-zero (tile rows) x 512-bit X tile is skipped
        if (!mult) {
#pragma unroll
            for (int i = 0; i < RF; i++)
#pragma unroll
                for (int j = 0; j < CF; j++) acc[i][j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        } else {
#ifdef QGTC_STAGGER_EXP   // timing-only experiment: the second wave of each SIMD (wv >= 4) out of phase with the first
            if ((cfg & (1u << 30)) && wv >= 4) __builtin_amdgcn_s_sleep(6);
            if ((cfg & (1u << 31))) { if (wv >= 4) __builtin_amdgcn_s_setprio(0); else __builtin_amdgcn_s_setprio(2); }
            if ((cfg & (1u << 27)) && wv >= 4) __builtin_amdgcn_s_sleep(12);
#endif
            uint32_t xo[RF][NDA][4][4], wo[4][4];
            int xs[RF][NDA][4], ws[4];
#pragma unroll
            for (int i = 0; i < RF; i++)
#pragma unroll
                for (int da = 0; da < NDA; da++) one_expand<NA>(xr[i], da, xo[i][da], xs[i][da]);
#pragma unroll
            for (int j = 0; j < CF; j++)
#pragma unroll
                for (int dw = 0; dw < NDW; dw++) {
                    one_expand<NW>(wr[j], dw, wo, ws);
#pragma unroll
                    for (int s = 0; s < 4; s++) {
                        // (an FP4 operand is the first 128 bits of the instruction's 256-bit register tuple: the upper half
                        // stays undefined - zeros there were four v_mov per operand)
                        const i32x4 b4 = {static_cast<int>(wo[s][0]), static_cast<int>(wo[s][1]), static_cast<int>(wo[s][2]), static_cast<int>(wo[s][3])};
                        const i32x8 b8 = __builtin_shufflevector(b4, b4, 0, 1, 2, 3, -1, -1, -1, -1);
#pragma unroll
                        for (int i = 0; i < RF; i++)
#pragma unroll
                            for (int da = 0; da < NDA; da++) {
                                const i32x4 a4 = {static_cast<int>(xo[i][da][s][0]), static_cast<int>(xo[i][da][s][1]), static_cast<int>(xo[i][da][s][2]), static_cast<int>(xo[i][da][s][3])};
                                const i32x8 a8 = __builtin_shufflevector(a4, a4, 0, 1, 2, 3, -1, -1, -1, -1);
                                // the accumulators START as the constant C operand of their first MFMA
                                const f32x4 c = (dw == 0 && s == 0 && da == 0) ? f32x4{0.0f, 0.0f, 0.0f, 0.0f} : acc[i][j];
                                // cbsz = blgp = 4: E2M1 operands. SWAP: D = W-fragment x X-fragment^T, i.e. lane (li, g) register r
                                // holds C[row 16 i + li][column 16 j + 4 g + r]; else C[row 16 i + 4 g + r][column 16 j + li]
                                if (SWAP) acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(b8, a8, c, 4, 4, 0, ws[s], 0, xs[i][da][s]);
                                else acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8, b8, c, 4, 4, 0, xs[i][da][s], 0, ws[s]);
                            }
                    }
                }
        }
#ifdef QGTC_STAMPS
        asm volatile("" : "+v"(acc[0][0]));
#endif
        ONE_STAMP(2);
#pragma unroll
        for (int i = 0; i < RF; i++)
#pragma unroll
            for (int j = 0; j < CF; j++) part[wv][i * CF + j][lane] = acc[i][j];
    }
    // ---- the reducer waves' epilogue plan, ahead of the barrier (what follows it is the exposed tail of the launch) and in the
    // reducer waves ONLY: the other waves are the second ones of their SIMDs - the ones the barrier waits for (3.20 -> 3.14 us
    // at 4096 x 4096 x 64, 1 bit; 3.85 -> 3.75 at 2 bits, 6.43 -> 6.33 at 8). Reducer wave f = fi CF + fj finishes fragment f; its
    // lane (li, g) owns "line" li of the fragment (a row for the rows layout / float32, a column for the cols layout) and four
    // consecutive elements 4 g .. 4 g + 3 along it
    int n_valid = 0;
    uint32_t vnib = 0u, oplane_bytes = 0u, o_off = 0xffffffffu, sh = 0u;
    bool store = false;
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(outp, 0, static_cast<int>(out_bytes), 0x00020000);
    if (wv < NF) {
        const int fi = wv / CF, fj = wv % CF;
        const int line = (SWAP ? m0 + 16 * fi : n0 + 16 * fj) + li;
        const int elem0 = (SWAP ? n0 + 16 * fj : m0 + 16 * fi) + 4 * g;
        const int line_lim = SWAP ? M : N, elem_lim = SWAP ? N : M;
        n_valid = line < line_lim ? min(max(elem_lim - elem0, 0), 4) : 0;   // leading elements inside the matrix
        vnib = (0xf0u >> n_valid) & 0xfu;                                   // element e at bit 3 - e of the nibble
        // bit modes: output word (line, elem / 32), element e at bit 31 - e (kernel.h:357-389 / :651-810 as intended). The
        // fragment holds elements 16 h .. 16 h + 15 of the word: the HIGH halfword (byte offset 2) for h = 0.
        const int pitch = SWAP ? step128(N) * 4 : step128(M) * 4;           // words per line
        const int n_lines = SWAP ? pad8(M) : pad128(N);                      // lines that exist in the output
        oplane_bytes = static_cast<uint32_t>(n_lines) * static_cast<uint32_t>(pitch) * 4u;
        const int frag0 = SWAP ? n0 + 16 * fj : m0 + 16 * fi;               // first element of the fragment
        const int word = frag0 >> 5;
        o_off = MODE == 2 ? (static_cast<uint32_t>(line) * static_cast<uint32_t>(N) + static_cast<uint32_t>(elem0)) * 4u
                          : (static_cast<uint32_t>(line) * static_cast<uint32_t>(pitch) + static_cast<uint32_t>(word)) * 4u + ((frag0 & 16) ? 0u : 2u);
        store = MODE == 2 ? line < M : (g == 0 && line < n_lines && word < pitch);
        if (!store) o_off = 0xffffffffu;                                     // the range check drops the store
        sh = 12u - 4u * static_cast<uint32_t>(g);
        asm volatile("" : "+v"(o_off) : "v"(vnib), "v"(sh), "s"(oplane_bytes));
    }
    ONE_STAMP(3);
    __syncthreads();
    ONE_STAMP(4);

    if (wv >= NF) {
        uint32_t *out = static_cast<uint32_t *>(outp);
        // ---- not a reducer: zero the output words no fragment computes
        const int t = tid - 64 * NF, nt = 64 * (ONE_WAVES - NF);
        if (MODE == 0) {   // row words past the last column tile (rows layout [ob][PAD8(M)][STEP128(N)*4])
            const int rows_pad = pad8(M), row_words = step128(N) * 4;
            const int w_next = (n0 + 16 * CF + 31) >> 5;
            // (the grid from the problem's shape, as launch_one sizes it: gridDim is a HIDDEN kernel argument - reading it makes the kernel's
            // argument segment 312 bytes instead of 56, five lines instead of one for hipLaunchKernel to write through the PCIe BAR:
            // tools/kernarg_probe.hip measured up to 0.9 us of host time per launch)
            if (static_cast<int>(blockIdx.y) == (N + 16 * CF - 1) / (16 * CF) - 1 && w_next < row_words) {
                const int nx = row_words - w_next;
                for (int e = t; e < ob * 16 * RF * nx; e += nt) {
                    const int x = e % nx, r = (e / nx) % (16 * RF), p = e / (nx * 16 * RF);
                    if (m0 + r < rows_pad) out[(static_cast<size_t>(p) * rows_pad + m0 + r) * row_words + w_next + x] = 0u;
                }
            }
        } else if (MODE == 1) {   // cols layout [ob][PAD128(N)][STEP128(M)*4]: words past the last row tile, lines past the last column tile
            const int lines = pad128(N), line_words = step128(M) * 4;
            const size_t oplane = static_cast<size_t>(lines) * line_words;
            const bool last_m = static_cast<int>(blockIdx.x) == (M + 16 * RF - 1) / (16 * RF) - 1, last_n = static_cast<int>(blockIdx.y) == (N + 16 * CF - 1) / (16 * CF) - 1;
            const int word1 = m0 >> 5, w_core1 = min(line_words, word1 + 1);
            if (last_m && w_core1 < line_words) {
                for (int e = t; e < ob * 16 * CF; e += nt) {
                    const int line = n0 + e % (16 * CF), p = e / (16 * CF);
                    if (line < lines)
                        for (int wi = w_core1; wi < line_words; wi++) out[p * oplane + static_cast<size_t>(line) * line_words + wi] = 0u;
                }
            }
            if (last_n && n0 + 16 * CF < lines) {
                const int nl = lines - (n0 + 16 * CF), w_end = last_m ? line_words : w_core1;
                for (int e = t; e < ob * nl; e += nt) {
                    const int line = n0 + 16 * CF + e % nl, p = e / nl;
                    for (int wi = word1; wi < w_end; wi++) out[p * oplane + static_cast<size_t>(line) * line_words + wi] = 0u;
                }
            }
        }
        return;
    }

    // ---- reducer wave f: fragment f of every partial tile, 8 linear 16-byte reads
    f32x4 sum;
    if (nwv == ONE_WAVES) {   // every partial exists: eight reads in flight, then a tree of adds
        f32x4 pv[ONE_WAVES];
#ifdef QGTC_ABL_HALFRED   // timing-only build (wrong sums): what a reducer wave would save if it read and added HALF of the partials -
                          // more than an all-eight-waves reduction can gain, which adds the cross-half exchange on top
#pragma unroll
        for (int p = 0; p < ONE_WAVES / 2; p++) pv[p] = part[p][wv][lane];
        sum = (pv[0] + pv[1]) + (pv[2] + pv[3]);
#else
#pragma unroll
        for (int p = 0; p < ONE_WAVES; p++) pv[p] = part[p][wv][lane];
        sum = ((pv[0] + pv[1]) + (pv[2] + pv[3])) + ((pv[4] + pv[5]) + (pv[6] + pv[7]));
#endif
    } else {
        sum = part[0][wv][lane];
        for (int p = 1; p < nwv; p++) sum += part[p][wv][lane];
    }
    ONE_STAMP(5);
    if (MODE == 2) {   // float32 [M,N] (reference kernel.h:915-930): four consecutive columns of a row
        // one store for a lane's four columns whatever N is (rows of 4 N bytes are dword-aligned only when N % 4 != 0 - the 10-class
        // output layer; buffer stores take any dword alignment), a two- and / or a one-float store for a row's tail
        const u32x4 v4 = __builtin_bit_cast(u32x4, sum);
        if (n_valid == 4) {
            __builtin_amdgcn_raw_buffer_store_b128(v4, ro, o_off, 0, 0);
        } else {
            if (n_valid >= 2) __builtin_amdgcn_raw_buffer_store_b64(u32x2{v4.x, v4.y}, ro, o_off, 0, 0);
            if (n_valid == 3) __builtin_amdgcn_raw_buffer_store_b32(v4.z, ro, o_off + 8u, 0, 0);
            if (n_valid == 1) __builtin_amdgcn_raw_buffer_store_b32(v4.x, ro, o_off, 0, 0);
        }
    } else {
        // requantise (kernel.h:31-37,350: c > 2^ob ? 2^ob - 1 : c; the sums are exact integers in [0, 2^24) and
        // ob <= 23, so the float compare of the reference is this integer compare), one value per byte
        const int maxi = 1 << ob;
        uint32_t P = 0u;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const int c = static_cast<int>(sum[e]);
            P = (P << 8) | (static_cast<uint32_t>(c > maxi ? maxi - 1 : c) & 255u);   // planes p < 8 come from the byte
        }
        int qhi[4];
        if (ob > 8) {
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int c = static_cast<int>(sum[e]);
                qhi[e] = c > maxi ? maxi - 1 : c;
            }
        }
        for (int p = 0; p < ob; p++, o_off += store ? oplane_bytes : 0u) {
            uint32_t t;   // bits 24, 16, 8, 0: plane p of elements 0..3
            if (p < 8) t = (P >> p) & 0x01010101u;
            else t = ((qhi[0] >> p) & 1) << 24 | ((qhi[1] >> p) & 1) << 16 | ((qhi[2] >> p) & 1) << 8 | ((qhi[3] >> p) & 1);
            uint32_t x = (((t >> 21) | (t >> 14)) | ((t >> 7) | t)) & vnib;   // element e at bit 3 - e
            x <<= sh;
            const auto s16 = __builtin_amdgcn_permlane16_swap(x, x, false, false);   // rows of 16 lanes: (0,1), (2,3)
            x = s16[0] | s16[1];
            x = or_with_partner_half(x);                                            // halves of the wave
            __builtin_amdgcn_raw_buffer_store_b16(static_cast<unsigned short>(x), ro, o_off, 0, 0);
        }
    }
#ifdef QGTC_STAMPS
    ONE_STAMP(6);
    if (tid == 0 && blockIdx.y == 0 && blockIdx.x < 1024)
        for (int i = 0; i < 8; i++) g_stamps[blockIdx.x * 16 + i] = st_[i];
#endif
#undef ONE_STAMP
}

}  // namespace
