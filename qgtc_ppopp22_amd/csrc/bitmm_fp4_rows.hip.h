// bitmm_fp4_rows.hip.h — part of libqgtc_hip.so (qgtc_fp4.hip).
// Grouped launches of the "A . (XW)" stages of a GNN epoch: sparse, block-structured left operands (cluster-batch
// adjacencies), right operands of at most 256 columns, rows-layout bits or float32 out.
#pragma once

namespace {

// ------------------------------------------------------------------------------------------
// On 128 x 128 tiles (bitmm_mfma.hip.h) a workgroup of these stages visits every k-quad in which ANY of its 128 rows
// has a bit: 54 % of them on the ogbn-arxiv-sized cluster batches, where only 19 % of the 32-row x 128-bit tiles are
// occupied. Here the unit is a 32-row block of one batch:
//   * a workgroup = one row block, wave j = its columns 32 j .. 32 j + 31 (N <= 128: four waves, N <= 256: eight); the
//     waves never synchronise and use no LDS;
//   * the k-quads to visit come from the occupancy bitmap (qgtc_tile_occupancy: one bit per 32-row tile and k-quad;
//     without a bitmap: all of them), in pairs: the lanes of half fh load the 16 bytes of the pair's k-quad fh of
//     their A row and T line, two pairs in flight before the first is used;
//   * per pair four v_mfma_scale_f32_32x32x64_f8f6f4 (word t of both k-quads), operands SWAPPED so that a lane owns one row and
//     16 of its 32 columns: half an output word per plane (byte-packed re-quantised values, one shift + AND per plane
//     and four values, one half-wave swap), stored straight from the registers - the four waves' words of a row are 16
//     contiguous bytes; float32 rows for the output layer.
// Rows past M and columns past N need no masks: their operand lines are read as zeros and requant(0) = 0.
// float32 sums: exact while K (2^a - 1)(2^w - 1) < 2^24 (the host checks it, a <= 4, w <= 8, K <= 8192).
// ------------------------------------------------------------------------------------------
constexpr int RS_CHUNK = 2;   // PAIRS of k-quads whose packed words a wave has in flight at once

// MODE 0 rows-layout bits / 1 cols-layout bits (single launches only: the operands are NOT swapped, a lane owns a column and 16
// of the block's 32 rows - half a word of the column's line; a workgroup per WORD of a line, the padding ones included) / 2
// float32; OB output planes (0 = any); CB column blocks of 32 per wave (2: half the waves per launch - a stage of the
// ogbn-arxiv-sized epoch then fits the chip in ONE round of waves instead of 1.4)
template <int NA, int NW, int MODE, int OB, int CB>
__device__ __forceinline__ void rows_block(const qgtc_problem &pr, const MMShape &sh, int rb, int batch) {
    constexpr int NDA = (NA + 1) / 2, NDW = (NW + 1) / 2;   // base-4 digits
    static_assert(MODE != 1 || CB == 1, "cols-layout output: one column block per wave");
#ifdef QGTC_STAMPS
    unsigned long long st_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define RW_STAMP(i) st_[i] = __builtin_amdgcn_s_memtime()
#else
#define RW_STAMP(i) do { } while (0)
#endif
    RW_STAMP(0);
    (void)batch;
    pin_problem(pr);
    const int M = pr.M, K = pr.K, N = pr.N;
    if (MODE == 1 ? rb >= step128(M) * 4 : 32 * rb >= M) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fl = lane & 31, fh = lane >> 5;
    const int n0 = 32 * CB * wv;
    const int row_words = step128(N) * 4;
    if (MODE == 0 ? CB * wv >= row_words : MODE == 1 ? n0 >= pad128(N) : n0 >= N) return;  // nothing of this wave's words / lines / columns exists
    const int kq = step128(K);
    const uint32_t row_bytes = static_cast<uint32_t>(kq) * 16u;
    // k-quads to visit (K <= 8192: one 64-bit word per 32-row tile)
    unsigned long long todo = kq >= 64 ? ~0ull : ((1ull << kq) - 1ull);
    if (pr.occ && 32 * rb < M) todo &= pr.occ[static_cast<size_t>(rb) * pr.occ_words];
    const bool cols_live = n0 < N;                           // (a padding word of the row: zeros, no arithmetic)
    if (!cols_live || 32 * rb >= M) todo = 0ull;             // (... or a padding word of a cols-layout line)
#ifdef QGTC_STAMPS
    asm volatile("" ::"s"(todo));
    st_[6] = __builtin_popcountll(todo);
#endif
    RW_STAMP(1);

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint32_t *>(pr.X), 0, static_cast<int>(static_cast<uint32_t>(pr.x_words) * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint32_t *>(pr.W), 0, static_cast<int>(static_cast<uint32_t>(pr.w_words) * 4u), 0x00020000);
    const int m = 32 * rb + fl;
    const uint32_t x_plane = static_cast<uint32_t>(pad8(M)) * row_bytes, w_plane = static_cast<uint32_t>(pr.w_lines) * row_bytes;
    const uint32_t x_base = m < M ? static_cast<uint32_t>(m) * row_bytes : 0xffffffffu;
    uint32_t w_base[CB];
#pragma unroll
    for (int j = 0; j < CB; j++) w_base[j] = n0 + 32 * j + fl < N ? static_cast<uint32_t>(n0 + 32 * j + fl) * row_bytes : 0xffffffffu;

    f32x16 acc[CB];
#pragma unroll
    for (int j = 0; j < CB; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[j][r] = 0.0f;
    // The visited k-quads are taken in PAIRS: the lanes of half fh load the 16 bytes of the pair's k-quad fh of their
    // A row and T line (one 16-byte load per lane, operand and plane - every byte used; 4-byte loads at a 160-byte row
    // stride cost the address unit 16 cycles each for a quarter of the data), and MFMA t = 0..3 of the pair multiplies
    // word t of both k-quads: which 64 elements of K an instruction covers is free as long as A and T agree.
    // Two register sets in turn: the loads of the NEXT pair are issued before the current one is multiplied. (Written
    // as a loop over chunks of pairs with a `break` for the missing ones, hipcc sank the loads of the second pair below
    // the multiplication of the first - one more exposed memory latency for every row block with three k-quads or more.)
    unsigned todo_lo = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(todo)), todo_hi = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(todo >> 32));
    unsigned long long left = (static_cast<unsigned long long>(todo_hi) << 32) | todo_lo;   // (wave-uniform: scalar registers)
    struct Pair {
        int first;      // this lane's k-quad of the pair (-1: none)
        bool second;    // the pair exists (wave-uniform)
    };
    auto take = [&]() {
        const int qa = left != 0ull ? __builtin_ctzll(left) : -1;
        left &= left - 1ull;
        const int qb = left != 0ull ? __builtin_ctzll(left) : -1;
        left &= left - 1ull;
        return Pair{qa < 0 ? -1 : (fh ? qb : qa), qa >= 0};
    };
    auto load = [&](int q, u32x4 (&xl)[NA], u32x4 (&wl)[CB][NW]) {   // unconditional (a missing k-quad reads zeros): exact vmcnt waits
        const uint32_t ko = static_cast<uint32_t>(q) * 16u;
#pragma unroll
        for (int p = 0; p < NA; p++)
            xl[p] = __builtin_amdgcn_raw_buffer_load_b128(rx, (q >= 0 && x_base != 0xffffffffu && p < sh.a) ? x_base + static_cast<uint32_t>(p) * x_plane + ko : 0xffffffffu, 0, 0);
#pragma unroll
        for (int j = 0; j < CB; j++)
#pragma unroll
            for (int p = 0; p < NW; p++)
                wl[j][p] = __builtin_amdgcn_raw_buffer_load_b128(rw, (q >= 0 && w_base[j] != 0xffffffffu && p < sh.w) ? static_cast<uint32_t>(p) * w_plane + (w_base[j] + ko) : 0xffffffffu, 0, 0);
    };
    // An ODD k-quad out (the only one of an X . W product with K <= 128; the commonest row block of a cluster batch has one beside its
    // diagonal) is SHARED between the halves instead: half fh loads words 2 fh, 2 fh + 1 of both operands (8 bytes per lane and plane), TWO
    // MFMAs per pair of digits cover its 128 bits - as a pair with a missing partner it was four, half of every operand zeros
    // (tools/grouped_cols_sweep.py: the one-k-quad kernel k_bitmm_fp4_xw_rows was 20-40 % ahead of this one on K <= 128 for that reason).
    auto load_single = [&](int q, u32x2 (&xl)[NA], u32x2 (&wl)[CB][NW]) {
        const uint32_t ko = static_cast<uint32_t>(q) * 16u + 8u * static_cast<uint32_t>(fh);
#pragma unroll
        for (int p = 0; p < NA; p++)
            xl[p] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rx, (q >= 0 && x_base != 0xffffffffu && p < sh.a) ? x_base + static_cast<uint32_t>(p) * x_plane + ko : 0xffffffffu, 0, 0));
#pragma unroll
        for (int j = 0; j < CB; j++)
#pragma unroll
            for (int p = 0; p < NW; p++)
                wl[j][p] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rw, (q >= 0 && w_base[j] != 0xffffffffu && p < sh.w) ? static_cast<uint32_t>(p) * w_plane + (w_base[j] + ko) : 0xffffffffu, 0, 0));
    };
    auto multiply = [&](const auto &xl, const auto &wl, auto nt) {   // nt words of every lane's load: 4 of a pair's k-quad, 2 of a shared one
#pragma unroll
        for (int t = 0; t < decltype(nt)::value; t++) {
            uint32_t xw[NA];
#pragma unroll
            for (int p = 0; p < NA; p++) xw[p] = xl[p][t];
            i32x8 xa[NDA];
#pragma unroll
            for (int da = 0; da < NDA; da++) xa[da] = strip_operand<NA>(xw, da);
#pragma unroll
            for (int j = 0; j < CB; j++) {
                uint32_t ww[NW];
#pragma unroll
                for (int p = 0; p < NW; p++) ww[p] = wl[j][p][t];
#pragma unroll
                for (int dw = 0; dw < NDW; dw++) {
                    const i32x8 wb = strip_operand<NW>(ww, dw);
#pragma unroll
                    for (int da = 0; da < NDA; da++) {
                        if (MODE == 1)   // lane (fl, fh) register r holds C[row (r & 3) + 8 (r >> 2) + 4 fh][column fl]
                            acc[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(xa[da], wb, acc[j], 4, 4, 0, 128 + 2 * da, 0, 128 + 2 * dw);
                        else             // swapped: lane (fl, fh) register r holds C[row fl][column (r & 3) + 8 (r >> 2) + 4 fh]
                            acc[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wb, xa[da], acc[j], 4, 4, 0, 128 + 2 * dw, 0, 128 + 2 * da);
                    }
                }
            }
        }
    };
    {
        const std::integral_constant<int, 4> four;
        const std::integral_constant<int, 2> two;
        const bool odd = (__builtin_popcountll(left) & 1) != 0;   // (wave-uniform)
        int qs = -1;
        if (odd) {
            qs = __builtin_ctzll(left);
            left &= left - 1ull;
        }
        u32x2 xs_[NA], ws_[CB][NW];
        if (odd) load_single(qs, xs_, ws_);   // (wave-uniform)
        u32x4 xa_[NA], wa_[CB][NW], xb_[NA], wb_[CB][NW];
        auto pa = take();
        load(pa.first, xa_, wa_);
#ifdef QGTC_STAMPS
        asm volatile("" ::"v"(xa_[0]), "v"(wa_[0][0]));
        RW_STAMP(2);
#endif
        if (odd) multiply(xs_, ws_, two);   // (the first pair's loads are in flight)
        while (pa.second) {   // wave-uniform
            auto pb = take();
            if (pb.second) load(pb.first, xb_, wb_);
            multiply(xa_, wa_, four);
            if (!pb.second) break;
            pa = take();
            if (pa.second) load(pa.first, xa_, wa_);
            multiply(xb_, wb_, four);
        }
    }

#ifdef QGTC_STAMPS
    asm volatile("" ::"v"(acc[0][0]));
#endif
    RW_STAMP(3);
    if (MODE == 2) {   // float32 [M,N] (reference kernel.h:915-930): registers 4 g .. 4 g + 3 are four consecutive columns
        if (m < M) {
#pragma unroll
            for (int j = 0; j < CB; j++) {
                float *dst = static_cast<float *>(pr.out) + static_cast<size_t>(m) * N + n0 + 32 * j;
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const int e = 8 * g + 4 * fh;
                    if (n0 + 32 * j + e + 3 < N) {   // one 16-byte store (scalar stores of a column run over 32 rows x 512 bytes); rows of 4 N
                        // bytes are only dword-aligned when N % 4 != 0 (10 classes): the vector type says so, global stores take it
                        typedef float f32x4_u4 __attribute__((ext_vector_type(4), aligned(4)));
                        *reinterpret_cast<f32x4_u4 *>(dst + e) = f32x4_u4{acc[j][4 * g], acc[j][4 * g + 1], acc[j][4 * g + 2], acc[j][4 * g + 3]};
                    } else {
#pragma unroll
                        for (int t = 0; t < 4; t++)
                            if (n0 + 32 * j + e + t < N) dst[e + t] = acc[j][4 * g + t];
                    }
                }
            }
        }
        return;
    }
    // rows layout [ob][PAD8(M)][STEP128(N)*4] (reference kernel.h:357-389): word (row, wv); column e of the block at bit
    // 31 - e = 8 (3 - gq) + (7 - t - 4 fh) with t = r & 3, gq = r >> 2: the values of one t a byte each (byte 3 - gq),
    // plane p of the four = one shift + AND; the partner lane (fl, fh ^ 1) holds the other 16 bits of the word
    // cols layout [ob][PAD128(N)][STEP128(M)*4] (QGTC_device.cu:456): word (line n0 + fl, rb), the same expression with rows for columns
    const int ob = OB > 0 ? OB : sh.ob;
    const int rows_pad = pad8(M);
    const int line_words = step128(M) * 4, lines = pad128(N);
    const size_t oplane = MODE == 1 ? static_cast<size_t>(lines) * line_words : static_cast<size_t>(rows_pad) * row_words;
    const bool store = fh == 0 && (MODE == 1 ? n0 + fl < lines : m < rows_pad);
#pragma unroll
    for (int j = 0; j < CB; j++) {
        if (MODE == 0 && CB * wv + j >= row_words) break;   // (wave-uniform: the row has no such word)
        uint32_t qv[16], P[4];
        requant_pack16<OB>(acc[j], ob, P, qv);
        uint32_t *dst = static_cast<uint32_t *>(pr.out) + (MODE == 1 ? static_cast<size_t>(n0 + fl) * line_words + rb : static_cast<size_t>(m) * row_words + CB * wv + j);
#pragma unroll
        for (int p = 0; p < (OB > 0 ? OB : 32); p++) {
            if (OB == 0 && p >= ob) break;
            uint32_t x;
            if (OB > 0 || p < 8) {
                x = ((P[0] >> p) & 0x01010101u) << 3 | ((P[1] >> p) & 0x01010101u) << 2 | ((P[2] >> p) & 0x01010101u) << 1 | ((P[3] >> p) & 0x01010101u);
            } else {   // more than 8 output planes: from the full values
                x = 0u;
#pragma unroll
                for (int r = 0; r < 16; r++) x |= ((qv[r] >> p) & 1u) << (8 * (3 - (r >> 2)) + 3 - (r & 3));
            }
            x <<= 4u - 4u * static_cast<uint32_t>(fh);   // bits 7 - t - 4 fh of every byte
            x = or_with_partner_half(x);
            if (store) dst[p * oplane] = x;
        }
    }
#ifdef QGTC_STAMPS
    RW_STAMP(4);
    if (tid == 0) {
        const int slot = (batch * gridDim.x + rb) % 1024;
        for (int i = 0; i < 8; i++) g_stamps[slot * 16 + i] = st_[i];
    }
#endif
#undef RW_STAMP
}

template <int NA, int NW, int MODE, int OB, int CB>
__global__ __launch_bounds__(64 * 8) void k_bitmm_fp4_rows(const qgtc_problem *__restrict__ prs, MMShape sh) {
    pin_shape(sh);
    pin_grid();
    // sh.per != 0: the row blocks of a batch run on ONE XCD (they share its T lines and descriptor in that L2; spread
    // round-robin over the eight, every XCD fetched every batch's T: rocprofv3 counted 18.5 MB of fetches per launch for
    // 5.6 MB of operands)
    int rb = static_cast<int>(blockIdx.x), batch = static_cast<int>(blockIdx.y);
    if (sh.per) {
        const int v = xcd_consecutive(batch * static_cast<int>(gridDim.x) + rb, static_cast<int>(gridDim.x * gridDim.y));
        batch = v / static_cast<int>(gridDim.x);
        rb = v - batch * static_cast<int>(gridDim.x);
    }
    const qgtc_problem pr = prs[batch];
    rows_block<NA, NW, MODE, OB, CB>(pr, sh, rb, batch);
}

// the same row block for ONE problem handed over by value: single launches with three or four left-hand planes and a short K (the
// per-batch 4 x 4-bit products of the Batched-GIN chain, main_qgtc.py:132,134,138 - 599 x 50 x 64 and the like; the narrow-operand
// kernels of bitmm_fp4_one / _skinny take two left-hand planes at most, and one wave per 32 x 32 tile on 16 x 16 x 128 MFMAs
// multiplies mostly zeros when K is a single k-quad)
template <int NA, int NW, int MODE, int OB>
__global__ __launch_bounds__(64 * 8) void k_bitmm_fp4_rows_single(qgtc_problem pr, MMShape sh) {
    pin_shape(sh);
    rows_block<NA, NW, MODE, OB, 1>(pr, sh, static_cast<int>(blockIdx.x), 0);
}

// ------------------------------------------------------------------------------------------
// A grouped "X . W" stage with ONE k-quad of K (K <= 128, N <= 128, cols-layout bits out) at the epochs' widths, fixed shape:
// a workgroup = 32 rows of one batch, wave j = columns 32 j .. 32 j + 31,
// the X words straight from global memory (8 bytes per lane and plane), no LDS, no barrier, no line-assembly pass: the
// word of each of its 128 lines goes out as a 4-byte store, and those merge in the one L2 all row blocks of a batch share.
// ------------------------------------------------------------------------------------------
template <int NA, int NW, int OB>
__global__ __launch_bounds__(64 * 4) __attribute__((amdgpu_waves_per_eu(NA <= 2 ? 8 : 4, 8))) void k_bitmm_fp4_xw_rows(
    const qgtc_problem *__restrict__ prs, MMShape sh) {
    constexpr int NDA = (NA + 1) / 2, NDW = (NW + 1) / 2;   // base-4 digits
    pin_shape(sh);
    pin_grid();
    int rb = static_cast<int>(blockIdx.x), batch = static_cast<int>(blockIdx.y);
    if (sh.per) {
        const int v = xcd_consecutive(batch * static_cast<int>(gridDim.x) + rb, static_cast<int>(gridDim.x * gridDim.y));
        batch = v / static_cast<int>(gridDim.x);
        rb = v - batch * static_cast<int>(gridDim.x);
    }
    const qgtc_problem pr = prs[batch];
    pin_problem(pr);
    const int M = pr.M, N = pr.N;
    const int line_words = step128(M) * 4, lines = pad128(N);
    if (rb >= line_words) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fl = lane & 31, fh = lane >> 5;
    const int n = 32 * wv + fl, m = 32 * rb + fl;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint32_t *>(pr.X), 0, static_cast<int>(static_cast<uint32_t>(pr.x_words) * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint32_t *>(pr.W), 0, static_cast<int>(static_cast<uint32_t>(pr.w_words) * 4u), 0x00020000);
    const uint32_t x_plane = static_cast<uint32_t>(pad8(M)) * 16u, w_plane = static_cast<uint32_t>(pr.w_lines) * 16u;
    uint32_t xl[2][NA], wl[2][NW];   // [k half][plane]: words 2 fh, 2 fh + 1 of the lane's row / line
#pragma unroll
    for (int p = 0; p < NA; p++) {
        const u32x2 v = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rx, (m < M && p < sh.a) ? static_cast<uint32_t>(p) * x_plane + static_cast<uint32_t>(m) * 16u + 8u * fh : 0xffffffffu, 0, 0));
        xl[0][p] = v.x;
        xl[1][p] = v.y;
    }
#pragma unroll
    for (int p = 0; p < NW; p++) {
        const u32x2 v = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rw, (n < N && p < sh.w) ? static_cast<uint32_t>(p) * w_plane + static_cast<uint32_t>(n) * 16u + 8u * fh : 0xffffffffu, 0, 0));
        wl[0][p] = v.x;
        wl[1][p] = v.y;
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.0f;
#pragma unroll
    for (int h = 0; h < 2; h++)
#pragma unroll
        for (int da = 0; da < NDA; da++) {
            const i32x8 xa = strip_operand<NA>(xl[h], da);   // not swapped: lane (fl, fh) register r holds C[row (r & 3) + 8 (r >> 2) + 4 fh][column fl]
#pragma unroll
            for (int dw = 0; dw < NDW; dw++) {
                const i32x8 wb = strip_operand<NW>(wl[h], dw);
                acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(xa, wb, acc, 4, 4, 0, 128 + 2 * da, 0, 128 + 2 * dw);
            }
        }
    uint32_t qv[16], P[4];
    requant_pack16<OB>(acc, OB, P, qv);
    const size_t oplane = static_cast<size_t>(lines) * line_words;
    uint32_t *dst = static_cast<uint32_t *>(pr.out) + static_cast<size_t>(n) * line_words + rb;
#pragma unroll
    for (int p = 0; p < OB; p++) {
        uint32_t x = ((P[0] >> p) & 0x01010101u) << 3 | ((P[1] >> p) & 0x01010101u) << 2 | ((P[2] >> p) & 0x01010101u) << 1 | ((P[3] >> p) & 0x01010101u);
        x <<= 4u - 4u * static_cast<uint32_t>(fh);
        x = or_with_partner_half(x);
        if (fh == 0 && n < lines) dst[p * oplane] = x;
    }
}

}  // namespace
