// bitmm_fp4_chain.hip.h — part of libqgtc_hip.so (qgtc_fp4.hip).
// An "A . T" stage of a GNN epoch with the NEXT layer's "X . W" stage in its tail: two operators, one launch, no hand-off
// between workgroups.
#pragma once

namespace {

// ------------------------------------------------------------------------------------------
// out = requant(A . T) is the left operand of the next layer's T' = requant(out . W'), and T' is ROW-LOCAL: the 32 rows
// of T' that a row block contributes (one word per column and plane of the cols layout) depend only on the 32 rows of
// `out` the same workgroup has just computed, and on W' (a few KB, shared by everybody). So the workgroup of
// bitmm_fp4_rows.hip.h (one 32-row block of one batch, wave j = columns 32 j .. 32 j + 31 of `out`) carries on:
//   * every wave writes its packed word of `out` (per row and plane) to global memory as before AND into 16 bytes x 32
//     rows x planes of LDS; one workgroup barrier (four waves);
//   * wave j then multiplies the 32 x 128-bit block of `out` (8 bytes per lane and plane from LDS) with the 32 lines
//     32 j .. 32 j + 31 of W' (fetched at the very start of the kernel): two v_mfma_scale_f32_32x32x64_f8f6f4 per pair of
//     base-4 digits, NOT swapped (a lane owns one column of T' and 16 of the block's 32 rows), re-quantises, packs and
//     stores the block's word of its 32 lines of T' - and zeros for the lines past N' and the words past the last row;
//   * the partial-line stores (4 bytes at the line pitch) merge in ONE L2: all row blocks of a batch run on one XCD
//     (xcd_consecutive), which is also where the next stage reads T'.
// An epoch of the layout-correct Cluster-GCN chain is then four launches instead of six (X.W1 | A.T1 + X.W2 | A.T2 +
// X.W3 | A.T3 -> float32), a Batched-GIN epoch three (A.X + X.W1 | A.T1 + X.W2 | A.T2 + X.W3 -> float32).
// Conditions (host): one-plane A, N <= 128 (the four words of a row of `out` = the whole K of the next product),
// N' <= 128, the plane counts instantiated below.
// ------------------------------------------------------------------------------------------
// MODE2: 1 = T' as cols-layout bits (OB2 planes), 2 = float32 [M, N'] (the output layer: kernel.h:915-930; OB2 unused)
// DISC: `out` itself is not wanted (QGTC_CHAIN_DISCARD): it is neither packed nor stored - the re-quantised values go to
// the second product as the E2M1 codes they are (a nibble each, in the order the W' expansion uses: column e of a word
// = nibble 7 - e / 4 of dword 3 - e % 4), 16 instead of 36 + 28 VALU operations per wave between the two products
// CODES: bit 0 - T (the first product's right operand) arrives as E2M1 codes, bit 1 - T' is written as codes (4-bit
// products only, where the codes are exactly as large as the packed planes): [k-quad][line][word][dword], 16 bytes of
// codes per packed word, nibble 7 - e / 4 of dword 3 - e % 4 for element e - finished MFMA operands, written by the X.W
// phase of the launch before, never seen outside a chain of these launches (QGTC_CHAIN_CODES_IN / _OUT)
// sh.qmajor (the 2-bit chains' form of the same agreement): T / T' stay bit planes, but in QUAD-MAJOR order - word j of line n
// of a plane at ((j >> 2) lines + n) 4 + (j & 3) instead of n line_words + j: the 16 bytes a lane needs of its line and
// k-quad are then contiguous over the 32 lines of a wave (4 cache lines per load instead of 32), and so are the words the
// row blocks store. Same size. *Measured*: Cluster-GCN epoch 34.2 -> 29.5 us.
template <int NW, int OB, int NW2, int OB2, int MODE2, bool DISC, int CODES = 0>
__global__ __launch_bounds__(64 * 4) __attribute__((amdgpu_waves_per_eu(OB <= 2 ? 8 : 4, 8))) void k_bitmm_fp4_chain(const qgtc_problem *__restrict__ prs, const qgtc_problem *__restrict__ prs2,
                                                            MMShape sh, MMShape sh2) {
    constexpr int NDW = (NW + 1) / 2, NDX2 = (OB + 1) / 2, NDW2 = (NW2 + 1) / 2;   // base-4 digits
    constexpr bool CIN = (CODES & 1) != 0, COUT = (CODES & 2) != 0;
    static_assert(!CIN || NW == 4, "code-form T: 4-bit products");
    static_assert(!COUT || (OB2 == 4 && MODE2 == 1), "code-form T': 4-bit products, bits mode");
    __shared__ __attribute__((aligned(16))) uint32_t xchg[DISC ? 4 : OB][32][4];   // [plane][row of the block][word of the row]; DISC: [word][row][dword of codes]
    pin_shape(sh);
    int rb = static_cast<int>(blockIdx.x), batch = static_cast<int>(blockIdx.y);
    if (sh.per) {   // the row blocks of a batch on ONE XCD (bitmm_fp4_rows.hip.h)
        const int v = xcd_consecutive(batch * static_cast<int>(gridDim.x) + rb, static_cast<int>(gridDim.x * gridDim.y));
        batch = v / static_cast<int>(gridDim.x);
        rb = v - batch * static_cast<int>(gridDim.x);
    }
    const qgtc_problem pr = prs[batch], pr2 = prs2[batch];
    pin_problem(pr);
    pin_problem(pr2);
    const int M = pr.M, K = pr.K, N = pr.N, N2 = pr2.N;
    const int line_words2 = step128(M) * 4, lines2 = pad128(N2);
    if (MODE2 == 2 ? 32 * rb >= M : rb >= line_words2) return;   // (no rows / not even a padding word of T')
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fl = lane & 31, fh = lane >> 5;
    const int n0 = 32 * wv;

    // ---- W' first: nothing below depends on it until the second product
    uint32_t w2l[2][NW2];   // [k half][plane]: words 2 fh, 2 fh + 1 of line n2 (one k-quad: a packed line is 16 bytes)
    const int n2 = n0 + fl;
    {
        const __amdgpu_buffer_rsrc_t rw2 = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<uint32_t *>(pr2.W), 0, static_cast<int>(static_cast<uint32_t>(pr2.w_words) * 4u), 0x00020000);
        const uint32_t w2_plane = static_cast<uint32_t>(pr2.w_lines) * 16u;
#pragma unroll
        for (int p = 0; p < NW2; p++) {
            const u32x2 v = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rw2, (n2 < N2 && p < sh2.w) ? static_cast<uint32_t>(p) * w2_plane + static_cast<uint32_t>(n2) * 16u + 8u * fh : 0xffffffffu, 0, 0));
            w2l[0][p] = v.x;
            w2l[1][p] = v.y;
        }
    }

    // ---- first product: the row block of out = requant(A . T), exactly as k_bitmm_fp4_rows<1, NW, 0, OB, 1>
    const int ob = OB;
    const int m = 32 * rb + fl;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.0f;
    if (32 * rb < M) {   // (workgroup-uniform; a padding word of T' has no rows)
        const int kq = step128(K);
        const uint32_t row_bytes = static_cast<uint32_t>(kq) * 16u;
        unsigned long long todo = kq >= 64 ? ~0ull : ((1ull << kq) - 1ull);
        if (pr.occ) todo &= pr.occ[static_cast<size_t>(rb) * pr.occ_words];
        if (n0 >= N) todo = 0ull;                            // (a padding word of the row: zeros, no arithmetic)
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<uint32_t *>(pr.X), 0, static_cast<int>(static_cast<uint32_t>(pr.x_words) * 4u), 0x00020000);
        const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<uint32_t *>(pr.W), 0, static_cast<int>(static_cast<uint32_t>(pr.w_words) * 4u), 0x00020000);
        const uint32_t w_plane = static_cast<uint32_t>(pr.w_lines) * row_bytes;
        const uint32_t x_base = m < M ? static_cast<uint32_t>(m) * row_bytes : 0xffffffffu;
        const uint32_t w_base = n0 + fl < N ? static_cast<uint32_t>(n0 + fl) * row_bytes : 0xffffffffu;
        unsigned todo_lo = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(todo)), todo_hi = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(todo >> 32));
        unsigned long long left = (static_cast<unsigned long long>(todo_hi) << 32) | todo_lo;
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
        const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(   // (code-form T: 64 bytes per line and k-quad)
            const_cast<uint32_t *>(pr.W), 0, static_cast<int>(static_cast<uint32_t>(pad128(N)) * static_cast<uint32_t>(kq) * 64u), 0x00020000);
        auto load = [&](int q, u32x4 &xl, u32x4 (&wl)[NW]) {
            const uint32_t ko = static_cast<uint32_t>(q) * 16u;
            xl = __builtin_amdgcn_raw_buffer_load_b128(rx, (q >= 0 && x_base != 0xffffffffu) ? x_base + ko : 0xffffffffu, 0, 0);
#pragma unroll
            for (int p = 0; p < NW; p++) {
                if constexpr (CIN)   // wl[t] = the codes of word t of the k-quad (NW = 4 doubles as the word count)
                    wl[p] = __builtin_amdgcn_raw_buffer_load_b128(rc, (q >= 0 && w_base != 0xffffffffu) ? static_cast<uint32_t>(q * pad128(N) + n0 + fl) * 64u + 16u * p : 0xffffffffu, 0, 0);
                else
                    wl[p] = __builtin_amdgcn_raw_buffer_load_b128(rw, (q >= 0 && w_base != 0xffffffffu && p < sh.w) ? static_cast<uint32_t>(p) * w_plane + ((sh.qmajor & 1) ? static_cast<uint32_t>(q * pr.w_lines + n0 + fl) * 16u : w_base + ko) : 0xffffffffu, 0, 0);
            }
        };
        auto multiply = [&](const u32x4 &xl, const u32x4 (&wl)[NW]) {
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const uint32_t xw[1] = {xl[t]};
                const i32x8 xa = strip_operand<1>(xw, 0);
                uint32_t ww[NW];
#pragma unroll
                for (int p = 0; p < NW; p++) ww[p] = wl[p][t];
#pragma unroll
                for (int dw = 0; dw < NDW; dw++) {
                    i32x8 wb;   // swapped: lane (fl, fh) register r holds C[row fl][column (r & 3) + 8 (r >> 2) + 4 fh]
                    if constexpr (CIN) wb = i32x8{static_cast<int>((wl[t][0] >> (2 * dw)) & 0x33333333u), static_cast<int>((wl[t][1] >> (2 * dw)) & 0x33333333u),
                                                  static_cast<int>((wl[t][2] >> (2 * dw)) & 0x33333333u), static_cast<int>((wl[t][3] >> (2 * dw)) & 0x33333333u), 0, 0, 0, 0};
                    else wb = strip_operand<NW>(ww, dw);
                    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wb, xa, acc, 4, 4, 0, 128 + 2 * dw, 0, 128);
                }
            }
        };
        u32x4 xa_, wa_[NW], xb_, wb_[NW];
        auto pa = take();
        load(pa.first, xa_, wa_);
        while (pa.second) {   // wave-uniform; the next pair's loads are issued before the current pair is multiplied
            auto pb = take();
            if (pb.second) load(pb.first, xb_, wb_);
            multiply(xa_, wa_);
            if (!pb.second) break;
            pa = take();
            if (pa.second) load(pa.first, xa_, wa_);
            multiply(xb_, wb_);
        }
    }
    const bool live1 = n0 < N;   // (wave-uniform: a wave past the last column of `out` holds zeros - no arithmetic on them)
    if (!live1) {
        if constexpr (DISC) {
            if (fh == 0) *reinterpret_cast<u32x4 *>(&xchg[wv][fl][0]) = u32x4{0u, 0u, 0u, 0u};
        } else {
            const int row_words = step128(N) * 4, rows_pad = pad8(M);
            const size_t oplane = static_cast<size_t>(rows_pad) * row_words;
            uint32_t *dst = static_cast<uint32_t *>(pr.out) + static_cast<size_t>(m) * row_words + wv;
#pragma unroll
            for (int p = 0; p < OB; p++) {
                if (fh == 0 && m < rows_pad && wv < row_words) dst[p * oplane] = 0u;
                if (fh == 0) xchg[p][fl][wv] = 0u;
            }
        }
    } else if constexpr (DISC) {   // re-quantise; the values themselves (low OB bits: kernel.h:350 keeps c == 2^ob, which packs as 0) are the codes
        uint32_t qv[16], P[4];
        requant_pack16<OB>(acc, ob, P, qv);   // P[t] byte 3 - gq = value of column t + 8 gq + 4 fh
        uint32_t x[4];
#pragma unroll
        for (int t = 0; t < 4; t++) {
            x[t] = (P[t] & (OB <= 2 ? 0x03030303u : 0x0f0f0f0fu)) << (4u - 4u * static_cast<uint32_t>(fh));   // nibble 7 - 2 gq - fh of dword 3 - t
            x[t] = or_with_partner_half(x[t]);
        }
        if (fh == 0) *reinterpret_cast<u32x4 *>(&xchg[wv][fl][0]) = u32x4{x[3], x[2], x[1], x[0]};
    } else {   // re-quantise, pack, store the word of `out` (rows layout, kernel.h:357-389) and hand it to the second product
        const int row_words = step128(N) * 4, rows_pad = pad8(M);
        const size_t oplane = static_cast<size_t>(rows_pad) * row_words;
        const bool store = fh == 0 && m < rows_pad && wv < row_words;
        uint32_t qv[16], P[4];
        requant_pack16<OB>(acc, ob, P, qv);
        uint32_t *dst = static_cast<uint32_t *>(pr.out) + static_cast<size_t>(m) * row_words + wv;
#pragma unroll
        for (int p = 0; p < OB; p++) {
            uint32_t x = ((P[0] >> p) & 0x01010101u) << 3 | ((P[1] >> p) & 0x01010101u) << 2 | ((P[2] >> p) & 0x01010101u) << 1 | ((P[3] >> p) & 0x01010101u);
            x <<= 4u - 4u * static_cast<uint32_t>(fh);   // bits 7 - t - 4 fh of every byte
            x = or_with_partner_half(x);
            if (store) dst[p * oplane] = x;
            if (fh == 0) xchg[p][fl][wv] = x;
        }
    }
    __syncthreads();
    if (n0 >= N2) {   // (wave-uniform) no column of T' here: zero lines of the cols layout, nothing for float32
        if constexpr (COUT) {
            if (fh == 0 && n2 < lines2)
                *reinterpret_cast<u32x4 *>(static_cast<uint32_t *>(pr2.out) + ((static_cast<size_t>(rb >> 2) * lines2 + n2) * 4 + (rb & 3)) * 4) = u32x4{0u, 0u, 0u, 0u};
        } else if constexpr (MODE2 != 2) {
            const size_t oplane2 = static_cast<size_t>(lines2) * line_words2;
            uint32_t *dst = static_cast<uint32_t *>(pr2.out) + ((sh.qmajor & 2) ? (static_cast<size_t>(rb >> 2) * lines2 + n2) * 4 + (rb & 3) : static_cast<size_t>(n2) * line_words2 + rb);
#pragma unroll
            for (int p = 0; p < OB2; p++)
                if (fh == 0 && n2 < lines2) dst[p * oplane2] = 0u;
        }
        return;
    }
    i32x8 w2b[2][NDW2];   // (expanded after the barrier: before it, the codes would be live beside the first product's registers)
#pragma unroll
    for (int h = 0; h < 2; h++)
#pragma unroll
        for (int d = 0; d < NDW2; d++) w2b[h][d] = strip_operand<NW2>(w2l[h], d);

    // ---- second product: T'[32 rows of the block][columns n0 .. n0 + 31] = out . W' over the one k-quad (K' = N <= 128)
    f32x16 acc2;
#pragma unroll
    for (int r = 0; r < 16; r++) acc2[r] = 0.0f;
    if constexpr (DISC) {
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const u32x4 c = *reinterpret_cast<const u32x4 *>(&xchg[2 * fh + h][fl][0]);   // the codes of word 2 fh + h of row fl
#pragma unroll
            for (int da = 0; da < NDX2; da++) {
                i32x8 xa;
                if constexpr (OB <= 2) xa = i32x8{static_cast<int>(c[0]), static_cast<int>(c[1]), static_cast<int>(c[2]), static_cast<int>(c[3]), 0, 0, 0, 0};
                else xa = i32x8{static_cast<int>((c[0] >> (2 * da)) & 0x33333333u), static_cast<int>((c[1] >> (2 * da)) & 0x33333333u),
                                static_cast<int>((c[2] >> (2 * da)) & 0x33333333u), static_cast<int>((c[3] >> (2 * da)) & 0x33333333u), 0, 0, 0, 0};
#pragma unroll
                for (int dw = 0; dw < NDW2; dw++)
                    acc2 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(xa, w2b[h][dw], acc2, 4, 4, 0, 128 + 2 * da, 0, 128 + 2 * dw);
            }
        }
    } else {
        uint32_t xd[2][OB];   // [k half][plane]: words 2 fh, 2 fh + 1 of row fl
#pragma unroll
        for (int p = 0; p < OB; p++) {
            const u32x2 v = *reinterpret_cast<const u32x2 *>(&xchg[p][fl][2 * fh]);
            xd[0][p] = v.x;
            xd[1][p] = v.y;
        }
#pragma unroll
        for (int h = 0; h < 2; h++)
#pragma unroll
            for (int da = 0; da < NDX2; da++) {
                const i32x8 xa = strip_operand<OB>(xd[h], da);   // not swapped: lane (fl, fh) register r holds C[row (r & 3) + 8 (r >> 2) + 4 fh][column fl]
#pragma unroll
                for (int dw = 0; dw < NDW2; dw++)
                    acc2 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(xa, w2b[h][dw], acc2, 4, 4, 0, 128 + 2 * da, 0, 128 + 2 * dw);
            }
    }
    if constexpr (MODE2 == 2) {   // float32 rows: lane (fl, fh) register r is row (r & 3) + 8 (r >> 2) + 4 fh of the block, column n2
        // Branch-free (the range check drops the stores of the columns past N' and the rows past M): with the stores
        // under `if (n2 < N2)`, hipcc sank the MFMAs and the expansion of their operands into that branch - and an MFMA
        // reads the operand registers of ALL lanes, whatever EXEC says: rows past N' came out as garbage.
        const __amdgpu_buffer_rsrc_t ro2 = __builtin_amdgcn_make_buffer_rsrc(pr2.out, 0, static_cast<int>(static_cast<uint32_t>(M) * static_cast<uint32_t>(N2) * 4u), 0x00020000);
        const uint32_t base = n2 < N2 ? (static_cast<uint32_t>(32 * rb) * static_cast<uint32_t>(N2) + static_cast<uint32_t>(n2)) * 4u : 0xffffffffu;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * fh;
            const float v = acc2[r];
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), ro2, (base != 0xffffffffu && 32 * rb + row < M) ? base + static_cast<uint32_t>(row) * static_cast<uint32_t>(N2) * 4u : 0xffffffffu, 0, 0);
        }
    } else if constexpr (COUT) {   // the re-quantised values of word rb of line n2 as codes (see CODES above)
        uint32_t qv[16], P[4];
        requant_pack16<OB2>(acc2, OB2, P, qv);
        uint32_t x[4];
#pragma unroll
        for (int t = 0; t < 4; t++) {
            x[t] = (P[t] & 0x0f0f0f0fu) << (4u - 4u * static_cast<uint32_t>(fh));
            x[t] = or_with_partner_half(x[t]);
        }
        if (fh == 0 && n2 < lines2)
            *reinterpret_cast<u32x4 *>(static_cast<uint32_t *>(pr2.out) + ((static_cast<size_t>(rb >> 2) * lines2 + n2) * 4 + (rb & 3)) * 4) = u32x4{x[3], x[2], x[1], x[0]};
    } else {   // cols layout [ob'][PAD128(N')][STEP128(M) * 4] (kernel.h:651-810 as intended): word rb of line n2, rows past M
        // and lines past N' are zero (their operands were)
        uint32_t qv[16], P[4];
        requant_pack16<OB2>(acc2, OB2, P, qv);
        const size_t oplane2 = static_cast<size_t>(lines2) * line_words2;
        uint32_t *dst = static_cast<uint32_t *>(pr2.out) + ((sh.qmajor & 2) ? (static_cast<size_t>(rb >> 2) * lines2 + n2) * 4 + (rb & 3) : static_cast<size_t>(n2) * line_words2 + rb);
#pragma unroll
        for (int p = 0; p < OB2; p++) {
            uint32_t x = ((P[0] >> p) & 0x01010101u) << 3 | ((P[1] >> p) & 0x01010101u) << 2 | ((P[2] >> p) & 0x01010101u) << 1 | ((P[3] >> p) & 0x01010101u);
            x <<= 4u - 4u * static_cast<uint32_t>(fh);
            x = or_with_partner_half(x);
            if (fh == 0 && n2 < lines2) dst[p * oplane2] = x;
        }
    }
}

// ------------------------------------------------------------------------------------------
// The same second product on its own: a grouped "X . W" stage (K <= 128, N <= 128, cols-layout bits out) by ROW BLOCKS
// instead of column strips (bitmm_fp4_strip.hip.h) - a workgroup = 32 rows of one batch, wave j = columns 32 j .. 32 j + 31,
// the X words straight from global memory (8 bytes per lane and plane), no LDS, no barrier, no line-assembly pass: the
// word of each of its 128 lines goes out as a 4-byte store, and those merge in the one L2 all row blocks of a batch share.
// ------------------------------------------------------------------------------------------
template <int NA, int NW, int OB>
__global__ __launch_bounds__(64 * 4) __attribute__((amdgpu_waves_per_eu(NA <= 2 ? 8 : 4, 8))) void k_bitmm_fp4_xw_rows(
    const qgtc_problem *__restrict__ prs, MMShape sh) {
    constexpr int NDA = (NA + 1) / 2, NDW = (NW + 1) / 2;   // base-4 digits
    pin_shape(sh);
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
    uint32_t *dst = static_cast<uint32_t *>(pr.out) + ((sh.qmajor & 2) ? (static_cast<size_t>(rb >> 2) * lines + n) * 4 + (rb & 3) : static_cast<size_t>(n) * line_words + rb);   // (quad-major: see k_bitmm_fp4_chain)
#pragma unroll
    for (int p = 0; p < OB; p++) {
        uint32_t x = ((P[0] >> p) & 0x01010101u) << 3 | ((P[1] >> p) & 0x01010101u) << 2 | ((P[2] >> p) & 0x01010101u) << 1 | ((P[3] >> p) & 0x01010101u);
        x <<= 4u - 4u * static_cast<uint32_t>(fh);
        x = or_with_partner_half(x);
        if (fh == 0 && n < lines) dst[p * oplane] = x;
    }
}

}  // namespace
