// bitmm_fp4_strip.hip.h — part of libqgtc_hip.so (qgtc_fp4.hip).
// Grouped launches of the "X . W" stages of a GNN epoch: K <= 128 (the feature / hidden width: one k-quad), cols-layout
// output (bitMM2Bit_col: the product is the right operand of the following A . (XW)).
#pragma once

namespace {

// ------------------------------------------------------------------------------------------
// These stages are 75 products of ~1213 x 128 x 128 with ONE 128-bit step of K each: on 128 x 128 tiles every
// workgroup is fixed cost only (descriptor, one load round trip, expansion, 8 MFMAs, epilogue: ~5 us, 12 us per
// stage), and the cols layout [plane][column n][word m / 32] makes a row tile write 4 (or 16) bytes per column at a
// 160-byte stride: rocprofv3 counted 22 MB of HBM-side writes for 3 MB of output (profiles/r02/summary_epoch.json).
// Here a workgroup owns a STRIP of 32 columns of one batch for ALL rows:
//   * the strip's W lines (32 columns x 128 bits x planes) are loaded and expanded once per wave;
//   * the eight waves walk down the rows in blocks of 32 (wave v takes blocks v, v + 8, ..), the packed X words of six
//     blocks per wave requested up front (one memory round trip for 1536 rows): lane (fl, fh) loads words fh and
//     2 + fh of its row's 128 bits, so TWO v_mfma_scale_f32_32x32x64_f8f6f4 cover the whole K of a 32 x 32 block
//     (base-4 digits for more than two planes);
//   * a block's results are re-quantised from the accumulators: in the 32 x 32 C layout a lane owns 16 of the 32 rows
//     of ONE column, i.e. half an output word per plane (byte-packed values, one shift + AND per plane and four
//     values, one half-wave swap) - 95 VALU operations per block where 16 x 16 fragments needed 260 - and the word
//     goes to LDS at [plane][column][block];
//   * at the end the workgroup writes its 32 lines per plane as whole contiguous lines (160 bytes for 1213 rows).
// Strips past the last column (lines N .. PAD128(N) of the layout) are written as zeros by workgroups of their own.
// float32 sums are exact: K <= 128, a <= 4, w <= 8 gives at most 128 * 15 * 255 < 2^24.
// ------------------------------------------------------------------------------------------
constexpr int ST_WAVES = 8;
constexpr int ST_CHUNK = 6;   // row blocks whose packed X words a wave has in flight at once (8 x 6 x 32 = 1536 rows per pass)

// the lane's packed word (bits 32 g .. 32 g + 31 of its line) of up to NP planes -> E2M1 operand registers of digit d
template <int NP>
__device__ __forceinline__ i32x8 strip_operand(const uint32_t (&pl)[NP], int digit) {
    uint32_t wd[2], e[4];
    wd[0] = pl[2 * digit];
    wd[1] = 2 * digit + 1 < NP ? pl[(2 * digit + 1) % NP] : 0u;
    expand_word_fp4<2>(wd, 2, e);
    return i32x8{static_cast<int>(e[0]), static_cast<int>(e[1]), static_cast<int>(e[2]), static_cast<int>(e[3]), 0, 0, 0, 0};
}

// OB: output planes at compile time (1, 2, 4, 8: the widths the reference publishes; 0 = any, runtime loop)
// Re-quantise the 16 sums a lane holds of a 32 x 32 tile (kernel.h:31-37,350: c > 2^ob ? 2^ob - 1 : c) and pack them a
// byte each: P[t] byte 3 - gq = value of register 4 gq + t (OB = 1, 2: its low OB bits are, the rest of the byte is not
// meaningful). For OB = 1, 2, 4 the clamp happens on the float (an exact integer) and v_cvt_pk_u8_f32 converts AND inserts
// the byte: 2 - 3 operations per value where convert + compare + select + shift/mask/or took 4.5 (these kernels are bound
// by exactly this VALU work). Other widths keep the
// integer route (OB = 8 must map the sum 256 to the byte 0, which a saturating conversion cannot).
template <int OB>
__device__ __forceinline__ void requant_pack16(const f32x16 &acc, int ob, uint32_t (&P)[4], uint32_t (&qv)[16]) {
    const int maxi = 1 << ob;   // (host: ob <= 23, so the reference's float compare c > 2^ob is this integer compare)
    const uint32_t ones = static_cast<uint32_t>(maxi - 1);
    if constexpr (OB == 1 || OB == 2 || OB == 4) {
        const float lim = static_cast<float>(maxi), onesf = static_cast<float>(ones);
#pragma unroll
        for (int t = 0; t < 4; t++) {
            uint32_t pk = 0u;
#pragma unroll
            for (int gq = 0; gq < 4; gq++) {
                float f = acc[4 * gq + t];
                // Only the low OB bits of a byte are read below. OB = 1 / 2: ONE multiplication instead of compare + select -
                // by m = 85 / 53: m = 1 (mod 2^OB), so m c = c (mod 2^OB) for c <= 2^OB (c = 2^OB itself packs as 0,
                // kernel.h:350), and m (2^OB + 1) >= 255, so every larger sum saturates to 255 = 2^OB - 1 (mod 2^OB).
                // (OB = 4 has no such m: 17 m >= 255 and 16 m <= 255 leave m = 15 only, which is not 1 mod 16.)
                if constexpr (OB == 1) f *= 85.0f;
                else if constexpr (OB == 2) f *= 53.0f;
                else f = f > lim ? onesf : f;
                pk = __builtin_amdgcn_cvt_pk_u8_f32(f, 3 - gq, pk);
            }
            P[t] = pk;
        }
    } else {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int c = static_cast<int>(acc[r]);   // exact: the sums are integers below 2^24
            qv[r] = c > maxi ? ones : static_cast<uint32_t>(c);
        }
#pragma unroll
        for (int t = 0; t < 4; t++) P[t] = ((qv[t] & 255u) << 24) | ((qv[4 + t] & 255u) << 16) | ((qv[8 + t] & 255u) << 8) | (qv[12 + t] & 255u);
    }
}

template <int NA, int NW, int OB>
__global__ __launch_bounds__(64 * ST_WAVES) void k_bitmm_fp4_strip(const qgtc_problem *__restrict__ prs, MMShape sh) {
    constexpr int NDA = (NA + 1) / 2, NDW = (NW + 1) / 2;   // base-4 digits
    extern __shared__ __attribute__((aligned(16))) uint32_t strip_words[];   // [ob][32 columns][line_words | 1]
    pin_shape(sh);
    // sh.per != 0: the strips (and parts) of a batch run on ONE XCD - they all read the batch's X rows (bitmm_fp4_rows.hip.h)
    int strip = static_cast<int>(blockIdx.x), batch = static_cast<int>(blockIdx.y), part = static_cast<int>(blockIdx.z);
    if (sh.per) {
        const int gx = static_cast<int>(gridDim.x), gy = static_cast<int>(gridDim.y), gz = static_cast<int>(gridDim.z);
        const int v = xcd_consecutive(strip + gx * (batch + gy * part), gx * gy * gz);
        batch = v / (gx * gz);
        const int rem = v - batch * (gx * gz);
        part = rem / gx;
        strip = rem - part * gx;
    }
    const qgtc_problem pr = prs[batch];
    pin_problem(pr);
    const int M = pr.M, N = pr.N;
    const int n0 = strip * 32;
    const int lines = pad128(N), line_words = step128(M) * 4;
    if (n0 >= lines) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fl = lane & 31, fh = lane >> 5;               // the lane's line of a 32-line fragment, its half of every 64 bits of K
    const int ob = OB > 0 ? OB : sh.ob;
    uint32_t *out = static_cast<uint32_t *>(pr.out);
    const size_t oplane = static_cast<size_t>(lines) * line_words;
    const int nrb = (M + 31) >> 5;                           // row blocks = words of a line that hold rows
    // gridDim.z workgroups share a strip: this one takes the row blocks [rb0, rb1) and writes the words [rb0, w1) of its
    // lines (the last part also the zero words past the last row block). More, smaller workgroups balance better over
    // the CUs: the stage is bound by the VALU work of the re-quantise + pack epilogue, ~190 operations per 32 x 32 block.
    const int per = (line_words + static_cast<int>(gridDim.z) - 1) / static_cast<int>(gridDim.z);
    const int rb0 = part * per, w1 = min(line_words, rb0 + per), rb1 = min(nrb, w1);
    if (rb0 >= line_words) return;
    const int ps = per | 1;   // odd LDS pitch between a plane's columns: 32 lanes write one word each, 32 different banks
#ifdef QGTC_STAMPS
    unsigned long long st_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define ST_STAMP(i) st_[i] = __builtin_amdgcn_s_memtime()
#else
#define ST_STAMP(i) do { } while (0)
#endif
    ST_STAMP(0);
    if (n0 < N) {
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<uint32_t *>(pr.X), 0, static_cast<int>(static_cast<uint32_t>(pr.x_words) * 4u), 0x00020000);
        const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<uint32_t *>(pr.W), 0, static_cast<int>(static_cast<uint32_t>(pr.w_words) * 4u), 0x00020000);
        // one k-quad: a packed line is 16 bytes = words 0..3; lane (fl, fh) loads words 2 fh and 2 fh + 1 of line fl (ONE
        // 8-byte load; 4-byte loads cost the address unit the same 16 cycles for half the data) and supplies word 2 fh + h
        // to MFMA h (v_mfma_scale_f32_32x32x64_f8f6f4, h = 0, 1): which 64 elements of K an instruction covers is free as
        // long as X and W agree (words past K are zero padding)
        const uint32_t x_plane = static_cast<uint32_t>(pad8(M)) * 16u, w_plane = static_cast<uint32_t>(pr.w_lines) * 16u;
        uint32_t wl[2][NW];   // [k half][plane]
        {
            const int n = n0 + fl;
#pragma unroll
            for (int p = 0; p < NW; p++) {
                const u32x2 v = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rw, (n < N && p < sh.w) ? static_cast<uint32_t>(p) * w_plane + static_cast<uint32_t>(n) * 16u + 8u * fh : 0xffffffffu, 0, 0));
                wl[0][p] = v.x;
                wl[1][p] = v.y;
            }
        }
        uint32_t xl[ST_CHUNK][2][NA];   // [row block of the pass][k half][plane]
        auto issue = [&](int rb, uint32_t (&xd)[2][NA]) {   // unconditional: exact vmcnt waits
            const int m = 32 * rb + fl;
#pragma unroll
            for (int p = 0; p < NA; p++) {
                const u32x2 v = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rx, (rb < rb1 && m < M && p < sh.a) ? static_cast<uint32_t>(p) * x_plane + static_cast<uint32_t>(m) * 16u + 8u * fh : 0xffffffffu, 0, 0));
                xd[0][p] = v.x;
                xd[1][p] = v.y;
            }
        };
        // the first pass's X words are requested before anything waits for W: one memory round trip for both
#pragma unroll
        for (int c = 0; c < ST_CHUNK; c++) issue(rb0 + wv + ST_WAVES * c, xl[c]);
        i32x8 wb[2][NDW];   // [k half][digit]
#pragma unroll
        for (int h = 0; h < 2; h++)
#pragma unroll
            for (int d = 0; d < NDW; d++) wb[h][d] = strip_operand<NW>(wl[h], d);
#ifdef QGTC_STAMPS
        asm volatile("" ::"v"(wb[0][0][0]), "v"(wb[1][0][3]));
#endif
        ST_STAMP(1);
        // one row block = one 32 x 32 tile: two MFMAs (the halves of K) per pair of base-4 digits. Not swapped: lane
        // (fl, fh) register r holds C[row (r & 3) + 8 (r >> 2) + 4 fh][column fl]: 16 of the 32 bits of ONE output word
        auto mma = [&](const uint32_t (&xd)[2][NA], f32x16 &acc) {
#pragma unroll
            for (int r = 0; r < 16; r++) acc[r] = 0.0f;
#pragma unroll
            for (int h = 0; h < 2; h++)
#pragma unroll
                for (int da = 0; da < NDA; da++) {
                    const i32x8 xa = strip_operand<NA>(xd[h], da);
#pragma unroll
                    for (int dw = 0; dw < NDW; dw++)
                        acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(xa, wb[h][dw], acc, 4, 4, 0, 128 + 2 * da, 0, 128 + 2 * dw);
                }
        };
        // requantise (kernel.h:31-37,350); rows past M and columns past N are zero already (their operand lines were
        // loaded as zeros). Row e = t + 8 gq + 4 fh (t = r & 3, gq = r >> 2) sits at bit 31 - e = 8 (3 - gq) + (7 - t - 4 fh)
        // of the column's word: the values of one t are packed a byte each (byte 3 - gq) and plane p of the four is
        // ONE shift + AND (the scheme of bitmm_mfma.hip.h's epilogue); the partner lane (fl, fh ^ 1) holds the other 16 bits
        auto finish = [&](int rb, const f32x16 &acc) {
            uint32_t q[16], P[4];
            requant_pack16<OB>(acc, ob, P, q);
#pragma unroll
            for (int p = 0; p < (OB > 0 ? OB : 32); p++) {
                if (OB == 0 && p >= ob) break;
                uint32_t x;
                if (OB > 0 || p < 8) {
                    x = ((P[0] >> p) & 0x01010101u) << 3 | ((P[1] >> p) & 0x01010101u) << 2 | ((P[2] >> p) & 0x01010101u) << 1 | ((P[3] >> p) & 0x01010101u);
                } else {   // more than 8 output planes: from the full values
                    x = 0u;
#pragma unroll
                    for (int r = 0; r < 16; r++) x |= ((q[r] >> p) & 1u) << (8 * (3 - (r >> 2)) + 3 - (r & 3));
                }
                x <<= 4u - 4u * static_cast<uint32_t>(fh);   // bits 7 - t - 4 fh of every byte
                x = or_with_partner_half(x);
                if (fh == 0) strip_words[(p * 32 + fl) * ps + (rb - rb0)] = x;
            }
        };
        // a pass = ST_WAVES x ST_CHUNK row blocks: every packed X word of the pass is requested before the first is used
        // (these stages are latency-bound: one memory round trip per pass instead of one per block), then all the MFMAs
        // of the pass, then the epilogues (no epilogue waits on the MFMA it has just issued)
        for (int base = rb0; base < rb1; base += ST_WAVES * ST_CHUNK) {
            f32x16 acc[ST_CHUNK];
            if (base > rb0) {
#pragma unroll
                for (int c = 0; c < ST_CHUNK; c++) issue(base + wv + ST_WAVES * c, xl[c]);
            }
#pragma unroll
            for (int c = 0; c < ST_CHUNK; c++) mma(xl[c], acc[c]);
#ifdef QGTC_STAMPS
            asm volatile("" ::"v"(acc[0][0]), "v"(acc[ST_CHUNK - 1][15]));
#endif
            ST_STAMP(2);
#pragma unroll
            for (int c = 0; c < ST_CHUNK; c++) {
                const int rb = base + wv + ST_WAVES * c;
                if (rb < rb1) finish(rb, acc[c]);
            }
        }
        ST_STAMP(3);
        __syncthreads();
        ST_STAMP(4);
    }
    // ---- the strip's lines, whole and contiguous: [plane][line n0 .. n0 + 31][line_words]; words past the last row
    // block, and every word of a strip past the last column, are zero
    const bool live = n0 < N;
    const int nw = w1 - rb0;                                  // words of each line this workgroup writes
    uint32_t *dst0 = out + static_cast<size_t>(n0) * line_words + rb0;
    const uint32_t inv_nw = 0xffffffffu / static_cast<uint32_t>(nw) + 1u;   // ceil(2^32 / nw), once per workgroup
    for (int idx = tid; idx < ob * 32 * nw; idx += 64 * ST_WAVES) {
        // line = idx / nw by multiply-high (exact here: idx x nw < 2^32)
        const int line = nw == 1 ? idx : static_cast<int>(__umulhi(static_cast<uint32_t>(idx), inv_nw));   // (2^32 / 1 does not fit)
        const int wi = idx - line * nw, p = line >> 5;
        dst0[p * oplane + static_cast<size_t>(line & 31) * line_words + wi] =
            (live && rb0 + wi < rb1) ? strip_words[line * ps + wi] : 0u;
    }
#ifdef QGTC_STAMPS
    ST_STAMP(5);
    if (tid == 0) {
        const int slot = (batch * gridDim.x + strip) % 1024;
        for (int i = 0; i < 8; i++) g_stamps[slot * 16 + i] = st_[i];
    }
#endif
#undef ST_STAMP
}

}  // namespace
