// bitmm_fp4_wide.hip.h — part of libqgtc_hip.so (qgtc_wide.hip).
// The bit-GEMM on the matrix cores for WIDE right operands (N > 256), 1 / 2 / 4 / 8 planes on one side with 1 or 2 on
// the other: packed words
// staged in LDS as they are (LDS-DMA, no expansion pass, no expander waves), expanded in the registers of the
// multiplying waves with one AND per dword.
#pragma once

namespace {

// ------------------------------------------------------------------------------------------
// The 128 x 128-tile engine (bitmm_mfma.hip.h) expands both operands to E2M1 codes in LDS: 8 bytes of LDS traffic per
// packed byte, an expander / multiplier hand-over per k-quad, and 59 cycles per 16x16x128 MFMA at 8192 x 4096 x 1024
// where the matrix pipe needs 8.75 (tools/mfma_rates2.hip). Here:
//   * a workgroup (8 waves, two per SIMD: the VALU work of one overlaps the MFMAs of the other) owns 128 lines of the
//     left operand x 256 lines of the right one, a wave 64 x 64 = 4 x 4 fragments of v_mfma_scale_f32_16x16x128_f8f6f4
//     (64 x 256 or 128 x 128 per workgroup, 2 x 4 / 4 x 2 fragments per wave, when the grid would leave CUs idle);
//   * K is walked in groups of 1024 bits: 128 bytes of every line = one full cache line, fetched ONCE per workgroup by
//     LDS-DMA (buffer_load_dwordx4 .. lds, no registers, no write pass) in pieces of 8 lines x 128 bytes; the 16-byte
//     chunks of a line are XOR-swizzled on the SOURCE address so that a fragment read (16 lines x 4 chunks) touches
//     every bank once. Three stages: the DMA of group g + 2 is issued right after the barrier that publishes group g;
//   * lane (i, kg) of a fragment reads chunk 4 u + kg (u = 0, 1) of line i - four packed words - and MFMA s = 0..3
//     takes bits s, s + 4, .. of all four IN PLACE: nibble code 1 << s = 0.5, 1, 2 with the E8M0 scale 2^(1 - s)
//     (the fourth with one shift: code 8 is -0). Which 128 elements of K an instruction covers is free as long as both
//     operands agree. 2.5 VALU operations per MFMA for one-plane operands; two planes become the 2-bit code of one
//     nibble (6.5 per MFMA), four / eight planes two / four such base-4 digits (an MFMA per pair of digits);
//   * operands swapped (D = R-fragment x L-fragment^T): a lane owns ONE output line and four consecutive elements of it
//     per fragment; fragment fc of a wave is made of the right-hand lines 8 fc .. 8 fc + 7 and 32 + 8 fc .. 32 + 8 fc + 7
//     of the wave's 64, so that the 16 values a lane holds of a line are half of every byte of one output word: byte-
//     packed re-quantised values, one shift + AND per plane and four values, one v_permlane16_swap.
// "Left" / "right" are X / W for the rows layout and float32, W / X for the cols layout (bitMM2Bit_col): the two packed
// layouts are the same [plane][line][k word] arrays, and so are the two output layouts.
// float32 sums: exact while K (2^a - 1)(2^w - 1) < 2^24 (host).
// ------------------------------------------------------------------------------------------
constexpr int WD_WAVES = 8;               // 2 (left) x 4 (right) waves of RF x CF fragments each
constexpr int WD_LDS_MAX = 160 * 1024;
constexpr int WD_PIECE = 1024;            // bytes one LDS-DMA wave-instruction lands: 8 lines x 128 bytes

constexpr int wd_tl(int rf) { return 2 * 16 * rf; }   // lines of the left / right operand per workgroup
constexpr int wd_tr(int cf) { return 4 * 16 * cf; }
// gb = bytes of a line per group of K: 128 (one-plane operands: full cache lines, pieces of 8 lines) or 64 (more planes
// per stage: pieces of 16 lines = one fragment)
constexpr int wd_pieces(int nl, int nr, int rf, int cf, int gb) { return (nl * wd_tl(rf) + nr * wd_tr(cf)) * gb / WD_PIECE; }
constexpr int wd_spare(int nl, int nr, int rf, int cf, int gb) { return wd_pieces(nl, nr, rf, cf, gb) % WD_WAVES ? 1 : 0; }
// three stages (two groups in flight) where they fit the 160 KB of a CU, else two (four- and eight-plane operands)
constexpr int wd_stages(int nl, int nr, int rf, int cf, int gb) {
    return (3 * wd_pieces(nl, nr, rf, cf, gb) + wd_spare(nl, nr, rf, cf, gb)) * WD_PIECE <= WD_LDS_MAX ? 3 : 2;
}
constexpr int wd_lds_bytes(int nl, int nr, int rf, int cf, int gb) {
    return (wd_stages(nl, nr, rf, cf, gb) * wd_pieces(nl, nr, rf, cf, gb) + wd_spare(nl, nr, rf, cf, gb)) * WD_PIECE;
}

// one LDS-DMA instruction: lane i's 16 bytes at (voff + soff) of the buffer land at LDS byte lds_dst + 16 i.
// hipcc does not count this load: the kernel waits for it with its own s_waitcnt vmcnt.
__device__ __forceinline__ void wd_dma(uint32_t lds_dst, uint32_t voff, i32x4 rsrc, uint32_t soff) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "s"(lds_dst), "v"(voff), "s"(rsrc), "s"(soff)
                 : "memory");
}

// MODE 0: packed bits [ob][out_lines][STEP128(Rc) * 4], a word = 32 consecutive right-hand lines of one left-hand line
// (rows layout, kernel.h:357-389, or - operands exchanged by the host - the cols layout, kernel.h:651-810);
// MODE 2: float32 [Lc][Rc] (kernel.h:915-930).
// RF x CF fragments per wave: 4 x 4 when the grid fills the chip, else 2 x 4 (64 x 256 workgroup tiles, 3.75 instead of
// 2.5 VALU operations per MFMA) or 4 x 2 (128 x 128 tiles: the right-hand operand's expansions, the costlier ones when it
// has more planes, are still shared by four fragments).
// GB bytes of every line per group of K (128 for one-plane operands, 64 when the stage has to hold more planes).
template <int NL, int NR, int MODE, int RF, int CF, int GB>
__global__ __launch_bounds__(64 * WD_WAVES) void k_bitmm_fp4_wide(
    const uint32_t *__restrict__ Lp, const uint32_t *__restrict__ Rp, void *__restrict__ outp, uint32_t l_bytes,
    uint32_t r_bytes, uint32_t out_bytes, int Lc, int Rc, int K, int l_lines, int r_lines, int out_lines,
    uint32_t cfg /* ob | tiles along R << 8; host: ob <= 23, every byte count < 2^32 */) {
    static_assert((CF % 4 == 0 || CF == 2) && (RF == 2 || RF == 4) && (GB == 64 || GB == 128), "fragment grid of a wave, group of K");
    static_assert((NL == 1 || NL == 2 || NL == 4 || NL == 8) && (NR == 1 || NR == 2 || NR == 4 || NR == 8), "planes: one, or whole base-4 digits");
    constexpr int NDL = wd_digits(NL), NDR = wd_digits(NR);
    constexpr int WD_STAGES = wd_stages(NL, NR, RF, CF, GB);
    constexpr int TL = wd_tl(RF), TR = wd_tr(CF);
    constexpr int PL = GB == 128 ? 8 : 16;                          // lines per piece
    constexpr int GQ = GB / 16;                                     // k-quads per group
    constexpr int LPP = TL / PL, RPP = TR / PL;                     // pieces per plane
    constexpr int LPC = NL * LPP, RPC = NR * RPP, TOT = LPC + RPC;  // pieces per stage
    constexpr int STAGE = TOT * WD_PIECE;
    constexpr int DMAS = (TOT + WD_WAVES - 1) / WD_WAVES;           // per wave and group
    static_assert(DMAS >= 1 && DMAS <= 15, "vmcnt immediate");
    extern __shared__ __attribute__((aligned(1024))) unsigned char wd_lds[];
#ifdef QGTC_STAMPS
    unsigned long long st_[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define WD_STAMP(i) st_[i] = __builtin_amdgcn_s_memtime()
#else
#define WD_STAMP(i) do { } while (0)
#endif
    WD_STAMP(0);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kg = lane >> 4;
    const int wr = wv >> 2, wc = wv & 3;
    const int ob = cfg & 255u, nt_r = static_cast<int>(cfg >> 8);
    // workgroups that run on one XCD (ids congruent mod 8) take consecutive tiles: they share their left-hand lines in L2
    int tile;
    {
        const int nwg = static_cast<int>(gridDim.x), id = static_cast<int>(blockIdx.x);
        const int q = nwg >> 3, r = nwg & 7, xcd = id & 7;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
    }
    const int tl = tile / nt_r, tr = tile - tl * nt_r;
    const int kq = step128(K);
    const uint32_t row_bytes = static_cast<uint32_t>(kq) * 16u;
    const uint32_t lds0 = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(wd_lds));

    // ---- the DMA plan of this wave: pieces wv, wv + 8, .. of the stage's list [left planes][right planes]. The 16-byte
    // chunks of a line are XOR-swizzled on the SOURCE side (the LDS image of a piece is lane-linear) so that a fragment
    // read touches every bank once. GB = 128: a piece = 8 consecutive lines x 8 chunks, lane (rr, cc) fetches chunk
    // cc ^ (rr & 6). GB = 64: a piece = the 16 lines of one fragment x 4 chunks, lane (rr, cc) fetches chunk
    // cc ^ G(rr >> 2), G = (0, 2, 3, 1). Lines past the operand read whatever follows them (finite E2M1 codes) or are
    // dropped by the range check: the epilogue keeps nothing of them.
    const i32x4 rs_l = {static_cast<int>(reinterpret_cast<uintptr_t>(Lp)), static_cast<int>((reinterpret_cast<uintptr_t>(Lp) >> 32) & 0xffffu), static_cast<int>(l_bytes), 0x00020000};
    const i32x4 rs_r = {static_cast<int>(reinterpret_cast<uintptr_t>(Rp)), static_cast<int>((reinterpret_cast<uintptr_t>(Rp) >> 32) & 0xffffu), static_cast<int>(r_bytes), 0x00020000};
    uint32_t voff_l, voff_r;
    if constexpr (GB == 128) {
        const int rr = lane >> 3, cc = lane & 7;
        const uint32_t swz = static_cast<uint32_t>(cc ^ (rr & 6)) * 16u;
        voff_l = static_cast<uint32_t>(tl * TL + rr) * row_bytes + swz;
        voff_r = static_cast<uint32_t>(tr * TR + rr) * row_bytes + swz;
    } else {
        const int rr = lane >> 2, cc = lane & 3;
        const uint32_t swz = static_cast<uint32_t>(cc ^ ((0x78 >> (2 * (rr >> 2))) & 3)) * 16u;
        voff_l = static_cast<uint32_t>(tl * TL + rr) * row_bytes + swz;
        // (a right-hand fragment: two runs of 8 lines, or - CF = 2 - four runs of 4)
        voff_r = static_cast<uint32_t>(tr * TR + (CF == 2 ? (rr & 3) + 8 * (rr >> 2) : (rr & 7) + 32 * (rr >> 3))) * row_bytes + swz;
    }
    const uint32_t l_plane = static_cast<uint32_t>(l_lines) * row_bytes, r_plane = static_cast<uint32_t>(r_lines) * row_bytes;
    auto issue = [&](int g) {   // group g -> stage g % 3 (groups past the last: garbage into a stage nobody reads)
        const uint32_t base = lds0 + static_cast<uint32_t>(g % WD_STAGES) * STAGE;
        const uint32_t ko = static_cast<uint32_t>(g) * GB;
#pragma unroll
        for (int j = 0; j < DMAS; j++) {
            const int t = wv + WD_WAVES * j;   // (scalar)
            if (t < LPC) {
                const int p = t / LPP, q = t % LPP;
                wd_dma(base + static_cast<uint32_t>(t) * WD_PIECE, voff_l, rs_l, ko + static_cast<uint32_t>(p) * l_plane + static_cast<uint32_t>(PL * q) * row_bytes);
            } else if (TOT % WD_WAVES == 0 || t < TOT) {
                const int p = (t - LPC) / RPP, q = (t - LPC) % RPP;
                const int line = GB == 128 ? 8 * q : CF == 2 ? 32 * (q >> 1) + 4 * (q & 1) : 64 * (q >> 2) + 8 * (q & 3);
                wd_dma(base + static_cast<uint32_t>(t) * WD_PIECE, voff_r, rs_r, ko + static_cast<uint32_t>(p) * r_plane + static_cast<uint32_t>(line) * row_bytes);
            } else {   // (keeps the count of outstanding loads the same for every wave: lands in a spare piece)
                wd_dma(lds0 + static_cast<uint32_t>(WD_STAGES) * STAGE, voff_l, rs_l, 0xfffffff0u);
            }
        }
    };
    issue(0);
    if constexpr (WD_STAGES == 3) issue(1);
    WD_STAMP(1);

    // ---- the fragment reads of this wave: lane (li, kg) takes chunk 4 u + kg of line li of the fragment. Left-hand
    // fragment fr = lines 16 fr .. 16 fr + 15 of the wave's 16 RF; right-hand fragment fc = lines 8 (fc & 3) .. + 7 and
    // 32 + 8 (fc & 3) .. + 7 of block fc >> 2 of 64 lines (a lane's 16 values of a line and block = half of every byte of
    // one word); with CF = 2 (32 lines, one word per wave) fragment fc = lines 8 g + 4 fc .. + 3, g = 0..3: a lane's 8
    // values of a line are ONE byte of the word
    uint32_t la0, ra0;   // + the fragment's and the plane's pieces
    if constexpr (GB == 128) {
        const uint32_t frag_off = static_cast<uint32_t>(li & 7) * 128u + static_cast<uint32_t>(kg ^ (li & 6)) * 16u;
        la0 = static_cast<uint32_t>(2 * RF * wr + (li >> 3)) * WD_PIECE + frag_off;
        if constexpr (CF == 2)   // line 8 (li >> 2) + 4 fc + (li & 3) of the wave's 32: piece li >> 2, row 4 fc + (li & 3) of it
            ra0 = static_cast<uint32_t>(LPC + 4 * wc + (li >> 2)) * WD_PIECE + static_cast<uint32_t>(li & 3) * 128u + static_cast<uint32_t>(kg ^ (li & 2)) * 16u;
        else
            ra0 = static_cast<uint32_t>(LPC + 2 * CF * wc + 4 * (li >> 3)) * WD_PIECE + frag_off;
    } else {
        const uint32_t frag_off = static_cast<uint32_t>(li) * 64u + static_cast<uint32_t>(kg ^ ((0x78 >> (2 * (li >> 2))) & 3)) * 16u;
        la0 = static_cast<uint32_t>(RF * wr) * WD_PIECE + frag_off;
        ra0 = static_cast<uint32_t>(LPC + CF * wc) * WD_PIECE + frag_off;
    }
    auto l_addr = [&](int f, int p, int u) { return (la0 + static_cast<uint32_t>((GB == 128 ? 2 * f : f) + p * LPP) * WD_PIECE) ^ (64u * u); };
    auto r_addr = [&](int f, int p, int u) {
        if constexpr (GB == 128 && CF == 2)   // (row 4 f + .. of the piece: its swizzle term and the chunk's 4 u share bit 6)
            return (ra0 + static_cast<uint32_t>(p * RPP) * WD_PIECE + 512u * f) ^ (64u * (u ^ f));
        else
            return (ra0 + static_cast<uint32_t>((GB == 128 ? 8 * (f >> 2) + (f & 3) : f) + p * RPP) * WD_PIECE) ^ (64u * u);
    };

    f32x4 acc[RF][CF];
#pragma unroll
    for (int i = 0; i < RF; i++)
#pragma unroll
        for (int j = 0; j < CF; j++) acc[i][j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

    // group g has landed for every wave that passes the barrier; the stage of group g - 1 is free again
    auto publish = [&](int g) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((WD_STAGES - 2) * DMAS) : "memory");
        __builtin_amdgcn_s_barrier();
#ifdef QGTC_STAMPS
        if (g == 0) WD_STAMP(2); else if (g == 1) WD_STAMP(3); else if (g == 2) WD_STAMP(4); else if (g == 3) WD_STAMP(5);
        else if (g == 4) WD_STAMP(6); else if (g == 5) WD_STAMP(7); else if (g == 6) WD_STAMP(8); else if (g == 7) WD_STAMP(9);
#endif
        issue(g + WD_STAGES - 1);
    };
    // the GQ (or rem < GQ) k-quads of group g: GB / 64 rounds of (fragment reads, 4 x RF x CF MFMAs)
    auto body = [&](int g, int rem, auto tail_c) {
        constexpr bool TAIL = decltype(tail_c)::value;
        const unsigned char *stage = wd_lds + (g % WD_STAGES) * STAGE;
#pragma unroll
        for (int u = 0; u < GB / 64; u++) {
            if (TAIL && 4 * u >= rem) break;   // (workgroup-uniform)
            WdPrep<NL> lp[RF][NDL];
            WdPrep<NR> rp[CF][NDR];
            const bool live = 4 * u + kg < rem;   // a chunk past K: whatever the DMA found there must not count
#pragma unroll
            for (int f = 0; f < RF; f++) {
                u32x4 raw[NL];
#pragma unroll
                for (int p = 0; p < NL; p++) {
                    raw[p] = *reinterpret_cast<const u32x4 *>(stage + l_addr(f, p, u));
                    if (TAIL && !live) raw[p] = u32x4{0u, 0u, 0u, 0u};
                }
#pragma unroll
                for (int d = 0; d < NDL; d++) wd_prep<NL>(raw, d, lp[f][d]);
            }
#pragma unroll
            for (int f = 0; f < CF; f++) {
                u32x4 raw[NR];
#pragma unroll
                for (int p = 0; p < NR; p++)
                    raw[p] = *reinterpret_cast<const u32x4 *>(stage + r_addr(f, p, u));
#pragma unroll
                for (int d = 0; d < NDR; d++) wd_prep<NR>(raw, d, rp[f][d]);
            }
#pragma unroll
            for (int s = 0; s < 4; s++) {
                i32x8 lo[RF][NDL];
#pragma unroll
                for (int f = 0; f < RF; f++)
#pragma unroll
                    for (int d = 0; d < NDL; d++) lo[f][d] = wd_operand<NL>(lp[f][d], s);
#pragma unroll
                for (int fc = 0; fc < CF; fc++)
#pragma unroll
                    for (int dr = 0; dr < NDR; dr++) {
                        const i32x8 ro = wd_operand<NR>(rp[fc][dr], s);
#pragma unroll
                        for (int fr = 0; fr < RF; fr++)   // lane (li, kg) register j: line 16 fr + li of the wave's lines, right-hand element 4 kg + j of fragment fc
#pragma unroll
                            for (int dl = 0; dl < NDL; dl++)
                                acc[fr][fc] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(ro, lo[fr][dl], acc[fr][fc], 4, 4, 0, wd_scale<NR>(s, dr), 0, wd_scale<NL>(s, dl));
                    }
            }
        }
    };
    const int ng_full = kq / GQ;
    for (int g = 0; g < ng_full; g++) {
        publish(g);
        body(g, GQ, std::false_type{});
    }
    if (kq % GQ) {   // the last, shorter group
        publish(ng_full);
        body(ng_full, kq % GQ, std::true_type{});
    }
#ifdef QGTC_STAMPS
    asm volatile("" : "+v"(acc[0][0]));
#endif
    WD_STAMP(10);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the two groups issued past the end)
    WD_STAMP(11);

    // ---- epilogue, from the accumulators
    const __amdgpu_buffer_rsrc_t ro_ = __builtin_amdgcn_make_buffer_rsrc(outp, 0, static_cast<int>(out_bytes), 0x00020000);
    const int line0 = tl * TL + 16 * RF * wr + li;
    if constexpr (CF == 2) {
        // element 4 kg + j of fragment fc is the right-hand line 8 kg + 4 fc + j of the wave's 32: byte 3 - kg of ONE word
        const int c0 = tr * TR + 32 * wc;
        if (MODE == 2) {
#pragma unroll
            for (int fr = 0; fr < RF; fr++) {
                const int line = line0 + 16 * fr;
#pragma unroll
                for (int fc = 0; fc < 2; fc++) {
                    const int c = c0 + 8 * kg + 4 * fc;
                    const uint32_t off = (static_cast<uint32_t>(line) * static_cast<uint32_t>(Rc) + static_cast<uint32_t>(c)) * 4u;
                    if (line < Lc && c + 3 < Rc && (Rc & 3) == 0) {
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[fr][fc]), ro_, off, 0, 0);
                    } else if (line < Lc) {
#pragma unroll
                        for (int j = 0; j < 4; j++)
                            if (c + j < Rc) {
                                const float v = acc[fr][fc][j];
                                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), ro_, off + 4u * j, 0, 0);
                            }
                    }
                }
            }
        } else {
            const int pitch = step128(Rc) * 4, word = c0 >> 5;
            const uint32_t oplane_bytes = static_cast<uint32_t>(out_lines) * static_cast<uint32_t>(pitch) * 4u;
            const int valid = min(max(Rc - c0, 0), 32);
            const uint32_t wmask = valid >= 32 ? 0xffffffffu : ~(0xffffffffu >> valid);
            const int maxi = 1 << ob;
            const float lim = static_cast<float>(maxi), onesf = static_cast<float>(maxi - 1);
#pragma unroll
            for (int fr = 0; fr < RF; fr++) {
                const int line = line0 + 16 * fr;
                const bool store = kg == 0 && line < out_lines && word < pitch;
                uint32_t off = store ? (static_cast<uint32_t>(line) * static_cast<uint32_t>(pitch) + static_cast<uint32_t>(word)) * 4u : 0xffffffffu;
                const uint32_t keep = line < Lc ? wmask : 0u;
                uint32_t P[2] = {0u, 0u};   // byte 3 - j of P[fc] = re-quantised value (fc, j), planes below 8
                int q[2][4];
#pragma unroll
                for (int fc = 0; fc < 2; fc++)
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        if (ob <= 4) {      // (compare and select on the float, v_cvt_pk_u8_f32 converts and inserts: requant_pack16)
                            float f = acc[fr][fc][j];
                            f = f > lim ? onesf : f;
                            P[fc] = __builtin_amdgcn_cvt_pk_u8_f32(f, 3 - j, P[fc]);
                        } else {
                            const int c = static_cast<int>(acc[fr][fc][j]);
                            q[fc][j] = c > maxi ? maxi - 1 : c;
                            P[fc] |= (static_cast<uint32_t>(q[fc][j]) & 255u) << (8 * (3 - j));
                        }
                    }
                for (int p = 0; p < ob; p++, off += store ? oplane_bytes : 0u) {
                    uint32_t byte;
                    if (p < 8) {
                        const uint32_t t0 = (P[0] >> p) & 0x01010101u, t1 = (P[1] >> p) & 0x01010101u;
                        const uint32_t n0 = (((t0 >> 21) | (t0 >> 14)) | ((t0 >> 7) | t0)) & 0xfu, n1 = (((t1 >> 21) | (t1 >> 14)) | ((t1 >> 7) | t1)) & 0xfu;
                        byte = n0 << 4 | n1;
                    } else {
                        byte = 0u;
#pragma unroll
                        for (int fc = 0; fc < 2; fc++)
#pragma unroll
                            for (int j = 0; j < 4; j++) byte |= ((static_cast<uint32_t>(q[fc][j]) >> p) & 1u) << (7 - 4 * fc - j);
                    }
                    uint32_t x = byte << (8u * (3u - static_cast<uint32_t>(kg)));
                    const auto s16 = __builtin_amdgcn_permlane16_swap(x, x, false, false);   // lanes (li, kg) and (li, kg ^ 1)
                    x = s16[0] | s16[1];
                    x = or_with_partner_half(x) & keep;                                      // and (li, kg ^ 2)
                    __builtin_amdgcn_raw_buffer_store_b32(x, ro_, off, 0, 0);
                }
            }
        }
    } else {
    // element 4 kg + j of fragment fc is the right-hand line 64 (fc >> 2) + 32 (kg >> 1) + 8 (fc & 3) + 4 (kg & 1) + j
    const int col0 = tr * TR + 16 * CF * wc + 32 * (kg >> 1);   // first right-hand line of the lane's output word of block 0
    if (MODE == 2) {
#pragma unroll
        for (int fr = 0; fr < RF; fr++) {
            const int line = line0 + 16 * fr;
#pragma unroll
            for (int fc = 0; fc < CF; fc++) {
                const int c = col0 + 64 * (fc >> 2) + 8 * (fc & 3) + 4 * (kg & 1);
                const uint32_t off = (static_cast<uint32_t>(line) * static_cast<uint32_t>(Rc) + static_cast<uint32_t>(c)) * 4u;
                if (line < Lc && c + 3 < Rc && (Rc & 3) == 0) {
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[fr][fc]), ro_, off, 0, 0);
                } else if (line < Lc) {
#pragma unroll
                    for (int j = 0; j < 4; j++)
                        if (c + j < Rc) {
                            const float v = acc[fr][fc][j];
                            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), ro_, off + 4u * j, 0, 0);
                        }
                }
            }
        }
    } else {
        const int pitch = step128(Rc) * 4;                       // words per output line
        const uint32_t oplane_bytes = static_cast<uint32_t>(out_lines) * static_cast<uint32_t>(pitch) * 4u;
        auto finish = [&](auto ob_c) {
            constexpr int OB = decltype(ob_c)::value;
#pragma unroll
            for (int h = 0; h < CF / 4; h++) {
                const int c0 = col0 + 64 * h, word = c0 >> 5;
                const int valid = min(max(Rc - c0, 0), 32);       // leading bits of the word that are right-hand lines
                const uint32_t wmask = valid >= 32 ? 0xffffffffu : ~(0xffffffffu >> valid);
#pragma unroll
                for (int fr = 0; fr < RF; fr++) {
                    const int line = line0 + 16 * fr;
                    f32x16 v;
#pragma unroll
                    for (int fc = 0; fc < 4; fc++)
#pragma unroll
                        for (int j = 0; j < 4; j++) v[4 * fc + j] = acc[fr][4 * h + fc][j];
                    uint32_t qv[16], P[4];
                    requant_pack16<OB>(v, ob, P, qv);   // P[j] byte 3 - fc = value (fc, j)
                    const bool store = (kg & 1) == 0 && line < out_lines && word < pitch;
                    uint32_t off = store ? (static_cast<uint32_t>(line) * static_cast<uint32_t>(pitch) + static_cast<uint32_t>(word)) * 4u : 0xffffffffu;
                    const uint32_t keep = line < Lc ? wmask : 0u;   // lines past the operand are zeros in both layouts
#pragma unroll
                    for (int p = 0; p < (OB > 0 ? OB : 32); p++) {
                        if (OB == 0 && p >= ob) break;
                        uint32_t x;
                        if (OB > 0 || p < 8) {
                            x = ((P[0] >> p) & 0x01010101u) << 3 | ((P[1] >> p) & 0x01010101u) << 2 | ((P[2] >> p) & 0x01010101u) << 1 | ((P[3] >> p) & 0x01010101u);
                        } else {
                            x = 0u;
#pragma unroll
                            for (int r = 0; r < 16; r++) x |= ((qv[r] >> p) & 1u) << (8 * (3 - (r >> 2)) + 3 - (r & 3));
                        }
                        x <<= 4u - 4u * static_cast<uint32_t>(kg & 1);   // element 4 (kg & 1) + j of byte 3 - fc at bit 7 - 4 (kg & 1) - j
                        const auto s16 = __builtin_amdgcn_permlane16_swap(x, x, false, false);   // lanes (li, kg) and (li, kg ^ 1)
                        x = (s16[0] | s16[1]) & keep;
                        __builtin_amdgcn_raw_buffer_store_b32(x, ro_, off, 0, 0);
                        off += store ? oplane_bytes : 0u;
                    }
                }
            }
        };
        if (ob == 1) finish(std::integral_constant<int, 1>{});
        else if (ob == 2) finish(std::integral_constant<int, 2>{});
        else finish(std::integral_constant<int, 0>{});
    }
    }
#ifdef QGTC_STAMPS
    WD_STAMP(12);
    if (tid == 0) for (int i = 0; i < 16; i++) g_stamps[blockIdx.x % 1024 * 16 + i] = st_[i];
#endif
#undef WD_STAMP
}

}  // namespace
