// bitmm_mfma.hip.h — part of libqgtc_hip.so (included by qgtc_hip.hip, one translation unit).
// The bit-GEMM, matrix-core engine (opt-in): bit planes expanded to int8 on the fly,
// v_mfma_i32_32x32x32_i8, same words as the popcount engine.
#pragma once

namespace {

// ------------------------------------------------------------------------------------------
// The bit-GEMM on the matrix cores (opt-in engine, QGTC_ENGINE_MFMA): same operands, same
// results, but the bit planes of both operands are expanded on the fly to int8 VALUES (plane p
// contributes bit p of the byte, so a planes x w planes collapse into ONE product) and multiplied
// with v_mfma_i32_32x32x32_i8, int32 accumulation, exact. gfx950 has no 1-bit MFMA; expanding
// costs ~0.8 VALU operations per operand byte, which only pays when an expanded byte is reused by
// several MFMA tiles: a workgroup owns a 128 x 128 output tile (four waves, 64 x 64 each), so this
// engine is for wide N (>= 128) and/or several planes; the popcount kernels stay the default and
// remain the faster path at N = 64. Needs a, w <= 8. An 8-plane operand does
// not fit a non-negative int8: its plane 7 is inverted on the way in (byte = value - 128 as int8) and
// the product is corrected in the epilogue with the line sums of the other operand,
//   sum x w = sum x' w' + 128 [w has 8 planes] sum x + 128 [x has 8 planes] sum w - 16384 [both] K',
// x', w' the bytes as multiplied, the sums over the K' = 128 * (visited k-quads) positions - exact in
// int32 arithmetic mod 2^32, like everything else here. The sums are popcounts of the packed words.
//
// Per k-quad (128 bits of K) one thread of the expander waves expands one row of X or one column of W:
// dword d of a word's 32 bytes takes the word's bits d, d+8, d+16, d+24 (a shift and an AND per plane and
// dword - the order of the elements inside a word is irrelevant to the product as long as both operands
// share it); the 128 bytes go to LDS ([line][144-byte pitch]: the 16-byte MFMA fragment reads of 32 lines are
// conflict-free). The packed words of the next k-quad are loaded (range-checked buffer loads)
// while the current one is multiplied. The three epilogues (rows-layout bits, cols-layout bits, float32)
// work straight from the accumulators of the multiplier waves (the rows layout multiplies with the
// operands swapped, so that a lane owns a row); with 128-wide tiles every output word belongs to
// exactly one workgroup, so there is no padding to zero-fill separately.
// ------------------------------------------------------------------------------------------
//
// FP4 form (plane counts <= 2, sums below 2^24): E2M1 codes 0..3 are 0, 0.5, 1, 1.5, so a 2-bit value v
// stored as the nibble v means v / 2 and v_mfma_scale_f32_32x32x64_f8f6f4 with the E8M0 scale 2 on both
// operands returns the integer product exactly in float32 (tools/fp4_probe.hip). Half the expanded bytes
// (64 per line and k-quad), half the expansion work (dword d of a word's 16 bytes takes bits d, d+4, ..),
// half the fragment reads and twice the MFMA rate of the int8 form.
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int MF_PITCH4 = 80;      // FP4 form: bytes between the expanded lines (64 + 16: conflict-free 16-byte reads)
constexpr int MF_T = 128;          // tile edge
constexpr int MF_PITCH = 144;      // bytes between the expanded lines of one operand
constexpr int MF_STAGE = 2 * MF_T * MF_PITCH;                 // one staging buffer (int8 form): X lines, then W lines
constexpr int MF_SUMS = 0;  // line sums of 8-plane operands (128 X rows, 128 W columns): over the dead staging buffers
__host__ __device__ constexpr int mf_lds_bytes(bool fp4) { return 2 * 2 * MF_T * (fp4 ? 80 : MF_PITCH); }  // two staging buffers

// 32 packed elements of `planes` planes -> 32 bytes (8 dwords), byte = sum_p bit_p << p.
// The dot product over k does not care in which ORDER the 32 elements of a word become bytes as long as
// X and W use the same order, so the cheapest bijection is taken: dword d holds the word's bits d, d+8,
// d+16, d+24 (one per byte) - a shift and an AND with 0x01010101 << p per (plane, dword), no bit reversal,
// no per-nibble multiply. (Plane 7 of an 8-plane operand arrives inverted: byte = value - 128.)
template <int MAXP>
__device__ __forceinline__ void expand_word(const uint32_t (&wd)[MAXP], int planes, uint32_t (&out)[8]) {
#pragma unroll
    for (int d = 0; d < 8; d++) out[d] = 0u;
#pragma unroll
    for (int p = 0; p < MAXP; p++) {
        if (p >= planes) break;
        const uint32_t r = p == 7 ? ~wd[p] : wd[p];
#pragma unroll
        for (int d = 0; d < 8; d++) {
            // bits d + 8i of r to bits p + 8i (p + 24 <= 31: a left shift loses nothing that is kept)
            const uint32_t t = d > p ? r >> (d - p) : (d < p ? r << (p - d) : r);
            out[d] |= t & (0x01010101u << p);
        }
    }
}

// FP4 form: 32 packed elements of 1 or 2 planes -> 32 nibbles (4 dwords), nibble = b0 + 2 b1 (= E2M1 v / 2)
template <int MAXP>
__device__ __forceinline__ void expand_word_fp4(const uint32_t (&wd)[MAXP], int planes, uint32_t (&out)[4]) {
#pragma unroll
    for (int d = 0; d < 4; d++) out[d] = 0u;
#pragma unroll
    for (int p = 0; p < MAXP; p++) {
        if (p >= planes || p >= 2) break;
        const uint32_t r = wd[p];
#pragma unroll
        for (int d = 0; d < 4; d++) {
            const uint32_t t = d > p ? r >> (d - p) : (d < p ? r << (p - d) : r);  // bits d + 4i to bits p + 4i
            out[d] |= t & (0x11111111u << p);
        }
    }
}

// x | (x of lane ^ 32): v_permlane32_swap (gfx950) exchanges the wave's halves in the VALU, no LDS round trip
__device__ __forceinline__ uint32_t or_with_partner_half(uint32_t x) {
    const auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);  // {lo, lo} and {hi, hi} of x
    return r[0] | r[1];
}

// EXPW expander waves: 8 (two threads per line) when a CU holds one workgroup - a lone expander wave per
// SIMD is latency-bound - or 4 (one thread per line, fewer registers per workgroup) when the grid is large
// enough for two workgroups per CU to overlap each other.
//
// Zero-tile jumping: when the problem carries the occupancy bitmap of its left operand (pr.occ, one
// 64-bit word per 32-row tile, i.e. K <= 8192) the workgroup ORs the words of its four row tiles and
// visits only the k-quads whose 128-row x 128-bit X tile has a bit set - every wave derives the same
// sequence from the same scalar loads, so the expander / multiplier hand-over needs nothing extra.
// WCOH: the right operand was written by other workgroups of THIS launch (fused layer, stage 2): its loads carry
// agent scope (sc1: served from the memory side, never from a line this XCD's L2 filled before the writer was done).
template <int MAXP, int EXPW, bool FP4 = false, bool PUB = false, bool WCOH = false>
__device__ __forceinline__ void mf_tile(const qgtc_problem &pr, const MMShape &sh, int tm, int tn,
                                        unsigned char *smem) {
    static_assert(!FP4 || MAXP <= 2, "the FP4 form holds 2-bit values at most");
    constexpr int PITCH = FP4 ? MF_PITCH4 : MF_PITCH;   // bytes between expanded lines
    constexpr int STAGE = 2 * MF_T * PITCH;             // one staging buffer: X lines, then W lines
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int M = pr.M, K = pr.K, N = pr.N;
    const int m0 = tm * MF_T, n0 = tn * MF_T;
    const int kq = step128(K);
    // the k-quads this tile visits: all kq of them in order, or the set bits of `kmask`
    const bool jump = pr.occ != nullptr && pr.occ_words == 1;
    unsigned long long kmask = 0ull;
    if (jump) {
        // four independent scalar loads (a row tile past the end re-reads the last one: OR is idempotent)
        const int rt0 = 4 * tm, rt_last = ((M + 31) >> 5) - 1;
        const unsigned long long o0 = pr.occ[rt0], o1 = pr.occ[min(rt0 + 1, rt_last)];
        const unsigned long long o2 = pr.occ[min(rt0 + 2, rt_last)], o3 = pr.occ[min(rt0 + 3, rt_last)];
        kmask = (o0 | o1) | (o2 | o3);
    }
    const int nq = jump ? __builtin_popcountll(kmask) : kq;   // steps of the main loop
    // Waves 0-3 multiply, waves 4.. expand: waves v, v+4 (and v+8) share a SIMD, so the matrix pipe
    // (multiplying k-quad q) and the vector pipe (expanding k-quad q+1) of every SIMD run side by
    // side. Two staging buffers, one barrier per k-quad.
    const bool expander = wv >= 4;
#ifdef QGTC_STAMPS
    unsigned long long st_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define MF_STAMP(i) st_[i] = __builtin_amdgcn_s_memtime()
#else
#define MF_STAMP(i) do { } while (0)
#endif
    MF_STAMP(0);

    i32x16 acc[2][2];
    f32x16 accf[2][2];  // FP4 form (the unused set is dead code)
    const int mw = wv & 3, wr = mw >> 1, wc = mw & 1;   // multiplier wave (wr, wc): a 64 x 64 quarter, 2 x 2 MFMA tiles
    const int fl = lane & 31, fh = lane >> 5;           // fragment line, k half (16 bytes each)
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                acc[i][j][r] = 0;
                accf[i][j][r] = 0.0f;
            }

    if (expander) {
        const uint32_t kw = static_cast<uint32_t>(kq) * 4u;
        const uint32_t x_plane = static_cast<uint32_t>(pad8(M)) * kw, w_plane = static_cast<uint32_t>(pr.w_lines) * kw;
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<uint32_t *>(pr.X), 0, static_cast<int>(static_cast<uint32_t>(pr.x_words) * 4u), 0x00020000);
        const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<uint32_t *>(pr.W), 0, static_cast<int>(static_cast<uint32_t>(pr.w_words) * 4u), 0x00020000);
        // this thread's expansion unit: one line (row of X / column of W) of the tile; the wave-level
        // split (waves 4,5: X rows, waves 6,7: W columns) keeps the descriptor choice wave-uniform
        // two threads per line (each expands two of the four words of a k-quad): eight expander waves, so
        // that every SIMD has two of them to interleave - one expander wave per SIMD is latency-bound
        constexpr int TPL = EXPW / 4;  // threads per line
        const int u = (tid - 256) / TPL, hw = TPL == 2 ? (tid & 1) : 0;
        const bool is_x = wv < 4 + EXPW / 2;
        const int line = is_x ? u : u - MF_T;
        const int gline = (is_x ? m0 : n0) + line;
        const bool line_ok = gline < (is_x ? M : N);
        const int planes = is_x ? sh.a : sh.w;
        const uint32_t plane_words = is_x ? x_plane : w_plane;
        const uint32_t base = static_cast<uint32_t>(gline) * kw * 4u;  // byte offset of the line inside a plane
        unsigned char *my_stage = smem + (is_x ? 0 : MF_T * PITCH) + line * PITCH;
        // Packed words are loaded GQ k-quads at a time per line (GQ * 16 contiguous bytes per lane): one
        // k-quad per load instruction touches 64 different 128-byte lines for 16 bytes each and the L1
        // (32 KB) does not keep them until the next k-quad - measured: 1300 cycles per k-quad, all of
        // it L2 -> L1 traffic. Two register sets: group g+1 is in flight while group g is expanded.
        constexpr int GQ = MAXP <= 2 ? 4 : (MAXP <= 4 ? 2 : 1);
        u32x4 grp[2][GQ][MAXP];
        // groups are requested strictly in order, so the k-quad stream is a running cursor
        int k_seq = 0;
        unsigned long long k_left = kmask;
        auto next_k = [&]() -> int {  // next k-quad to load, kq when the stream is exhausted
            if (!jump) return k_seq < kq ? k_seq++ : kq;
            if (k_left == 0ull) return kq;
            const int k = __builtin_ctzll(k_left);
            k_left &= k_left - 1ull;
            return k;
        };
        auto issue_group = [&](u32x4 (&dst)[GQ][MAXP]) {
#pragma unroll
            for (int j = 0; j < GQ; j++) {
                const int q = next_k();
#pragma unroll
                for (int p = 0; p < MAXP; p++) {
                    const bool ok = line_ok && p < planes && q < kq;
                    const uint32_t off = ok ? static_cast<uint32_t>(p) * plane_words * 4u + base + static_cast<uint32_t>(q) * 16u : 0xffffffffu;
                    dst[j][p] = is_x ? __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0)
                                     : __builtin_amdgcn_raw_buffer_load_b128(rw, off, 0, WCOH ? AUX_SC1 : 0);
                }
            }
        };
        // 8-plane operands: the epilogue needs this line's sum of values when the OTHER operand is the offset one
        const bool need_sum = MAXP == 8 && (is_x ? sh.w == 8 : sh.a == 8);
        uint32_t cnt[MAXP];  // set bits of the line, per plane
#pragma unroll
        for (int p = 0; p < MAXP; p++) cnt[p] = 0u;
        auto expand = [&](int q, const u32x4 (&src)[MAXP]) {  // packed words of step q -> bytes in staging buffer q & 1
            unsigned char *stage = my_stage + (q & 1) * STAGE;
#pragma unroll
            for (int cc = 0; cc < 4 / TPL; cc++) {
                const int c = (4 / TPL) * hw + cc;  // this thread's words of the k-quad
                uint32_t wd[MAXP], out[8];
#pragma unroll
                for (int p = 0; p < MAXP; p++) wd[p] = (TPL == 2 && hw) ? src[p][2 + cc] : src[p][cc];
                if (MAXP == 8 && need_sum) {
#pragma unroll
                    for (int p = 0; p < MAXP; p++) cnt[p] += __popc(wd[p]);
                }
                if constexpr (FP4) {
                    uint32_t o4[4];
                    expand_word_fp4<MAXP>(wd, planes, o4);
                    *reinterpret_cast<u32x4 *>(stage + c * 16) = u32x4{o4[0], o4[1], o4[2], o4[3]};
                    continue;
                }
#ifdef QGTC_MF_NOEXPAND  // timing-only build
#pragma unroll
                for (int d = 0; d < 8; d++) out[d] = wd[0];
#else
                expand_word<MAXP>(wd, planes, out);
#endif
#ifdef QGTC_MF_NOWRITE  // timing-only build
                asm volatile("" ::"v"(out[0]), "v"(out[1]), "v"(out[2]), "v"(out[3]), "v"(out[4]), "v"(out[5]), "v"(out[6]), "v"(out[7]));
#else
                *reinterpret_cast<u32x4 *>(stage + c * 32) = u32x4{out[0], out[1], out[2], out[3]};
                *reinterpret_cast<u32x4 *>(stage + c * 32 + 16) = u32x4{out[4], out[5], out[6], out[7]};
#endif
            }
        };
        issue_group(grp[0]);
        issue_group(grp[1]);
        MF_STAMP(1);
        if (nq > 0) expand(0, grp[0][0]);
        if (GQ == 1) issue_group(grp[0]);
        MF_STAMP(2);
        __syncthreads();
        MF_STAMP(3);
        // step J of a block of 2*GQ: step q0+J is being multiplied; expand step e = q0+J+1 (set (e/GQ)&1,
        // slot e%GQ); after the last slot of a set, refill the set with the group two ahead.
        // The refill is UNCONDITIONAL (an exhausted k stream loads zeros from offset 0xffffffff) and the
        // block is left with `break`, not skipped step by step: with loads under a condition hipcc must
        // assume at every later wait that they were not issued, and then waits for the just-issued group
        // (a full memory latency every GQ-th step) instead of the one four steps old.
#ifdef QGTC_STEP_TRACE  // the same for an X expander wave (256) and a W expander wave (last wave)
#define MF_TRACE(Q, W) do { if (blockIdx.x == 0 && (Q) < 40 && (tid == 256 || tid == 64 * (4 + EXPW) - 64)) g_stamps[8192 + (tid == 256 ? 128 : 256) + 2 * (Q) + (W)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define MF_TRACE(Q, W) do { } while (0)
#endif
#define QGTC_MF_STEP(J)                                                                   \
    if (J < 2 * GQ) {                                                                     \
        if (q0 + J >= nq) break;                                                          \
        constexpr int E = (J + 1) % (2 * GQ);                                             \
        if (q0 + J + 1 < nq) expand(q0 + J + 1, grp[E / GQ][E % GQ]);                     \
        if (E % GQ == GQ - 1) issue_group(grp[E / GQ]);                                   \
        if (q0 + J == 8) MF_STAMP(6);                                                     \
        MF_TRACE(q0 + J, 0);                                                              \
        __syncthreads();                                                                  \
        MF_TRACE(q0 + J, 1);                                                              \
        if (q0 + J == 8) MF_STAMP(7);                                                     \
    }
        for (int q0 = 0; q0 < nq; q0 += 2 * GQ) {
            if (q0 == 2 * GQ) MF_STAMP(4);
            QGTC_MF_STEP(0)
            QGTC_MF_STEP(1)
            QGTC_MF_STEP(2)
            QGTC_MF_STEP(3)
            QGTC_MF_STEP(4)
            QGTC_MF_STEP(5)
            QGTC_MF_STEP(6)
            QGTC_MF_STEP(7)
        }
#undef QGTC_MF_STEP
#undef MF_TRACE
        MF_STAMP(5);
        if (MAXP == 8 && need_sum) {  // (the staging buffers are dead: the last barrier is behind every read)
            uint32_t tot = 0u;
#pragma unroll
            for (int p = 0; p < MAXP; p++) tot += cnt[p] << p;
            if (TPL == 2) tot += static_cast<uint32_t>(__shfl_xor(static_cast<int>(tot), 1));
            if (hw == 0) reinterpret_cast<uint32_t *>(smem + MF_SUMS)[(is_x ? 0 : MF_T) + line] = tot;
        }
    } else {
        __syncthreads();
        MF_STAMP(3);
        // Rows-layout output (mode 0) multiplies with the operands SWAPPED: the accumulator tile is then the
        // transpose (lane = row of C, registers = 16 of its columns), which is what a row word needs.
        auto main_loop = [&](auto swap_c) {
            constexpr bool SWAP = decltype(swap_c)::value;
        for (int q = 0; q < nq; q++) {
                if (q == 8) MF_STAMP(4);
                const unsigned char *xs = smem + (q & 1) * STAGE + (64 * wr + fl) * PITCH + 16 * fh;
                const unsigned char *ws = smem + (q & 1) * STAGE + MF_T * PITCH + (64 * wc + fl) * PITCH + 16 * fh;
                // fragments of k sub-step s+1 are read from LDS while sub-step s is multiplied
                constexpr int NSUB = FP4 ? 2 : 4;   // FP4: 64 elements of K per MFMA, int8: 32; 16 bytes per lane either way
                i32x4 af[2][2], bf[2][2];
    #pragma unroll
                for (int i = 0; i < 2; i++) {
                    af[0][i] = *reinterpret_cast<const i32x4 *>(xs + 32 * i * PITCH);
                    bf[0][i] = *reinterpret_cast<const i32x4 *>(ws + 32 * i * PITCH);
                }
    #pragma unroll
                for (int sub = 0; sub < NSUB; sub++) {
                    if (sub + 1 < NSUB) {
    #pragma unroll
                        for (int i = 0; i < 2; i++) {
                            af[(sub + 1) & 1][i] = *reinterpret_cast<const i32x4 *>(xs + 32 * i * PITCH + 32 * (sub + 1));
                            bf[(sub + 1) & 1][i] = *reinterpret_cast<const i32x4 *>(ws + 32 * i * PITCH + 32 * (sub + 1));
                        }
                    }
    #pragma unroll
                    for (int i = 0; i < 2; i++)
    #pragma unroll
                        for (int j = 0; j < 2; j++) {
    #ifdef QGTC_MF_NOMFMA  // timing-only build
                            asm volatile("" ::"v"(af[sub & 1][i]), "v"(bf[sub & 1][j]));
    #else
                            if constexpr (FP4) {
                                const i32x4 fa = SWAP ? bf[sub & 1][j] : af[sub & 1][i], fb = SWAP ? af[sub & 1][i] : bf[sub & 1][j];
                                const i32x8 a8 = {fa.x, fa.y, fa.z, fa.w, 0, 0, 0, 0}, b8 = {fb.x, fb.y, fb.z, fb.w, 0, 0, 0, 0};
                                // cbsz = blgp = 4: E2M1 operands; E8M0 scale 128 = x2 on each: nibble v counts as v
                                accf[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, accf[i][j], 4, 4, 0, 128, 0, 128);
                            } else {
                                acc[i][j] = SWAP ? __builtin_amdgcn_mfma_i32_32x32x32_i8(bf[sub & 1][j], af[sub & 1][i], acc[i][j], 0, 0, 0)
                                                 : __builtin_amdgcn_mfma_i32_32x32x32_i8(af[sub & 1][i], bf[sub & 1][j], acc[i][j], 0, 0, 0);
                            }
    #endif
                        }
                }
                if (q == 8) MF_STAMP(1);
    #ifdef QGTC_STEP_TRACE  // diagnostic: when each step's MFMAs were issued / its barrier passed (workgroup 0, wave 0)
                if (tid == 0 && blockIdx.x == 0 && q < 40) g_stamps[8192 + 2 * q] = __builtin_amdgcn_s_memtime();
    #endif
                __syncthreads();
    #ifdef QGTC_STEP_TRACE
                if (tid == 0 && blockIdx.x == 0 && q < 40) g_stamps[8192 + 2 * q + 1] = __builtin_amdgcn_s_memtime();
    #endif
                if (q == 8) MF_STAMP(2);
            }
        };
        if (sh.mode == 0) main_loop(std::true_type{});
        else main_loop(std::false_type{});
        MF_STAMP(5);
        if constexpr (FP4) {  // the float32 sums are exact integers
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 2; j++)
#pragma unroll
                    for (int r = 0; r < 16; r++) acc[i][j][r] = static_cast<int>(accf[i][j][r]);
        }
    }
    __syncthreads();  // (8-plane operands: the expanders' line sums are visible)
    if (!expander) MF_STAMP(6);

    // ---- epilogue, straight from the accumulators (multiplier waves; the expanders are done).
    // MFMA C/D layout of a 32 x 32 tile: lane l owns "line" l & 31 and, in register r, "element"
    // e(r) = (r & 3) + 8 (r >> 2) + 4 (l >> 5). Not swapped: line = column of C, element = row (what a
    // cols-layout word - 32 rows of a column - and coalesced float rows need); swapped: line = row, element =
    // column (a rows-layout word). A lane therefore holds 16 of the 32 bits of one output word per plane and
    // its partner lane l ^ 32 the other 16. Bit of element e: 31 - e = 8 (3 - g) + (7 - t - 4 fh) with
    // g = r >> 2, t = r & 3: the re-quantised values of one t are packed a byte each (byte 3 - g) and plane p of
    // the four is ONE shift + AND; the four t are merged with constant shifts.
    if (!expander) {
        const int *sums = reinterpret_cast<const int *>(smem + MF_SUMS);
        const bool fix = MAXP == 8 && (sh.a == 8 || sh.w == 8);   // 8-plane operands (header)
        const int fix_x = sh.w == 8 ? 128 : 0, fix_w = sh.a == 8 ? 128 : 0;
        const int fix_k = (sh.a == 8 && sh.w == 8) ? static_cast<int>(0u - 2097152u * static_cast<uint32_t>(nq)) : 0;  // -16384 K' mod 2^32
        const bool swapped = sh.mode == 0;
        if (MAXP == 8 && fix) {
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    // tile-local row / column of the lane's line and of element 0
                    const int row_l = 64 * wr + 32 * i, col_l = 64 * wc + 32 * j;
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        const int e = (r & 3) + 8 * (r >> 2) + 4 * fh;
                        const int row = swapped ? row_l + fl : row_l + e, col = swapped ? col_l + e : col_l + fl;
                        acc[i][j][r] += fix_x * sums[row] + fix_w * sums[MF_T + col] + fix_k;
                    }
                }
        }
        if (sh.mode == 2) {  // float32 [M,N] (reference kernel.h:915-930): 32 lanes = 32 consecutive columns of a row
            float *outf = static_cast<float *>(pr.out);
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    const int col = n0 + 64 * wc + 32 * j + fl;
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        const int row = m0 + 64 * wr + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * fh;
                        if (row < M && col < N) outf[static_cast<size_t>(row) * N + col] = static_cast<float>(acc[i][j][r]);
                    }
                }
        } else {
            // rows layout [ob][PAD8(M)][STEP128(N)*4] (kernel.h:357-389): word (m, n / 32), swapped tiles;
            // cols layout [ob][PAD128(N)][STEP128(M)*4] (kernel.h:651-810 as intended): word (n, m / 32)
            const int rows_pad = pad8(M), row_words = step128(N) * 4, line_words = step128(M) * 4;
            const size_t oplane = swapped ? static_cast<size_t>(rows_pad) * row_words : static_cast<size_t>(pad128(N)) * line_words;
            const int pitch = swapped ? row_words : line_words;              // output words per line
            const int line_lim = swapped ? M : N, elem_lim = swapped ? N : M; // lines / elements inside the matrix
            const int line_store = swapped ? rows_pad : pad128(N);           // lines that exist in the output
            uint32_t *out = static_cast<uint32_t *>(pr.out);
            const uint32_t lane_sh = 4u - 4u * static_cast<uint32_t>(fh);
            auto finish = [&](auto ob_c) {
                constexpr int OB = decltype(ob_c)::value;   // 0: any ob (runtime loop, per-plane bit extraction)
                const int maxi = 1 << (sh.ob & 31);
                const uint32_t ones = static_cast<uint32_t>(maxi - 1);
                const bool int_rq = sh.ob <= 23;  // float(c) > 2^ob  <=>  c > 2^ob for every int c >= 0
#pragma unroll
                for (int i = 0; i < 2; i++)
#pragma unroll
                    for (int j = 0; j < 2; j++) {
                        // tile coordinates along the lane dimension (lines) and the register dimension (elements)
                        const int line0 = swapped ? m0 + 64 * wr + 32 * i : n0 + 64 * wc + 32 * j;
                        const int elem0 = swapped ? n0 + 64 * wc + 32 * j : m0 + 64 * wr + 32 * i;
                        const int line = line0 + fl;
                        const int nvalid = min(max(elem_lim - elem0, 0), 32);       // leading elements inside the matrix
                        uint32_t emask = nvalid == 32 ? 0xffffffffu : ~(0xffffffffu >> nvalid);  // element e at bit 31 - e
                        if (line >= line_lim) emask = 0u;
                        uint32_t q[16];
#pragma unroll
                        for (int r = 0; r < 16; r++) {
                            const int c = acc[i][j][r];
                            if (OB > 0) {  // the low OB bits of requant(c) = c < 0 ? 1 : (c > 2^ob ? all ones : c)  (kernel.h:31-37,350)
                                uint32_t t = c > maxi ? ones : static_cast<uint32_t>(c);
                                if (!sh.nowrap) t = c < 0 ? 1u : t;
                                q[r] = OB < 8 ? t : (t & 255u);
                            } else {
                                q[r] = static_cast<uint32_t>(int_rq ? (c < 0 ? 1 : (c > maxi ? maxi - 1 : c)) : requant(c, sh.maxv, sh.maxm1));
                            }
                        }
                        uint32_t *dst = out + static_cast<size_t>(line) * pitch + (elem0 >> 5);
                        const bool store = fh == 0 && line < line_store;
                        if constexpr (OB > 0) {
                            uint32_t P[4];  // values of t = r & 3, one per byte: g = r >> 2 in byte 3 - g
#pragma unroll
                            for (int t = 0; t < 4; t++) P[t] = (q[t] << 24) | (q[4 + t] << 16) | (q[8 + t] << 8) | q[12 + t];
#pragma unroll
                            for (int p = 0; p < OB; p++, dst += oplane) {
                                uint32_t w = ((P[0] >> p) & 0x01010101u) << 3 | ((P[1] >> p) & 0x01010101u) << 2 |
                                             ((P[2] >> p) & 0x01010101u) << 1 | ((P[3] >> p) & 0x01010101u);
                                w <<= lane_sh;                       // bits 7 - t - 4 fh of every byte
                                w = or_with_partner_half(w);
                                if (store) st_word<PUB>(dst, w & emask);
                            }
                        } else {
                            for (int p = 0; p < sh.ob; p++, dst += oplane) {
                                uint32_t w = 0u;
#pragma unroll
                                for (int r = 0; r < 16; r++) w |= ((q[r] >> p) & 1u) << (31 - ((r & 3) + 8 * (r >> 2)));
                                w >>= 4 * fh;
                                w = or_with_partner_half(w);
                                if (store) st_word<PUB>(dst, w & emask);
                            }
                        }
                    }
            };
            if (sh.ob == 1) finish(std::integral_constant<int, 1>{});
            else if (sh.ob == 2) finish(std::integral_constant<int, 2>{});
            else if (sh.ob == 4) finish(std::integral_constant<int, 4>{});
            else if (sh.ob == 8) finish(std::integral_constant<int, 8>{});
            else finish(std::integral_constant<int, 0>{});
        }
    }
#ifdef QGTC_STAMPS
    if (!expander) MF_STAMP(7);
    if ((tid == 0 || tid == 256) && blockIdx.x < 512)
        for (int i = 0; i < 8; i++) g_stamps[blockIdx.x * 16 + (tid ? 8 : 0) + i] = st_[i];
#endif
#undef MF_STAMP
}

template <int MAXP, int EXPW, bool FP4 = false>
__global__ __launch_bounds__(64 * (4 + EXPW)) void k_bitmm_mfma(qgtc_problem pr, MMShape sh, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    mf_tile<MAXP, EXPW, FP4>(pr, sh, blockIdx.x / tiles_n, blockIdx.x % tiles_n, smem);
}

// grouped launch: blockIdx.y = problem, blockIdx.x = 128 x 128 tile (surplus tiles exit at once)
template <int MAXP, int EXPW, bool FP4 = false>
__global__ __launch_bounds__(64 * (4 + EXPW)) void k_bitmm_mfma_batched(const qgtc_problem *__restrict__ prs, MMShape sh) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const qgtc_problem pr = prs[blockIdx.y];
    const int tiles_m = (pr.M + MF_T - 1) / MF_T, tiles_n = (pr.N + MF_T - 1) / MF_T;
    const int tile = blockIdx.x;
    if (tile >= tiles_m * tiles_n) return;
    mf_tile<MAXP, EXPW, FP4>(pr, sh, tile / tiles_n, tile % tiles_n, smem);
}

}  // namespace
