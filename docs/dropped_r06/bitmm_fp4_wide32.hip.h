// bitmm_fp4_wide32.hip.h — part of libqgtc_hip.so (qgtc_wide.hip).
// The wide-operand FP4 kernel (bitmm_fp4_wide.hip.h) for ONE-plane operands on v_mfma_scale_f32_32x32x64_f8f6f4: the wide columns of the
// reference's adjacency-size study (5_9_adjmatrix_size.py:15-18: N = 512 / 1024 at 1 bit; QGTC_module/logs/profile_new.log:26).
#pragma once

namespace {

// ------------------------------------------------------------------------------------------
// Same staging as k_bitmm_fp4_wide<1, 1, .., 4, 4, 128>: a workgroup (8 waves as 2 x 4) owns 128 left-hand lines x 256 right-hand lines, K
// is walked in groups of 128 bytes of every line fetched once per workgroup by LDS-DMA in pieces of 8 lines x 128 bytes (chunks XOR-swizzled
// on the source address), three stages. What differs is the instruction: an MFMA holds its SIMD's vector issue for 8 cycles whatever its
// shape (MI355X guide, 'vector-instruction ISSUE cost'), and the in-place expansion costs 5 VALU operations per 32 x 32 x 64 MFMA (2.5 per
// 16 x 16 x 128): 8 + 2.5 x 4 = 18 issue cycles against 16 of the pipe for the small shape, 8 + 5 x 4 = 28 against 32 for the large one -
// tools/mfma_overlap.hip: 37 cycles per MFMA and SIMD for the large shape's step where the small one's ran 2 x 27. The wave's tile stays
// 64 x 64 (2 x 2 fragments of 32 lines): lane (fl, hf) reads chunk 2 u + hf (u = 0..3) of line fl of a fragment - eight consecutive lanes
// read eight lines of one piece: every bank once - and MFMA s = 0..3 of a step takes the bits s, s + 4, .. of the lane's four words in
// place. The fragment reads of a step are issued under the MFMAs of the step before. Epilogue: the waves' tiles go through LDS once (the stages' place) as int32, then a thread re-quantises four
// consecutive right-hand elements of a line and eight lanes OR their nibbles into a word.
// MODE 0: packed bits [ob][out_lines][STEP128(Rc) * 4] (rows layout, or - operands exchanged by the host - the cols layout); 2: float32.
// ------------------------------------------------------------------------------------------
constexpr int W32_TL = 128, W32_TR = 256, W32_GB = 128;
constexpr int W32_LP = W32_TL / 8, W32_RP = W32_TR / 8, W32_TOT = W32_LP + W32_RP;   // pieces per stage: 16 + 32
constexpr int W32_STAGE = W32_TOT * WD_PIECE, W32_STAGES = 3, W32_DMAS = W32_TOT / WD_WAVES;
constexpr int W32_PITCH = W32_TR + 4;                                                 // ints between the rows of the epilogue's tile
constexpr int w32_lds_bytes() { return W32_STAGES * W32_STAGE > W32_TL * W32_PITCH * 4 ? W32_STAGES * W32_STAGE : W32_TL * W32_PITCH * 4; }

template <int MODE>
__global__ __launch_bounds__(64 * WD_WAVES) void k_bitmm_fp4_wide32(
    const uint32_t *__restrict__ Lp, const uint32_t *__restrict__ Rp, void *__restrict__ outp, uint32_t l_bytes, uint32_t r_bytes, uint32_t out_bytes,
    int Lc, int Rc, int K, int l_lines, int r_lines, int out_lines, uint32_t cfg /* ob | tiles along R << 8; host: ob <= 23, every byte count < 2^32 */, int n_wg) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char w32_lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fl = lane & 31, hf = lane >> 5;
    const int wr = wv >> 2, wc = wv & 3;
    const int ob = cfg & 255u, nt_r = static_cast<int>(cfg >> 8);
    // workgroups that run on one XCD (ids congruent mod 8) take consecutive tiles: they share their left-hand lines in L2
    const int tile = xcd_consecutive(static_cast<int>(blockIdx.x), n_wg);
    const int tl = tile / nt_r, tr = tile - tl * nt_r;
    const int kq = step128(K);
    const uint32_t row_bytes = static_cast<uint32_t>(kq) * 16u;
    const uint32_t lds0 = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(w32_lds));
    (void)l_lines; (void)r_lines; (void)out_bytes;

    const i32x4 rs_l = {static_cast<int>(reinterpret_cast<uintptr_t>(Lp)), static_cast<int>((reinterpret_cast<uintptr_t>(Lp) >> 32) & 0xffffu), static_cast<int>(l_bytes), 0x00020000};
    const i32x4 rs_r = {static_cast<int>(reinterpret_cast<uintptr_t>(Rp)), static_cast<int>((reinterpret_cast<uintptr_t>(Rp) >> 32) & 0xffffu), static_cast<int>(r_bytes), 0x00020000};
    const int rr = lane >> 3, cc = lane & 7;
    const uint32_t swz = static_cast<uint32_t>(cc ^ (rr & 6)) * 16u;
    const uint32_t voff_l = static_cast<uint32_t>(tl * W32_TL + rr) * row_bytes + swz, voff_r = static_cast<uint32_t>(tr * W32_TR + rr) * row_bytes + swz;
    const int ng = (kq + 7) / 8;
    auto issue = [&](int g) {   // group g -> stage g % 3: this wave's pieces wv, wv + 8, .. of [left line groups | right line groups]
        if (g >= ng) return;
        const uint32_t base = lds0 + static_cast<uint32_t>(g % W32_STAGES) * W32_STAGE;
        const uint32_t ko = static_cast<uint32_t>(g) * W32_GB;
#pragma unroll
        for (int j = 0; j < W32_DMAS; j++) {
            const int p = wv + WD_WAVES * j;
            if (WD_WAVES * j < W32_LP) wd_dma(base + static_cast<uint32_t>(p) * WD_PIECE, voff_l, rs_l, ko + static_cast<uint32_t>(8 * p) * row_bytes);
            else wd_dma(base + static_cast<uint32_t>(p) * WD_PIECE, voff_r, rs_r, ko + static_cast<uint32_t>(8 * (p - W32_LP)) * row_bytes);
        }
    };
    issue(0);
    issue(1);

    // fragment reads: left fragment i = lines 64 wr + 32 i + fl, right fragment j = lines 64 wc + 32 j + fl; step u: chunk 2 u + hf
    const uint32_t frag_off = static_cast<uint32_t>(fl & 7) * 128u + static_cast<uint32_t>(hf ^ (fl & 6)) * 16u;
    const uint32_t la0 = static_cast<uint32_t>(8 * wr + (fl >> 3)) * WD_PIECE + frag_off;              // + 4 i pieces; step u: ^ 32 u
    const uint32_t ra0 = static_cast<uint32_t>(W32_LP + 8 * wc + (fl >> 3)) * WD_PIECE + frag_off;     // + 4 j pieces

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.0f;

    auto publish = [&](int g) {   // group g has landed for every wave that passes the barrier; the stage of group g - 1 (in registers by now) is free
        if (g + 1 < ng) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(W32_DMAS) : "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        issue(g + 2);
    };
    auto fetch = [&](int g, int u, u32x4 (&xr)[2], u32x4 (&wr_)[2]) {   // step u of group g: chunk 2 u + hf of the lane's lines
        const unsigned char *stage = w32_lds + (g % W32_STAGES) * W32_STAGE;
#pragma unroll
        for (int i = 0; i < 2; i++) xr[i] = *reinterpret_cast<const u32x4 *>(stage + ((la0 + static_cast<uint32_t>(4 * i) * WD_PIECE) ^ (32u * u)));
#pragma unroll
        for (int j = 0; j < 2; j++) wr_[j] = *reinterpret_cast<const u32x4 *>(stage + ((ra0 + static_cast<uint32_t>(4 * j) * WD_PIECE) ^ (32u * u)));
        if (g == ng - 1 && 8 * g + 2 * u + hf >= kq) {   // a chunk past K: whatever the DMA found there must not count
#pragma unroll
            for (int i = 0; i < 2; i++) xr[i] = u32x4{0u, 0u, 0u, 0u};
        }
    };
    auto multiply = [&](const u32x4 (&xr)[2], const u32x4 (&wr_)[2]) {
#pragma unroll
        for (int s = 0; s < 4; s++) {
            const uint32_t mask = s < 3 ? 0x11111111u << s : 0x11111111u;
            const int sc = s < 3 ? 128 - s : 128;   // E8M0: code 1 << s counts as 1
            i32x8 b8[2];
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const u32x4 v = s < 3 ? wr_[j] : wr_[j] >> 3;
                const i32x4 b4 = {static_cast<int>(v.x & mask), static_cast<int>(v.y & mask), static_cast<int>(v.z & mask), static_cast<int>(v.w & mask)};
                b8[j] = __builtin_shufflevector(b4, b4, 0, 1, 2, 3, -1, -1, -1, -1);
            }
#pragma unroll
            for (int i = 0; i < 2; i++) {
                const u32x4 v = s < 3 ? xr[i] : xr[i] >> 3;
                const i32x4 a4 = {static_cast<int>(v.x & mask), static_cast<int>(v.y & mask), static_cast<int>(v.z & mask), static_cast<int>(v.w & mask)};
                const i32x8 a8 = __builtin_shufflevector(a4, a4, 0, 1, 2, 3, -1, -1, -1, -1);
#pragma unroll
                for (int j = 0; j < 2; j++)
                    // lane (fl, hf) register r holds C[left line 32 i + 8 (r >> 2) + 4 hf + (r & 3)][right line 32 j + fl]
                    acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8[j], acc[i][j], 4, 4, 0, sc, 0, sc);
            }
        }
    };
    // the fragments of step u + 1 are read under the MFMAs of step u (two register sets in turns); a group's first read follows its barrier
    u32x4 xa[2], wa[2], xb[2], wb[2];
    for (int g = 0; g < ng; g++) {
        publish(g);
        fetch(g, 0, xa, wa);
        fetch(g, 1, xb, wb);
        multiply(xa, wa);
        fetch(g, 2, xa, wa);
        multiply(xb, wb);
        fetch(g, 3, xb, wb);
        multiply(xa, wa);
        multiply(xb, wb);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();   // (every wave has read its last fragments: the stages' place is free)

    // ---- epilogue through an int32 tile in LDS
    int (*tile_)[W32_PITCH] = reinterpret_cast<int (*)[W32_PITCH]>(w32_lds);
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) tile_[64 * wr + 32 * i + 8 * (r >> 2) + 4 * hf + (r & 3)][64 * wc + 32 * j + fl] = static_cast<int>(acc[i][j][r]);
    __syncthreads();
    const int maxi = 1 << ob;   // (host: ob <= 23 - float(c) > 2^ob <=> c > 2^ob for every int 0 <= c < 2^24)
    const int pitch = step128(Rc) * 4;                       // words per output line
    const size_t oplane = static_cast<size_t>(out_lines) * pitch;
    for (int it = tid; it < W32_TL * (W32_TR / 4); it += 64 * WD_WAVES) {   // (line, quad): four consecutive right-hand elements of a line
        const int row = it >> 6, quad = it & 63;
        const i32x4 c4 = *reinterpret_cast<const i32x4 *>(&tile_[row][4 * quad]);
        const int line = tl * W32_TL + row, col = tr * W32_TR + 4 * quad;
        if (MODE == 2) {   // float32 [Lc][Rc] (kernel.h:915-930)
            if (line < Lc) {
                float *dst = static_cast<float *>(outp) + static_cast<size_t>(line) * Rc + col;
                if (col + 3 < Rc && (Rc & 3) == 0) {
                    *reinterpret_cast<f32x4 *>(dst) = f32x4{static_cast<float>(c4.x), static_cast<float>(c4.y), static_cast<float>(c4.z), static_cast<float>(c4.w)};
                } else {
#pragma unroll
                    for (int e = 0; e < 4; e++)
                        if (col + e < Rc) dst[e] = static_cast<float>(c4[e]);
                }
            }
            continue;
        }
        uint32_t q[4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const int c = c4[e];
            q[e] = (line < Lc && col + e < Rc) ? static_cast<uint32_t>(c > maxi ? maxi - 1 : c) : 0u;   // kernel.h:31-37
        }
        const int word = (tr * W32_TR >> 5) + (quad >> 3);
        const bool store = (quad & 7) == 0 && line < out_lines && word < pitch;   // (the padding lines of the bit layouts are written as zeros)
        const uint32_t sh_n = 28u - 4u * static_cast<uint32_t>(quad & 7);
        uint32_t *dst = static_cast<uint32_t *>(outp) + static_cast<size_t>(line) * pitch + word;
        for (int p = 0; p < ob; p++, dst += oplane) {
            const uint32_t nib = (((q[0] >> p) & 1u) << 3) | (((q[1] >> p) & 1u) << 2) | (((q[2] >> p) & 1u) << 1) | ((q[3] >> p) & 1u);
            const uint32_t wrd = or_reduce8(nib << sh_n);
            if (store) dst[0] = wrd;
        }
    }
}

}  // namespace
