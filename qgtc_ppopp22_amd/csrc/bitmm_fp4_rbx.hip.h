// bitmm_fp4_rbx.hip.h — part of libqgtc_hip.so (qgtc_chainx.hip).
// The chain entries beyond the widths of bitmm_fp4_rbw.hip.h: one width of 5 .. 8 bits per chain (N <= 128), and 129 .. 256 columns at
// 1 .. 4 bits (main_qgtc.py:31,37: --n-hidden and --bit_width are free; 2_7c_QGTC_GEMM_INT8.py:15 sweeps 1 .. 8 bits). Same entry
// points (qgtc_chain_transform / qgtc_chain_aggregate), same descriptors, same results word for word after decoding.
#pragma once

namespace {

// ------------------------------------------------------------------------------------------
// One WAVE per 32-row block for the whole output width, as in bitmm_fp4_rbw.hip.h; what differs:
//   * VALUES OF 5 .. 8 BITS are four base-4 digits. T between the launches is TWO arrays of the 4-bit chain format, one behind the
//     other (qgtc_chain_words(M, N, bits) = twice the words): array hv holds bits 4 hv .. 4 hv + 3 of every value, two base-4 digits a
//     nibble, so digit dg = 2 hv + dd is (array hv >> 2 dd) & 0x33333333 and counts 4^dg through the E8M0 scale (128 + 2 dg). A 1-bit
//     adjacency times an 8-bit T is four FP4 MFMAs per 64 elements of K; the second product (8-bit aggregate x 8-bit W') sixteen.
//     ND = digits of a value: 1 (1 / 2 bits), 2 (3 / 4), 4 (5 .. 8; the top digits of 5- and 6-bit values are zero).
//   * UP TO EIGHT COLUMN BLOCKS (256 columns): the second product's left operand is still the aggregate's row as it sits in the
//     registers - MFMA m of its K takes the lane's values of column blocks 2 m, 2 m + 1 (rbw_column, m < 4) - and W' comes
//     pre-expanded in that order with four 64-column slices per column block (k_expand_weights, MS = 4).
//   * ONE code path per k-quad: every occupied k-quad of the adjacency row block is shared between the wave's halves (half fh takes
//     words 2 fh, 2 fh + 1 of both operands - the lone-k-quad form of k_rbw_chain); no diagonal staging, no pairs. These widths are off
//     the BASELINE epochs: the simple form, exact, bounded by the same launch gap + one dependent chain per wave.
//   * the re-quantisation takes its width at RUN time (ob): one kernel per (digits, column blocks), not per width. (Skipping the zero
//     top digits of 5- and 6-bit values with launch-uniform branches around the MFMAs was measured: 0.060 -> 0.058 ms per epoch at
//     5 / 6 bits, 0.060 -> 0.068 at 7 / 8 - the branches cost the full-width case its MFMA interleaving. Not kept.)
// ------------------------------------------------------------------------------------------

// 16 sums -> bytes: P[t] byte 3 - gq = requant(register 4 gq + t) & (2^ob - 1) (kernel.h:31-37,350: c > 2^ob ? 2^ob - 1 : c; c == 2^ob
// keeps its bits, of which only the low ob are packed - so it packs as 0; ob = 8: the saturating conversion would make 256 a 255,
// hence the explicit 256 -> 0)
__device__ __forceinline__ void rbx_requant(const f32x16 &acc, int ob, uint32_t (&P)[4]) {
    const float lim = static_cast<float>(1 << ob), onesf = lim - 1.0f, wrap = ob == 8 ? 256.0f : -1.0f;
    const uint32_t mask = ((1u << ob) - 1u) * 0x01010101u;
#pragma unroll
    for (int t = 0; t < 4; t++) {
        uint32_t pk = 0u;
#pragma unroll
        for (int gq = 0; gq < 4; gq++) {
            float f = acc[4 * gq + t];
            f = f > lim ? onesf : f;
            f = f == wrap ? 0.0f : f;
            pk = __builtin_amdgcn_cvt_pk_u8_f32(f, 3 - gq, pk);
        }
        P[t] = pk & mask;
    }
}

// bits 4 hv .. 4 hv + 3 of every byte, a byte each
__device__ __forceinline__ void rbx_half(const uint32_t (&P)[4], int hv, uint32_t (&H)[4]) {
#pragma unroll
    for (int t = 0; t < 4; t++) H[t] = (P[t] >> (4 * hv)) & 0x0f0f0f0fu;
}

// the codes of word (rb & 3) of line n of one array of T' from the 16 nibble values a lane holds of column n (not swapped: rows
// t + 8 gq + 4 fh): nibble 7 - 2 gq - fh of dword 3 - t (rbw_store_codes' placement)
__device__ __forceinline__ void rbx_store_codes(const uint32_t (&H)[4], uint32_t *__restrict__ dst, int fh, bool ok) {
    uint32_t x[4];
#pragma unroll
    for (int t = 0; t < 4; t++) x[t] = or_with_partner_half(H[t] << (4u - 4u * static_cast<uint32_t>(fh)));
    if (fh == 0 && ok) __builtin_nontemporal_store(u32x4{x[3], x[2], x[1], x[0]}, reinterpret_cast<u32x4 *>(dst));
}

// digit dd of the nibbles of an array (ND = 1: the nibble IS the digit)
template <int ND>
__device__ __forceinline__ i32x8 rbx_digit(const u32x4 &v, int dd) {
    if constexpr (ND == 1) return fp4_op(v);
    else return fp4_op((v[0] >> (2 * dd)) & 0x33333333u, (v[1] >> (2 * dd)) & 0x33333333u, (v[2] >> (2 * dd)) & 0x33333333u, (v[3] >> (2 * dd)) & 0x33333333u);
}

// ------------------------------------------------------------------------------------------
// T = requant(X . W): X packed rows-layout planes (sh_a of them, NA = capacity), W pre-expanded (order 0, NDW digits, a table per
// k-quad of K), T in the chain format (two arrays when ob > 4). A workgroup = four row blocks (one k-quad of T), a wave = one.
// ------------------------------------------------------------------------------------------
template <int NA, int NDW, int NCB>
__global__ __launch_bounds__(256) void k_rbx_xw(const qgtc_problem *__restrict__ prs, const u32x4 *__restrict__ w_codes, int per, int a_planes, int gx, int gy,
                                                int kq_tables, int ob, int table_blocks) {
    constexpr int NDA = (NA + 1) / 2;
    int grp, batch;
    rbw_ids(per, gx, gy, grp, batch);
    const qgtc_problem pr = prs[batch];
    rbw_pin(pr);
    if (grp >= step128(pr.M)) return;
    const int M = pr.M, N = pr.N;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fl = lane & 31, fh = lane >> 5;
    const int rb = 4 * grp + wv, m = 32 * rb + fl;
    const int lines = pad128(N);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint32_t *>(pr.X), 0, static_cast<int>(static_cast<uint32_t>(pr.x_words) * 4u), 0x00020000);
    const int kq_x = step128(pr.K), kq = kq_tables < kq_x ? kq_tables : kq_x;   // (the loop bound is the HOST's: w_codes is a raw pointer)
    const uint32_t row_bytes = static_cast<uint32_t>(kq_x) * 16u, x_plane = static_cast<uint32_t>(pad8(M)) * row_bytes;
    const size_t half_words = static_cast<size_t>(step128(M)) * lines * 16u;
    uint32_t *tbase = static_cast<uint32_t *>(pr.out) + static_cast<size_t>(grp * 4 + wv) * lines * 4;   // word wv of k-quad grp
    // PASSES of at most four column blocks: eight blocks' accumulators (128 registers) leave one wave per SIMD, and a launch of 750
    // four-wave workgroups then runs in three rounds on 256 CUs. A pass re-reads the lane's words of X (8 bytes a plane) and expands them again.
    // The k-quad's weight table goes through LDS once per workgroup: at 8 bits it is 32 KB, and every wave fetching it for itself was
    // 96 MB of L2 reads a launch (ogbn-arxiv-sized epoch: X . W1 at 8 bits 15.4 us).
    constexpr int JP = NCB > 4 ? 4 : NCB;
    constexpr int WN = JP * 2 * NDW * 64, NI = (WN + 255) / 256;   // a pass's part of the table: JP column blocks
    __shared__ __attribute__((aligned(16))) u32x4 w_lds[WN];
#pragma unroll
    for (int jp = 0; jp < NCB; jp += JP) {
        f32x16 accs[JP];
#pragma unroll
        for (int jn = 0; jn < JP; jn++) accs[jn] = f32x16_zero();
        for (int q = 0; q < kq; q++) {   // (workgroup-uniform)
            {
                const u32x4 *wq = w_codes + (static_cast<size_t>(q) * table_blocks + jp) * 2 * NDW * 64;
                u32x4 wreg[NI];
#pragma unroll
                for (int i = 0; i < NI; i++) wreg[i] = (i * 256 + tid < WN) ? wq[i * 256 + tid] : u32x4{0u, 0u, 0u, 0u};
                if (q > 0 || jp > 0) __syncthreads();      // (the previous part of the table is still being read)
#pragma unroll
                for (int i = 0; i < NI; i++)
                    if (i * 256 + tid < WN) w_lds[i * 256 + tid] = wreg[i];
            }
            uint32_t xl[2][NA];   // [k half][plane]: words 2 fh, 2 fh + 1 of k-quad q of the lane's row
#pragma unroll
            for (int p = 0; p < NA; p++) {
                const u32x2 v = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rx, (m < M && p < a_planes) ? static_cast<uint32_t>(p) * x_plane + static_cast<uint32_t>(m) * row_bytes + static_cast<uint32_t>(q) * 16u + 8u * fh : 0xffffffffu, 0, 0));
                xl[0][p] = v.x;
                xl[1][p] = v.y;
            }
            __syncthreads();
#pragma unroll
            for (int h = 0; h < 2; h++) {
                i32x8 xa[NDA];
#pragma unroll
                for (int da = 0; da < NDA; da++) xa[da] = fp4_op(strip_operand<NA>(xl[h], da));
#pragma unroll
                for (int jn = 0; jn < JP; jn++)
#pragma unroll
                    for (int dw = 0; dw < NDW; dw++) {
                        const u32x4 w = w_lds[((jn * 2 + h) * NDW + dw) * 64 + lane];
#pragma unroll
                        for (int da = 0; da < NDA; da++)
                            accs[jn] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(xa[da], fp4_op(w), accs[jn], 4, 4, 0, 128 + 2 * da, 0, 128 + 2 * dw);   // not swapped: lane = column 32 (jp + jn) + fl
                    }
            }
        }
#pragma unroll
        for (int jn = 0; jn < JP; jn++) {
            const int n = 32 * (jp + jn) + fl;
            uint32_t P[4], H[4];
            rbx_requant(accs[jn], ob, P);
            rbx_half(P, 0, H);
            rbx_store_codes(H, tbase + static_cast<size_t>(n) * 4, fh, n < lines);
            if (ob > 4) {   // (launch-uniform)
                rbx_half(P, 1, H);
                rbx_store_codes(H, tbase + half_words + static_cast<size_t>(n) * 4, fh, n < lines);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// An aggregation stage with the next layer's X . W in its tail, or the last aggregation (the MODE2 forms of k_rbw_chain):
//   MODE2 0: out = float32(A . T) [M, N]              MODE2 1: T' = requant(requant(A . T) . W')   (chain format)
//   MODE2 2: out = float32(requant(A . T) . W') [M, N2]
// ND: digits of T's values (and of the aggregate's and of W''s: a chain has one width). NCB1 / NCB2: column blocks of T / of the output.
// ------------------------------------------------------------------------------------------
// (which instantiations get 256 registers a wave instead of 168: the ones hipcc spills at three waves per SIMD - resource report of
// -Rpass-analysis=kernel-resource-usage; the eight-by-eight ones hold 64 KB of W' in LDS and run two workgroups a CU anyway)
constexpr bool rbx_two_waves(int nd, int mode2, int ncb1, int ncb2) {
    return (nd >= 2 && ncb1 * nd >= 16) || ncb1 * ncb2 == 64 || (nd == 1 && mode2 == 2 && ncb1 == 8 && ncb2 == 1);
}
template <int ND, int MODE2, int NCB1, int NCB2>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(rbx_two_waves(ND, MODE2, NCB1, NCB2) ? 2 : 3, 4))) void k_rbx_chain(const qgtc_problem *__restrict__ prs, const qgtc_problem *__restrict__ prs2, const u32x4 *__restrict__ w2_codes,
                                                   int per, int tiles, int gx, int gy, int ob) {
    constexpr int NH = ND > 2 ? 2 : 1;            // arrays of T (nibble planes of a value)
    constexpr int DPN = ND > 1 ? 2 : 1;           // digits in a nibble
    constexpr int MH = (NCB1 + 1) / 2;            // MFMAs (64 elements of K each) of the second product
    constexpr int MS = NCB1 > 4 ? 4 : 2;          // 64-column slices per column block in W''s tables (k_expand_weights)
    constexpr int W2N = MODE2 == 0 ? 1 : NCB2 * MS * ND * 64;
    __shared__ __attribute__((aligned(16))) u32x4 w2_lds[W2N];
    int grp, batch;
    rbw_ids(per, gx, gy, grp, batch);
    const qgtc_problem pr = prs[batch];
    const qgtc_problem pr2 = MODE2 == 0 ? pr : prs2[batch];
    rbw_pin(pr);
    if constexpr (MODE2 != 0) asm volatile("" ::"s"(pr2.out), "s"(pr2.N));
    if (grp >= step128(pr.M)) return;
    const int M = pr.M, K = pr.K, N = pr.N;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fl = lane & 31, fh = lane >> 5;
    const int rb = 4 * grp + wv, m = 32 * rb + fl;
    const int kq = step128(K), lines = pad128(N);
    // W' once per workgroup through LDS (every load issued before the first LDS write)
    if constexpr (MODE2 != 0) {
        constexpr int NI = (W2N + 255) / 256;
        u32x4 wreg[NI];
#pragma unroll
        for (int i = 0; i < NI; i++) {
            const int e = i * 256 + tid;
            wreg[i] = u32x4{0u, 0u, 0u, 0u};
            if (e < W2N && ((e / (64 * ND)) % MS) < MH) wreg[i] = w2_codes[e];
        }
#pragma unroll
        for (int i = 0; i < NI; i++) {
            const int e = i * 256 + tid;
            if (e < W2N) w2_lds[e] = wreg[i];
        }
    }
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint32_t *>(pr.X), 0, static_cast<int>(static_cast<uint32_t>(pr.x_words) * 4u), 0x00020000);
    // T: NH arrays of kq * lines * 64 bytes, never more than the descriptor says the buffer holds (reads past either: zeros)
    const uint32_t half_bytes = static_cast<uint32_t>(kq) * static_cast<uint32_t>(lines) * 64u, t_want = half_bytes * NH, t_have = static_cast<uint32_t>(pr.w_words) * 4u;
    const __amdgpu_buffer_rsrc_t rt = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(pr.W), 0, static_cast<int>(t_want < t_have ? t_want : t_have), 0x00020000);
    const bool a_tiles = tiles != 0;
    const uint32_t row_bytes = static_cast<uint32_t>(kq) * 16u, xq_bytes = a_tiles ? 512u : 16u;
    const uint32_t x_base = m >= M ? 0xffffffffu : (a_tiles ? (static_cast<uint32_t>(rb) * static_cast<uint32_t>(kq) * 32u + static_cast<uint32_t>(fl)) * 16u
                                                              : static_cast<uint32_t>(m) * row_bytes);
    unsigned long long todo = kq >= 64 ? ~0ull : ((1ull << kq) - 1ull);
    if (pr.occ && 32 * rb < M) todo &= pr.occ[static_cast<size_t>(rb) * pr.occ_words];
    if constexpr (MODE2 != 0) __syncthreads();
    if (MODE2 != 1 && 32 * rb >= M) return;   // (float32 rows: no rows here; T' still needs its padding words)

    // ---- first product: acc[j] = (A . T)[row fl][columns 32 j + t + 8 gq + 4 fh], swapped operands - in PASSES of at most four column
    // blocks of T (k_rbx_xw has the reason); a pass walks the occupied k-quads again (8 bytes of A a lane and k-quad, its own lines of T)
    bool any = false;   // (wave-uniform)
    unsigned long long occupied = 0ull;
    if (32 * rb < M) {
        const unsigned lo = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(todo)), hi = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(todo >> 32));
        occupied = (static_cast<unsigned long long>(hi) << 32) | lo;
        any = occupied != 0ull;
    }
    if (!any) {
        // No occupied k-quad, or a padding block of T': zeros out
        if constexpr (MODE2 == 0) {
            if (m < M)
                for (int c = fh; c < N; c += 2) static_cast<float *>(pr.out)[static_cast<size_t>(m) * N + c] = 0.0f;
        } else if constexpr (MODE2 == 2) {
            if (m < M)
                for (int c = fh; c < pr2.N; c += 2) static_cast<float *>(pr2.out)[static_cast<size_t>(m) * pr2.N + c] = 0.0f;
        } else {
            const int lines2 = pad128(pr2.N);
            const size_t half2 = static_cast<size_t>(step128(M)) * lines2 * 16u;
            uint32_t *tz = static_cast<uint32_t *>(pr2.out) + static_cast<size_t>(grp * 4 + wv) * lines2 * 4;
#pragma unroll
            for (int jn = 0; jn < NCB2; jn++)
                if (fh == 0) {
                    *reinterpret_cast<u32x4 *>(tz + (32 * jn + fl) * 4) = u32x4{0u, 0u, 0u, 0u};
                    if (ob > 4) *reinterpret_cast<u32x4 *>(tz + half2 + (32 * jn + fl) * 4) = u32x4{0u, 0u, 0u, 0u};
                }
        }
        return;
    }
    // the aggregate's row as the second product's left operand, straight from the registers: XA[hv][mm] = the nibble codes of bits
    // 4 hv .. 4 hv + 3 of the lane's 32 values of column blocks 2 mm, 2 mm + 1 (rbw_column's order)
    uint32_t XA[NH][MH][4];
    if constexpr (MODE2 != 0) {
#pragma unroll
        for (int hv = 0; hv < NH; hv++)
#pragma unroll
            for (int mm = 0; mm < MH; mm++)
#pragma unroll
                for (int d = 0; d < 4; d++) XA[hv][mm][d] = 0u;
    }
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(pr.out, 0, MODE2 == 0 ? static_cast<int>(static_cast<uint32_t>(M) * static_cast<uint32_t>(N) * 4u) : 0, 0x00020000);
    constexpr int JP = NCB1 > 4 ? 4 : NCB1;
#pragma unroll
    for (int jp = 0; jp < NCB1; jp += JP) {
        f32x16 acc[JP];
#pragma unroll
        for (int j = 0; j < JP; j++) acc[j] = f32x16_zero();
        unsigned long long left = occupied;
        while (left != 0ull) {
            const int q = __builtin_ctzll(left);
            left &= left - 1ull;
            const u32x2 xs = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rx, x_base != 0xffffffffu ? x_base + static_cast<uint32_t>(q) * xq_bytes + 8u * fh : 0xffffffffu, 0, 0));
            const uint32_t t_lane = (static_cast<uint32_t>(q) * 4u * static_cast<uint32_t>(lines) + static_cast<uint32_t>(2 * fh * lines + fl)) * 16u;   // word 2 fh of k-quad q, line fl
            u32x4 tl[JP][2][NH];
#pragma unroll
            for (int j = 0; j < JP; j++)
#pragma unroll
                for (int h = 0; h < 2; h++)
#pragma unroll
                    for (int hv = 0; hv < NH; hv++)
                        tl[j][h][hv] = __builtin_amdgcn_raw_buffer_load_b128(rt, t_lane + static_cast<uint32_t>(hv) * half_bytes + static_cast<uint32_t>(h * lines + 32 * (jp + j)) * 16u, 0, 0);
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const uint32_t xw[1] = {xs[h]};
                const i32x8 xa = fp4_op(strip_operand<1>(xw, 0));
#pragma unroll
                for (int j = 0; j < JP; j++)
#pragma unroll
                    for (int hv = 0; hv < NH; hv++)
#pragma unroll
                        for (int dd = 0; dd < DPN; dd++)
                            acc[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(rbx_digit<ND>(tl[j][h][hv], dd), xa, acc[j], 4, 4, 0, 128 + 2 * (2 * hv + dd), 0, 128);
            }
        }
        if constexpr (MODE2 == 0) {   // float32 [M, N] (kernel.h:915-930)
            const uint32_t row_off = m < M ? static_cast<uint32_t>(m) * static_cast<uint32_t>(N) * 4u : 0xffffffffu;
#pragma unroll
            for (int j = 0; j < JP; j++) rbw_store_f32_row(ro, row_off, acc[j], 32 * (jp + j), fh, N);
        } else {
#pragma unroll
            for (int j = 0; j < JP; j++) {
                uint32_t P[4];
                rbx_requant(acc[j], ob, P);
#pragma unroll
                for (int hv = 0; hv < NH; hv++) {
                    uint32_t H[4];
                    rbx_half(P, hv, H);
                    XA[hv][(jp + j) >> 1][2 * ((jp + j) & 1)] = H[0] | (H[1] << 4);
                    XA[hv][(jp + j) >> 1][2 * ((jp + j) & 1) + 1] = H[2] | (H[3] << 4);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);   // (a pass ends here: interleaved with the next one, the two passes' loads and accumulators were live together)
    }
    if constexpr (MODE2 != 0) {
        const int N2 = pr2.N, lines2 = pad128(N2);
        const size_t half2 = static_cast<size_t>(step128(M)) * lines2 * 16u;
        uint32_t *tbase = static_cast<uint32_t *>(pr2.out) + static_cast<size_t>(grp * 4 + wv) * lines2 * 4;   // word wv of k-quad grp
        const __amdgpu_buffer_rsrc_t ro2 = __builtin_amdgcn_make_buffer_rsrc(pr2.out, 0, MODE2 == 2 ? static_cast<int>(static_cast<uint32_t>(M) * static_cast<uint32_t>(N2) * 4u) : 0, 0x00020000);
        constexpr int G2 = NCB2 > 4 ? 4 : NCB2;   // column blocks of the output whose MFMAs run before their epilogues
#pragma unroll
        for (int g0 = 0; g0 < NCB2; g0 += G2) {
            f32x16 acc2s[G2];
#pragma unroll
            for (int g = 0; g < G2; g++) {
                const int jn = g0 + g;
                f32x16 acc2 = f32x16_zero();
#pragma unroll
                for (int mm = 0; mm < MH; mm++)
#pragma unroll
                    for (int da = 0; da < ND; da++) {
                        const u32x4 xv = {XA[da / DPN][mm][0], XA[da / DPN][mm][1], XA[da / DPN][mm][2], XA[da / DPN][mm][3]};
                        const i32x8 xa = rbx_digit<ND>(xv, da % DPN);
#pragma unroll
                        for (int dw = 0; dw < ND; dw++) {
                            const i32x8 wb = fp4_op(w2_lds[((jn * MS + mm) * ND + dw) * 64 + lane]);
                            // T' (MODE2 1): not swapped - lane = column 32 jn + fl of T'; float32 rows (MODE2 2): swapped - lane = row fl
                            if constexpr (MODE2 == 2) acc2 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wb, xa, acc2, 4, 4, 0, 128 + 2 * dw, 0, 128 + 2 * da);
                            else acc2 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(xa, wb, acc2, 4, 4, 0, 128 + 2 * da, 0, 128 + 2 * dw);
                        }
                    }
                acc2s[g] = acc2;
            }
#pragma unroll
            for (int g = 0; g < G2; g++) {
                const int jn = g0 + g, n2 = 32 * jn + fl;
                if constexpr (MODE2 == 2) {
                    rbw_store_f32_row(ro2, m < M ? static_cast<uint32_t>(m) * static_cast<uint32_t>(N2) * 4u : 0xffffffffu, acc2s[g], 32 * jn, fh, N2);
                } else {
                    uint32_t P[4], H[4];
                    rbx_requant(acc2s[g], ob, P);
                    rbx_half(P, 0, H);
                    rbx_store_codes(H, tbase + static_cast<size_t>(n2) * 4, fh, n2 < lines2);
                    if (ob > 4) {
                        rbx_half(P, 1, H);
                        rbx_store_codes(H, tbase + half2 + static_cast<size_t>(n2) * 4, fh, n2 < lines2);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);   // (a group of output column blocks ends here)
        }
    }
}

}  // namespace
