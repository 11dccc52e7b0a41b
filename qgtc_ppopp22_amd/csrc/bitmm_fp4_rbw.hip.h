// bitmm_fp4_rbw.hip.h — part of libqgtc_hip.so (qgtc_fp4.hip).
// The 2-bit Cluster-GCN chain of a device-filled epoch plan, ONE WAVE PER 32-ROW BLOCK for the whole output width.
#pragma once

namespace {

// ------------------------------------------------------------------------------------------
// Why. The row-block kernels of bitmm_fp4_rows.hip.h / bitmm_fp4_chain.hip.h give every 32 x 32 output tile its own wave
// (four waves per row block at 128 columns). rocprofv3 (profiles/r02): 54 VALU instructions per MFMA, 319 per wave - these
// launches are bound by VALU issue, and most of that work is REPEATED or CONVERSION work:
//   * every one of a row block's four waves expands the same adjacency words to E2M1 codes (28 operations per pair of
//     k-quads), loads and expands W' (28), and goes through a barrier and LDS to hand its quarter of the aggregate's
//     row to the other three for the second product;
//   * T - written by one launch of the chain, read by the next and by nobody else - travels as bit planes that the
//     producer packs (28 per tile) and the consumer expands again (56 per tile and pair of k-quads).
// Here a wave owns a row block for ALL column blocks (NCB x 32 columns):
//   * the adjacency word is expanded once per row block;
//   * T / T' are CODES between the launches of a chain (two E2M1 nibbles per byte, the finished MFMA operand: no packing in
//     the producer, no expansion in the consumer) - `fmt` 1 of qgtc_stage, twice the bytes of two bit planes;
//   * the weights arrive PRE-EXPANDED (k_expand_weights, once per plan) in exactly the lane order the products use;
//   * the second product's left operand never leaves the registers: after the first product lane (fl, fh) holds, of row
//     fl of the aggregate, the 64 columns c with ((c >> 2) & 1) == fh - which 32 elements of K an MFMA's lane supplies is
//     free as long as both operands agree, so MFMA m takes the lane's values of column blocks 2 m and 2 m + 1 as they sit
//     (two nibbles per byte from v_cvt_pk_u8_f32 + one v_lshl_or), and W' was expanded in that order. No LDS, no barrier,
//     no cross-lane operation between the two products.
// Per row block of the 128-column chained launch: ~420 VALU operations instead of 4 x 319.
//
// Chain format of T ("codes", fmt 1): [k-quad q of the producing product's rows][word j of the k-quad][line n < pad128(N)]
// [4 dwords]: element e (row 32 (4 q + j) + e) at nibble 7 - e / 4 of dword 3 - e % 4, the order expand_word_fp4 gives the
// bits of a packed A word, so the consumer's A expansion and these codes agree on K. Lines are the INNER index: the 16
// bytes the 32 lanes of a half-wave load (or store) for one (q, j) are 512 contiguous bytes - 8 cache lines per load
// instruction; with the words of a line together ([q][n][j], the first form) every instruction touched 32 lines and the
// four instructions of a column block touched the same 32: the waves of a CU then queued 4.4 k cycles at the address unit
// (in-kernel stamps, tools/rbw_bench.hip). Size: step128(M) * pad128(N) * 16 words.
// ------------------------------------------------------------------------------------------


// ------------------------------------------------------------------------------------------
// Pre-expanded weights. order 0: the right operand of X . W where X arrives as packed words (K <= 128: one k-quad) - table
// [column block jn][k half h][lane] of 16 bytes: the codes of word 2 fh + h of line 32 jn + fl (what strip_operand gives).
// order 1: the right operand of (aggregate) . W' in the register order above - table [jn][m][lane]: dword d nibble i = the
// value of W'[rbw_column(m, fh, d, i)][32 jn + fl]. One table per base-4 digit (planes 2 dg, 2 dg + 1) of the weight:
// entry ((jn * MS + s) * ND + dg) * 64 + lane, ND = chain_digits(nbits) (1- and 2-bit weights: one digit; 3 / 4-bit: two; 5 .. 8-bit: four),
// MS = 2 (order 0: the halves of a k-quad; order 1 with K <= 128: the two MFMAs of 64 columns) or 4 (order 1, 128 < K <= 256).
// ------------------------------------------------------------------------------------------
struct ExpandJob {
    const uint32_t *W;
    uint32_t *codes;
    unsigned long long w_words;
    int K, N, w_lines, nbits, order, ncb, ms;
};
struct ExpandJobs {
    ExpandJob job[QGTC_MAX_WEIGHTS];
};

__global__ __launch_bounds__(64) void k_expand_weights(ExpandJobs jobs) {
    const ExpandJob j = jobs.job[blockIdx.y];
    const int jn = static_cast<int>(blockIdx.x) / j.ms, s = static_cast<int>(blockIdx.x) % j.ms;
    const int q = static_cast<int>(blockIdx.z);   // order 0 with K > 128: one table per k-quad of K, [k-quad][column block][k half][digit][lane]
    if (jn >= j.ncb || q >= (j.order == 0 ? step128(j.K) : 1)) return;
    const int lane = threadIdx.x, fl = lane & 31, fh = lane >> 5;
    const int n = 32 * jn + fl;
    const int line_words = step128(j.K) * 4;
    const size_t plane = static_cast<size_t>(j.w_lines) * line_words;
    auto bit = [&](int p, int c) -> uint32_t {   // bit of plane p for K index c of line n
        if (n >= j.N || c >= j.K || p >= j.nbits) return 0u;
        const size_t wi = p * plane + static_cast<size_t>(n) * line_words + (c >> 5);
        return wi < j.w_words ? (j.W[wi] >> (31 - (c & 31))) & 1u : 0u;
    };
    const int nd = chain_digits(j.nbits);
    for (int dg = 0; dg < nd; dg++) {
        uint32_t out[4] = {0u, 0u, 0u, 0u};
        for (int d = 0; d < 4; d++)
            for (int i = 0; i < 8; i++) {
                int c;
                if (j.order == 0) c = 128 * q + 32 * (2 * fh + s) + 31 - (d + 4 * i);   // bit d + 4 i of word 2 fh + s of k-quad q = element 31 - (d + 4 i)
                else c = rbw_column(s, fh, d, i);
                out[d] |= (bit(2 * dg, c) | (bit(2 * dg + 1, c) << 1)) << (4 * i);
            }
        *reinterpret_cast<u32x4 *>(j.codes + (((static_cast<size_t>(q) * j.ncb * j.ms + blockIdx.x) * nd + dg) * 64 + lane) * 4) = u32x4{out[0], out[1], out[2], out[3]};
    }
}

// A cols-layout operand (public format: [plane][line n][word], QGTC_device.cu:97) -> the chain format, for right operands
// the DATA LOADER supplies (Batched-GIN's first product is A . X, main_qgtc.py:131, X packed by sampler.py:99): done once
// beside the packing. One thread per (k-quad, word, line): nibble i of dword d = sum_p bit (d + 4 i) of plane p's word << p.
// More than four planes (5 .. 8 bits, bitmm_fp4_rbx.hip.h): a second array of the same shape behind the first holds planes 4 .. 7.
__device__ __forceinline__ u32x4 chain_nibbles(const uint32_t (&r)[4], int np) {
    uint32_t out[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int p = 0; p < 4; p++) {
        if (p >= np) break;
#pragma unroll
        for (int d = 0; d < 4; d++) {
            const uint32_t sh = d > p ? r[p] >> (d - p) : (d < p ? r[p] << (p - d) : r[p]);   // bits d + 4 i to bits p + 4 i
            out[d] |= sh & (0x11111111u << p);
        }
    }
    return u32x4{out[0], out[1], out[2], out[3]};
}
__global__ __launch_bounds__(256) void k_cols_to_chain(const uint32_t *__restrict__ cols, unsigned long long words, int H, int W, int nbits,
                                                       uint32_t *__restrict__ chain) {
    const int lines = pad128(W), line_words = step128(H) * 4;
    const size_t plane = static_cast<size_t>(lines) * line_words, total = static_cast<size_t>(line_words) * lines;
    for (size_t t = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; t < total; t += static_cast<size_t>(gridDim.x) * blockDim.x) {
        const int n = static_cast<int>(t % lines), wj = static_cast<int>(t / lines);   // wj = 4 q + j
        for (int hv = 0; 4 * hv < nbits && hv < 2; hv++) {
            uint32_t r[4];
#pragma unroll
            for (int p = 0; p < 4; p++) {
                const size_t wi = (4 * hv + p) * plane + static_cast<size_t>(n) * line_words + wj;
                r[p] = (4 * hv + p < nbits && n < W && wi < words) ? cols[wi] : 0u;
            }
            *reinterpret_cast<u32x4 *>(chain + (static_cast<size_t>(hv) * total + static_cast<size_t>(wj) * lines + n) * 4) = chain_nibbles(r, nbits - 4 * hv);
        }
    }
}

// the same for every batch of a grouped pack (qgtc_load_batches): blockIdx.y = batch, cols = the batch's X just written
__global__ __launch_bounds__(256) void k_cols_to_chain_batched(const qgtc_loader_batch *__restrict__ tb, int W, int nbits) {
    const qgtc_loader_batch b = tb[blockIdx.y];
    if (!b.XC || !b.X) return;
    const int H = b.n, lines = pad128(W), line_words = step128(H) * 4;
    const size_t plane = static_cast<size_t>(lines) * line_words, total = static_cast<size_t>(line_words) * lines;
    for (size_t t = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; t < total; t += static_cast<size_t>(gridDim.x) * blockDim.x) {
        const int n = static_cast<int>(t % lines), wj = static_cast<int>(t / lines);
        for (int hv = 0; 4 * hv < nbits && hv < 2; hv++) {
            uint32_t r[4];
#pragma unroll
            for (int p = 0; p < 4; p++) r[p] = (4 * hv + p < nbits && n < W) ? b.X[(4 * hv + p) * plane + static_cast<size_t>(n) * line_words + wj] : 0u;
            *reinterpret_cast<u32x4 *>(b.XC + (static_cast<size_t>(hv) * total + static_cast<size_t>(wj) * lines + n) * 4) = chain_nibbles(r, nbits - 4 * hv);
        }
    }
}

// A rows-layout adjacency (one plane, [PAD8(M)][STEP128(K) * 4 words]: QGTC_device.cu:83) as 512-byte TILES
// [row block of 32][k-quad][32 rows][4 words] - what the aggregation kernels below read when told so (RbwShape::tiles). In the
// rows layout the 32 lanes of a half-wave that load one k-quad touch 32 different rows (160 bytes apart in a 1213-node batch):
// every launch pulled ALL of A out of HBM although a fifth of its tiles is occupied, and each load instruction cost the
// address unit 32 cache lines instead of 4. Done once by the data loader beside the packing. One thread per 16 bytes.
__global__ __launch_bounds__(256) void k_rows_to_tiles(const uint32_t *__restrict__ rows, unsigned long long words, int M, int K, uint32_t *__restrict__ tiles) {
    const int kq = step128(K);
    const size_t total = static_cast<size_t>((M + 31) / 32) * kq * 32;
    for (size_t t = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; t < total; t += static_cast<size_t>(gridDim.x) * blockDim.x) {
        const int r = static_cast<int>(t & 31);
        const size_t tq = t >> 5;
        const int q = static_cast<int>(tq % kq), rb = static_cast<int>(tq / kq);
        const size_t src = (static_cast<size_t>(32 * rb + r) * kq + q) * 4u;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (32 * rb + r < pad8(M) && src + 4u <= words) v = *reinterpret_cast<const u32x4 *>(rows + src);
        *reinterpret_cast<u32x4 *>(tiles + t * 4u) = v;
    }
}

// codes of word (rb & 3) of line n2 of T' from the 16 values a lane holds of column n2 (not swapped: rows t + 8 gq + 4 fh)
template <int OB>
__device__ __forceinline__ void rbw_store_codes(const f32x16 &acc, uint32_t *__restrict__ dst, int fh, bool ok) {
    uint32_t x[4];
#ifdef QGTC_ABL_NOEPI   // timing-only (wrong codes): what do the re-quantisation, the packing and the half-wave exchange of a column block cost?
#pragma unroll
    for (int t = 0; t < 4; t++) x[t] = __float_as_uint(acc[t]);
#else
    uint32_t qv[16], P[4];
    requant_pack16<OB>(acc, OB, P, qv);
#pragma unroll
    for (int t = 0; t < 4; t++) {
        x[t] = (P[t] & (((1u << OB) - 1u) * 0x01010101u)) << (4u - 4u * static_cast<uint32_t>(fh));   // nibble 7 - 2 gq - fh of dword 3 - t
        x[t] = or_with_partner_half(x[t]);
    }
#endif
    // (non-temporal: nobody in THIS launch reads T', and what a launch leaves dirty in the L2s is written back at its end, before
    // the next launch may start - 6 MB of T' per 128-column launch: 6.20 -> 5.90 us. -DQGTC_RBW_PLAIN_STORES: the A/B build)
#ifdef QGTC_RBW_PLAIN_STORES
    if (fh == 0 && ok) *reinterpret_cast<u32x4 *>(dst) = u32x4{x[3], x[2], x[1], x[0]};
#else
    if (fh == 0 && ok) __builtin_nontemporal_store(u32x4{x[3], x[2], x[1], x[0]}, reinterpret_cast<u32x4 *>(dst));
#endif
}


// ------------------------------------------------------------------------------------------
// T = requant(X . W) for every cluster batch (main_qgtc.py:147, layout-correct form): X = packed rows-layout planes (K <=
// 128), W pre-expanded (order 0), T in the chain format. A workgroup = four row blocks (one k-quad of T), a wave = one.
// ------------------------------------------------------------------------------------------
template <int NA, int OB, int NCB>
__device__ __forceinline__ void rbw_xw_body(const qgtc_problem &pr, const u32x4 *__restrict__ w_codes, const RbwShape &sh, int grp, int kq_tables) {
    constexpr int NDA = (NA + 1) / 2;
    constexpr int NDW = OB > 2 ? 2 : 1;   // base-4 digits of W (a chain has ONE width: W has as many planes as T)
    const int M = pr.M, N = pr.N;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fl = lane & 31, fh = lane >> 5;
    const int rb = 4 * grp + wv, m = 32 * rb + fl;
    const int lines = pad128(N);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint32_t *>(pr.X), 0, static_cast<int>(static_cast<uint32_t>(pr.x_words) * 4u), 0x00020000);
    // k-quads: the count the HOST stated (the K the weight tables were expanded for - w_codes is a raw pointer, so the loop bound must
    // not come from a descriptor). X's own row stride is the descriptor's; where the two disagree the buffer range check of rx and the
    // clamp below keep every read inside what exists. One k-quad for the epochs' feature widths; more - K up to 8192 - take the loop.
    const int kq_x = step128(pr.K), kq = kq_tables < kq_x ? kq_tables : kq_x;
    const uint32_t row_bytes = static_cast<uint32_t>(kq_x) * 16u, x_plane = static_cast<uint32_t>(pad8(M)) * row_bytes;
    uint32_t xl[2][NA];   // [k half][plane]: words 2 fh, 2 fh + 1 of the lane's row
#pragma unroll
    for (int p = 0; p < NA; p++) {
        const u32x2 v = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rx, (m < M && p < sh.a) ? static_cast<uint32_t>(p) * x_plane + static_cast<uint32_t>(m) * row_bytes + 8u * fh : 0xffffffffu, 0, 0));
        xl[0][p] = v.x;
        xl[1][p] = v.y;
    }
    u32x4 wc[NCB][2][NDW];
#pragma unroll
    for (int jn = 0; jn < NCB; jn++)
#pragma unroll
        for (int h = 0; h < 2; h++)
#pragma unroll
            for (int dw = 0; dw < NDW; dw++) wc[jn][h][dw] = w_codes[((jn * 2 + h) * NDW + dw) * 64 + lane];
    i32x8 xa[2][NDA];
#pragma unroll
    for (int h = 0; h < 2; h++)
#pragma unroll
        for (int da = 0; da < NDA; da++) xa[h][da] = fp4_op(strip_operand<NA>(xl[h], da));
    uint32_t *tbase = static_cast<uint32_t *>(pr.out) + static_cast<size_t>(grp * 4 + wv) * lines * 4;   // word wv of k-quad grp
    f32x16 accs[NCB];   // (every column block's MFMAs before the first epilogue: see the second product of rbw_chain_body)
#pragma unroll
    for (int jn = 0; jn < NCB; jn++) {
        f32x16 acc = f32x16_zero();   // (a constant C operand of the first MFMA, not sixteen v_mov)
#pragma unroll
        for (int h = 0; h < 2; h++)
#pragma unroll
            for (int da = 0; da < NDA; da++)
#pragma unroll
                for (int dw = 0; dw < NDW; dw++)
                    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(xa[h][da], fp4_op(wc[jn][h][dw]), acc, 4, 4, 0, 128 + 2 * da, 0, 128 + 2 * dw);   // not swapped: lane = column 32 jn + fl
        accs[jn] = acc;
    }
    for (int q = 1; q < kq; q++) {   // (wave-uniform) the further k-quads of a wide feature matrix: its words, that k-quad's weight table
#pragma unroll
        for (int p = 0; p < NA; p++) {
            const u32x2 v = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rx, (m < M && p < sh.a) ? static_cast<uint32_t>(p) * x_plane + static_cast<uint32_t>(m) * row_bytes + static_cast<uint32_t>(q) * 16u + 8u * fh : 0xffffffffu, 0, 0));
            xl[0][p] = v.x;
            xl[1][p] = v.y;
        }
        const u32x4 *wq = w_codes + static_cast<size_t>(q) * (NCB == 3 ? 4 : NCB) * 2 * NDW * 64;   // (a table has weight_table_blocks(N) column blocks)
#pragma unroll
        for (int h = 0; h < 2; h++)
#pragma unroll
            for (int da = 0; da < NDA; da++) xa[h][da] = fp4_op(strip_operand<NA>(xl[h], da));
#pragma unroll
        for (int jn = 0; jn < NCB; jn++)
#pragma unroll
            for (int h = 0; h < 2; h++)
#pragma unroll
                for (int dw = 0; dw < NDW; dw++) {
                    const u32x4 w = wq[((jn * 2 + h) * NDW + dw) * 64 + lane];
#pragma unroll
                    for (int da = 0; da < NDA; da++)
                        accs[jn] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(xa[h][da], fp4_op(w), accs[jn], 4, 4, 0, 128 + 2 * da, 0, 128 + 2 * dw);
                }
    }
#pragma unroll
    for (int jn = 0; jn < NCB; jn++) {
        const int n = 32 * jn + fl;
        rbw_store_codes<OB>(accs[jn], tbase + static_cast<size_t>(n) * 4, fh, n < lines);
    }
}

template <int NA, int OB, int NCB>
__global__ __launch_bounds__(256) void k_rbw_xw(const qgtc_problem *__restrict__ prs, const u32x4 *__restrict__ w_codes, int per, int a_planes, int gx, int gy, int kq_tables) {
    // (scalar kernel arguments only: a struct by value is not preloaded into SGPRs - its fields were s_load round trips of their own)
    int grp, batch;
    rbw_ids(per, gx, gy, grp, batch);
    const qgtc_problem pr = prs[batch];
    rbw_pin(pr);
    if (grp >= step128(pr.M)) return;
    const RbwShape sh{per, a_planes, 0};
    rbw_xw_body<NA, OB, NCB>(pr, w_codes, sh, grp, kq_tables);
}

// ------------------------------------------------------------------------------------------
// An aggregation stage with the next layer's X . W in its tail (main_qgtc.py:148-149 / :150-151), or the last aggregation
// (main_qgtc.py:154: float32 out). prs: {A (rows layout, one plane), T (chain format, OB planes' worth of values), out
// unused unless MODE2 == 0}; prs2 (MODE2 1 / 2): {-, -, T' (chain format) or float32 [M, N2]}.
//   MODE2 0: out = float32(A . T) [M, N]              (no second product; NCB2 unused)
//   MODE2 1: T' = requant(requant(A . T) . W')        chain format
//   MODE2 2: out = float32(requant(A . T) . W')       [M, N2]
// ------------------------------------------------------------------------------------------
// -DQGTC_ABL_HALFWORK (timing-only, wrong sums; tools/rbw_bench): a wave does HALF of its matrix-core work and re-quantises half of its
// column blocks - the ceiling of what a 16-row-block form of these kernels (twice the waves, half the chain a wave)
// could take off a launch: its loads, its barrier and its dependent round trips stay as they are.
#ifdef QGTC_ABL_HALFWORK
#define RBW_HALF(cond) if (cond)
#else
#define RBW_HALF(cond)
#endif
#ifdef QGTC_RBW_STAMPS   // tools/rbw_bench.hip: s_memtime at the phases of a wave, kept in scalar registers
#define RBW_STAMP(i) st_[i] = __builtin_amdgcn_s_memtime()
#else
#define RBW_STAMP(i) do { } while (0)
#endif

// AUX: cache policy of the loads of T (0, or AUX_SC1 = agent scope: past the CU's L1 - the whole-epoch kernel below reads a T that
// other workgroups wrote earlier in the SAME launch). w2_lds / t_lds: the workgroup's staging areas (NCB2 x 2 x ND x 64 and 512 u32x4).
template <int OB, int OB2, int MODE2, int NCB1, int NCB2, int AUX>
__device__ __forceinline__ void rbw_chain_body(const qgtc_problem &pr, const qgtc_problem &pr2, const u32x4 *__restrict__ w2_codes,
                                               int grp, int batch, u32x4 *__restrict__ w2_lds, u32x4 *__restrict__ t_lds, bool a_tiles
#ifdef QGTC_RBW_STAMPS
                                               , unsigned long long (&st_)[10]
#endif
) {
    constexpr int MH = (NCB1 + 1) / 2;   // MFMAs (64 elements of K each) of the second product
    const int M = pr.M, K = pr.K, N = pr.N;
#ifdef QGTC_RBW_STAMPS
    asm volatile("" ::"s"(M), "s"(K));
#endif
    RBW_STAMP(1);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fl = lane & 31, fh = lane >> 5;
    const int rb = 4 * grp + wv, m = 32 * rb + fl;
    // ---- what the whole workgroup needs, fetched FIRST and once, into LDS:
    //  * W' (pre-expanded, NCB2 x MH x 1 KB): fetched per wave after the first product - two 16-byte loads per column block
    //    out of L2, one behind the other for want of registers - it cost 0.7 us per column block;
    //  * the DIAGONAL k-quad of T (8 KB of codes): the four row blocks of this workgroup are rows 128 grp .. 128 grp + 127, and
    //    a cluster batch's adjacency is block-diagonal-dominant - their occupancy words almost always name k-quad grp. Every
    //    wave loading it for itself is 16 load instructions x 1 KB per wave, and the CU's one address unit (64 bytes a clock)
    //    was where the waves queued (in-kernel stamps: 2.4 k cycles from the occupancy word to the last load issued).
    constexpr int ND = OB > 2 ? 2 : 1;   // base-4 digits of the aggregate's values and of W' (the epochs' widths: 2 / 2 or 4 / 4 bits)
    const int kq = step128(K);
    const int lines = 128;   // pad128(N), N <= 128
    const bool diag_ok = grp < kq;   // (workgroup-uniform; A is square in the epochs, so the diagonal k-quad exists)
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint32_t *>(pr.X), 0, static_cast<int>(static_cast<uint32_t>(pr.x_words) * 4u), 0x00020000);
    // (the extent of T: what the chain format of a K x N operand takes, and never more than the descriptor says the buffer holds -
    // reads past either come back as zeros)
    const uint32_t t_bytes = static_cast<uint32_t>(kq) * static_cast<uint32_t>(lines) * 64u, t_have = static_cast<uint32_t>(pr.w_words) * 4u;
    const __amdgpu_buffer_rsrc_t rt = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(pr.W), 0, static_cast<int>(t_bytes < t_have ? t_bytes : t_have), 0x00020000);
    const uint32_t row_bytes = static_cast<uint32_t>(kq) * 16u;
    // the lane's row of the adjacency: k-quad q at x_base + q * xq_bytes (rows layout: 16 bytes on along the row; tiles
    // [row block][k-quad][32 rows][16 bytes]: one tile on)   (a_tiles is launch-uniform: scalar selects)
    const uint32_t xq_bytes = a_tiles ? 512u : 16u;
    const uint32_t x_base = m >= M ? 0xffffffffu : (a_tiles ? (static_cast<uint32_t>(rb) * static_cast<uint32_t>(kq) * 32u + static_cast<uint32_t>(fl)) * 16u
                                                              : static_cast<uint32_t>(m) * row_bytes);
#define RBW_XQ(q) (static_cast<uint32_t>(q) * xq_bytes)
    {
        u32x4 td[2];
#pragma unroll
        for (int i = 0; i < 2; i++)
            td[i] = __builtin_amdgcn_raw_buffer_load_b128(rt, (diag_ok && ((tid + 256 * i) & 127) < 32 * NCB1) ? (static_cast<uint32_t>(grp) * 512u + static_cast<uint32_t>(tid + 256 * i)) * 16u : 0xffffffffu, 0, AUX);   // (only the lines the first product reads)
        // (every load of the staging issued before the first LDS write: written as "w2_lds[e] = w2_codes[e]" per piece, each piece
        // waited for ALL outstanding loads - the 8 KB of W' were two round trips one behind the other)
        constexpr int NI = MODE2 == 0 ? 1 : (NCB2 * 2 * ND * 64 + 255) / 256;
        u32x4 wreg[NI];
        if constexpr (MODE2 != 0) {
#pragma unroll
            for (int i = 0; i < NI; i++) {
                const int e = i * 256 + tid;
                wreg[i] = u32x4{0u, 0u, 0u, 0u};
                if (e < NCB2 * 2 * ND * 64 && ((e / (64 * ND)) & 1) < MH) wreg[i] = w2_codes[e];
            }
        }
#pragma unroll
        for (int i = 0; i < 2; i++) t_lds[tid + 256 * i] = td[i];
        if constexpr (MODE2 != 0) {
#pragma unroll
            for (int i = 0; i < NI; i++) {
                const int e = i * 256 + tid;
                if (e < NCB2 * 2 * ND * 64) w2_lds[e] = wreg[i];
            }
        }
    }
    // the lane's words 2 fh, 2 fh + 1 of the diagonal k-quad of its adjacency row, and the occupancy word: in flight over the barrier
    u32x2 xd = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rx, (diag_ok && x_base != 0xffffffffu) ? x_base + RBW_XQ(grp) + 8u * fh : 0xffffffffu, 0, 0));
    unsigned long long todo = kq >= 64 ? ~0ull : ((1ull << kq) - 1ull);
    if (pr.occ && 32 * rb < M) todo &= pr.occ[static_cast<size_t>(rb) * pr.occ_words];
    __syncthreads();   // (every wave of the workgroup is still at its start)
    if (MODE2 != 1 && 32 * rb >= M) return;   // (float32 rows: no rows here; T' still needs its padding words)

    // ---- first product: acc[j] = (A . T)[row fl][columns 32 j + t + 8 gq + 4 fh], swapped operands
    f32x16 acc[NCB1];
    bool any = false;   // (wave-uniform)
    if (32 * rb < M) {
        unsigned todo_lo = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(todo)), todo_hi = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(todo >> 32));
        unsigned long long left = (static_cast<unsigned long long>(todo_hi) << 32) | todo_lo;   // (wave-uniform: scalar registers)
        const bool has_diag = diag_ok && ((left >> grp) & 1ull) != 0ull;
        if (diag_ok) left &= ~(1ull << grp);
#ifdef QGTC_RBW_STAMPS
        asm volatile("" ::"s"(left));
#endif
        RBW_STAMP(2);
        any = has_diag || left != 0ull;
        if (any) {
            // the OTHER occupied k-quads, in PAIRS: the lanes of half fh take the pair's k-quad fh. ONE per-lane offset per
            // pair - word 0 of the lane's k-quad, line fl - and the (word, column block) part as the instruction's scalar
            // offset (per load the first build spent seven VALU operations on its address). A missing k-quad (q < 0) keeps
            // the lane offset at 0xffffffff: out of range whatever is added. Lines up to 32 NCB1 - 1 exist and were written
            // (zeros past N) by the launch that produced T.
            u32x4 xl, tl[NCB1][4];
            auto load_pair = [&]() {
                const int qa = __builtin_ctzll(left);
                left &= left - 1ull;
                const int qb = left != 0ull ? __builtin_ctzll(left) : -1;
                left &= left - 1ull;
                const int q = fh ? qb : qa;
                xl = __builtin_amdgcn_raw_buffer_load_b128(rx, (q >= 0 && x_base != 0xffffffffu) ? x_base + RBW_XQ(q) : 0xffffffffu, 0, 0);
                const uint32_t t_lane = q >= 0 ? (static_cast<uint32_t>(q) * 512u + static_cast<uint32_t>(fl)) * 16u : 0xffffffffu;
#pragma unroll
                for (int j = 0; j < NCB1; j++)
#pragma unroll
                    for (int t = 0; t < 4; t++) tl[j][t] = __builtin_amdgcn_raw_buffer_load_b128(rt, t_lane, (t * 128 + 32 * j) * 16, AUX);
            };
            auto multiply_pair = [&]() {
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    RBW_HALF(t < 2) {
                    const uint32_t xw[1] = {xl[t]};
                    const i32x8 xa = fp4_op(strip_operand<1>(xw, 0));
#pragma unroll
                    for (int j = 0; j < NCB1; j++) {
                        if constexpr (OB <= 2) {
                            acc[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fp4_op(tl[j][t]), xa, acc[j], 4, 4, 0, 128, 0, 128);
                        } else {   // 4-bit values: two base-4 digits in a nibble
                            acc[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fp4_op(tl[j][t][0] & 0x33333333u, tl[j][t][1] & 0x33333333u, tl[j][t][2] & 0x33333333u, tl[j][t][3] & 0x33333333u),
                                                                                     xa, acc[j], 4, 4, 0, 128, 0, 128);
                            acc[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fp4_op((tl[j][t][0] >> 2) & 0x33333333u, (tl[j][t][1] >> 2) & 0x33333333u, (tl[j][t][2] >> 2) & 0x33333333u, (tl[j][t][3] >> 2) & 0x33333333u),
                                                                                     xa, acc[j], 4, 4, 0, 130, 0, 128);
                        }
                    }
                    }
                }
            };
            RBW_STAMP(3);
            // the diagonal k-quad from LDS: MFMA h takes word 2 fh + h of both operands. Always issued - a block whose
            // occupancy word does not name it multiplies zeros - so that the accumulators START here, as the constant C
            // operand of these MFMAs (64 v_mov less per row block) on every path
            if (!has_diag) xd = u32x2{0u, 0u};
#pragma unroll
            for (int h = 0; h < 2; h++) RBW_HALF(h == 0) {
                const uint32_t xw[1] = {xd[h]};
                const i32x8 xa = fp4_op(strip_operand<1>(xw, 0));
#pragma unroll
                for (int j = 0; j < NCB1; j++) {
                    const u32x4 tc = t_lds[(2 * fh + h) * 128 + 32 * j + fl];
                    if constexpr (OB <= 2) {
                        acc[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fp4_op(tc), xa, h == 0 ? f32x16_zero() : acc[j], 4, 4, 0, 128, 0, 128);
                    } else {   // 4-bit values: two base-4 digits in a nibble
                        acc[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fp4_op(tc[0] & 0x33333333u, tc[1] & 0x33333333u, tc[2] & 0x33333333u, tc[3] & 0x33333333u), xa,
                                                                                 h == 0 ? f32x16_zero() : acc[j], 4, 4, 0, 128, 0, 128);
                        acc[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fp4_op((tc[0] >> 2) & 0x33333333u, (tc[1] >> 2) & 0x33333333u, (tc[2] >> 2) & 0x33333333u, (tc[3] >> 2) & 0x33333333u), xa,
                                                                                 acc[j], 4, 4, 0, 130, 0, 128);
                    }
                }
            }
            RBW_STAMP(4);
            // (The first pair's loads issued AHEAD of the diagonal step - in flight during it - were measured: 151 registers
            // instead of 104, three waves per SIMD instead of four, 6.78 against 6.64 us per 128 x 128 chained launch.)
#ifdef QGTC_RBW_NO_SINGLE   // timing-only switch of tools/rbw_bench: a last k-quad without a partner runs as a pair
            while (left != 0ull) {
                load_pair();
                multiply_pair();
            }
#endif
            while ((left & (left - 1ull)) != 0ull) {   // (wave-uniform) at least two left
                load_pair();
                multiply_pair();
            }
            if (left != 0ull) {
                // ONE k-quad left (the commonest row block of a cluster batch has exactly one beside its diagonal): as a "pair"
                // with a missing partner it was 16 load instructions and 4 NCB1 MFMAs with half the lanes on zeros. Shared between
                // the halves like the diagonal step instead - half fh takes words 2 fh, 2 fh + 1 of both operands: 2 NCB1 loads of
                // T, 2 NCB1 MFMAs.
                const int q = __builtin_ctzll(left);
                const u32x2 xs = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rx, x_base != 0xffffffffu ? x_base + RBW_XQ(q) + 8u * fh : 0xffffffffu, 0, 0));
                const uint32_t t_lane = (static_cast<uint32_t>(q) * 512u + static_cast<uint32_t>(256 * fh + fl)) * 16u;
#pragma unroll
                for (int j = 0; j < NCB1; j++)
#pragma unroll
                    for (int h = 0; h < 2; h++) tl[j][h] = __builtin_amdgcn_raw_buffer_load_b128(rt, t_lane, (h * 128 + 32 * j) * 16, AUX);
#pragma unroll
                for (int h = 0; h < 2; h++) RBW_HALF(h == 0) {
                    const uint32_t xw[1] = {xs[h]};
                    const i32x8 xa = fp4_op(strip_operand<1>(xw, 0));
#pragma unroll
                    for (int j = 0; j < NCB1; j++) {
                        if constexpr (OB <= 2) {
                            acc[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fp4_op(tl[j][h]), xa, acc[j], 4, 4, 0, 128, 0, 128);
                        } else {
                            acc[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fp4_op(tl[j][h][0] & 0x33333333u, tl[j][h][1] & 0x33333333u, tl[j][h][2] & 0x33333333u, tl[j][h][3] & 0x33333333u),
                                                                                     xa, acc[j], 4, 4, 0, 128, 0, 128);
                            acc[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fp4_op((tl[j][h][0] >> 2) & 0x33333333u, (tl[j][h][1] >> 2) & 0x33333333u, (tl[j][h][2] >> 2) & 0x33333333u, (tl[j][h][3] >> 2) & 0x33333333u),
                                                                                     xa, acc[j], 4, 4, 0, 130, 0, 128);
                        }
                    }
                }
            }
        }
    }
    if (!any) {
        // No occupied k-quad, or a padding block of T': the aggregate's rows are zero, so is everything after them - the
        // wave stores its zeros and is done. (As a join of two paths into the products below, the accumulators had to be
        // zeroed up front on BOTH: 64 v_mov per wave.)
        if constexpr (MODE2 == 0) {
            if (m < M)
                for (int c = fh; c < N; c += 2) static_cast<float *>(pr.out)[static_cast<size_t>(m) * N + c] = 0.0f;
        } else {
            const qgtc_problem &pz = pr2;
            if constexpr (MODE2 == 2) {
                if (m < M)
                    for (int c = fh; c < pz.N; c += 2) static_cast<float *>(pz.out)[static_cast<size_t>(m) * pz.N + c] = 0.0f;
            } else {
                uint32_t *tz = static_cast<uint32_t *>(pz.out) + static_cast<size_t>(grp * 4 + wv) * 128 * 4;
#pragma unroll
                for (int jn = 0; jn < NCB2; jn++)
                    if (fh == 0) *reinterpret_cast<u32x4 *>(tz + (32 * jn + fl) * 4) = u32x4{0u, 0u, 0u, 0u};
            }
        }
        return;
    }

#ifdef QGTC_RBW_STAMPS
    asm volatile("" ::"v"(acc[0][0]), "v"(acc[NCB1 - 1][15]));
#endif
    RBW_STAMP(5);
    if constexpr (MODE2 == 0) {   // float32 [M, N] (kernel.h:915-930): registers 4 g .. 4 g + 3 are four consecutive columns of row m
        const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(pr.out, 0, static_cast<int>(static_cast<uint32_t>(M) * static_cast<uint32_t>(N) * 4u), 0x00020000);
        const uint32_t row_off = m < M ? static_cast<uint32_t>(m) * static_cast<uint32_t>(N) * 4u : 0xffffffffu;
#pragma unroll
        for (int j = 0; j < NCB1; j++) rbw_store_f32_row(ro, row_off, acc[j], 32 * j, fh, N);
        return;
    } else {
        // ---- the aggregate's row as the second product's left operand, straight from the registers
        uint32_t XA[MH][4];
#pragma unroll
        for (int mm = 0; mm < MH; mm++)
#pragma unroll
            for (int d = 0; d < 4; d++) XA[mm][d] = 0u;
#pragma unroll
        for (int j = 0; j < NCB1; j++) RBW_HALF(NCB1 == 1 || (j & 1) == 0) {
            uint32_t qv[16], P[4];
            requant_pack16<OB>(acc[j], OB, P, qv);
            rbw_nibbles<OB>(P, XA[j >> 1][2 * (j & 1)], XA[j >> 1][2 * (j & 1) + 1]);
        }
#ifdef QGTC_RBW_STAMPS
        asm volatile("" ::"v"(XA[0][0]), "v"(XA[MH - 1][3]));
#endif
        RBW_STAMP(6);
        const int N2 = pr2.N, lines2 = pad128(N2);
        uint32_t *tbase = static_cast<uint32_t *>(pr2.out) + static_cast<size_t>(grp * 4 + wv) * lines2 * 4;   // word wv of k-quad grp
        const __amdgpu_buffer_rsrc_t ro2 = __builtin_amdgcn_make_buffer_rsrc(pr2.out, 0, MODE2 == 2 ? static_cast<int>(static_cast<uint32_t>(M) * static_cast<uint32_t>(N2) * 4u) : 0, 0x00020000);
        // ALL column blocks' MFMAs first, into accumulators of their own (the first product's are dead by now): written as "multiply a
        // block, re-quantise it, store it" per block, every block reused the same sixteen registers and its MFMAs waited for the previous
        // block's epilogue - 0.45 k cycles per block one behind the other (in-kernel stamps: 1.9 k for four blocks)
        f32x16 acc2s[NCB2];
#pragma unroll
        for (int jn = 0; jn < NCB2; jn++) {
            f32x16 acc2 = f32x16_zero();
#pragma unroll
            for (int mm = 0; mm < MH; mm++) RBW_HALF(NCB2 == 1 || (jn & 1) == 0) {
                // T' (MODE2 1): not swapped - lane = column 32 jn + fl of T', 16 rows in its registers: what a chain-format line takes.
                // float32 rows (MODE2 2): SWAPPED - lane = row fl, registers 4 g .. 4 g + 3 = four consecutive columns, stored as vectors
                // (rbw_store_f32_row; unswapped, a lane stored its column's 16 rows one float at a time: 16 store instructions per
                // column block, 40 useful bytes each at 10 classes)
                if constexpr (ND == 1) {
                    const i32x8 xa = fp4_op(XA[mm][0], XA[mm][1], XA[mm][2], XA[mm][3]), wb = fp4_op(w2_lds[(jn * 2 + mm) * 64 + lane]);
                    if constexpr (MODE2 == 2) acc2 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wb, xa, acc2, 4, 4, 0, 128, 0, 128);
                    else acc2 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(xa, wb, acc2, 4, 4, 0, 128, 0, 128);
                } else {   // 4-bit aggregate x 4-bit W': two base-4 digits each, the E8M0 scales carry 4^(da + dw)
#pragma unroll
                    for (int da = 0; da < 2; da++) {
                        const i32x8 xa = fp4_op((XA[mm][0] >> (2 * da)) & 0x33333333u, (XA[mm][1] >> (2 * da)) & 0x33333333u, (XA[mm][2] >> (2 * da)) & 0x33333333u, (XA[mm][3] >> (2 * da)) & 0x33333333u);
#pragma unroll
                        for (int dw = 0; dw < 2; dw++) {
                            const i32x8 wb = fp4_op(w2_lds[((jn * 2 + mm) * 2 + dw) * 64 + lane]);
                            if constexpr (MODE2 == 2) acc2 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wb, xa, acc2, 4, 4, 0, 128 + 2 * dw, 0, 128 + 2 * da);
                            else acc2 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(xa, wb, acc2, 4, 4, 0, 128 + 2 * da, 0, 128 + 2 * dw);
                        }
                    }
                }
            }
            acc2s[jn] = acc2;
        }
#pragma unroll
        for (int jn = 0; jn < NCB2; jn++) RBW_HALF(NCB2 == 1 || (jn & 1) == 0) {
            const f32x16 &acc2 = acc2s[jn];
            const int n2 = 32 * jn + fl;
            if constexpr (MODE2 == 2) {   // float32 rows of the swapped product: lane = row m, columns 32 jn + 8 g + 4 fh + t in register 4 g + t
                rbw_store_f32_row(ro2, m < M ? static_cast<uint32_t>(m) * static_cast<uint32_t>(N2) * 4u : 0xffffffffu, acc2, 32 * jn, fh, N2);
            } else {
                rbw_store_codes<OB2>(acc2, tbase + static_cast<size_t>(n2) * 4, fh, n2 < lines2);
            }
        }
        RBW_STAMP(8);
#ifdef QGTC_RBW_STAMPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        RBW_STAMP(9);
        if (tid == 0) {
            const int slot = (batch * static_cast<int>(gridDim.x) + grp) % 1024;
            for (int i = 0; i < 10; i++) g_stamps[slot * 16 + i] = st_[i];
        }
#endif
    }
}

template <int OB, int OB2, int MODE2, int NCB1, int NCB2>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 4))) void k_rbw_chain(const qgtc_problem *__restrict__ prs, const qgtc_problem *__restrict__ prs2,
                                                                                                const u32x4 *__restrict__ w2_codes, int per, int tiles, int gx, int gy) {
    constexpr int ND = OB > 2 ? 2 : 1;
    __shared__ __attribute__((aligned(16))) u32x4 w2_lds[MODE2 == 0 ? 1 : NCB2 * 2 * ND * 64];
    __shared__ __attribute__((aligned(16))) u32x4 t_lds[4 * 128];   // [word t of the k-quad][line n]
#ifdef QGTC_RBW_STAMPS
    unsigned long long st_[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
    RBW_STAMP(0);
    int grp, batch;
    rbw_ids(per, gx, gy, grp, batch);
    // both descriptors at once, ahead of everything (the second stage's was fetched after the first product: a scalar-load round trip
    // between the two products of every wave)
    const qgtc_problem pr = prs[batch];
    const qgtc_problem pr2 = MODE2 == 0 ? pr : prs2[batch];
    rbw_pin(pr);
    if constexpr (MODE2 != 0) asm volatile("" ::"s"(pr2.out), "s"(pr2.N));
    if (grp >= step128(pr.M)) return;
    rbw_chain_body<OB, OB2, MODE2, NCB1, NCB2, 0>(pr, pr2, w2_codes, grp, batch, w2_lds, t_lds, tiles != 0
#ifdef QGTC_RBW_STAMPS
                                                  , st_
#endif
    );
}

}  // namespace
