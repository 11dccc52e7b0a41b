// bitmm_planes.hip.h — part of libqgtc_hip.so (included by qgtc_hip.hip, one translation unit).
// The bit-GEMM, popcount engine, for MANY bit planes of which few hold a set bit: more than eight planes on a side (9 .. 32: beyond every
// matrix-core form and every fixed-plane kernel of bitmm_popcount.hip.h).
#pragma once

namespace {

// ------------------------------------------------------------------------------------------
// The reference's checked-in epoch script launches exactly this (0_7a_eval_QGTC_cluster_GCN.py:10: bitwidth = 32, so a = w = ob = 32 in
// every product of main_qgtc.py:147-154): N(0,1) features quantise to 0 .. 4, all-ones weights to 1, sums to a few hundred - three of X's
// 32 planes, one of W's and ten of an intermediate's are non-zero, and the reference pays all 1024 b1 MMAs a tile (kernel.h:292-341).
// The generic kernel (k_bitmm<1, 0, 0>: plane blocks of 8 x 8 walked as stages, each a dependent memory round trip and an 8 x 8 loop of
// plane tests) took 27 - 34 us a product there: 6 us to find its first stage, 2.4 us a stage, 17 us to store 32 output planes word by
// word from one lane in eight. This kernel is built around what such operands are:
//   * CENSUS first: every granule (16 bytes) of the tile's lines of BOTH operands is read once, many loads a lane in flight, and ORed into
//     a 32-bit mask of the planes that hold a set bit (ballots; one LDS atomic a plane and wave);
//   * only the set planes are staged in LDS - eight a side and eight k-quads at a time, one barrier pair a chunk - and multiplied: thread
//     (row, column group) owns four outputs, AND + v_bcnt on 16-byte granules, shifted by the planes' own numbers (kernel.h:295,340:
//     b_opt = pa + pw; a shift of 32 and more contributes nothing);
//   * the tile's re-quantised values go through LDS once, so that a thread builds whole output words of a LINE - 32 rows of a column or 32
//     columns of a row, the same code for both packed layouts - for its share of the planes; planes above the tile's largest value are
//     stored as zeros without looking at the values (the sums of a 32-bit epoch need ten of 32 planes).
// Same words as every other kernel: the int32 sums wrap (uint32 arithmetic), the re-quantisation is kernel.h:31-37's float compare.
// A workgroup = 256 threads = one 32 x 32 output tile for the whole K; grouped launches: blockIdx.y = problem.
// ------------------------------------------------------------------------------------------
constexpr int PL_THREADS = 256;
constexpr int PL_PB = 8;               // set planes of an operand staged at a time
constexpr int PL_KC = 8;               // k-quads staged at a time
constexpr int PL_PITCH = PL_KC + 1;    // granules between staged lines (odd: eight lines apart land on eight different bank quads)

struct PlanesLds {
    u32x4 x[PL_PB][32][PL_PITCH];
    u32x4 w[PL_PB][32][PL_PITCH];
    int val[32][33];
    uint32_t xmask, wmask, need;
    int xl[32], wl[32];   // the set planes of X / W, ascending
};

// planes of `planes` (mask bit p) with a set bit in the tile's lines line0 .. line0 + 31 (< lines_in), every k-quad: the waves take the
// planes round robin, a lane up to 16 independent loads at a time
__device__ __forceinline__ void planes_census(const __amdgpu_buffer_rsrc_t &rs, int planes, int line0, int lines_in, uint32_t plane_words, uint32_t kw, int kq,
                                              int wv, int lane, uint32_t *mask) {
    const int mine_planes = (planes - wv + 3) / 4;   // planes wv, wv + 4, ..
    if (mine_planes <= 0) return;
    const int per_plane = 32 * kq, items = mine_planes * per_plane;
    uint32_t nz = 0u;   // bit j: the lane saw a set bit in plane wv + 4 j
    for (int it0 = 0; it0 < items; it0 += 64 * 16) {
        u32x4 v[16];
        int pj[16];
#pragma unroll
        for (int z = 0; z < 16; z++) {
            const int it = it0 + 64 * z + lane;
            const int j = it / per_plane, rem = it - j * per_plane, r = rem / kq, q = rem - r * kq;
            pj[z] = j;
            const bool ok = it < items && line0 + r < lines_in;
            v[z] = __builtin_amdgcn_raw_buffer_load_b128(rs, ok ? (static_cast<uint32_t>(wv + 4 * j) * plane_words + static_cast<uint32_t>(line0 + r) * kw) * 4u +
                                                                 static_cast<uint32_t>(q) * 16u : 0xffffffffu, 0, 0);
        }
#pragma unroll
        for (int z = 0; z < 16; z++)
            if (((v[z].x | v[z].y) | (v[z].z | v[z].w)) != 0u) nz |= 1u << pj[z];
    }
    uint32_t found = 0u;
    for (int j = 0; j < mine_planes; j++)
        if (__ballot((nz >> j) & 1u) != 0ull) found |= 1u << (wv + 4 * j);
    if (lane == 0 && found) atomicOr(mask, found);
}

template <bool BATCHED>
__global__ __launch_bounds__(PL_THREADS) void k_bitmm_planes(const qgtc_problem *__restrict__ prs, qgtc_problem pr1, MMShape sh) {
    extern __shared__ __attribute__((aligned(16))) unsigned char pl_smem[];
    PlanesLds &L = *reinterpret_cast<PlanesLds *>(pl_smem);
    const qgtc_problem pr = BATCHED ? prs[blockIdx.y] : pr1;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int M = pr.M, K = pr.K, N = pr.N;
    const int tiles_m = (M + 31) / 32, tiles_n = (N + 31) / 32;
    const int tile = static_cast<int>(blockIdx.x);
    if (tile >= tiles_m * tiles_n) return;   // (grouped launches are sized for the largest problem)
    const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
    const int m0 = tm * 32, n0 = tn * 32;
    const int kq = step128(K);
    const uint32_t kw = static_cast<uint32_t>(kq) * 4u;
    const uint32_t x_plane = static_cast<uint32_t>(pad8(M)) * kw, w_plane = static_cast<uint32_t>(pr.w_lines) * kw;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(pr.X), 0, static_cast<int>(static_cast<uint32_t>(pr.x_words) * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(pr.W), 0, static_cast<int>(static_cast<uint32_t>(pr.w_words) * 4u), 0x00020000);
    const int a = sh.a, w = sh.w;

    // ---- census
    if (tid == 0) {
        L.xmask = a == 1 ? 1u : 0u;
        L.wmask = w == 1 ? 1u : 0u;
        L.need = 0u;
    }
    __syncthreads();
    if (a > 1) planes_census(rx, a, m0, M, x_plane, kw, kq, wv, lane, &L.xmask);
    if (w > 1) planes_census(rw, w, n0, N, w_plane, kw, kq, wv, lane, &L.wmask);
    __syncthreads();
    const uint32_t xmask = L.xmask, wmask = L.wmask;
    const int nx = __builtin_popcount(xmask), nw = __builtin_popcount(wmask);
    if (tid < 32) {   // the i-th set plane of each operand
        uint32_t mx = xmask, mw = wmask;
        for (int i = 0; i < tid; i++) {
            mx &= mx - 1u;
            mw &= mw - 1u;
        }
        L.xl[tid] = mx ? __builtin_ctz(mx) : 0;
        L.wl[tid] = mw ? __builtin_ctz(mw) : 0;
    }
    // (published by the first barrier of the chunk loop, or the one below when there is nothing to multiply)

    // ---- the set planes, eight a side and eight k-quads at a time
    const int row = tid >> 3, cj = tid & 7;   // the thread's outputs: row m0 + row, columns n0 + cj + 8 c
    uint32_t tot[4] = {0u, 0u, 0u, 0u};       // unsigned: the reference's int32 accumulation wraps on overflow
    if (nx == 0 || nw == 0) __syncthreads();
    for (int xb = 0; xb < nx; xb += PL_PB)
        for (int wb = 0; wb < nw; wb += PL_PB)
            for (int kc = 0; kc < kq; kc += PL_KC) {
                const int nxb = min(PL_PB, nx - xb), nwb = min(PL_PB, nw - wb), nk = min(PL_KC, kq - kc);
                __syncthreads();   // (the plane lists; the previous chunk's reads)
                const int gx = nxb * 32 * nk, gtot = gx + nwb * 32 * nk;
                for (int it0 = 0; it0 < gtot; it0 += PL_THREADS * 8) {   // eight loads a thread in flight, then their LDS stores (a load and its
                    u32x4 v[8];                                         // store per trip of a loop is a memory round trip per granule)
                    u32x4 *dst[8];
#pragma unroll
                    for (int z = 0; z < 8; z++) {
                        const int it = it0 + PL_THREADS * z + tid;
                        const bool is_x = it < gx;
                        const int u = is_x ? it : it - gx;
                        const int pi = u / (32 * nk), rem = u - pi * (32 * nk), ln = rem / nk, q = rem - ln * nk;
                        const int plane = is_x ? L.xl[xb + (pi & 7)] : L.wl[wb + (pi & 7)];
                        const int gl = (is_x ? m0 : n0) + ln;
                        const bool ok = it < gtot && gl < (is_x ? M : N);
                        const uint32_t off = (static_cast<uint32_t>(plane) * (is_x ? x_plane : w_plane) + static_cast<uint32_t>(gl) * kw) * 4u + static_cast<uint32_t>(kc + q) * 16u;
                        v[z] = __builtin_amdgcn_raw_buffer_load_b128(is_x ? rx : rw, ok ? off : 0xffffffffu, 0, 0);
                        dst[z] = it < gtot ? (is_x ? &L.x[pi][ln][q] : &L.w[pi][ln][q]) : nullptr;
                    }
#pragma unroll
                    for (int z = 0; z < 8; z++)
                        if (dst[z]) *dst[z] = v[z];
                }
                __syncthreads();
                for (int pi = 0; pi < nxb; pi++) {
                    const int pa = L.xl[xb + pi];
                    for (int pj = 0; pj < nwb; pj++) {
                        const int s = pa + L.wl[wb + pj];   // reference kernel.h:295,340
                        if (s >= 32) continue;              // (workgroup-uniform)
                        uint32_t cnt[4] = {0u, 0u, 0u, 0u};
                        for (int q = 0; q < nk; q++) {
                            const u32x4 xg = L.x[pi][row][q];
#pragma unroll
                            for (int c = 0; c < 4; c++) {
                                const u32x4 wg = L.w[pj][cj + 8 * c][q];
                                cnt[c] += __builtin_popcount(xg.x & wg.x) + __builtin_popcount(xg.y & wg.y) + __builtin_popcount(xg.z & wg.z) + __builtin_popcount(xg.w & wg.w);
                            }
                        }
#pragma unroll
                        for (int c = 0; c < 4; c++) tot[c] += cnt[c] << s;
                    }
                }
            }

    // ---- epilogue
    const int m = m0 + row;
    if (sh.mode == 2) {   // float32 [M,N] (reference kernel.h:915-930)
        if (m < M) {
            float *dst = static_cast<float *>(pr.out) + static_cast<size_t>(m) * N;
#pragma unroll
            for (int c = 0; c < 4; c++)
                if (n0 + cj + 8 * c < N) dst[n0 + cj + 8 * c] = static_cast<float>(static_cast<int>(tot[c]));
        }
        return;
    }
    uint32_t allq = 0u;
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const int r = requant(static_cast<int>(tot[c]), sh.maxv, sh.maxm1);   // kernel.h:31-37
        const uint32_t v = (m < M && n0 + cj + 8 * c < N) ? static_cast<uint32_t>(r) : 0u;
        L.val[row][cj + 8 * c] = static_cast<int>(v);
        allq |= v;
    }
    if (__ballot(allq != 0u) != 0ull) {   // planes up to the highest set bit of the tile's values: the rest are stored as zeros
        int top = 0;
        for (int sft = 16; sft >= 1; sft >>= 1)
            if (__ballot((allq >> (top + sft)) != 0u) != 0ull) top += sft;
        if (lane == 0) atomicMax(&L.need, static_cast<uint32_t>(top + 1));
    }
    __syncthreads();
    const int ob = sh.ob, np = min(static_cast<int>(L.need), ob);
    const int line = tid & 31, grp = tid >> 5;   // the thread builds the words of one line of the tile for planes grp, grp + 8, ..
    uint32_t vals[32];
#pragma unroll
    for (int e = 0; e < 32; e++) vals[e] = static_cast<uint32_t>(sh.mode == 0 ? L.val[line][e] : L.val[e][line]);
    uint32_t *out = static_cast<uint32_t *>(pr.out);
    const bool last_m = tm == tiles_m - 1, last_n = tn == tiles_n - 1;
    if (sh.mode == 0) {   // rows layout [ob][PAD8(M)][STEP128(N) * 4] (reference kernel.h:357-389): word (m, n0 / 32)
        const int rows_pad = pad8(M), row_words = step128(N) * 4;
        const size_t oplane = static_cast<size_t>(rows_pad) * row_words;
        const int extra = last_n ? row_words - (n0 >> 5) - 1 : 0;   // the last column tile also zeroes the row words past it
        const int mm = m0 + line;
        if (mm < rows_pad) {
            uint32_t *dst = out + static_cast<size_t>(mm) * row_words + (n0 >> 5);
            const bool vec = tiles_n == 1 && row_words == 4;        // (a whole packed row: one 16-byte store)
            for (int p = grp; p < ob; p += PL_THREADS / 32) {
                uint32_t word = 0u;
                if (p < np) {
#pragma unroll
                    for (int e = 0; e < 32; e++) word |= ((vals[e] >> p) & 1u) << (31 - e);
                }
                uint32_t *o = dst + static_cast<size_t>(p) * oplane;
                if (vec) {
                    *reinterpret_cast<u32x4 *>(o) = u32x4{word, 0u, 0u, 0u};
                } else {
                    o[0] = word;
                    for (int x = 1; x <= extra; x++) o[x] = 0u;
                }
            }
        }
        return;
    }
    // cols layout [ob][PAD128(N)][STEP128(M) * 4] (intended semantics of kernel.h:651-810): word (n, m0 / 32)
    const int lines = pad128(N), line_words = step128(M) * 4;
    const size_t oplane = static_cast<size_t>(lines) * line_words;
    const int n = n0 + line;
    if (n < lines) {
        uint32_t *dst = out + static_cast<size_t>(n) * line_words + (m0 >> 5);
        for (int p = grp; p < ob; p += PL_THREADS / 32) {
            uint32_t word = 0u;
            if (p < np) {
#pragma unroll
                for (int e = 0; e < 32; e++) word |= ((vals[e] >> p) & 1u) << (31 - e);
            }
            dst[static_cast<size_t>(p) * oplane] = word;
        }
    }
    // zero what no tile computes: words past the last row tile, lines past the last column tile
    const int w_core0 = m0 >> 5, w_core1 = min(line_words, w_core0 + 1);
    if (last_m && w_core1 < line_words) {
        for (int e = tid; e < ob * 32; e += PL_THREADS) {
            const int ln2 = n0 + (e & 31), p = e >> 5;
            if (ln2 < lines)
                for (int wi = w_core1; wi < line_words; wi++) out[p * oplane + static_cast<size_t>(ln2) * line_words + wi] = 0u;
        }
    }
    if (last_n && n0 + 32 < lines) {
        const int nl = lines - (n0 + 32), w_end = last_m ? line_words : w_core1;
        for (int e = tid; e < ob * nl; e += PL_THREADS) {
            const int ln2 = n0 + 32 + e % nl, p = e / nl;
            for (int wi = w_core0; wi < w_end; wi++) out[p * oplane + static_cast<size_t>(ln2) * line_words + wi] = 0u;
        }
    }
}

}  // namespace
