// loader_kernels.hip.h — part of libqgtc_hip.so (included by qgtc_hip.hip).
// The data loader's packing for ALL cluster batches of an iterator in a handful of launches (qgtc_load_batches): what
// sampler.py:76-106 does per batch - adjacency from the batch's edges, `QGTC.val2bit(A, 1, False, False)`,
// `QGTC.val2bit(X, b, True, False)` - plus the formats the grouped epoch reads (rows-layout X, 512-byte adjacency tiles,
// occupancy bitmaps, X in the chain format), each written by the kernel that has the data in registers anyway.
// Round 3 issued these per batch: 75 x (2 memsets + edge count + edge finish + 2 val2bit + rows->tiles) + one bitmap launch
// = 2.6 ms of 4 - 8 us launches for the ogbn-arxiv-sized iterator (profiles/r03/kernel_stats_epoch.csv).
//   blockIdx.y = batch everywhere; the per-batch table (qgtc_loader_batch) lives in device memory.
#pragma once

namespace {

// Edges -> the three unary multiplicity bitmaps of k_edge_list_count (pack_kernels.hip.h), t1 = the batch's A buffer, t2 / t3 =
// its scratch halves. Indices are local to the batch (row = src, col = dst: sampler.py:80-89).
__global__ __launch_bounds__(256) void k_load_edges(const qgtc_loader_batch *__restrict__ tb, const int64_t *__restrict__ src,
                                                    const int64_t *__restrict__ dst, int *__restrict__ bad) {
    const qgtc_loader_batch b = tb[blockIdx.y];
    const int n = b.n, row_words = step128(n) * 4;
    const size_t words = static_cast<size_t>(pad8(n)) * row_words;
    uint32_t *t1 = b.A, *t2 = b.scratch, *t3 = b.scratch + words;
    for (unsigned long long e = static_cast<unsigned long long>(blockIdx.x) * blockDim.x + threadIdx.x; e < b.n_edges;
         e += static_cast<unsigned long long>(gridDim.x) * blockDim.x) {
        const int64_t r = src[b.edge_off + e], c = dst[b.edge_off + e];
        if (r < 0 || r >= n || c < 0 || c >= n) {
            if (bad) *bad = 1;
            continue;
        }
        const uint32_t bit = 1u << (31 - (c & 31));
        const size_t wi = static_cast<size_t>(r) * row_words + (c >> 5);
        if (atomicOr(t1 + wi, bit) & bit)
            if (atomicOr(t2 + wi, bit) & bit) atomicOr(t3 + wi, bit);
    }
}

// plane 0 = t1 & (~t2 | t3) (k_edge_list_finish), written back to the rows layout AND - from the same registers - as
// 512-byte tiles (k_rows_to_tiles: [32-row block][k-quad][32 rows][4 words]) and into the occupancy bitmap
// (k_tile_occupancy: bit q of word [row block][q / 64]); the occupied-tile count goes to stats[0]. One wave per (32-row
// block, bitmap word); lanes = k-quads, and with at most 16 k-quads (n <= 2048) four rows at a time so that the wave's
// lanes are not mostly idle.
__global__ __launch_bounds__(256) void k_load_finish(const qgtc_loader_batch *__restrict__ tb, unsigned long long *__restrict__ stats) {
    const qgtc_loader_batch b = tb[blockIdx.y];
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int n = b.n, kq = step128(n), tiles_m = (n + TM - 1) / TM, ow = (kq + 63) / 64;
    if (wave >= tiles_m * ow) return;   // whole waves
    const int tm = wave / ow, wi = wave - tm * ow;
    const bool narrow = kq <= 16;       // wave-uniform
    const int q = wi * 64 + (narrow ? (lane & 15) : lane), rp = narrow ? 4 : 1;
    const size_t kw = static_cast<size_t>(kq) * 4u, words = static_cast<size_t>(pad8(n)) * kw;
    uint32_t *t1 = b.A;
    const uint32_t *t2 = b.scratch, *t3 = b.scratch + words;
    uint32_t any = 0u;
    if (q < kq) {
#pragma unroll 4
        for (int r = narrow ? (lane >> 4) : 0; r < TM; r += rp) {
            const int m = tm * TM + r;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (m < pad8(n)) {
                const size_t off = static_cast<size_t>(m) * kw + static_cast<size_t>(q) * 4u;
                const u32x4 a = *reinterpret_cast<const u32x4 *>(t1 + off);
                const u32x4 c2 = *reinterpret_cast<const u32x4 *>(t2 + off);
                const u32x4 c3 = *reinterpret_cast<const u32x4 *>(t3 + off);
                v = u32x4{a.x & (~c2.x | c3.x), a.y & (~c2.y | c3.y), a.z & (~c2.z | c3.z), a.w & (~c2.w | c3.w)};
                *reinterpret_cast<u32x4 *>(t1 + off) = v;
            }
            if (b.AT) *reinterpret_cast<u32x4 *>(b.AT + ((static_cast<size_t>(tm) * kq + q) * 32u + r) * 4u) = v;
            any |= (v.x | v.y) | (v.z | v.w);
        }
    }
    if (narrow) {   // the four row groups of a k-quad sit 16 lanes apart
        any |= __shfl_xor(any, 16);
        any |= __shfl_xor(any, 32);
    }
    unsigned long long mask = __ballot(any != 0u);
    if (narrow) mask &= 0xffffull;
    if (lane == 0) {
        if (b.occ) b.occ[static_cast<size_t>(tm) * ow + wi] = mask;
        if (stats && mask) atomicAdd(stats, static_cast<unsigned long long>(__popcll(mask)));
    }
}

// val2bit of every batch's features (rows feat_row .. feat_row + n - 1 of `feats`, F columns): the cols layout the reference
// packs (sampler.py:99) and the rows layout the layout-correct chain's first X.W reads. Same device code as the single
// launches (pack_kernels.hip.h), so the same words; V4 = the float4 path (F % 4 == 0).
template <int NB>
__global__ __launch_bounds__(256) void k_load_x_cols(const qgtc_loader_batch *__restrict__ tb, const float *__restrict__ feats, int F, int nbits,
                                                     float ub, float ubm1) {
    const qgtc_loader_batch b = tb[blockIdx.y];
    if (!b.X) return;
    val2bit_cols_body<NB>(feats + b.feat_row * static_cast<size_t>(F), b.n, F, nbits, ub, ubm1, b.X, pad128(F), step128(b.n) * 4,
                          (static_cast<long>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6, (static_cast<long>(gridDim.x) * blockDim.x) >> 6);
}

template <bool V4>
__global__ __launch_bounds__(256) void k_load_x_rows(const qgtc_loader_batch *__restrict__ tb, const float *__restrict__ feats, int F, int nbits,
                                                     float ub, float ubm1) {
    const qgtc_loader_batch b = tb[blockIdx.y];
    if (!b.XR) return;
    const float *x = feats + b.feat_row * static_cast<size_t>(F);
    if constexpr (V4)
        val2bit_rows_v4_body<2>(x, b.n, F, nbits, ub, ubm1, b.XR, pad8(b.n), step128(F) * 4, (blockIdx.x * blockDim.x + threadIdx.x) >> 6,
                                (gridDim.x * blockDim.x) >> 6);
    else
        val2bit_rows_body(x, b.n, F, nbits, ub, ubm1, b.XR, pad8(b.n), step128(F) * 4,
                          (static_cast<long>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6, (static_cast<long>(gridDim.x) * blockDim.x) >> 6);
}

}  // namespace
