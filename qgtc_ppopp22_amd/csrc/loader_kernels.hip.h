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


// ------------------------------------------------------------------------------------------
// The adjacency WITHOUT dense multiplicity bitmaps (round 5; VERDICT r4 weak 10: k_load_finish moved 73 MB for 14 MB of adjacency).
// The route above sets one device-scope atomic per edge into a zeroed bitmap (537 k atomics: 18 us on the ogbn-arxiv-sized iterator -
// tools/atomic_probe.hip: the memory side takes ~18 G of them a second wherever they land), needs the three bitmaps cleared
// (41 MB of memset) and reads all three back densely to write rows + tiles (k_load_finish: 43 us). Here:
//   k_load_sort   one workgroup per batch buckets the batch's edges by 32-ROW BLOCK: histogram in LDS, scan, scatter of the edges as
//                 32-bit (row in block, column) words into the caller's work buffer - no global atomic, the edge list read twice;
//   k_load_tiles  one wave per (batch, row block) builds the block's 32 x n bits in LDS from its bucket (LDS atomics; the three
//                 multiplicity bitmaps of k_edge_list_count live THERE: a few KB) and writes every word of the block's rows, its
//                 512-byte tiles and its occupancy word exactly once: nothing to clear first, nothing read back.
// Same words as the route above (tests/test_loader_gpu.py compares both with the oracle). Limits: max_n <= LOAD_SORT_MAX_N (the three
// bitmaps of a row block in 64 KB of LDS); larger batches take the route above.
// ------------------------------------------------------------------------------------------
constexpr int LOAD_SORT_MAX_KQ = 40, LOAD_SORT_MAX_N = LOAD_SORT_MAX_KQ * 128;   // 3 x 40 x 512 bytes = 60 KB of LDS a row block
constexpr int LOAD_SORT_THREADS = 1024;

// Runs of equal keys over the lanes of a wave (every lane must be active): the lane where this lane's run starts and the run's length.
__device__ __forceinline__ void wave_runs(int key, int &run_start, int &run_len) {
    const int lane = threadIdx.x & 63;
    const int prev = __shfl_up(key, 1);
    const unsigned long long starts = __ballot(lane == 0 || key != prev);
    const unsigned long long upto = lane == 63 ? ~0ull : ((2ull << lane) - 1ull);   // lanes 0 .. lane
    run_start = 63 - __builtin_clzll(starts & upto);
    const unsigned long long later = starts & ~upto;
    run_len = (later ? __builtin_ctzll(later) : 64) - run_start;
}

// work: [count x (RB + 1) bucket offsets | the buckets of batch 0 | batch 1 | .. (total_edges_pad words: the edge count rounded up to
// even) | count x RB occupied-tile counts, one per row block] with RB = row blocks of the largest batch; batch b's edges sit at the offset its edge
// list has in src / dst. A batch whose buckets would not fit is skipped and reported.
__global__ __launch_bounds__(LOAD_SORT_THREADS) void k_load_sort(const qgtc_loader_batch *__restrict__ tb, const int64_t *__restrict__ src,
                                                                 const int64_t *__restrict__ dst, uint32_t *__restrict__ work,
                                                                 unsigned long long work_words, int rb_max, int count, unsigned long long total_edges_pad,
                                                                 int *__restrict__ bad) {
    __shared__ unsigned hist[LOAD_SORT_MAX_N / 32 + 1];
    __shared__ unsigned wave_sum[LOAD_SORT_THREADS / 64];
    const qgtc_loader_batch b = tb[blockIdx.x];
    const int n = b.n, tid = threadIdx.x, tiles_m = (n + 31) / 32;
    uint32_t *offs = work + static_cast<size_t>(blockIdx.x) * (rb_max + 1);
    uint32_t *bucket = work + static_cast<size_t>(count) * (rb_max + 1) + b.edge_off;
    const bool fits = b.edge_off + b.n_edges <= total_edges_pad && tiles_m <= rb_max;   // (the host sized the three parts of `work`)
    for (int i = tid; i <= rb_max; i += LOAD_SORT_THREADS) hist[i] = 0u;
    __syncthreads();
    if (!fits) {
        if (tid == 0 && bad) *bad = 1;
        for (int i = tid; i <= rb_max; i += LOAD_SORT_THREADS) offs[i] = 0u;   // (empty buckets: the batch's adjacency comes out zero)
        return;
    }
    const int64_t *s = src + b.edge_off, *d = dst + b.edge_off;
    // A batch of at most EPT x 1024 edges (the cluster batches of the reference's datasets have 2 - 70 k) is read ONCE: every load issued
    // before the first is used, the packed edges stay in registers over the scan. Longer lists are read twice, edge by edge.
    constexpr int EPT = 8, CH = 4;
    const bool resident = b.n_edges <= static_cast<unsigned long long>(EPT) * LOAD_SORT_THREADS;   // (workgroup-uniform)
    uint32_t pk[EPT];
    int bucket_of[EPT];
    if (resident) {
        int64_t rr[EPT], cc[EPT];
#pragma unroll
        for (int k = 0; k < EPT; k++) {
            const unsigned long long e = static_cast<unsigned long long>(k) * LOAD_SORT_THREADS + tid;
            rr[k] = e < b.n_edges ? s[e] : -1;
            cc[k] = e < b.n_edges ? d[e] : 0;
        }
#pragma unroll
        for (int k = 0; k < EPT; k++) {
            const unsigned long long e = static_cast<unsigned long long>(k) * LOAD_SORT_THREADS + tid;
            const bool ok = rr[k] >= 0 && rr[k] < n && cc[k] >= 0 && cc[k] < n;
            if (e < b.n_edges && !ok && bad) *bad = 1;
            bucket_of[k] = ok ? static_cast<int>(rr[k]) >> 5 : -1;
            pk[k] = (static_cast<uint32_t>(rr[k]) & 31u) << 27 | static_cast<uint32_t>(cc[k]);
            // one LDS atomic per RUN of equal buckets in the wave, not per edge: an edge list is row-sorted inside a partition, so the 64
            // lanes of a wave hold a few rows - 64 atomics on two or three LDS words serialised (rocprofv3: SQ_LDS_BANK_CONFLICT 975 k of
            // 1058 k LDS cycles, 16 us for the ogbn-arxiv-sized iterator)
            int run_start, run_len;
            wave_runs(bucket_of[k], run_start, run_len);
            if (ok && (tid & 63) == run_start) atomicAdd(&hist[bucket_of[k]], static_cast<unsigned>(run_len));
        }
    } else {
        // (in passes of CH x 1024 edges, every load of a pass issued before the first is used, and the same one-atomic-per-run histogram:
        // edge by edge with an atomic each, the ppi-sized iterator's 17 k-edge batches took 24 us - SQ_LDS_BANK_CONFLICT 1.39 M cycles)
        for (unsigned long long e0 = 0; e0 < b.n_edges; e0 += static_cast<unsigned long long>(CH) * LOAD_SORT_THREADS) {   // (workgroup-uniform)
            int64_t rr[CH], cc[CH];
#pragma unroll
            for (int k = 0; k < CH; k++) {
                const unsigned long long e = e0 + static_cast<unsigned long long>(k) * LOAD_SORT_THREADS + tid;
                rr[k] = e < b.n_edges ? s[e] : -1;
                cc[k] = e < b.n_edges ? d[e] : 0;
            }
#pragma unroll
            for (int k = 0; k < CH; k++) {
                const unsigned long long e = e0 + static_cast<unsigned long long>(k) * LOAD_SORT_THREADS + tid;
                const bool ok = rr[k] >= 0 && rr[k] < n && cc[k] >= 0 && cc[k] < n;
                if (e < b.n_edges && !ok && bad) *bad = 1;
                const int bk = ok ? static_cast<int>(rr[k]) >> 5 : -1;
                int run_start, run_len;
                wave_runs(bk, run_start, run_len);
                if (ok && (tid & 63) == run_start) atomicAdd(&hist[bk], static_cast<unsigned>(run_len));
            }
        }
    }
    __syncthreads();
    // exclusive scan of the (at most LOAD_SORT_MAX_N / 32 = 160) bucket sizes: one element a thread, wave scans + the waves' totals
    {
        const unsigned v = tid < tiles_m ? hist[tid] : 0u;
        unsigned incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned t = __shfl_up(incl, o);
            if ((tid & 63) >= o) incl += t;
        }
        if ((tid & 63) == 63) wave_sum[tid >> 6] = incl;
        __syncthreads();
        unsigned before = 0u;
        for (int w = 0; w < (tid >> 6); w++) before += wave_sum[w];
        __syncthreads();
        if (tid <= rb_max) {
            const unsigned start = tid < tiles_m ? before + incl - v : 0u;
            hist[tid] = start;            // now the bucket's cursor
            if (tid < tiles_m) offs[tid] = start;
        }
        if (tid == tiles_m - 1) {         // the end of the last bucket; the unused tail of the table points there too
            for (int i = tiles_m; i <= rb_max; i++) offs[i] = before + incl;
        }
    }
    __syncthreads();
    if (resident) {
#pragma unroll
        for (int k = 0; k < EPT; k++) {
            int run_start, run_len;
            wave_runs(bucket_of[k], run_start, run_len);
            unsigned base = 0u;
            if (bucket_of[k] >= 0 && (tid & 63) == run_start) base = atomicAdd(&hist[bucket_of[k]], static_cast<unsigned>(run_len));
            base = __shfl(base, run_start);                       // the run's first slot, from its first lane
            if (bucket_of[k] >= 0) bucket[base + static_cast<unsigned>((tid & 63) - run_start)] = pk[k];
        }
    } else {
        for (unsigned long long e0 = 0; e0 < b.n_edges; e0 += static_cast<unsigned long long>(CH) * LOAD_SORT_THREADS) {
            int64_t rr[CH], cc[CH];
#pragma unroll
            for (int k = 0; k < CH; k++) {
                const unsigned long long e = e0 + static_cast<unsigned long long>(k) * LOAD_SORT_THREADS + tid;
                rr[k] = e < b.n_edges ? s[e] : -1;
                cc[k] = e < b.n_edges ? d[e] : 0;
            }
#pragma unroll
            for (int k = 0; k < CH; k++) {
                const bool ok = rr[k] >= 0 && rr[k] < n && cc[k] >= 0 && cc[k] < n;
                const int bk = ok ? static_cast<int>(rr[k]) >> 5 : -1;
                int run_start, run_len;
                wave_runs(bk, run_start, run_len);
                unsigned base = 0u;
                if (ok && (tid & 63) == run_start) base = atomicAdd(&hist[bk], static_cast<unsigned>(run_len));
                base = __shfl(base, run_start);
                if (ok) bucket[base + static_cast<unsigned>((tid & 63) - run_start)] = (static_cast<uint32_t>(rr[k]) & 31u) << 27 | static_cast<uint32_t>(cc[k]);
            }
        }
    }
}

// One wave (a 64-thread workgroup) per (row block, batch). LDS: t1 | t2 | t3, each [32 rows][kq k-quads][4 words] of the block.
__global__ __launch_bounds__(64) void k_load_tiles(const qgtc_loader_batch *__restrict__ tb, const uint32_t *__restrict__ work, int rb_max, int count,
                                                   unsigned long long total_edges_pad, unsigned long long *__restrict__ stats) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const qgtc_loader_batch b = tb[blockIdx.y];
    const int n = b.n, kq = step128(n), tiles_m = (n + 31) / 32, rb = blockIdx.x, lane = threadIdx.x;
    // the block's occupied-tile count goes to its own word behind the buckets (k_load_stats sums them): no atomic
    uint32_t *tile_count = const_cast<uint32_t *>(work) + static_cast<size_t>(count) * (rb_max + 1) + total_edges_pad + static_cast<size_t>(blockIdx.y) * rb_max + rb;
    if (rb >= tiles_m) {
        if (lane == 0 && stats) *tile_count = 0u;
        return;
    }
    const int row_words = kq * 4, tw = 32 * row_words;   // words of one bitmap of the block
    uint32_t *t1 = lds, *t2 = lds + tw, *t3 = lds + 2 * tw;
    const uint32_t *offs = work + static_cast<size_t>(blockIdx.y) * (rb_max + 1);
    const uint32_t *bucket = work + static_cast<size_t>(count) * (rb_max + 1) + b.edge_off;
    const unsigned e0 = offs[rb], e1 = offs[rb + 1];
    // the first 256 edges of the bucket (a row block of the epochs' graphs has ~190) are IN FLIGHT while the bitmaps are cleared: loaded
    // behind the barrier they were a memory round trip of their own in every wave's chain
    constexpr int PRE = 4;
    uint32_t pre[PRE];
#pragma unroll
    for (int i = 0; i < PRE; i++) pre[i] = e0 + 64u * i + lane < e1 ? bucket[e0 + 64u * i + lane] : 0xffffffffu;
    for (int i = lane; i < 3 * tw / 4; i += 64) reinterpret_cast<u32x4 *>(lds)[i] = u32x4{0u, 0u, 0u, 0u};
    __syncthreads();
    auto place = [&](uint32_t pk) {
        const int r = static_cast<int>(pk >> 27), c = static_cast<int>(pk & 0x07ffffffu);
        const uint32_t bit = 1u << (31 - (c & 31));
        const int wi = r * row_words + (c >> 5);
        if (atomicOr(&t1[wi], bit) & bit)                        // multiplicities 1, 2, >= 3 land in t1, t2, t3 (k_edge_list_count)
            if (atomicOr(&t2[wi], bit) & bit) atomicOr(&t3[wi], bit);
    };
#pragma unroll
    for (int i = 0; i < PRE; i++)
        if (e0 + 64u * i + lane < e1) place(pre[i]);
    for (unsigned e = e0 + 64u * PRE + lane; e < e1; e += 64) place(bucket[e]);
    __syncthreads();
    // rows layout: [row][k-quad] - consecutive units of a row are contiguous (every row below pad8(n) is written, zeros past the edges)
    const int units = 32 * kq, rows_here = min(32, pad8(n) - 32 * rb);
    uint32_t *rows_out = b.A + static_cast<size_t>(32 * rb) * row_words;
    for (int u = lane; u < units; u += 64) {
        const int r = u / kq;
        const u32x4 a = reinterpret_cast<const u32x4 *>(t1)[u], c2 = reinterpret_cast<const u32x4 *>(t2)[u], c3 = reinterpret_cast<const u32x4 *>(t3)[u];
        const u32x4 v = {a.x & (~c2.x | c3.x), a.y & (~c2.y | c3.y), a.z & (~c2.z | c3.z), a.w & (~c2.w | c3.w)};   // the 1-bit quantiser's image of 1, 2, >= 3: 1, 0, 1
        reinterpret_cast<u32x4 *>(t1)[u] = v;
        if (r < rows_here) __builtin_nontemporal_store(v, reinterpret_cast<u32x4 *>(rows_out) + u);   // (nobody in this launch reads it back)
    }
    __syncthreads();
    // tiles [k-quad][32 rows][4 words] (512 contiguous bytes a tile) and the occupancy bits: lanes 0 .. 31 = the rows of k-quad 2 i, 32 .. 63 of 2 i + 1
    unsigned long long mask_lo = 0ull;   // (kq <= 40: one 64-bit word per row block)
    uint32_t *tile_out = b.AT ? b.AT + static_cast<size_t>(rb) * kq * 128u : nullptr;
    for (int q0 = 0; q0 < kq; q0 += 2) {
        const int q = q0 + (lane >> 5), r = lane & 31;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (q < kq) v = reinterpret_cast<const u32x4 *>(t1)[r * kq + q];
        if (tile_out && q < kq) __builtin_nontemporal_store(v, reinterpret_cast<u32x4 *>(tile_out) + (q * 32 + r));
        const unsigned long long bal = __ballot(((v.x | v.y) | (v.z | v.w)) != 0u);
        if (bal & 0xffffffffull) mask_lo |= 1ull << q0;
        if (bal >> 32) mask_lo |= 1ull << (q0 + 1);
    }
    if (lane == 0) {
        if (b.occ) b.occ[rb] = mask_lo;
        // the occupied-tile count: a plain store to the block's own word; k_load_stats, behind this launch, adds them all to stats[0].
        // Round 5's first forms: every row block adding to stats[0] itself = 2850 device-scope atomics on ONE address, serialised at
        // ~15 ns each: 42 of 45 us (and of round 4's k_load_finish); then per-batch 64-bit counters whose returned old value named the
        // batch's last row block: an atomic that RETURNS is a round trip to the memory side at the very end of every wave - 5.6 of
        // 16 us (timing build without it: 10.4); the same adds not waited for still cost 3.5.
        if (stats) *tile_count = static_cast<uint32_t>(__popcll(mask_lo));
    }
}

// stats[0] += the occupied-tile counts k_load_tiles left per row block (`words` of them; one workgroup of 1024; four loads a thread in
// flight at once - as a plain `s += counts[i]` loop of 256 threads the twelve loads of a thread went out one behind the other: 4.9 us)
__global__ __launch_bounds__(1024) void k_load_stats(const uint32_t *__restrict__ counts, int words, unsigned long long *__restrict__ stats) {
    __shared__ unsigned part[16];
    unsigned s = 0u;
    for (int i0 = threadIdx.x; i0 < words; i0 += 4096) {
        unsigned v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) v[k] = i0 + 1024 * k < words ? counts[i0 + 1024 * k] : 0u;
        s += (v[0] + v[1]) + (v[2] + v[3]);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x < 16) {
        unsigned t = part[threadIdx.x];
#pragma unroll
        for (int d = 8; d >= 1; d >>= 1) t += __shfl_xor(t, d);
        if (threadIdx.x == 0) atomicAdd(stats, static_cast<unsigned long long>(t));
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

// both layouts of X from ONE read of the features (val2bit_cols_rows_body); batches without an XR take the cols-only body
template <int NB>
__global__ __launch_bounds__(256) void k_load_x_both(const qgtc_loader_batch *__restrict__ tb, const float *__restrict__ feats, int F, int nbits,
                                                     float ub, float ubm1) {
    const qgtc_loader_batch b = tb[blockIdx.y];
    if (!b.X) return;
    const long wave = (static_cast<long>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6, nwaves = (static_cast<long>(gridDim.x) * blockDim.x) >> 6;
    if (b.XR)
        val2bit_cols_rows_body<NB>(feats + b.feat_row * static_cast<size_t>(F), b.n, F, nbits, ub, ubm1, b.X, pad128(F), step128(b.n) * 4, b.XR, pad8(b.n),
                                   step128(F) * 4, wave, nwaves);
    else
        val2bit_cols_body<NB>(feats + b.feat_row * static_cast<size_t>(F), b.n, F, nbits, ub, ubm1, b.X, pad128(F), step128(b.n) * 4, wave, nwaves);
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
