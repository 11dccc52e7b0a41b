// tile_stats_kernels.hip.h — part of libqgtc_hip.so (included by qgtc_hip.hip, one translation unit).
// Occupancy bitmaps (zero-tile jumping) and the reference's tile counters.
#pragma once

namespace {

// ------------------------------------------------------------------------------------------
// Occupancy bitmap of a rows-layout operand: bit q of word (tile, q/64) says whether the 32-row x
// 128-bit tile (row tile, k-quad q) has a bit set in any plane. One wave per (row tile, word):
// lane = k-quad, 32 x planes coalesced 16-byte loads per lane, one ballot.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_tile_occupancy(const uint32_t *__restrict__ X, unsigned x_bytes,
                                                        int M, int K, int a, unsigned long long *__restrict__ occ,
                                                        int occ_words, int tiles_m) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (wave >= tiles_m * occ_words) return;  // whole waves
    const int tm = wave / occ_words, wi = wave % occ_words;
    const int kq = step128(K), q = wi * 64 + lane;
    const uint32_t kw = static_cast<uint32_t>(kq) * 4u, x_plane = static_cast<uint32_t>(pad8(M)) * kw;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(X), 0,
                                                                       static_cast<int>(x_bytes), 0x00020000);
    uint32_t any = 0u;
    for (int p = 0; p < a; p++)
#pragma unroll 8
        for (int r = 0; r < TM; r++) {
            const int m = tm * TM + r;
            const uint32_t off = (q < kq && m < M) ? (p * x_plane + m * kw + q * 4u) * 4u : 0xffffffffu;
            const u32x4 g = __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0);
            any |= (g.x | g.y) | (g.z | g.w);
        }
    const unsigned long long m = __ballot(any != 0u);
    if (lane == 0) occ[static_cast<size_t>(tm) * occ_words + wi] = m;
}

// The same for every problem of a grouped launch (blockIdx.y = problem): the bitmaps go where the
// descriptors' `occ` fields point. One launch instead of one per cluster batch.
__global__ __launch_bounds__(256) void k_tile_occupancy_batched(const qgtc_problem *__restrict__ prs, int a) {
    const qgtc_problem pr = prs[blockIdx.y];
    if (!pr.occ) return;
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int M = pr.M, K = pr.K, tiles_m = (M + TM - 1) / TM, occ_words = pr.occ_words;
    if (wave >= tiles_m * occ_words) return;  // whole waves
    const int tm = wave / occ_words, wi = wave % occ_words;
    const int kq = step128(K), q = wi * 64 + lane;
    const uint32_t kw = static_cast<uint32_t>(kq) * 4u, x_plane = static_cast<uint32_t>(pad8(M)) * kw;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint32_t *>(pr.X), 0, static_cast<int>(static_cast<uint32_t>(pr.x_words) * 4u), 0x00020000);
    uint32_t any = 0u;
    for (int p = 0; p < a; p++)
#pragma unroll 8
        for (int r = 0; r < TM; r++) {
            const int m = tm * TM + r;
            const uint32_t off = (q < kq && m < M) ? (p * x_plane + m * kw + q * 4u) * 4u : 0xffffffffu;
            const u32x4 g = __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0);
            any |= (g.x | g.y) | (g.z | g.w);
        }
    const unsigned long long m = __ballot(any != 0u);
    if (lane == 0) const_cast<unsigned long long *>(reinterpret_cast<const unsigned long long *>(pr.occ))[static_cast<size_t>(tm) * occ_words + wi] = m;
}

// Decide ON THE DEVICE whether a grouped launch should jump: count the occupied 32-row x 128-bit
// tiles of all problems; if more than max_fraction of them are occupied the bitmaps only cost a
// dependent load ahead of every tile's first loads, so the descriptors' `occ` fields are cleared (the
// kernels then visit every k-quad). stats[0] = occupied tiles, stats[1] = all tiles (for the host, lazily).
// One workgroup; no host round trip inside the plan build.
__global__ __launch_bounds__(1024) void k_occupancy_decide(qgtc_problem *__restrict__ prs, int count,
                                                           float max_fraction, unsigned long long *__restrict__ stats) {
    __shared__ unsigned long long part[16];
    __shared__ int drop;
    unsigned long long set = 0ull, all = 0ull;
    for (int i = 0; i < count; i++) {
        const int tiles_m = (prs[i].M + TM - 1) / TM, ow = prs[i].occ_words;
        const unsigned long long *occ = reinterpret_cast<const unsigned long long *>(prs[i].occ);
        if (!occ) continue;
        for (int e = threadIdx.x; e < tiles_m * ow; e += blockDim.x) set += __popcll(occ[e]);
        all += static_cast<unsigned long long>(tiles_m) * step128(prs[i].K);
    }
    for (int o = 32; o > 0; o >>= 1) set += __shfl_xor(set, o);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = set;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long tot = 0ull;
        for (int w = 0; w < static_cast<int>(blockDim.x >> 6); w++) tot += part[w];
        stats[0] = tot;
        stats[1] = all;
        drop = all == 0ull || static_cast<double>(tot) > static_cast<double>(max_fraction) * static_cast<double>(all);
    }
    __syncthreads();
    if (drop)
        for (int i = threadIdx.x; i < count; i += blockDim.x) {
            prs[i].occ = nullptr;
            prs[i].occ_words = 0;
        }
}

// ------------------------------------------------------------------------------------------
// tile counters (reference kernel.h:452, :574-592): one thread per (plane, 8-row block, k-step)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_tile_counters(const uint32_t *__restrict__ X,
                                                       unsigned long long x_words, int M, int K,
                                                       int a, unsigned long long mult_total,
                                                       unsigned long long mult_nz,
                                                       unsigned long long *__restrict__ counters) {
    const int gdx = step8(M), gdk = step128(K);
    const size_t kw = static_cast<size_t>(gdk) * 4;
    const size_t x_plane = static_cast<size_t>(pad8(M)) * kw;
    const size_t total = static_cast<size_t>(a) * gdx * gdk;
    unsigned long long local = 0;
    for (size_t t = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; t < total;
         t += static_cast<size_t>(gridDim.x) * blockDim.x) {
        const size_t i = t % gdk, bx = (t / gdk) % gdx, pa = t / (static_cast<size_t>(gdk) * gdx);
        uint32_t any = 0;
        for (int r = 0; r < 8; r++) {
            const uint4 g = ldg4(X, x_words, pa * x_plane + (bx * 8 + r) * kw + i * 4);
            any |= g.x | g.y | g.z | g.w;
        }
        local += any ? 1u : 0u;
    }
    // wave reduce, then one atomic per wave
    for (int off = 32; off > 0; off >>= 1) local += __shfl_down(local, off);
    if ((threadIdx.x & 63) == 0 && local) atomicAdd(&counters[1], local * mult_nz);
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&counters[0], mult_total);
}

}  // namespace
