// bitmm_layer.hip.h — part of libqgtc_hip.so (included by qgtc_mfma.hip and qgtc_fp4.hip).
// One quantised GNN layer for a group of cluster batches in ONE launch (qgtc_gcn_layer_batched):
//   stage 1   T_b   = cols-layout pack of requant(X_b . W)          (what bitMM2Bit_col produces)
//   stage 2   out_b = A_b . T_b                                     (bitMM2Bit rows-layout bits, or bitMM2Int float32)
// i.e. the reference's per-layer pair (QGTC_conv.py:14-22; main_qgtc.py:147-154 issues it as two extension calls per
// cluster batch) without the launch boundary between the two products.
#pragma once

namespace {

// ------------------------------------------------------------------------------------------
// Stage 2 of batch b needs ALL of T_b, which the stage-1 workgroups of batch b write - a dependency between
// workgroups of one launch. It is carried by one arrival counter per batch:
//   * the launch is a 1-D grid cut into SLOTS of t1 + t2 workgroups; slot s holds the stage-1 tiles of batch s and
//     the stage-2 tiles of batch s - delay. Workgroups are dispatched in id order (per XCD, and a workgroup's XCD is
//     a function of its id), so every producer a stage-2 workgroup waits for has a LOWER id: it is resident or done
//     whenever the consumer runs, and resident workgroups make progress on their own - the wait cannot deadlock.
//     `delay` slots of other work sit between producer and consumer, so the consumer usually finds its counter full;
//   * a producer stores T with agent scope (st_word<true>: write-through, the consumer may sit on another XCD whose
//     L2 is not coherent with ours), drains its stores (vmcnt(0)), meets its workgroup at a barrier and thread 0 adds
//     1 to the batch's counter (relaxed, agent scope);
//   * a consumer's thread 0 polls the counter (relaxed agent-scope loads, s_sleep between polls) until it reaches
//     epoch x (stage-1 tiles of the batch) - the counters are never reset, `epoch` counts the launches of the plan -
//     then the workgroup passes a barrier and reads T with agent-scope (sc1) loads: always from the memory side,
//     never from an L2 line filled before the last writer was done.
// MEASURED on MI355X (tools/epoch_stages.py, the ogbn-arxiv-sized epoch, 75 batches, F = H = 128, 2-bit): a wide layer
// takes 35 us this way (31 us even with plain stores and loads, which are only safe when a batch stays on one XCD)
// against 30 us for the two grouped launches (12 + 18); the 10-class layer 37 against 19. The hand-off - write-through
// stores, the producers' drain, polls and arrivals at the memory side (with all counters in one cache line the polls
// alone tripled the launch: hence the 256-byte stride), consumers holding CU slots while they wait - costs more than
// the ~1.5 us launch boundary it removes. qgtc_gcn_layer_batched therefore issues the two grouped launches by default
// and this form only on request (QGTC_LAYER_ONE_LAUNCH); it stays tested word for word against the oracle.
// The tile bodies are the library's own: mf_tile (128 x 128 tiles on the matrix cores, bitmm_mfma.hip.h) when the
// layer is wide, fw_tile (one wave per 32 x 32 tile, bitmm_fp4_wave.hip.h) when it is narrow.
// ------------------------------------------------------------------------------------------
struct LayerShape {
    MMShape sh1, sh2;      // per-stage constants (planes, output bits, mode)
    int t1, t2;            // workgroups per slot: stage-1 / stage-2 tiles of the largest batch
    int count, delay;      // batches; slots between a batch's two stages
    uint32_t epoch;        // 1-based launch number of this plan (arrival counters are monotonic)
    int zero_skip;
};

__device__ __forceinline__ void layer_arrive(uint32_t *counter) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's write-through stores have reached memory
    __syncthreads();                                    // ... and every other wave's of the workgroup
    if (threadIdx.x == 0) __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ void layer_wait(const uint32_t *counter, uint32_t target) {
    if (threadIdx.x == 0) {
        // agent-scope polls and arrivals are served at the memory side, one channel per counter line: every counter has
        // its own 256-byte line (QGTC_ARRIVAL_STRIDE) and a waiting workgroup polls about once per microsecond
        while (static_cast<int>(__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0)
            __builtin_amdgcn_s_sleep(40);
    }
    __syncthreads();
    // no acquire fence: it would invalidate this XCD's L2 for every workgroup on it (measured: the launch took 3x
    // longer). The consumer reads T with agent-scope loads instead (WCOH), and nothing else it reads was written
    // inside this launch.
}

// slot / role of this workgroup; returns false when it has nothing to do. A slot holds EIGHT batches (an "octet"),
// workgroup ids interleaved so that id % 8 is the batch's position in its octet: under round-robin placement of
// workgroups over the 8 XCDs every workgroup of a batch - producers and consumers - runs on one XCD and shares one L2.
__device__ __forceinline__ bool layer_role(const LayerShape &ls, int &batch, int &tile, bool &second) {
    const int per = ls.t1 + ls.t2;
    const int id = static_cast<int>(blockIdx.x);
    const int slot = id / (8 * per), r8 = id % (8 * per);
    const int x = r8 & 7, r = r8 >> 3;
    second = r >= ls.t1;
    tile = second ? r - ls.t1 : r;
    batch = 8 * (second ? slot - ls.delay : slot) + x;
    return batch >= 0 && batch < ls.count;
}

#ifdef QGTC_LAYER_MFMA
// wide layers: 128 x 128 tiles, four multiplier + four expander waves (bitmm_mfma.hip.h)
template <int MAXP, bool FP4>
__global__ __launch_bounds__(512) void k_layer_mfma(const qgtc_problem *__restrict__ p1, const qgtc_problem *__restrict__ p2,
                                                    uint32_t *__restrict__ arrival, LayerShape ls) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int batch, tile;
    bool second;
    if (!layer_role(ls, batch, tile, second)) return;
    if (!second) {
        const qgtc_problem pr = p1[batch];
        const int tiles_m = (pr.M + MF_T - 1) / MF_T, tiles_n = (pr.N + MF_T - 1) / MF_T;
        if (tile >= tiles_m * tiles_n) return;
        mf_tile<MAXP, 4, FP4, true>(pr, ls.sh1, tile / tiles_n, tile % tiles_n, smem);
        layer_arrive(arrival + QGTC_ARRIVAL_STRIDE * batch);
    } else {
        const qgtc_problem pr = p2[batch];
        const int tiles_m = (pr.M + MF_T - 1) / MF_T, tiles_n = (pr.N + MF_T - 1) / MF_T;
        if (tile >= tiles_m * tiles_n) return;
        const int m1 = p1[batch].M, n1 = p1[batch].N;
        const uint32_t need = static_cast<uint32_t>(((m1 + MF_T - 1) / MF_T) * ((n1 + MF_T - 1) / MF_T));
        layer_wait(arrival + QGTC_ARRIVAL_STRIDE * batch, ls.epoch * need);
        mf_tile<MAXP, 4, FP4, false, true>(pr, ls.sh2, tile / tiles_n, tile % tiles_n, smem);
    }
}
#endif

#ifdef QGTC_LAYER_WAVE
// narrow layers: one wave per 32 x 32 tile (bitmm_fp4_wave.hip.h); MODE2 = 0 (rows-layout bits) or 2 (float32)
template <int NA, int NW, int MODE2>
__global__ __launch_bounds__(64) void k_layer_wave(const qgtc_problem *__restrict__ p1, const qgtc_problem *__restrict__ p2,
                                                   uint32_t *__restrict__ arrival, LayerShape ls) {
    int batch, tile;
    bool second;
    if (!layer_role(ls, batch, tile, second)) return;
    if (!second) {
        const qgtc_problem pr = p1[batch];
        if (tile >= ((pr.M + 31) / 32) * ((pr.N + 31) / 32)) return;
        fw_tile<NA, NW, 1, 2, 2, true>(pr, ls.sh1, ls.zero_skip, tile);
        layer_arrive(arrival + QGTC_ARRIVAL_STRIDE * batch);
    } else {
        const qgtc_problem pr = p2[batch];
        if (tile >= ((pr.M + 31) / 32) * ((pr.N + 31) / 32)) return;
        const int m1 = p1[batch].M, n1 = p1[batch].N;
        const uint32_t need = static_cast<uint32_t>(((m1 + 31) / 32) * ((n1 + 31) / 32));
        layer_wait(arrival + QGTC_ARRIVAL_STRIDE * batch, ls.epoch * need);
        fw_tile<NA, NW, MODE2, 2, 2, false, true>(pr, ls.sh2, ls.zero_skip, tile);
    }
}
#endif

}  // namespace
