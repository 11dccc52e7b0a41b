// epoch_plan.hip.h — part of libqgtc_hip.so (qgtc_epoch.hip).
// What an epoch needs besides its products, done ON THE DEVICE so that the host's share of the reference's epoch clock
// (main_qgtc.py:96-159: weights packed, outputs allocated and every operator launched inside it) is a handful of calls:
//   * k_val2bit_jobs      - val2bit of several small matrices in one launch (the three weight matrices, main_qgtc.py:100-110)
//   * k_epoch_plan_fill   - the qgtc_problem descriptors of every stage of a grouped epoch from the data loader's
//                           per-batch table (qgtc_batch) and the stage recipes (qgtc_stage): outputs carved out of one pool
//   * k_check_descriptors - QGTC_CHECK_DESCRIPTORS: the grouped entries' preconditions, checked where the descriptors live
#pragma once

namespace {

// ------------------------------------------------------------------------------------------
// val2bit jobs: one thread per packed WORD of a job's output (all planes of it). Weights are a few thousand words; the
// point is one launch instead of three or four, not bandwidth. Same words as k_val2bit_rows / k_val2bit_cols: element i of
// a line at word i >> 5, bit 31 - (i & 31); lines and words past the matrix are zero (reference kernel.h:75-106, :204-242).
// ------------------------------------------------------------------------------------------
struct PackJobs {
    qgtc_pack_job job[QGTC_MAX_PACK_JOBS];
    int n;
};

__global__ __launch_bounds__(256) void k_val2bit_jobs(PackJobs jobs) {
    const qgtc_pack_job j = jobs.job[blockIdx.y];
    const int H = j.H, W = j.W, nbits = j.nbits;
    // lines x words of one plane; `along` = the matrix dimension packed into a line's bits
    const int lines = j.col_major ? (j.output_layer ? pad8(W) : pad128(W)) : pad8(H);
    const int line_words = step128(j.col_major ? H : W) * 4;
    const size_t plane = static_cast<size_t>(lines) * line_words;
    const float ub = __builtin_ldexpf(1.0f, nbits), ubm1 = ub - 1.0f;
    for (size_t t = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; t < plane; t += static_cast<size_t>(gridDim.x) * blockDim.x) {
        // cols layout: consecutive threads take consecutive COLUMNS (lines) of one word index - coalesced reads of x rows
        const int line = j.col_major ? static_cast<int>(t % lines) : static_cast<int>(t / line_words);
        const int wi = j.col_major ? static_cast<int>(t / lines) : static_cast<int>(t % line_words);
        uint32_t q[32];
#pragma unroll
        for (int e = 0; e < 32; e++) {
            const int pos = wi * 32 + e;
            const int r = j.col_major ? pos : line, c = j.col_major ? line : pos;
            q[e] = (r < H && c < W) ? quant1(j.x[static_cast<size_t>(r) * W + c], ub, ubm1) : 0u;
        }
        for (int p = 0; p < nbits; p++) {
            uint32_t word = 0u;
#pragma unroll
            for (int e = 0; e < 32; e++) word |= ((q[e] >> p) & 1u) << (31 - e);
            j.out[p * plane + static_cast<size_t>(line) * line_words + wi] = word;
        }
    }
}

// ------------------------------------------------------------------------------------------
// Epoch plan. The pool rule shared with the host (qgtc_epoch_pool_layout): outputs in (stage, batch) order, each a
// multiple of four words (16-byte aligned views).
// ------------------------------------------------------------------------------------------
__host__ __device__ inline unsigned long long stage_out_words(const qgtc_stage &st, int n) {
    if (st.mode == 2) return (static_cast<unsigned long long>(n) * st.N + 3ull) & ~3ull;
    if (st.mode == 1 && st.fmt == 1) return static_cast<unsigned long long>(step128(n)) * pad128(st.N) * 16ull * (st.ob > 4 ? 2ull : 1ull);   // chain format (qgtc_chain_words)
    if (st.mode == 1) return static_cast<unsigned long long>(st.ob) * step128(n) * 4ull * pad128(st.N);
    return static_cast<unsigned long long>(st.ob) * pad8(n) * step128(st.N) * 4ull;
}

struct PlanArgs {
    qgtc_stage stage[QGTC_MAX_STAGES];
    qgtc_operand weight[QGTC_MAX_WEIGHTS];
    int n_stages, n_weights, count;
};

constexpr int PLAN_THREADS = 256;   // (1024 threads cap a thread at 128 registers: the descriptor assembly spilled)
__global__ __launch_bounds__(PLAN_THREADS) void k_epoch_plan_fill(const qgtc_batch *__restrict__ batches, PlanArgs pa, uint32_t *__restrict__ pool,
                                                          unsigned long long pool_words, qgtc_problem *__restrict__ descs, int *__restrict__ record) {
    // One workgroup; per (stage, 256 batches) an exclusive scan of the outputs' sizes: inside a wave with shuffles, across the four
    // waves through four LDS words - two barriers per pass. `carry` is the same number in every thread.
    __shared__ unsigned long long wave_total[PLAN_THREADS / 64];
    // the recipes and weights in LDS: indexed by run-time stage / source numbers below, and a by-value kernel argument indexed that
    // way is copied to SCRATCH (200 bytes a thread here: the launch then waits for the queue's scratch allocation - 36 us under
    // the tracer for a kernel with 3 us of work, inside the epoch clock)
    __shared__ qgtc_stage sst[QGTC_MAX_STAGES];
    __shared__ qgtc_operand swt[QGTC_MAX_WEIGHTS];
    const int tid = threadIdx.x, count = pa.count, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) {
#pragma unroll
        for (int j = 0; j < QGTC_MAX_STAGES; j++) sst[j] = pa.stage[j];
#pragma unroll
        for (int j = 0; j < QGTC_MAX_WEIGHTS; j++) swt[j] = pa.weight[j];
    }
    __syncthreads();
    unsigned long long carry = 0ull;
    // Up to 256 batches (one pass per stage): a thread keeps its batch's table line and the outputs it has handed out in registers -
    // re-reading both from memory per stage (a 96-byte line, then a pointer this thread stored one stage earlier) was two exposed
    // round trips per stage.
    const bool one_pass = count <= PLAN_THREADS;
    qgtc_batch mine{};
    if (one_pass && tid < count) mine = batches[tid];
    const uint32_t *handed[QGTC_MAX_STAGES];
#pragma unroll
    for (int j = 0; j < QGTC_MAX_STAGES; j++) handed[j] = nullptr;
    for (int s = 0; s < pa.n_stages; s++) {
        const qgtc_stage st = sst[s];
        for (int b0 = 0; b0 < count; b0 += PLAN_THREADS) {
            const int b = b0 + tid;
            qgtc_batch bt = mine;
            if (!one_pass && b < count) bt = batches[b];
            const bool n_ok = bt.n > 0;   // (the table is device memory: a batch with no nodes gets M = 0 descriptors and a record)
            const unsigned long long words = (b < count && n_ok) ? stage_out_words(st, bt.n) : 0ull;
            unsigned long long incl = words;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {   // inclusive scan inside the wave
                const unsigned long long v = __shfl_up(incl, o);
                if (lane >= o) incl += v;
            }
            if (lane == 63) wave_total[wv] = incl;
            __syncthreads();
            unsigned long long before = 0ull, all = 0ull;
#pragma unroll
            for (int w = 0; w < PLAN_THREADS / 64; w++) {
                const unsigned long long t = wave_total[w];
                all += t;
                if (w < wv) before += t;
            }
            unsigned long long off = carry + before + incl - words;
            if (b < count) {
                bool fits = n_ok;
                if (!n_ok) atomicMin(record, b * 8 + QGTC_VIOL_M);
                if (off + words > pool_words) {   // a pool smaller than qgtc_epoch_pool_layout says: recorded; the descriptor gets M = 0 (no
                    atomicMin(record, b * 8 + QGTC_VIOL_POINTER);   // kernel touches it) and a pointer inside the pool, 16-byte aligned
                    fits = false;
                    off = words <= pool_words ? ((pool_words - words) & ~3ull) : 0ull;
                }
                // (everything below in scalars: a struct handed to a lambda by reference, or selected whole by a ternary, lives in scratch)
                const uint32_t *op_ptr[2];
                unsigned long long op_words[2];
#pragma unroll
                for (int side = 0; side < 2; side++) {
                    const int src = side == 0 ? st.left : st.right;
                    const uint32_t *ptr = nullptr;
                    unsigned long long w = 0ull;
                    if (src >= QGTC_SRC_STAGE) {   // the output of an earlier stage of the SAME batch: this thread wrote that descriptor
                        const int j = src - QGTC_SRC_STAGE;
                        if (one_pass) {
#pragma unroll
                            for (int jj = 0; jj < QGTC_MAX_STAGES; jj++)
                                if (jj == j) ptr = handed[jj];
                        } else {
                            ptr = static_cast<const uint32_t *>(descs[static_cast<size_t>(j) * count + b].out);
                        }
                        w = stage_out_words(sst[j], bt.n);
                    } else if (src >= QGTC_SRC_WEIGHT) {
                        ptr = swt[src - QGTC_SRC_WEIGHT].ptr;
                        w = swt[src - QGTC_SRC_WEIGHT].words;
                    } else {
                        ptr = src == QGTC_SRC_A ? bt.A.ptr : (src == QGTC_SRC_X ? bt.X.ptr : (src == QGTC_SRC_XR ? bt.XR.ptr : (src == QGTC_SRC_XC ? bt.XC.ptr : bt.AT.ptr)));
                        w = src == QGTC_SRC_A ? bt.A.words : (src == QGTC_SRC_X ? bt.X.words : (src == QGTC_SRC_XR ? bt.XR.words : (src == QGTC_SRC_XC ? bt.XC.words : bt.AT.words)));
                    }
                    op_ptr[side] = ptr;
                    op_words[side] = w;
                }
                qgtc_problem pr{};
                pr.X = op_ptr[0];
                pr.W = op_ptr[1];
                pr.x_words = op_words[0];
                pr.w_words = op_words[1];
                pr.out = pool + off;
                pr.M = fits ? bt.n : 0;
                pr.K = st.K == QGTC_DIM_NODES ? bt.n : st.K;
                pr.N = st.N;
                pr.w_lines = (st.mode == 2 && !st.pad128) ? pad8(st.N) : pad128(st.N);
                const bool occ = st.use_occ && bt.occ != nullptr;
                pr.occ = occ ? bt.occ : nullptr;
                pr.occ_words = occ ? bt.occ_words : 0;
                descs[static_cast<size_t>(s) * count + b] = pr;
#pragma unroll
                for (int jj = 0; jj < QGTC_MAX_STAGES; jj++)
                    if (jj == s) handed[jj] = static_cast<const uint32_t *>(pr.out);
            }
            carry += all;
            __syncthreads();   // (wave_total is rewritten by the next pass)
        }
    }
}

// ------------------------------------------------------------------------------------------
// QGTC_CHECK_DESCRIPTORS. kind 0: one stage; 1: layer (stage 2's W is stage 1's output); 2: chain (stage 2's X is stage
// 1's output, K2 = N1); 3: one stage whose `out` is not used; 4: the pair of qgtc_chain_aggregate with a second product - p1 = the
// aggregation (its `out` unused), p2 = the transform whose OUTPUT is all the entry uses (non-NULL, 16-byte aligned: the kernels
// store 16 bytes a lane; M = p1's M). exact_N1 / exact_N2 != 0: the chain entries size their stores from the HOST's N, so the
// descriptors' N must EQUAL it; exact_K1 != 0 (qgtc_chain_transform: the weight tables were expanded for the host's K) likewise for
// stage 1's K. The first violation (lowest problem index, then lowest field code) wins: record = problem * 8 +
// field, kept with atomicMin (INT_MAX = none).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_check_descriptors(const qgtc_problem *__restrict__ p1, const qgtc_problem *__restrict__ p2, int count,
                                                           int max_M, int max_K1, int max_N1, int max_K2, int max_N2, int kind,
                                                           int exact_N1, int exact_N2, int exact_K1, int *__restrict__ record) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x) {
        int field = QGTC_VIOL_NONE;
        auto one = [&](const qgtc_problem &p, int mk, int mn, int exact_n, bool out_used, int exact_k = 0) {
            if (field) return;
            if (p.M <= 0 || p.M > max_M) field = QGTC_VIOL_M;
            else if (p.K <= 0 || p.K > mk || (exact_k && p.K != exact_k)) field = QGTC_VIOL_K;
            else if (p.N <= 0 || p.N > mn || (exact_n && p.N != exact_n)) field = QGTC_VIOL_N;
            else if (!p.X || !p.W || (reinterpret_cast<uintptr_t>(p.X) & 15u) || (reinterpret_cast<uintptr_t>(p.W) & 15u) ||
                     (out_used && (!p.out || (reinterpret_cast<uintptr_t>(p.out) & 15u))) || p.x_words >= (1ull << 30) || p.w_words >= (1ull << 30))
                field = QGTC_VIOL_POINTER;
        };
        const qgtc_problem a = p1[i];
        one(a, max_K1, max_N1, exact_N1, kind != 3 && kind != 4, exact_K1);
        if (p2 && kind == 4) {
            const qgtc_problem b = p2[i];
            if (!field) {
                if (b.M != a.M) field = QGTC_VIOL_CHAINING;
                else if (b.N <= 0 || b.N > max_N2 || (exact_N2 && b.N != exact_N2)) field = QGTC_VIOL_N;
                else if (!b.out || (reinterpret_cast<uintptr_t>(b.out) & 15u)) field = QGTC_VIOL_POINTER;
            }
        } else if (p2) {
            const qgtc_problem b = p2[i];
            one(b, max_K2, max_N2, exact_N2, true);
            if (!field) {
                if (kind == 1 && (b.W != static_cast<const uint32_t *>(a.out) || a.M != b.M || a.N != b.N)) field = QGTC_VIOL_CHAINING;
                if (kind == 2 && (b.X != static_cast<const uint32_t *>(a.out) || a.M != b.M || a.N != b.K)) field = QGTC_VIOL_CHAINING;
            }
        }
        if (field) atomicMin(record, i * 8 + field);
    }
}

}  // namespace
