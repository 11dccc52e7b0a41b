/*
 * qgtc.h — C-ABI of libqgtc_hip.so, the MI355X (gfx950) implementation of the QGTC bit-GEMM
 * hot path. This is the drop-in boundary: plain pointers and sizes, no torch types. The
 * `QGTC` PyTorch extension (qgtc_ppopp22_amd/csrc/qgtc_torch.cpp) is a thin binding over
 * these entry points; INTEGRATION.md shows the binding a reference maintainer would add.
 *
 * Each entry point names the reference interface it replaces (paths relative to the
 * reference checkout, YukeWang96/QGTC_PPoPP22 @ v1).
 *
 * Conventions
 *   - all pointers are DEVICE pointers on the device that is current when the call is made;
 *   - `stream` is a hipStream_t passed as void* (NULL = the legacy default stream); every
 *     call is asynchronous on that stream unless stated otherwise;
 *   - packed tensors are flat arrays of 32-bit words; element i of a packed line lives in
 *     word i>>5, bit 31-(i&31) (reference kernel.h:98,234);
 *   - "rows layout"  of an HxW matrix with b planes: [b][PAD8(H)][STEP128(W)*4] words
 *     (QGTC_device.cu:115);  "cols layout": [b][PAD128(W)][STEP128(H)*4] words
 *     (QGTC_device.cu:97), or [b][PAD8(W)][STEP128(H)*4] with output_layer (QGTC_device.cu:83);
 *   - every packed pointer must be 16-byte aligned (torch allocations are);
 *   - reads are bounds-safe: a word index >= the stated *_words reads as 0, so mis-sized or
 *     mis-laid operands (the reference reads raw memory there) can never fault;
 *   - bit widths (nbits, bit1, bit2, output_bit) must lie in [1, 32]; the reference publishes
 *     results for 1..8 only;
 *   - return value: QGTC_OK or a QGTC_E* code; qgtc_strerror() describes it. The reference
 *     printf()s and exit(-1)s instead (QGTC_device.cu:67-71).
 */
#ifndef QGTC_H
#define QGTC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QGTC_ABI_VERSION 11

enum {
    QGTC_OK = 0,
    QGTC_EINVAL = 1,   /* bad dimension / bit width / NULL pointer            */
    QGTC_ESIZE = 2,    /* output buffer smaller than the op's result          */
    QGTC_EALIGN = 3,   /* packed pointer not 16-byte aligned                  */
    QGTC_EHIP = 4,     /* HIP runtime error (see qgtc_last_hip_error())       */
    QGTC_ENODEVICE = 5 /* no gfx950 device / kernels not loadable             */
};

/* flags for qgtc_bitmm2bit */
#define QGTC_OUT_COLS 0x1u     /* pack the result in the cols layout (bitMM2Bit_col)      */
#define QGTC_NO_ZERO_SKIP 0x2u /* do not skip all-zero X tiles (result is identical)      */
#define QGTC_ZERO_JUMP 0x4u    /* qgtc_bitmm_batched: the problems carry occupancy bitmaps */
#define QGTC_ENGINE_MFMA 0x8u  /* qgtc_bitmm2bit / qgtc_bitmm2int / qgtc_bitmm_batched: expand the bit planes to
                                  int8 values and multiply on the matrix cores (bit1, bit2 <= 8; otherwise
                                  ignored). Same results; pays for wide N and several planes, not for N = 64.
                                  Grouped launches whose problems carry a one-word occupancy bitmap
                                  (K <= 8192) jump all-zero 128-row x 128-bit tiles */
#define QGTC_ENGINE_AUTO 0x10u /* let rules fitted to MI355X measurements choose between the two engines */
#define QGTC_CHAIN_DISCARD 0x40u    /* qgtc_gcn_chain_batched: the caller does not need stage_a's output itself (a hint: it is written anyway) */
/* (0x80, 0x100: flags of ABI <= 9 for a private T format between the one-launch chained pairs of rounds 2-3; that kernel family is gone -
 * qgtc_chain_transform / qgtc_chain_aggregate are the chained form) */
#define QGTC_CHAIN_ADJ_TILES 0x400u /* qgtc_chain_aggregate: the adjacencies (stage_a[b].X) are in the tile format of qgtc_adj_tiles_from_rows */
#define QGTC_CHECK_DESCRIPTORS 0x200u /* grouped entry points: a small kernel (one thread per descriptor) ahead of the product compares every DEVICE
                                  descriptor with the stated max_M / max_K / max_N (and the chaining rules of the two-stage
                                  entries) and records the first violation on the device; qgtc_last_batched_violation() reads it */

int qgtc_abi_version(void);
const char *qgtc_strerror(int code);
const char *qgtc_last_hip_error(void);

/* Shape algebra — utility.h:33-45 (STEP8/STEP128/PAD8/PAD128) and the allocation rules of
 * QGTC_device.cu:83,97,115,223,456. */
size_t qgtc_rows_words(int H, int W, int nbits);
size_t qgtc_cols_words(int H, int W, int nbits, int output_layer);

/* val2bit — replaces val2bit_cuda (QGTC_device.cu:44-130; binding QGTC_host.cpp:229-238):
 * Quantize_val (kernel.h:49-71) fused with QGTC_layer_input (kernel.h:204-242, rows layout)
 * or PackFcWeight128 (kernel.h:75-106, cols layout). x: float32 [H,W] row-major.
 * Writes every word of `out` (padding included); out_words must be >= qgtc_rows_words /
 * qgtc_cols_words. */
int qgtc_val2bit(const float *x, int H, int W, int nbits, int col_major, int output_layer,
                 uint32_t *out, size_t out_words, void *stream);

/* bit2val — replaces bit2val_cuda (QGTC_device.cu:135-206; QGTC_host.cpp:244-256):
 * UnPackFcOutput128 (kernel.h:173-201) / UnPackFcWeight128 (kernel.h:109-139).
 * out: int32 [H,W] row-major, fully written. */
int qgtc_bit2val(const uint32_t *bits, size_t bits_words, int nbits, int H, int W,
                 int col_major, int output_layer, int32_t *out, void *stream);

/* bitMM2Bit / bitMM2Bit_col — replaces bitMM2Bit_cuda (QGTC_device.cu:211-266) and
 * bitMM2Bit_col_cuda (QGTC_device.cu:441-489), i.e. kernels QGTC_layer_hidden
 * (kernel.h:245-391) and QGTC_layer_hidden_col (kernel.h:651-810):
 *   C = sum_{pa<bit1,pw<bit2} 2^(pa+pw) * popc-product(X plane pa, W plane pw)   (int32)
 *   out = pack(requant(C, output_bit))  in the rows layout, or the cols layout with
 *   QGTC_OUT_COLS.  X: rows layout of MxK with bit1 planes; W: cols layout of KxN with bit2
 *   planes (plane stride STEP128(K)*PAD128(N)*4). Writes every word of `out`. */
int qgtc_bitmm2bit(const uint32_t *X, size_t x_words, const uint32_t *W, size_t w_words,
                   int M, int K, int N, int bit1, int bit2, int output_bit,
                   uint32_t *out, size_t out_words, unsigned flags, void *stream);

/* bitMM2Int — replaces bitMM2Int_cuda (QGTC_device.cu:495-542): QGTC_layer_output_PAD8
 * (kernel.h:816-932; W plane stride STEP128(K)*PAD8(N)*4) when pad_128 == 0, else
 * QGTC_layer_output_PAD128 (kernel.h:938-1054). out: float32 [M,N] = (float)C. */
int qgtc_bitmm2int(const uint32_t *X, size_t x_words, const uint32_t *W, size_t w_words,
                   int M, int K, int N, int bit1, int bit2, int pad_128,
                   float *out, size_t out_elems, unsigned flags, void *stream);

/* bitMM2Bit_profile — replaces bitMM2Bit_cuda_profile (QGTC_device.cu:379-434): `reps`
 * back-to-back launches between two events; BLOCKS until they finish and returns the elapsed
 * milliseconds in *elapsed_ms (the reference hard-codes reps = 200 and printf()s TFLOPs). */
int qgtc_bitmm2bit_profile(const uint32_t *X, size_t x_words, const uint32_t *W, size_t w_words,
                           int M, int K, int N, int bit1, int bit2, int output_bit,
                           uint32_t *out, size_t out_words, unsigned flags, int reps,
                           float *elapsed_ms, void *stream);

/* Tile counters — replaces the `counter_global` / `counter` device globals that
 * QGTC_layer_hidden_base_cnt (kernel.h:394-512, :452) and QGTC_layer_hidden_zerojump_cnt
 * (kernel.h:516-648, :574-592) bump; per call, in the reference's 8-row x 128-bit tile units:
 *   counters[0] = STEP8(M)*STEP8(N)*STEP128(K)*bit1*bit2
 *   counters[1] = steps whose 8x128-bit X tile is non-zero (x STEP8(N)*bit2)
 * `counters` is a device pointer to two uint64; it is overwritten (not accumulated). */
int qgtc_tile_counters(const uint32_t *X, size_t x_words, int M, int K, int N, int bit1,
                       int bit2, uint64_t *counters, void *stream);

/* Batched bit-GEMM: `count` independent products in ONE launch (the Cluster-GCN / Batched-GIN
 * epoch loop of main_qgtc.py:112-155 issues one bitMM2Bit per cluster batch; on MI355X the
 * launch boundary dominates those, so the engine groups them). `problems` is a DEVICE array.
 * mode: 0 = rows-layout bits, 1 = cols-layout bits, 2 = float32 (bitMM2Int). */
typedef struct qgtc_problem {
    const uint32_t *X;
    const uint32_t *W;
    void *out;
    uint64_t x_words, w_words;
    int32_t M, K, N;
    int32_t w_lines; /* lines per W plane: PAD128(N), or PAD8(N) for bitMM2Int pad_128=0 */
    int32_t occ_words; /* 64-bit words per row tile of `occ`: ceil(STEP128(K) / 64) */
    const uint64_t *occ; /* optional (NULL = visit every k-quad): occupancy bitmap of X from
                            qgtc_tile_occupancy; zero X tiles are then neither loaded nor multiplied */
} qgtc_problem;

/* Occupancy bitmap of a rows-layout left operand, for zero-tile JUMPING in qgtc_bitmm_batched (the
 * reference only counts what jumping would save: QGTC_layer_hidden_zerojump_cnt, kernel.h:516-648).
 * Bit q of word [row_tile][q / 64] is set when the 32-row x 128-bit tile (rows 32*row_tile..+31,
 * k-quad q) has a set bit in any of the bit1 planes. qgtc_occupancy_words(M, K) 64-bit words.
 * Products are bit-identical with and without the bitmap. */
size_t qgtc_occupancy_words(int M, int K);
int qgtc_tile_occupancy(const uint32_t *X, size_t x_words, int M, int K, int bit1, uint64_t *occ,
                        size_t occ_words, void *stream);

/* The bitmaps of all problems of a grouped launch in ONE launch: every problem with a non-NULL `occ`
 * gets its bitmap written where `occ` points (occ_words must be ceil(STEP128(K) / 64)). With `stats`
 * (device pointer to two uint64) a second, one-workgroup kernel counts stats[0] = occupied and
 * stats[1] = all 32-row x 128-bit tiles and, when more than `max_fraction` of them are occupied, clears
 * the `occ` fields of the DEVICE descriptors (jumping then only costs a dependent load): the decision
 * needs no host round trip. Pass QGTC_ZERO_JUMP to qgtc_bitmm_batched either way. */
int qgtc_tile_occupancy_batched(qgtc_problem *problems, int count, int max_M, int max_K, int bit1,
                                float max_fraction, uint64_t *stats, void *stream);

/* Only the counting / clearing step, for descriptors whose bitmaps already exist (the adjacency bitmaps of
 * an epoch are shared by all its A.(...) stages). */
int qgtc_tile_occupancy_decide(qgtc_problem *problems, int count, float max_fraction, uint64_t *stats,
                               void *stream);

/* PRECONDITIONS the library cannot check (the descriptors live in device memory): max_M / max_K / max_N are at
 * least every problem's M / K / N - the grid, the split-K plan and the choice between the float32 (FP4) and int32
 * kernels are derived from them, so a larger problem than stated gets unwritten tiles or inexact sums; `occ` is
 * NULL or a valid bitmap from qgtc_tile_occupancy* for the SAME X (the matrix-core kernels follow a non-NULL `occ`
 * with or without QGTC_ZERO_JUMP). The PyTorch binding computes the maxima itself (BatchedGemm). */
int qgtc_bitmm_batched(const qgtc_problem *problems, int count, int max_M, int max_K, int max_N,
                       int bit1, int bit2, int output_bit, int mode, unsigned flags,
                       void *stream);

/* One quantised GNN layer for `count` cluster batches in ONE call - the grouped form of the reference's per-layer
 * pair (QGTC_conv.py:14-22: X.W, then A.(XW); main_qgtc.py:147-154 issues it as two extension calls per batch):
 *   stage 1   T_b   = bitMM2Bit_col(X_b, W, x_bits, w_bits, t_bits)        (cols-layout bits, written to stage1[b].out)
 *   stage 2   out_b = bitMM2Bit(A_b, T_b, a_bits, t_bits, output_bit)      mode 0: rows-layout bits
 *                   = bitMM2Int(A_b, T_b, a_bits, t_bits, pad_128 = 1)     mode 2: float32
 * stage1[b] = {X_b, W, T_b, .., M = n_b, K = f_in, N = f_out}, stage2[b] = {A_b, T_b, out_b, .., M = n_b, K = n_b,
 * N = f_out} (stage2[b].W must be stage1[b].out; w_lines = PAD128(N) in both). Runs as two grouped launches on `stream`
 * (word for word qgtc_bitmm_batched(stage1, mode 1) then qgtc_bitmm_batched(stage2, mode)); QGTC_ZERO_JUMP applies to
 * stage 2. max_* are hard preconditions as for qgtc_bitmm_batched. (Up to ABI 8 the entry also had a one-launch form
 * with in-kernel arrival counters; it measured slower and was removed - DESIGN.md appendix.) */
int qgtc_gcn_layer_batched(const qgtc_problem *stage1, const qgtc_problem *stage2, int count, int max_M, int max_K1,
                           int max_K2, int max_N, int x_bits, int w_bits, int t_bits, int a_bits, int output_bit,
                           int mode, unsigned flags, void *stream);

/* An aggregation stage and the NEXT layer's feature-transform stage in one call (main_qgtc.py:148-153: t1 = MM2Bit(bA, t0),
 * t2 = MM2Bit(t1, bW2), as the layout-correct chain issues them: the second with a cols-layout output). For every cluster
 * batch i: stage_a[i] is  out_i = requant(A_i . T_i)  (rows-layout bits, act_bits planes; A_i rows layout with a_bits
 * planes, T_i cols layout with t_bits planes), stage_xw[i] is  T'_i = requant(out_i . W')  (out_mode 1: cols-layout bits,
 * out_bits planes) or the output layer  float32(out_i . W')  (out_mode 2: [M, N'] floats, out_bits ignored; w_lines of the
 * descriptor as for qgtc_bitmm2int); W' cols layout with w_bits planes: stage_xw[i].X must be stage_a[i].out,
 * stage_xw[i].K = stage_a[i].N, stage_xw[i].M = stage_a[i].M. Word for word the result of qgtc_bitmm_batched(stage_a,
 * mode 0) followed by qgtc_bitmm_batched(stage_xw, mode out_mode), which is what it issues (ABI <= 9 had a one-launch kernel for some
 * shapes; the epoch's chained form is qgtc_chain_transform / qgtc_chain_aggregate below). QGTC_ZERO_JUMP applies to stage_a (its .occ
 * bitmaps).
 * max_* are hard preconditions as for qgtc_bitmm_batched. */
int qgtc_gcn_chain_batched(const qgtc_problem *stage_a, const qgtc_problem *stage_xw, int count, int max_M, int max_K,
                           int max_N1, int max_N2, int a_bits, int t_bits, int act_bits, int w_bits, int out_bits,
                           int out_mode, unsigned flags, void *stream);

/* Which kernel family a call takes - host only, no device work: the SAME rule functions the launchers use (qgtc_hip.hip:
 * single_route / batched_route over the predicates of launch_common.hip.h), so documentation and tests can name the kernel
 * behind a call shape (DESIGN.md section 5 is generated from these by tools/routing_table.py). mode 0 rows-layout bits, 1
 * cols-layout bits, 2 float32; flags = the engine / zero-jump flags of the launch. Returns a static string: a kernel
 * family name ("k_bitmm", "k_bitmm_fp4_one", ...) or "invalid". */
const char *qgtc_bitmm_route(int M, int K, int N, int bit1, int bit2, int output_bit, int mode, unsigned flags);
const char *qgtc_bitmm_batched_route(int max_M, int max_K, int max_N, int bit1, int bit2, int output_bit, int mode, unsigned flags);

/* Adjacency bit planes from an edge list — replaces the dense detour of sampler.py:80-101
 * (torch.sparse.FloatTensor(...).to_dense() then QGTC.val2bit(A, nbits, False, False)): the n x n
 * float matrix (5.9 MB for a 1213-node batch) is never materialised. `cells[i]` = row * W + col of
 * a DISTINCT non-zero cell (negative = skip), `counts[i]` its multiplicity (NULL = all 1); the
 * value is quantised like Quantize_val (kernel.h:39-44,49-71) and packed in the rows layout.
 * Result is word-for-word what qgtc_val2bit(dense A, rows layout) produces. */
int qgtc_pack_edges(const int64_t *cells, const int32_t *counts, size_t n_cells, int H, int W,
                    int nbits, uint32_t *out, size_t out_words, void *stream);

/* The 1-bit case of the above from the RAW edge list: src[i], dst[i] may repeat, nothing is sorted or
 * made unique first (three bitmaps count multiplicities 1, 2, >= 3 with atomic ORs; plane 0 = once or
 * three-times-and-more, the 1-bit quantiser's image of the summed matrix). `scratch`: 2 x
 * qgtc_rows_words(H, W, 1) words. Out-of-range indices are skipped and, if `bad_index` (device int) is
 * given, reported there (1) - no host round trip unless the caller wants one. */
int qgtc_pack_edge_list(const int64_t *src, const int64_t *dst, size_t n_edges, int H, int W, uint32_t *out,
                        size_t out_words, uint32_t *scratch, size_t scratch_words, int *bad_index, void *stream);

/* int8 MFMA GEMM, the comparison path — MI355X analogue of the reference's cuBLAS INT8
 * benchmark (cuBLASGemmEX/cublas_main.cu:123-172: cublasGemmEx, CUDA_R_8I in, CUDA_R_32F out).
 * C[M,N] (float32, row-major) = A[M,K] x B[K,N]; A is int8 row-major, B is passed as Bt[N,K]
 * (int8, K contiguous); int32 accumulation (v_mfma_i32_16x16x64_i8), exact. K % 16 == 0.
 * qgtc_i8gemm_profile times `reps` launches between two events like cublas_main.cu:123-172. */
int qgtc_i8gemm(const int8_t *A, const int8_t *Bt, int M, int K, int N, float *C, size_t c_elems,
                void *stream);
int qgtc_i8gemm_profile(const int8_t *A, const int8_t *Bt, int M, int K, int N, float *C,
                        size_t c_elems, int reps, float *elapsed_ms, void *stream);

/* ---- Detecting (not just documenting) the grouped entries' preconditions ------------------------------------------
 * The descriptors of qgtc_bitmm_batched / qgtc_gcn_layer_batched / qgtc_gcn_chain_batched live in device memory, so the
 * host side cannot compare them with the stated maxima (from which the grid, the split-K plan and the choice between the
 * float32 and int32 kernels are derived). With QGTC_CHECK_DESCRIPTORS in `flags` those entries first launch a
 * small kernel (one thread per descriptor) on `stream` that checks every descriptor: M <= max_M, K <= max_K, N <= max_N, all
 * positive, non-NULL 16-byte aligned operands AND outputs (the kernels store 16 bytes a lane), for the two-stage entries that
 * stage 2 really reads stage 1's output, and for the chain entries (qgtc_chain_transform / qgtc_chain_aggregate, whose stores
 * are sized from the HOST's N / N2) that every descriptor's N EQUALS the stated width - for qgtc_chain_aggregate's second
 * product the stage_xw descriptors are checked too (output pointer, N = N2, M = stage_a's M). The
 * product still runs (its results for an offending problem are unspecified, exactly as without the flag); the first
 * violation is kept in a per-device record until it is read. qgtc_epoch_plan_fill records there as well: a pool smaller
 * than qgtc_epoch_pool_layout's figure (QGTC_VIOL_POINTER) or a batch with n <= 0 (QGTC_VIOL_M) - such descriptors get M = 0,
 * which every grouped kernel skips.
 * qgtc_last_batched_violation(): waits for `stream`, returns QGTC_OK when no checked launch since the last call found a
 * violation, else QGTC_EINVAL with *problem = index of the first offending descriptor and *field = one of QGTC_VIOL_*;
 * the record is cleared. `problem` / `field` may be NULL. */
enum { QGTC_VIOL_NONE = 0, QGTC_VIOL_M = 1, QGTC_VIOL_K = 2, QGTC_VIOL_N = 3, QGTC_VIOL_POINTER = 4, QGTC_VIOL_CHAINING = 5 };
int qgtc_last_batched_violation(int *problem, int *field, void *stream);

/* val2bit of several matrices in ONE launch (the three weight matrices an epoch packs inside its clock,
 * main_qgtc.py:100-110: `QGTC.val2bit(W1.cuda(), w_bit, True, False)` x 3). `jobs` is a HOST array of at most
 * QGTC_MAX_PACK_JOBS entries; each job is exactly one qgtc_val2bit call (same words, every padding word written). */
#define QGTC_MAX_PACK_JOBS 8
typedef struct qgtc_pack_job {
    const float *x;      /* float32 [H, W] row-major, device */
    uint32_t *out;       /* packed result, device */
    uint64_t out_words;  /* capacity of `out` */
    int32_t H, W, nbits, col_major, output_layer;
    int32_t reserved;
} qgtc_pack_job;
int qgtc_val2bit_batched(const qgtc_pack_job *jobs, int n_jobs, void *stream);

/* ---- Epoch plans: the descriptors of every grouped launch of an epoch, filled ON THE DEVICE by one kernel -------------
 * The reference's epoch clock (main_qgtc.py:96-159) starts before the weights are packed and covers every per-batch
 * output allocation and launch. A grouped epoch needs, per stage, one qgtc_problem per cluster batch whose pointers chain
 * the stages' outputs; filling those on the host (75 descriptors x 6 stages, six uploads) cost more than the epoch's
 * kernels. Here the data loader's part - one qgtc_batch per cluster batch: the packed adjacency and features it built
 * (sampler.py:92-105) and the adjacency's occupancy bitmap - is made once beside the packing, and inside the clock ONE
 * launch turns `stages` (what each of the epoch's operators multiplies, main_qgtc.py:131-154) into device descriptors:
 * every stage's outputs are carved out of one pool, in batch order, each 16-byte aligned; stage s's descriptors are
 * descs[s * count .. s * count + count - 1].
 *   left / right: where an operand comes from - QGTC_SRC_A / _X / _XR / _XC of the batch, QGTC_SRC_WEIGHT + k = weights[k] (shared
 *   by all batches), QGTC_SRC_STAGE + j = the output of stage j < s of the same batch.
 *   M is the batch's node count n; K is `K`, or n when K == QGTC_DIM_NODES; N is `N`.
 *   mode / ob / pad128 as qgtc_bitmm_batched / qgtc_bitmm2int; use_occ: the descriptors carry the batch's bitmap.
 * qgtc_epoch_pool_layout (host only, no device work) gives the pool size in 32-bit words for the same arguments and,
 * optionally, every output's offset (offsets[s * count + b], in words) - the fill kernel uses the same rule. */
enum { QGTC_SRC_A = 0, QGTC_SRC_X = 1, QGTC_SRC_XR = 2, QGTC_SRC_XC = 3, QGTC_SRC_AT = 4, QGTC_SRC_WEIGHT = 16, QGTC_SRC_STAGE = 32 };
#define QGTC_DIM_NODES (-1)
#define QGTC_MAX_STAGES 8
#define QGTC_MAX_WEIGHTS 8
typedef struct qgtc_operand {
    const uint32_t *ptr;
    uint64_t words;
} qgtc_operand;
typedef struct qgtc_batch {
    qgtc_operand A;      /* adjacency, rows layout [n, n] */
    qgtc_operand X;      /* features, cols layout [n, F] (a right operand: sampler.py:99) */
    qgtc_operand XR;     /* features, rows layout [n, F] (a left operand), or {NULL, 0} */
    qgtc_operand XC;     /* features in the chain format of qgtc_chain_* (qgtc_chain_from_cols of X), or {NULL, 0} */
    qgtc_operand AT;     /* adjacency in the tile format of qgtc_chain_aggregate (qgtc_adj_tiles_from_rows of A), or {NULL, 0} */
    const uint64_t *occ; /* occupancy bitmap of A (qgtc_tile_occupancy) or NULL */
    int32_t n;           /* nodes of the batch */
    int32_t occ_words;   /* 64-bit words per row tile of `occ` */
} qgtc_batch;
typedef struct qgtc_stage {
    int32_t left, right; /* QGTC_SRC_* */
    int32_t K, N;        /* K: a number or QGTC_DIM_NODES */
    int32_t bit1, bit2, ob;
    int32_t mode;        /* 0 rows-layout bits, 1 cols-layout bits, 2 float32 */
    int32_t pad128;      /* mode 2: the right operand's planes have PAD128(N) lines (else PAD8(N)) */
    int32_t use_occ;     /* carry the batch's occupancy bitmap (left must be QGTC_SRC_A or QGTC_SRC_AT) */
    int32_t fmt;         /* mode 1 only: 0 = the cols layout, 1 = the chain format of qgtc_chain_* (qgtc_chain_words(n, N, ob) words) */
} qgtc_stage;
size_t qgtc_epoch_pool_layout(const int32_t *nodes, int count, const qgtc_stage *stages, int n_stages, uint64_t *offsets);
int qgtc_epoch_plan_fill(const qgtc_batch *batches, int count, const qgtc_stage *stages, int n_stages,
                         const qgtc_operand *weights, int n_weights, void *pool, size_t pool_words,
                         qgtc_problem *descs, void *stream);

/* ---- The chain entries: one wave per 32-row block for the whole output width (the grouped epochs at 1 .. 4 bits) ------------
 * Between the launches of a layout-correct epoch (X.W1 | A.T1 + .W2 | A.T2 + .W3 | A.T3, main_qgtc.py:147-154 with every
 * right operand in the cols layout) T is written by one launch and read by the next and by nobody else. These entries keep
 * it in a private CHAIN FORMAT - the finished matrix-core operand, qgtc_chain_words(M, N, bits) words per batch, unspecified to
 * the caller - and take the weights PRE-EXPANDED (qgtc_expand_weights, once per plan; qgtc_weight_codes_words(K, N, nbits,
 * order) words each, stated to the entry as the job's `codes_words`: a table too small for the job is QGTC_ESIZE, never a write). Word for word (after decoding) the results of the public entries; only the last call's float32 output is public.
 *   qgtc_chain_transform:  T_b = requant(X_b . W)                stage[b] = {X_b rows layout (x_bits planes, K <= 8192), -, T_b}
 *   qgtc_chain_aggregate:  out_mode 0: out_b = float32(A_b . T_b)                      stage_a[b] = {A_b, T_b, out_b}; stage_xw = NULL
 *                          out_mode 1: T'_b  = requant(requant(A_b . T_b) . W')        stage_a[b] = {A_b, T_b, -}, stage_xw[b] = {-, -, T'_b}
 *                          out_mode 2: out_b = float32(requant(A_b . T_b) . W')        stage_xw[b] = {-, -, out_b [M, N2]}
 * A_b: rows layout (or, with QGTC_CHAIN_ADJ_TILES, the tile format of qgtc_adj_tiles_from_rows), ONE plane, K <= 8192
 * (occupancy bitmaps of the descriptors are followed); t_bits / act_bits /
 * out_bits = bits of T / of the aggregate / of T': 1 .. 4, act_bits == out_bits (= the planes of the weights: a chain has one
 * width, main_qgtc.py's --bit_width), t_bits in the same format class (1 / 2 bits: one base-4 digit a nibble; 3 / 4 bits: two),
 * N, N2 <= 128; qgtc_chain_transform: out_bits 1 .. 4, x_bits <= 2 (out_bits <= 2) or <= 4 (out_bits 3 / 4).
 * ABI 11 widens both entries (bitmm_fp4_rbx.hip.h): one width of 5 .. 8 bits per chain (x_bits <= 8; N, N2 <= 128; the X . W product's
 * float32 sums must stay exact: K (2^x_bits - 1)(2^out_bits - 1) < 2^24, i.e. K <= 258 at 8 x 8 bits), or 1 .. 4 bits with up to 256
 * columns on either side (--n-hidden up to 256); qgtc_expand_weights accordingly nbits <= 8, N <= 256, K <= 256 for order 1.
 * QGTC_EINVAL outside that range: callers fall back to qgtc_gcn_chain_batched. w_codes: qgtc_expand_weights order 0 for
 * qgtc_chain_transform (the left operand arrives as packed words), order 1 for qgtc_chain_aggregate (the left operand is
 * the aggregate in the registers of the wave that computed it). max_M is a hard precondition (QGTC_CHECK_DESCRIPTORS).
 * qgtc_chain_transform's K is the K the weights were expanded for and EVERY descriptor's K must equal it: the kernel takes its
 * k-quad count and the stride of the weight tables from this argument, never from a descriptor (a larger descriptor K cannot
 * walk past the tables; QGTC_CHECK_DESCRIPTORS reports the mismatch). */
typedef struct qgtc_expand_job {
    const uint32_t *W;   /* cols layout [K, N], nbits planes of w_lines lines */
    uint32_t *codes;     /* out: qgtc_weight_codes_words(K, N, nbits, order) words (order 0: a table per k-quad of K) */
    uint64_t w_words;
    int32_t K, N, nbits, w_lines, order;
    uint32_t codes_words; /* capacity of `codes` in 32-bit words (ABI 11; was `reserved`): checked against the line above */
} qgtc_expand_job;
size_t qgtc_weight_codes_words(int K, int N, int nbits, int order);   /* (ABI 10 took (N, nbits) and left the per-k-quad factor to the caller) */
size_t qgtc_chain_words(int M, int N, int bits);   /* (ABI 11: 5 .. 8-bit values take two arrays of the 4-bit form) */
/* A cols-layout right operand (the public format: X of sampler.py:99, [H, W] with nbits <= 4 planes) in the chain format:
 * what a data loader does once beside the packing when the epoch's FIRST product is an aggregation (Batched-GIN: A . X,
 * main_qgtc.py:131). chain: qgtc_chain_words(H, W, nbits) words; nbits <= 8. */
int qgtc_chain_from_cols(const uint32_t *cols, size_t cols_words, int H, int W, int nbits, uint32_t *chain, size_t chain_words,
                         void *stream);
int qgtc_expand_weights(const qgtc_expand_job *jobs, int n_jobs, void *stream);
/* A one-plane rows-layout adjacency (the public format: bit_A of sampler.py:98, [M, K]) as 512-byte tiles
 * [32-row block][k-quad][32 rows][4 words], qgtc_adj_tiles_words(M, K) words: what a data loader makes once beside the
 * packing for qgtc_chain_aggregate(QGTC_CHAIN_ADJ_TILES). In the rows layout the 32 rows of a tile are a whole row apart, so a
 * launch that reads only the OCCUPIED tiles still pulls every cache line of A; as tiles it reads what it uses. Rows past M are
 * zero. The occupancy bitmaps (qgtc_tile_occupancy of the rows layout) describe both. */
size_t qgtc_adj_tiles_words(int M, int K);
int qgtc_adj_tiles_from_rows(const uint32_t *rows, size_t rows_words, int M, int K, uint32_t *tiles, size_t tiles_words, void *stream);
int qgtc_chain_transform(const qgtc_problem *stage, int count, int max_M, int K, int N, int x_bits, int out_bits,
                         const uint32_t *w_codes, unsigned flags, void *stream);
int qgtc_chain_aggregate(const qgtc_problem *stage_a, const qgtc_problem *stage_xw, int count, int max_M, int max_K, int N1,
                         int N2, int t_bits, int act_bits, int out_bits, int out_mode, const uint32_t *w2_codes,
                         unsigned flags, void *stream);

/* ---- The data loader's packing for ALL cluster batches of an iterator in a handful of launches -------------------------
 * sampler.py:76-106 packs every batch on its own: the batch's dense float adjacency from its edges, `QGTC.val2bit(A, 1, False,
 * False)`, `QGTC.val2bit(X, bit_width, True, False)`. Here one call packs `count` batches: per batch the 1-bit adjacency in the
 * rows layout straight from its edge list (the words qgtc_pack_edge_list gives: multiplicities 1, 2, >= 3 quantise to 1, 0, 1),
 * the features in the cols layout (the reference's bit_X) and - each optional, NULL = not wanted - the features in the rows
 * layout (left operand of the layout-correct chain's first X.W), the adjacency as 512-byte tiles (qgtc_adj_tiles_from_rows),
 * its occupancy bitmap (qgtc_tile_occupancy) and the features in the chain format (qgtc_chain_from_cols; x_bits <= 4).
 * Every output is word for word what the single-batch entry gives for that batch.
 *   `batches`: DEVICE array of `count` entries (the caller uploads it); edge indices are LOCAL to the batch (row = src,
 *   col = dst in [0, n)), int64, batch b's edges at src/dst[edge_off .. edge_off + n_edges); its features are rows
 *   feat_row .. feat_row + n - 1 of `feats` (float32, F columns, row-major).
 *   `zero` / `zero_bytes`: ONE region the call clears with one memset. It always contains `stats`; on the route WITHOUT a work
 *   buffer it must also contain every A and every scratch buffer (scratch: 2 x qgtc_rows_words(n, n, 1) words per batch - the
 *   multiplicity bitmaps of qgtc_pack_edge_list). With `work` (ABI 11; qgtc_load_work_words(count, max_n, total edges) words, 0 = not
 *   available for this iterator: max_n above 5120) the edges are bucketed by 32-row block there and every word of every A / AT / occ
 *   is written exactly once from LDS: A needs no clearing, `scratch` may be NULL. Same words either way. qgtc_load_work_words decides
 *   the route: a work buffer that cannot serve (max_n above 5120: QGTC_EINVAL; not 8-byte aligned: QGTC_EALIGN; fewer words than the
 *   fixed part of the layout: QGTC_ESIZE) is an error - the call never falls back to the route whose cleared region it was not given.
 *   max_n / max_edges: at least every batch's n / n_edges (grid sizes; hard preconditions like the grouped GEMM's maxima).
 *   stats (optional, inside `zero`): stats[0] += occupied 32-row x 128-bit adjacency tiles of all batches.
 *   bad_index (optional device int, cleared by the call): set to 1 when an edge index is out of range (such edges are skipped).
 *   formats: which of the optional feature formats ANY batch asks for (QGTC_LOAD_X_ROWS, QGTC_LOAD_X_CHAIN): a launch nobody
 *   needs is not made (the table itself lives on the device). */
#define QGTC_LOAD_X_ROWS 0x1u
#define QGTC_LOAD_X_CHAIN 0x2u
typedef struct qgtc_loader_batch {
    uint64_t edge_off, n_edges;
    uint64_t feat_row;
    int32_t n;
    int32_t reserved;
    uint32_t *A;       /* out: rows layout [n, n], one plane, qgtc_rows_words(n, n, 1) words (inside `zero` unless a work buffer is given) */
    uint32_t *scratch; /* 2 x qgtc_rows_words(n, n, 1) words (inside `zero`); NULL with a work buffer */
    uint32_t *AT;      /* out or NULL: qgtc_adj_tiles_words(n, n) words */
    uint64_t *occ;     /* out or NULL: qgtc_occupancy_words(n, n) 64-bit words */
    uint32_t *X;       /* out or NULL: cols layout [n, F], x_bits planes, qgtc_cols_words(n, F, x_bits, 0) words */
    uint32_t *XR;      /* out or NULL: rows layout [n, F], x_bits planes, qgtc_rows_words(n, F, x_bits) words */
    uint32_t *XC;      /* out or NULL: chain format, qgtc_chain_words(n, F, x_bits) words (needs X) */
} qgtc_loader_batch;
int qgtc_load_batches(const qgtc_loader_batch *batches, int count, int max_n, uint64_t max_edges, const int64_t *src,
                      const int64_t *dst, const float *feats, int F, int x_bits, void *zero, size_t zero_bytes,
                      uint64_t *stats, int *bad_index, unsigned formats, uint32_t *work, size_t work_words, void *stream);
size_t qgtc_load_work_words(int count, int max_n, uint64_t total_edges);

#ifdef __cplusplus
}
#endif
#endif
