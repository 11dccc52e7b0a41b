// qgtc_hip.hip — MI355X (gfx950 / CDNA4) kernels and C-ABI of the QGTC bit-GEMM hot path.
//
// What is here (reference file:line each piece replaces is in include/qgtc.h):
//   * fused quantise + bit-plane pack          (val2bit, rows and cols layouts)
//   * bit-plane unpack                         (bit2val)
//   * adjacency bit planes straight from an edge list (pack_edges), occupancy bitmaps
//   * multi-plane 1-bit GEMM: AND + popcount (v_and_b32 / v_bcnt_u32_b32) with shift-accumulate
//     into int32, and the same product on the matrix cores (bit planes expanded to FP4 / int8 codes, exact);
//     in-workgroup split-K, zero-tile skipping / jumping, and a fused epilogue that either re-quantises and
//     re-packs (rows / cols layout) or converts to float32; one entry per GNN layer
//   * tile counters, the 200-rep profile loop, a grouped (batched) launch
//   * the int8 MFMA comparison GEMM.
//
// Layout of the sources (one translation unit; the .hip.h files are included below):
//   common.hip.h             vector types, shape algebra, quantiser, DPP OR
//   pack_kernels.hip.h       val2bit (rows / cols), bit2val, pack_edges
//   i8gemm_kernel.hip.h      int8 MFMA comparison GEMM
//   tile_stats_kernels.hip.h occupancy bitmaps, tile counters
//   loader_kernels.hip.h     the data loader's packing for all cluster batches at once (qgtc_load_batches)
//   bitmm_popcount.hip.h     the bit-GEMM on AND + popcount (engine "popcount", and every plane count the matrix-core
//                            kernels do not cover) - start at the comment above `mm_tile`
//   bitmm_mfma.hip.h         the bit-GEMM on the matrix cores, 128 x 128 tiles (wide right operands)
//   fp4_expand.hip.h         packed words -> E2M1 MFMA operands in place (one AND per dword; digits)
//   bitmm_fp4_one.hip.h      the same for narrow right operands (N <= 256) and K <= 4096: the headline kernel
//   bitmm_fp4_skinny.hip.h   the same for longer K: no LDS staging
//   fp4_rowblock.hip.h       what the row-block kernels share (digit operands, the re-quantise + pack epilogue)
//   bitmm_fp4_rows.hip.h     one workgroup per 32-row block: grouped stages (sparse left operands, narrow or many-plane products), single
//                            launches with three to eight left-hand planes; all three outputs
//   launch_common.hip.h      launch constants, kernel-family predicates (shared with qgtc_mfma.hip / qgtc_fp4.hip)
//   launch.hip.h             split-K plan, kernel selection, launchers
//   qgtc_mfma.hip            second translation unit: the 128 x 128-tile matrix-core engine and its launchers
//   qgtc_fp4.hip             third translation unit: the FP4 narrow-operand / grouped kernels and their launchers
//   qgtc_wide.hip            fourth translation unit: bitmm_fp4_wide.hip.h (wide right operands: LDS-DMA staging) + launch_wide.hip.h
//   qgtc_hip.hip             the C-ABI of include/qgtc.h (this file)
// Design notes live in DESIGN.md.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <utility>

#include "qgtc.h"

#include "common.hip.h"
#include "pack_kernels.hip.h"
#include "i8gemm_kernel.hip.h"
#include "tile_stats_kernels.hip.h"
#include "loader_kernels.hip.h"
#include "bitmm_popcount.hip.h"
#include "bitmm_planes.hip.h"
#include "bitmm_mfma.hip.h"
#include "launch_common.hip.h"
#ifdef QGTC_SINGLE_TU   // tools/kbench.hip: everything in one translation unit
#include "bitmm_fp4_skinny.hip.h"
#include "bitmm_fp4_one.hip.h"
#include "fp4_rowblock.hip.h"
#include "bitmm_fp4_rows.hip.h"
#include "fp4_rbw_common.hip.h"
#include "bitmm_fp4_rbw.hip.h"
#include "bitmm_fp4_rbx.hip.h"
#include "fp4_expand.hip.h"
#include "bitmm_fp4_wide.hip.h"
#include "bitmm_fp4_stream.hip.h"
#include "launch_fp4.hip.h"
#include "launch_chainx.hip.h"
#include "launch_mfma.hip.h"
#include "launch_wide.hip.h"
#include "launch_stream.hip.h"
#endif
#include "launch.hip.h"

thread_local char qgtc_g_hip_err[256] = "";

namespace {
// the two timing events of a profile loop; destroyed on every path out of the scope
struct EventPair {
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t create() {
        hipError_t e = hipEventCreate(&e0);
        return e == hipSuccess ? hipEventCreate(&e1) : e;
    }
    ~EventPair() {
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
    }
};
}  // namespace

// ============================================================================================
// C-ABI
// ============================================================================================
extern "C" {

int qgtc_abi_version(void) { return QGTC_ABI_VERSION; }

const char *qgtc_strerror(int code) {
    switch (code) {
        case QGTC_OK: return "ok";
        case QGTC_EINVAL: return "invalid argument (dimension, bit width or NULL pointer)";
        case QGTC_ESIZE: return "output buffer too small";
        case QGTC_EALIGN: return "packed tensor pointer is not 16-byte aligned";
        case QGTC_EHIP: return "HIP runtime error";
        case QGTC_ENODEVICE: return "no usable gfx950 device";
        default: return "unknown error";
    }
}

const char *qgtc_last_hip_error(void) { return qgtc_g_hip_err; }

size_t qgtc_rows_words(int H, int W, int nbits) {
    return static_cast<size_t>(nbits) * pad8(H) * step128(W) * 4u;
}

size_t qgtc_cols_words(int H, int W, int nbits, int output_layer) {
    return static_cast<size_t>(nbits) * step128(H) * 4u * (output_layer ? pad8(W) : pad128(W));
}

int qgtc_val2bit(const float *x, int H, int W, int nbits, int col_major, int output_layer,
                 uint32_t *out, size_t out_words, void *stream) {
    if (!x || !out || H <= 0 || W <= 0 || !bits_ok(nbits)) return QGTC_EINVAL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const float ub = std::ldexp(1.0f, nbits), ubm1 = ub - 1.0f;
    if (!col_major) {
        if (out_words < qgtc_rows_words(H, W, nbits)) return QGTC_ESIZE;
        const int rows_pad = pad8(H), row_words = step128(W) * 4;
        const size_t units = static_cast<size_t>(rows_pad) * ((row_words + 7) / 8);
        if ((W & 3) == 0 && aligned16(x) && units < (1ull << 30)) {
            // Grid: one block per 4 x UNROLL units up to 8192 blocks, two units in flight per wave while that covers the
            // matrix, four beyond. Measured (tools/pack_exp.py): 4096^2 15.1 us (cap 2048, 4 units) -> 13.2 us = 5.2 TB/s,
            // 8192^2 50.4 -> 45.6 us = 6.1 TB/s: more, shorter waves hide the HBM latency better than deeper unrolling
            if (units <= 8192u * 8u) {
                const unsigned g = grid_for((units + 1) / 2, 4, 8192);
                QGTC_LAUNCH(k_val2bit_rows_v4<2>, dim3(g), dim3(256), 0, st, x, H, W, nbits, ub, ubm1, out, rows_pad, row_words, static_cast<int>(4u * g));
            } else {
                const unsigned g = grid_for((units + 3) / 4, 4, 8192);
                QGTC_LAUNCH(k_val2bit_rows_v4<4>, dim3(g), dim3(256), 0, st, x, H, W, nbits, ub, ubm1, out, rows_pad, row_words, static_cast<int>(4u * g));
            }
        } else {
            const unsigned g = grid_for(units, 4);
            QGTC_LAUNCH(k_val2bit_rows, dim3(g), dim3(256), 0, st, x, H, W, nbits, ub, ubm1, out, rows_pad, row_words, static_cast<int>(4u * g));
        }
    } else {
        if (out_words < qgtc_cols_words(H, W, nbits, output_layer)) return QGTC_ESIZE;
        const int lines = output_layer ? pad8(W) : pad128(W), line_words = step128(H) * 4;
        const size_t units = static_cast<size_t>((lines + 63) / 64) * line_words;
        const dim3 g(grid_for(units, 4, 8192)), b(256);   // (cap 2048 -> 8192: 8192^2 49.1 -> 41.4 us = 6.7 TB/s)
        const int nw = static_cast<int>(4u * g.x);
        if (nbits <= 1)
            QGTC_LAUNCH(k_val2bit_cols<1>, g, b, 0, st, x, H, W, nbits, ub, ubm1, out, lines, line_words, nw);
        else if (nbits <= 2)
            QGTC_LAUNCH(k_val2bit_cols<2>, g, b, 0, st, x, H, W, nbits, ub, ubm1, out, lines, line_words, nw);
        else if (nbits <= 4)
            QGTC_LAUNCH(k_val2bit_cols<4>, g, b, 0, st, x, H, W, nbits, ub, ubm1, out, lines, line_words, nw);
        else if (nbits <= 8)
            QGTC_LAUNCH(k_val2bit_cols<8>, g, b, 0, st, x, H, W, nbits, ub, ubm1, out, lines, line_words, nw);
        else
            QGTC_LAUNCH(k_val2bit_cols<32>, g, b, 0, st, x, H, W, nbits, ub, ubm1, out, lines, line_words, nw);
    }
    HIP_TRY(launch_status());
    return QGTC_OK;
}

int qgtc_bit2val(const uint32_t *bits, size_t bits_words, int nbits, int H, int W, int col_major,
                 int output_layer, int32_t *out, void *stream) {
    if (!bits || !out || H <= 0 || W <= 0 || !bits_ok(nbits)) return QGTC_EINVAL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    size_t plane;
    int line_words;
    if (col_major) {
        line_words = step128(H) * 4;
        plane = static_cast<size_t>(output_layer ? pad8(W) : pad128(W)) * line_words;
    } else {
        line_words = step128(W) * 4;
        plane = static_cast<size_t>(pad8(H)) * line_words;
    }
    const size_t total = static_cast<size_t>(H) * W;
    const unsigned g = grid_for(total, 256);
    QGTC_LAUNCH(k_bit2val, dim3(g), dim3(256), 0, st, bits, static_cast<unsigned long long>(bits_words), nbits, H, W, col_major, plane, line_words, out, 256u * g);
    HIP_TRY(launch_status());
    return QGTC_OK;
}

// ---- which kernel family a single launch takes (one rule set for the launchers and for qgtc_bitmm_route) ---------------
enum SingleRoute { RT_FP4_NARROW = 0, RT_FP4_STREAM, RT_FP4_ROWS, RT_FP4_WIDE, RT_MFMA_128, RT_POPCOUNT };
static SingleRoute single_route(const qgtc_problem &pr, int a, int w, int ob, int mode, unsigned flags) {
    const bool mf = (flags & QGTC_ENGINE_MFMA) != 0u, au = (flags & QGTC_ENGINE_AUTO) != 0u;
    if (stream_ok(pr, a, w, ob, mode) && (mf || (au && auto_prefers_stream(pr.M, pr.K, pr.N)))) return RT_FP4_STREAM;
    if (skinny_ok(pr.K, pr.N, a, w) && (mf || (au && auto_prefers_skinny(pr.M, pr.K, pr.N, a, w)))) return RT_FP4_NARROW;
    if (rows_single_ok(pr, a, w, ob, mode) && (mf || au)) return RT_FP4_ROWS;
    if (wide_ok(pr, a, w, ob, mode) && (mf || (au && auto_prefers_wide(pr.M, pr.K, pr.N, a, w, mode)))) return RT_FP4_WIDE;
    if ((mf && mfma_ok(a, w)) || (au && auto_prefers_mfma(pr.M, pr.K, pr.N, a, w))) return RT_MFMA_128;
    return RT_POPCOUNT;
}
static int launch_single_route(const qgtc_problem &pr, int a, int w, int ob, int mode, unsigned flags, hipStream_t st) {
    switch (single_route(pr, a, w, ob, mode, flags)) {
        case RT_FP4_NARROW: return qgtc_launch_skinny(pr, a, w, ob, mode, !(flags & QGTC_NO_ZERO_SKIP), st);
        case RT_FP4_STREAM: return qgtc_launch_stream(pr, ob, mode, !(flags & QGTC_NO_ZERO_SKIP), st);
        case RT_FP4_ROWS: return qgtc_launch_rows_single(pr, a, w, ob, mode, st);
        case RT_FP4_WIDE: return qgtc_launch_wide(pr, a, w, ob, mode, st);
        case RT_MFMA_128: return qgtc_launch_mfma(pr, a, w, ob, mode, st);
        default: break;
    }
    if (flags & QGTC_NO_ZERO_SKIP) return dispatch_single<false>(pr, pr.K, a, w, ob, mode, st);
    return dispatch_single<true>(pr, pr.K, a, w, ob, mode, st);
}

const char *qgtc_bitmm_route(int M, int K, int N, int bit1, int bit2, int output_bit, int mode, unsigned flags) {
    if (M <= 0 || K <= 0 || N <= 0 || !bits_ok(bit1) || !bits_ok(bit2) || mode < 0 || mode > 2 || (mode != 2 && !bits_ok(output_bit))) return "invalid";
    qgtc_problem pr{nullptr, nullptr, nullptr, qgtc_rows_words(M, K, bit1), qgtc_cols_words(K, N, bit2, 0), M, K, N, pad128(N), 0, nullptr};
    const int ob = mode == 2 ? 1 : output_bit;
    switch (single_route(pr, bit1, bit2, ob, mode, flags)) {
        case RT_FP4_NARROW: return qgtc_skinny_is_one(pr, ob, mode) ? "k_bitmm_fp4_one" : "k_bitmm_fp4_skinny";
        case RT_FP4_STREAM: return "k_bitmm_fp4_stream";
        case RT_FP4_ROWS: return "k_bitmm_fp4_rows_single";
        case RT_FP4_WIDE: return "k_bitmm_fp4_wide";
        case RT_MFMA_128: return "k_bitmm_mfma";
        default: return "k_bitmm";
    }
}

int qgtc_bitmm2bit(const uint32_t *X, size_t x_words, const uint32_t *W, size_t w_words, int M,
                   int K, int N, int bit1, int bit2, int output_bit, uint32_t *out,
                   size_t out_words, unsigned flags, void *stream) {
    int rc = check_mm_args(X, W, out, M, K, N, bit1, bit2);
    if (rc != QGTC_OK) return rc;
    if (!bits_ok(output_bit) || !words_ok(x_words, w_words)) return QGTC_EINVAL;
    const bool cols = flags & QGTC_OUT_COLS;
    const size_t need = cols ? qgtc_cols_words(M, N, output_bit, 0) : qgtc_rows_words(M, N, output_bit);
    if (out_words < need) return QGTC_ESIZE;
    qgtc_problem pr{X, W, out, x_words, w_words, M, K, N, pad128(N), 0, nullptr};
    return launch_single_route(pr, bit1, bit2, output_bit, cols ? 1 : 0, flags, static_cast<hipStream_t>(stream));
}

int qgtc_bitmm2int(const uint32_t *X, size_t x_words, const uint32_t *W, size_t w_words, int M,
                   int K, int N, int bit1, int bit2, int pad_128, float *out, size_t out_elems,
                   unsigned flags, void *stream) {
    int rc = check_mm_args(X, W, out, M, K, N, bit1, bit2);
    if (rc != QGTC_OK) return rc;
    if (!words_ok(x_words, w_words)) return QGTC_EINVAL;
    if (out_elems < static_cast<size_t>(M) * N) return QGTC_ESIZE;
    qgtc_problem pr{X, W, out, x_words, w_words, M, K, N, pad_128 ? pad128(N) : pad8(N), 0, nullptr};
    return launch_single_route(pr, bit1, bit2, 1, 2, flags, static_cast<hipStream_t>(stream));
}

int qgtc_bitmm2bit_profile(const uint32_t *X, size_t x_words, const uint32_t *W, size_t w_words,
                           int M, int K, int N, int bit1, int bit2, int output_bit, uint32_t *out,
                           size_t out_words, unsigned flags, int reps, float *elapsed_ms,
                           void *stream) {
    if (reps <= 0 || !elapsed_ms) return QGTC_EINVAL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    // one untimed launch validates the arguments and sets the kernel attributes
    int rc = qgtc_bitmm2bit(X, x_words, W, w_words, M, K, N, bit1, bit2, output_bit, out, out_words,
                            flags, stream);
    if (rc != QGTC_OK) return rc;
    EventPair ev;   // destroyed on every path out
    HIP_TRY(ev.create());
    HIP_TRY(hipEventRecord(ev.e0, st));
    for (int i = 0; i < reps && rc == QGTC_OK; i++)
        rc = qgtc_bitmm2bit(X, x_words, W, w_words, M, K, N, bit1, bit2, output_bit, out, out_words,
                            flags, stream);
    hipError_t e = hipEventRecord(ev.e1, st);
    if (e == hipSuccess) e = hipEventSynchronize(ev.e1);
    float ms = 0.0f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, ev.e0, ev.e1);
    if (rc != QGTC_OK) return rc;
    if (e != hipSuccess) return hip_fail(e, "profile events");
    *elapsed_ms = ms;
    return QGTC_OK;
}

int qgtc_tile_counters(const uint32_t *X, size_t x_words, int M, int K, int N, int bit1, int bit2,
                       uint64_t *counters, void *stream) {
    if (!X || !counters || M <= 0 || K <= 0 || N <= 0 || !bits_ok(bit1) || !bits_ok(bit2))
        return QGTC_EINVAL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    HIP_TRY(hipMemsetAsync(counters, 0, 2 * sizeof(uint64_t), st));
    const unsigned long long gdx = step8(M), gdy = step8(N), gdk = step128(K);
    const unsigned long long total = gdx * gdy * gdk * bit1 * bit2;
    const size_t items = static_cast<size_t>(bit1) * gdx * gdk;
    hipLaunchKernelGGL(k_tile_counters, dim3(grid_for(items, 256)), dim3(256), 0, st, X,
                       static_cast<unsigned long long>(x_words), M, K, bit1, total,
                       gdy * static_cast<unsigned long long>(bit2),
                       reinterpret_cast<unsigned long long *>(counters));
    HIP_TRY(hipGetLastError());
    return QGTC_OK;
}

// ---- which kernel family a grouped launch takes (one rule set for qgtc_bitmm_batched and qgtc_bitmm_batched_route) ------
enum BatchedRoute { BR_XW_ROWS = 0, BR_ROWS, BR_MFMA_128, BR_POPCOUNT };
static BatchedRoute batched_route(int max_M, int max_K, int max_N, int bit1, int bit2, int ob_, int mode, unsigned flags) {
    const bool engine = (flags & (QGTC_ENGINE_MFMA | QGTC_ENGINE_AUTO)) != 0u;
    const bool rows_route = engine && ((flags & QGTC_ZERO_JUMP) || max_N <= 64 || max_K <= 256) && rows_ok(max_K, max_N, bit1, bit2, ob_, mode);
    if (engine && mode == 1 && xw_rows_ok(max_K, max_N, bit1, bit2, ob_)) return BR_XW_ROWS;     // X . W stages: row blocks
    if (rows_route) return BR_ROWS;   // sparse left operands / narrow outputs / one or two k-quads: one workgroup per 32-row block
    // every other cols-layout stage the row blocks can take (tools/grouped_cols_sweep.py, 75 ragged batches: level with or ahead of the
    // column-strip and one-wave-per-tile kernels of rounds 1-2 on every shape tried - 3 x 3 bits K = 128: 16.3 -> 10-13.5 us, 4 x 8 bits
    // 24 -> 19-22, K = 256 N = 64 at 2 x 2 bits 12.6 -> 8.0 - which are deleted)
    if (engine && mode == 1 && rows_cols_ok(max_K, max_N, bit1, bit2, ob_)) return BR_ROWS;
    if (((flags & QGTC_ENGINE_MFMA) && mfma_ok(bit1, bit2)) ||  // problems with a one-word bitmap jump zero tiles
        ((flags & QGTC_ENGINE_AUTO) && auto_prefers_mfma_batched(max_M, max_N, bit1, bit2)))
        return BR_MFMA_128;
    return BR_POPCOUNT;
}

int qgtc_bitmm_batched(const qgtc_problem *problems, int count, int max_M, int max_K, int max_N,
                       int bit1, int bit2, int output_bit, int mode, unsigned flags,
                       void *stream) {
    if (!problems || count <= 0 || max_M <= 0 || max_K <= 0 || max_N <= 0) return QGTC_EINVAL;
    if (!bits_ok(bit1) || !bits_ok(bit2) || mode < 0 || mode > 2) return QGTC_EINVAL;
    if (mode != 2 && !bits_ok(output_bit)) return QGTC_EINVAL;
    if (count > 65535) return QGTC_EINVAL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (flags & QGTC_CHECK_DESCRIPTORS) {
        const int crc = qgtc_launch_check_descriptors(problems, nullptr, count, max_M, max_K, max_N, 0, 0, 0, st);
        if (crc != QGTC_OK) return crc;
        flags &= ~QGTC_CHECK_DESCRIPTORS;
    }
    const int ob_ = mode == 2 ? 1 : output_bit;
    const BatchedRoute route = batched_route(max_M, max_K, max_N, bit1, bit2, ob_, mode, flags);
    switch (route) {
        case BR_XW_ROWS: return qgtc_launch_xw_rows(problems, count, max_M, bit1, bit2, ob_, st);
        case BR_ROWS: return qgtc_launch_rows(problems, count, max_M, max_N, bit1, bit2, ob_, mode, st);
        case BR_MFMA_128: return qgtc_launch_mfma_batched(problems, count, max_M, max_K, max_N, bit1, bit2, ob_, mode, st);
        default: break;
    }
    // K is per problem; the split-K factor and chunk size are chosen for the longest K. Any choice is correct; this only affects speed.
    if (flags & QGTC_NO_ZERO_SKIP) return dispatch_batched<false, false>(problems, count, max_M, max_N, max_K, bit1, bit2, ob_, mode, st);
    if (flags & QGTC_ZERO_JUMP)  // the descriptors carry occupancy bitmaps (qgtc_tile_occupancy)
        return dispatch_batched<true, true>(problems, count, max_M, max_N, max_K, bit1, bit2, ob_, mode, st);
    return dispatch_batched<true, false>(problems, count, max_M, max_N, max_K, bit1, bit2, ob_, mode, st);
}

const char *qgtc_bitmm_batched_route(int max_M, int max_K, int max_N, int bit1, int bit2, int output_bit, int mode, unsigned flags) {
    if (max_M <= 0 || max_K <= 0 || max_N <= 0 || !bits_ok(bit1) || !bits_ok(bit2) || mode < 0 || mode > 2 || (mode != 2 && !bits_ok(output_bit))) return "invalid";
    switch (batched_route(max_M, max_K, max_N, bit1, bit2, mode == 2 ? 1 : output_bit, mode, flags)) {
        case BR_XW_ROWS: return "k_bitmm_fp4_xw_rows";
        case BR_ROWS: return "k_bitmm_fp4_rows";
        case BR_MFMA_128: return "k_bitmm_mfma_batched";
        default: return "k_bitmm_batched";
    }
}

int qgtc_gcn_layer_batched(const qgtc_problem *stage1, const qgtc_problem *stage2, int count, int max_M, int max_K1,
                           int max_K2, int max_N, int x_bits, int w_bits, int t_bits, int a_bits, int output_bit,
                           int mode, unsigned flags, void *stream) {
    if (!stage1 || !stage2 || count <= 0 || count > 65535) return QGTC_EINVAL;
    if (max_M <= 0 || max_K1 <= 0 || max_K2 <= 0 || max_N <= 0) return QGTC_EINVAL;
    if (!bits_ok(x_bits) || !bits_ok(w_bits) || !bits_ok(t_bits) || !bits_ok(a_bits)) return QGTC_EINVAL;
    if (mode != 0 && mode != 2) return QGTC_EINVAL;
    if (mode == 0 && !bits_ok(output_bit)) return QGTC_EINVAL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (flags & QGTC_CHECK_DESCRIPTORS) {
        const int crc = qgtc_launch_check_descriptors(stage1, stage2, count, max_M, max_K1, max_N, max_K2, max_N, 1, st);
        if (crc != QGTC_OK) return crc;
        flags &= ~QGTC_CHECK_DESCRIPTORS;
    }
    // two grouped launches on one stream (an in-launch hand-off between the stages was built in round 2 and measured
    // slower: 35 against 30 us for a 75-batch 128-wide layer)
    int rc = qgtc_bitmm_batched(stage1, count, max_M, max_K1, max_N, x_bits, w_bits, t_bits, 1, flags & ~QGTC_ZERO_JUMP, stream);
    if (rc != QGTC_OK) return rc;
    return qgtc_bitmm_batched(stage2, count, max_M, max_K2, max_N, a_bits, t_bits, mode == 2 ? 1 : output_bit, mode, flags, stream);
}

int qgtc_gcn_chain_batched(const qgtc_problem *stage_a, const qgtc_problem *stage_xw, int count, int max_M, int max_K,
                           int max_N1, int max_N2, int a_bits, int t_bits, int act_bits, int w_bits, int out_bits,
                           int out_mode, unsigned flags, void *stream) {
    if (!stage_a || !stage_xw || count <= 0 || count > 65535) return QGTC_EINVAL;
    if (max_M <= 0 || max_K <= 0 || max_N1 <= 0 || max_N2 <= 0) return QGTC_EINVAL;
    if (!bits_ok(a_bits) || !bits_ok(t_bits) || !bits_ok(act_bits) || !bits_ok(w_bits)) return QGTC_EINVAL;
    if (out_mode != 1 && out_mode != 2) return QGTC_EINVAL;
    if (out_mode == 1 && !bits_ok(out_bits)) return QGTC_EINVAL;
    if (flags & QGTC_CHECK_DESCRIPTORS) {
        const int crc = qgtc_launch_check_descriptors(stage_a, stage_xw, count, max_M, max_K, max_N1, max_N1, max_N2, 2, static_cast<hipStream_t>(stream));
        if (crc != QGTC_OK) return crc;
        flags &= ~QGTC_CHECK_DESCRIPTORS;
    }
    // (rounds 2-3 had a one-launch kernel for the pair; the chain entries below superseded it)
    int rc = qgtc_bitmm_batched(stage_a, count, max_M, max_K, max_N1, a_bits, t_bits, act_bits, 0, flags, stream);
    if (rc != QGTC_OK) return rc;
    return qgtc_bitmm_batched(stage_xw, count, max_M, max_N1, max_N2, act_bits, w_bits, out_mode == 2 ? 1 : out_bits, out_mode, flags & ~QGTC_ZERO_JUMP, stream);
}

size_t qgtc_weight_codes_words(int K, int N, int nbits, int order) {
    if (K <= 0 || N <= 0 || N > 256 || nbits < 1 || nbits > 8 || (order != 0 && order != 1) || (order == 1 && K > 256)) return 0u;
    const size_t table = static_cast<size_t>(weight_table_blocks(N)) * weight_table_slices(K, order) * chain_digits(nbits) * 64u * 4u;
    return table * (order == 0 ? static_cast<size_t>(step128(K)) : 1u);   // order 0: a table per k-quad of K
}

int qgtc_chain_from_cols(const uint32_t *cols, size_t cols_words, int H, int W, int nbits, uint32_t *chain, size_t chain_words,
                         void *stream) {
    if (!cols || !chain || H <= 0 || W <= 0 || nbits < 1 || nbits > 8) return QGTC_EINVAL;
    if (chain_words < qgtc_chain_words(H, W, nbits)) return QGTC_ESIZE;
    if (!aligned16(chain)) return QGTC_EALIGN;
    return qgtc_launch_cols_to_chain(cols, cols_words, H, W, nbits, chain, static_cast<hipStream_t>(stream));
}

size_t qgtc_chain_words(int M, int N, int bits) {   // 5 .. 8 bits: two arrays of the 4-bit form (bitmm_fp4_rbx.hip.h)
    return (M > 0 && N > 0 && bits >= 1 && bits <= 8) ? static_cast<size_t>(step128(M)) * pad128(N) * 16u * (bits > 4 ? 2u : 1u) : 0u;
}

int qgtc_expand_weights(const qgtc_expand_job *jobs, int n_jobs, void *stream) {
    if (!jobs || n_jobs <= 0 || n_jobs > QGTC_MAX_WEIGHTS) return QGTC_EINVAL;
    for (int i = 0; i < n_jobs; i++) {
        const qgtc_expand_job &j = jobs[i];
        if (!j.W || !j.codes || j.K <= 0 || j.N <= 0 || j.N > 256 || j.nbits < 1 || j.nbits > 8 || j.w_lines < j.N || (j.order != 0 && j.order != 1)) return QGTC_EINVAL;
        if ((j.order == 0 && j.K > 8192) || (j.order == 1 && j.K > 256)) return QGTC_EINVAL;
        if (!aligned16(j.codes)) return QGTC_EALIGN;
        if (j.codes_words < qgtc_weight_codes_words(j.K, j.N, j.nbits, j.order)) return QGTC_ESIZE;   // (ABI 11: the capacity travels with the job)
    }
    return qgtc_launch_expand_weights(jobs, n_jobs, static_cast<hipStream_t>(stream));
}

int qgtc_chain_transform(const qgtc_problem *stage, int count, int max_M, int K, int N, int x_bits, int out_bits,
                         const uint32_t *w_codes, unsigned flags, void *stream) {
    if (!stage || !w_codes || count <= 0 || count > 65535 || max_M <= 0) return QGTC_EINVAL;
    const bool narrow = rbw_xw_ok(K, N, x_bits, out_bits);   // (1 .. 4 bits, N <= 128: k_rbw_xw; else 5 .. 8 bits / up to 256 columns: k_rbx_xw)
    if ((!narrow && !rbx_xw_ok(K, N, x_bits, out_bits)) || getenv_flag("QGTC_NO_RBW")) return QGTC_EINVAL;
    if (!aligned16(w_codes)) return QGTC_EALIGN;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (flags & QGTC_CHECK_DESCRIPTORS) {
        const int crc = qgtc_launch_check_descriptors(stage, nullptr, count, max_M, K, N, 0, 0, 0, st, N, 0, K);   // (every descriptor's K == the K of the weight tables)
        if (crc != QGTC_OK) return crc;
    }
    if (!narrow) return qgtc_launch_rbx_xw(stage, count, max_M, K, N, x_bits, out_bits, w_codes, st);
    return qgtc_launch_rbw_xw(stage, count, max_M, K, N, x_bits, out_bits, w_codes, st);
}

int qgtc_chain_aggregate(const qgtc_problem *stage_a, const qgtc_problem *stage_xw, int count, int max_M, int max_K, int N1,
                         int N2, int t_bits, int act_bits, int out_bits, int out_mode, const uint32_t *w2_codes,
                         unsigned flags, void *stream) {
    if (!stage_a || count <= 0 || count > 65535 || max_M <= 0) return QGTC_EINVAL;
    if (out_mode < 0 || out_mode > 2 || (out_mode != 0 && (!stage_xw || !w2_codes))) return QGTC_EINVAL;
    const bool narrow = rbw_chain_ok(max_K, N1, N2, t_bits, act_bits, out_bits, out_mode);
    if ((!narrow && !rbx_chain_ok(max_K, N1, N2, t_bits, act_bits, out_bits, out_mode)) || getenv_flag("QGTC_NO_RBW")) return QGTC_EINVAL;
    if (w2_codes && !aligned16(w2_codes)) return QGTC_EALIGN;
    // (float32 outputs are written through a buffer descriptor with a 32-bit extent)
    if (out_mode != 1 && static_cast<unsigned long long>(max_M) * static_cast<unsigned long long>(out_mode == 0 ? N1 : N2) * 4ull >= (1ull << 31)) return QGTC_EINVAL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (flags & QGTC_CHECK_DESCRIPTORS) {
        const int crc = out_mode == 0 ? qgtc_launch_check_descriptors(stage_a, nullptr, count, max_M, max_K, N1, 0, 0, 0, st, N1, 0)
                                      : qgtc_launch_check_descriptors(stage_a, stage_xw, count, max_M, max_K, N1, N1, N2, 4, st, N1, N2);
        if (crc != QGTC_OK) return crc;
    }
    if (!narrow)
        return qgtc_launch_rbx_chain(stage_a, out_mode == 0 ? nullptr : stage_xw, count, max_M, N1, N2, t_bits, act_bits, out_mode, w2_codes,
                                     (flags & QGTC_CHAIN_ADJ_TILES) != 0u, st);
    return qgtc_launch_rbw_chain(stage_a, out_mode == 0 ? nullptr : stage_xw, count, max_M, N1, N2, t_bits, act_bits, out_bits, out_mode, w2_codes,
                                 (flags & QGTC_CHAIN_ADJ_TILES) != 0u, st);
}

size_t qgtc_adj_tiles_words(int M, int K) { return (M > 0 && K > 0) ? static_cast<size_t>((M + 31) / 32) * step128(K) * 128u : 0u; }

int qgtc_adj_tiles_from_rows(const uint32_t *rows, size_t rows_words, int M, int K, uint32_t *tiles, size_t tiles_words, void *stream) {
    if (!rows || !tiles || M <= 0 || K <= 0) return QGTC_EINVAL;
    if (tiles_words < qgtc_adj_tiles_words(M, K)) return QGTC_ESIZE;
    if (!aligned16(rows) || !aligned16(tiles)) return QGTC_EALIGN;
    return qgtc_launch_rows_to_tiles(rows, rows_words, M, K, tiles, static_cast<hipStream_t>(stream));
}

size_t qgtc_occupancy_words(int M, int K) {
    return static_cast<size_t>((M + TM - 1) / TM) * ((step128(K) + 63) / 64);
}

int qgtc_tile_occupancy(const uint32_t *X, size_t x_words, int M, int K, int bit1, uint64_t *occ,
                        size_t occ_words, void *stream) {
    if (!X || !occ || M <= 0 || K <= 0 || !bits_ok(bit1)) return QGTC_EINVAL;
    if (!aligned16(X)) return QGTC_EALIGN;
    if (x_words >= (1ull << 30)) return QGTC_EINVAL;
    if (occ_words < qgtc_occupancy_words(M, K)) return QGTC_ESIZE;
    const int tiles_m = (M + TM - 1) / TM, ow = (step128(K) + 63) / 64;
    const int waves = tiles_m * ow;
    hipLaunchKernelGGL(k_tile_occupancy, dim3((waves + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream),
                       X, static_cast<unsigned>(x_words * 4), M, K, bit1,
                       reinterpret_cast<unsigned long long *>(occ), ow, tiles_m);
    HIP_TRY(hipGetLastError());
    return QGTC_OK;
}

int qgtc_tile_occupancy_batched(qgtc_problem *problems, int count, int max_M, int max_K, int bit1,
                                float max_fraction, uint64_t *stats, void *stream) {
    if (!problems || count <= 0 || count > 65535 || max_M <= 0 || max_K <= 0 || !bits_ok(bit1)) return QGTC_EINVAL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int waves = ((max_M + TM - 1) / TM) * ((step128(max_K) + 63) / 64);
    hipLaunchKernelGGL(k_tile_occupancy_batched, dim3((waves + 3) / 4, count), dim3(256), 0, st, problems, bit1);
    HIP_TRY(hipGetLastError());
    if (stats) {
        hipLaunchKernelGGL(k_occupancy_decide, dim3(1), dim3(1024), 0, st, problems, count, max_fraction,
                           reinterpret_cast<unsigned long long *>(stats));
        HIP_TRY(hipGetLastError());
    }
    return QGTC_OK;
}

int qgtc_tile_occupancy_decide(qgtc_problem *problems, int count, float max_fraction, uint64_t *stats,
                               void *stream) {
    if (!problems || !stats || count <= 0 || count > 65535) return QGTC_EINVAL;
    hipLaunchKernelGGL(k_occupancy_decide, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream), problems,
                       count, max_fraction, reinterpret_cast<unsigned long long *>(stats));
    HIP_TRY(hipGetLastError());
    return QGTC_OK;
}

int qgtc_pack_edges(const int64_t *cells, const int32_t *counts, size_t n_cells, int H, int W,
                    int nbits, uint32_t *out, size_t out_words, void *stream) {
    if (!out || H <= 0 || W <= 0 || !bits_ok(nbits) || (n_cells && !cells)) return QGTC_EINVAL;
    if (out_words < qgtc_rows_words(H, W, nbits)) return QGTC_ESIZE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    HIP_TRY(hipMemsetAsync(out, 0, qgtc_rows_words(H, W, nbits) * sizeof(uint32_t), st));
    if (n_cells) {
        const float ub = std::ldexp(1.0f, nbits), ubm1 = ub - 1.0f;
        hipLaunchKernelGGL(k_pack_edges, dim3(grid_for(n_cells, 256)), dim3(256), 0, st, cells, counts,
                           n_cells, H, W, nbits, ub, ubm1, out, pad8(H), step128(W) * 4);
        HIP_TRY(hipGetLastError());
    }
    return QGTC_OK;
}

int qgtc_pack_edge_list(const int64_t *src, const int64_t *dst, size_t n_edges, int H, int W, uint32_t *out,
                        size_t out_words, uint32_t *scratch, size_t scratch_words, int *bad_index, void *stream) {
    if (!out || !scratch || H <= 0 || W <= 0 || (n_edges && (!src || !dst))) return QGTC_EINVAL;
    const size_t words = qgtc_rows_words(H, W, 1);
    if (out_words < words || scratch_words < 2 * words) return QGTC_ESIZE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (scratch == out + words) {   // one allocation [out | scratch] (the PyTorch binding's): one memset
        HIP_TRY(hipMemsetAsync(out, 0, 3 * words * sizeof(uint32_t), st));
    } else {
        HIP_TRY(hipMemsetAsync(out, 0, words * sizeof(uint32_t), st));
        HIP_TRY(hipMemsetAsync(scratch, 0, 2 * words * sizeof(uint32_t), st));
    }
    if (bad_index) HIP_TRY(hipMemsetAsync(bad_index, 0, sizeof(int), st));
    if (n_edges) {
        hipLaunchKernelGGL(k_edge_list_count, dim3(grid_for(n_edges, 256)), dim3(256), 0, st, src, dst, n_edges, H, W,
                           out, scratch, scratch + words, step128(W) * 4, bad_index);
        hipLaunchKernelGGL(k_edge_list_finish, dim3(grid_for(words, 256)), dim3(256), 0, st, out, scratch,
                           scratch + words, words);
        HIP_TRY(hipGetLastError());
    }
    return QGTC_OK;
}

// work buffer of the bucketed loader: [count x (RB + 1) bucket offsets | buckets | count 64-bit per-batch tile counters]; the bucket
// region is padded by one word when that makes the counters 8-byte aligned
static size_t load_work_offsets(int count, int max_n) { return static_cast<size_t>(count) * ((max_n + 31) / 32 + 1); }
static size_t load_work_counts(int count, int max_n) { return static_cast<size_t>(count) * ((max_n + 31) / 32); }
static size_t load_work_edges(int count, int max_n, uint64_t total_edges) { return static_cast<size_t>(total_edges) + ((load_work_offsets(count, max_n) + total_edges) & 1u); }
size_t qgtc_load_work_words(int count, int max_n, uint64_t total_edges) {
    // 0: no bucketed route for this iterator. THE decider of the route: a caller that gets 0 here passes work = NULL and the
    // cleared region with A and scratch inside; one that passes a work buffer gets the bucketed route or an error, never the other route
    // silently. QGTC_NO_LOAD_SORT (tests / tools: the bitmap route on iterators that would be bucketed) is read here, on every call, and
    // nowhere else.
    if (count <= 0 || max_n <= 0 || max_n > LOAD_SORT_MAX_N || total_edges >= (1ull << 32) || getenv_flag_now("QGTC_NO_LOAD_SORT")) return 0u;
    // [bucket offsets | buckets (an even number of words) | one occupied-tile count per row block of every batch]
    return load_work_offsets(count, max_n) + load_work_edges(count, max_n, total_edges) + load_work_counts(count, max_n);
}

int qgtc_load_batches(const qgtc_loader_batch *batches, int count, int max_n, uint64_t max_edges, const int64_t *src,
                      const int64_t *dst, const float *feats, int F, int x_bits, void *zero, size_t zero_bytes,
                      uint64_t *stats, int *bad_index, unsigned formats, uint32_t *work, size_t work_words, void *stream) {
    if (!batches || count <= 0 || count > 65535 || max_n <= 0 || !zero || zero_bytes == 0) return QGTC_EINVAL;
    if (max_edges && (!src || !dst)) return QGTC_EINVAL;
    if (feats && (F <= 0 || !bits_ok(x_bits))) return QGTC_EINVAL;
    if (max_edges >= (1ull << 40)) return QGTC_EINVAL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    HIP_TRY(hipMemsetAsync(zero, 0, zero_bytes, st));
    if (bad_index) HIP_TRY(hipMemsetAsync(bad_index, 0, sizeof(int), st));
    const int rb_max = (max_n + TM - 1) / TM;
    const size_t work_fixed = load_work_offsets(count, max_n) + load_work_counts(count, max_n);
    if (work) {   // a work buffer that cannot serve is an error, not a reason to take the route whose `zero` region the caller did not supply
        if (max_n > LOAD_SORT_MAX_N) return QGTC_EINVAL;
        if (reinterpret_cast<uintptr_t>(work) & 7u) return QGTC_EALIGN;
        if (work_words < work_fixed) return QGTC_ESIZE;
    }
    if (work) {
        size_t edges_pad = work_words - work_fixed;                                    // what the caller left for the buckets (a batch that does not fit is reported)
        if ((load_work_offsets(count, max_n) + edges_pad) & 1u) edges_pad -= 1u;       // (an even number of words, as qgtc_load_work_words counts them)
        // the bucketed route (loader_kernels.hip.h): edges by row block, then every word of rows + tiles + bitmaps written once from LDS
        hipLaunchKernelGGL(k_load_sort, dim3(count), dim3(LOAD_SORT_THREADS), 0, st, batches, src, dst, work, static_cast<unsigned long long>(work_words),
                           rb_max, count, static_cast<unsigned long long>(edges_pad), bad_index);
        HIP_TRY(hipGetLastError());
        const size_t lds = static_cast<size_t>(3) * 32 * step128(max_n) * 16;
        hipLaunchKernelGGL(k_load_tiles, dim3(rb_max, count), dim3(64), lds, st, batches, work, rb_max, count, static_cast<unsigned long long>(edges_pad),
                           reinterpret_cast<unsigned long long *>(stats));
        if (stats)
            hipLaunchKernelGGL(k_load_stats, dim3(1), dim3(1024), 0, st, work + static_cast<size_t>(count) * (rb_max + 1) + edges_pad, count * rb_max,
                               reinterpret_cast<unsigned long long *>(stats));
        HIP_TRY(hipGetLastError());
    } else {
        if (max_edges) {
            hipLaunchKernelGGL(k_load_edges, dim3(grid_for(static_cast<size_t>(max_edges), 256, 4096), count), dim3(256), 0, st, batches, src, dst, bad_index);
            HIP_TRY(hipGetLastError());
        }
        // one wave per (32-row block, bitmap word) of the largest batch
        const int waves = ((max_n + TM - 1) / TM) * ((step128(max_n) + 63) / 64);
        hipLaunchKernelGGL(k_load_finish, dim3((waves + 3) / 4, count), dim3(256), 0, st, batches, reinterpret_cast<unsigned long long *>(stats));
        HIP_TRY(hipGetLastError());
    }
    if (feats) {
        const float ub = std::ldexp(1.0f, x_bits), ubm1 = ub - 1.0f;
        // cols layout (k_val2bit_cols): a wave per (64-line chunk, 32-row word); with the rows layout wanted and at most 8 planes the same
        // waves write it too (k_load_x_both: the features are read once), else a second pass (k_val2bit_rows_v4 / k_val2bit_rows)
        const size_t units = static_cast<size_t>((pad128(F) + 63) / 64) * (step128(max_n) * 4);
        const dim3 g(grid_for(units, 4, 8192), count), b(256);
        // (k_load_x_both reads a batch's features through a buffer resource with 32-bit offsets, the padding rows' included)
        const bool both = (formats & QGTC_LOAD_X_ROWS) && x_bits <= 8 && static_cast<size_t>(pad128(max_n) + 32) * F * 4u < (1ull << 31) &&
                          !getenv_flag("QGTC_NO_LOAD_BOTH");
        if (both) {
            if (x_bits <= 1) hipLaunchKernelGGL(k_load_x_both<1>, g, b, 0, st, batches, feats, F, x_bits, ub, ubm1);
            else if (x_bits <= 2) hipLaunchKernelGGL(k_load_x_both<2>, g, b, 0, st, batches, feats, F, x_bits, ub, ubm1);
            else if (x_bits <= 4) hipLaunchKernelGGL(k_load_x_both<4>, g, b, 0, st, batches, feats, F, x_bits, ub, ubm1);
            else hipLaunchKernelGGL(k_load_x_both<8>, g, b, 0, st, batches, feats, F, x_bits, ub, ubm1);
        } else {
            if (x_bits <= 1) hipLaunchKernelGGL(k_load_x_cols<1>, g, b, 0, st, batches, feats, F, x_bits, ub, ubm1);
            else if (x_bits <= 2) hipLaunchKernelGGL(k_load_x_cols<2>, g, b, 0, st, batches, feats, F, x_bits, ub, ubm1);
            else if (x_bits <= 4) hipLaunchKernelGGL(k_load_x_cols<4>, g, b, 0, st, batches, feats, F, x_bits, ub, ubm1);
            else if (x_bits <= 8) hipLaunchKernelGGL(k_load_x_cols<8>, g, b, 0, st, batches, feats, F, x_bits, ub, ubm1);
            else hipLaunchKernelGGL(k_load_x_cols<32>, g, b, 0, st, batches, feats, F, x_bits, ub, ubm1);
            if (formats & QGTC_LOAD_X_ROWS) {   // rows layout (k_val2bit_rows_v4 / k_val2bit_rows): a wave per (row, 256-column chunk)
                const size_t runits = static_cast<size_t>(pad8(max_n)) * ((step128(F) * 4 + 7) / 8);
                if ((F & 3) == 0 && aligned16(feats))
                    hipLaunchKernelGGL(k_load_x_rows<true>, dim3(grid_for((runits + 1) / 2, 4, 8192), count), dim3(256), 0, st, batches, feats, F, x_bits, ub, ubm1);
                else
                    hipLaunchKernelGGL(k_load_x_rows<false>, dim3(grid_for(runits, 4, 8192), count), dim3(256), 0, st, batches, feats, F, x_bits, ub, ubm1);
            }
        }
        HIP_TRY(hipGetLastError());
        if (formats & QGTC_LOAD_X_CHAIN) {   // the chain format of the entries that have an XC (from the cols layout just written)
            if (x_bits > 8) return QGTC_EINVAL;
            const int rc = qgtc_launch_cols_to_chain_batched(batches, count, max_n, F, x_bits, st);
            if (rc != QGTC_OK) return rc;
        }
    }
    return QGTC_OK;
}

int qgtc_i8gemm(const int8_t *A, const int8_t *Bt, int M, int K, int N, float *C, size_t c_elems,
                void *stream) {
    if (!A || !Bt || !C || M <= 0 || K <= 0 || N <= 0) return QGTC_EINVAL;
    if (K % 16 != 0) return QGTC_EINVAL;  // 16-byte MFMA fragments
    if (!aligned16(A) || !aligned16(Bt)) return QGTC_EALIGN;
    if (static_cast<size_t>(M) * K >= (1ull << 32) || static_cast<size_t>(N) * K >= (1ull << 32)) return QGTC_EINVAL;
    if (c_elems < static_cast<size_t>(M) * N) return QGTC_ESIZE;
    const int tiles_m = (M + I8_TM - 1) / I8_TM, tiles_n = (N + I8_TN - 1) / I8_TN;
    hipLaunchKernelGGL(k_i8gemm, dim3(tiles_m * tiles_n), dim3(64 * I8_WAVES), 0,
                       static_cast<hipStream_t>(stream), A, Bt, M, K, N, C, tiles_n);
    HIP_TRY(hipGetLastError());
    return QGTC_OK;
}

int qgtc_i8gemm_profile(const int8_t *A, const int8_t *Bt, int M, int K, int N, float *C,
                        size_t c_elems, int reps, float *elapsed_ms, void *stream) {
    if (reps <= 0 || !elapsed_ms) return QGTC_EINVAL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    int rc = qgtc_i8gemm(A, Bt, M, K, N, C, c_elems, stream);
    if (rc != QGTC_OK) return rc;
    EventPair ev;
    HIP_TRY(ev.create());
    HIP_TRY(hipEventRecord(ev.e0, st));
    for (int i = 0; i < reps && rc == QGTC_OK; i++) rc = qgtc_i8gemm(A, Bt, M, K, N, C, c_elems, stream);
    hipError_t e = hipEventRecord(ev.e1, st);
    if (e == hipSuccess) e = hipEventSynchronize(ev.e1);
    float ms = 0.0f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, ev.e0, ev.e1);
    if (rc != QGTC_OK) return rc;
    if (e != hipSuccess) return hip_fail(e, "profile events");
    *elapsed_ms = ms;
    return QGTC_OK;
}

}  // extern "C"

#ifdef QGTC_SINGLE_TU   // tools/*.hip: the plan / check translation unit too
#include "qgtc_epoch.hip"
#endif
