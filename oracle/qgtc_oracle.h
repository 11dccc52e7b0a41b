/*
 * qgtc_oracle.h — CPU restatement of the QGTC bit-GEMM hot path (TEST INFRASTRUCTURE ONLY).
 *
 * This is the parity oracle for the HIP path in qgtc_ppopp22_amd/csrc. It is plain C, written
 * from the closed-form semantics of the reference's CUDA kernels; every function cites the
 * reference file:line it restates (paths relative to the reference checkout).
 *
 * Who may use it: tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg — as the
 * checker / reported baseline only. The product path (libqgtc_hip.so, the QGTC extension)
 * never links, imports or falls back to anything in oracle/.
 *
 * PARITY PINNING STATUS: the reference records NO expected outputs anywhere (no tests dir, no
 * asserts, print-only unitest.py) and its arithmetic is CUDA-only (nvcuda::wmma b1 fragments,
 * PTX inline asm) so it can neither be compiled (oracle/_ref is unbuildable: needs nvcc + an
 * NVIDIA GPU) nor imported (main_qgtc.py/sampler.py need dgl, ogb and the CUDA extension) in
 * this image. The oracle is therefore pinned by (1) the known answers derivable in closed form
 * from the all-ones inputs of QGTC_module/unitest.py (tests/test_oracle_kat.py) and (2) an
 * independent warp-level emulation of the reference kernels' ballot/brev/byte-store data
 * movement (oracle/warp_emulation.py). Beyond those: PARITY UNPINNED — there is no
 * reference-produced vector to compare against.
 */
#ifndef QGTC_ORACLE_H
#define QGTC_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* utility.h:33-45 — STEPk(x)=ceil(x/k), PADk(x)=k*ceil(x/k) */
int qo_step8(int x);
int qo_step128(int x);
int qo_pad8(int x);
int qo_pad128(int x);

/* Packed buffer sizes in 32-bit words.
 * rows layout  [b][PAD8(H)][STEP128(W)*4]                     QGTC_device.cu:115
 * cols layout  [b][PAD128(W) or PAD8(W)][STEP128(H)*4]        QGTC_device.cu:83,97 */
size_t qo_rows_words(int H, int W, int nbits);
size_t qo_cols_words(int H, int W, int nbits, int output_layer);

/* kernel.h:39-44 (clip) + :49-71 (Quantize_val): float -> int32 quantised value. */
int32_t qo_quantize_one(float x, int nbits);
void qo_quantize(const float *x, size_t n, int nbits, int32_t *q);

/* kernel.h:204-242 QGTC_layer_input — row-major bit-plane pack of int32 [H,W]. */
void qo_pack_rows(const int32_t *q, int H, int W, int nbits, uint32_t *out);
/* kernel.h:75-106 PackFcWeight128 — transposed (col-major) bit-plane pack of int32 [H,W]. */
void qo_pack_cols(const int32_t *q, int H, int W, int nbits, int output_layer, uint32_t *out);
/* kernel.h:173-201 UnPackFcOutput128 / :109-139 UnPackFcWeight128 — inverses. */
void qo_unpack_rows(const uint32_t *bits, int nbits, int H, int W, int32_t *out);
void qo_unpack_cols(const uint32_t *bits, int nbits, int H, int W, int output_layer, int32_t *out);

/* QGTC_device.cu:44-130 val2bit_cuda / :135-206 bit2val_cuda */
void qo_val2bit(const float *x, int H, int W, int nbits, int col_major, int output_layer,
                uint32_t *out);
void qo_bit2val(const uint32_t *bits, int nbits, int H, int W, int col_major, int output_layer,
                int32_t *out);

/* kernel.h:292-341 — int32 accumulators of the multi-plane AND+popcount product.
 * X : rows layout of an M x K matrix with `a` planes.
 * Wt: cols layout of a  K x N matrix with `w` planes, plane stride = STEP128(K)*4*w_lines words
 *     (w_lines = PAD128(N) for the hidden / PAD128 kernels, PAD8(N) for the PAD8 kernel).
 * Words at index >= x_words / w_words read as 0 (bounds-safe; the reference reads raw memory). */
void qo_acc(const uint32_t *X, size_t x_words, const uint32_t *Wt, size_t w_words,
            int M, int K, int N, int a, int w, int w_lines, int32_t *acc /* [M*N] */);

/* kernel.h:31-37 quantize() as called at :350 — the re-quantisation of an accumulator. */
int32_t qo_requant(int32_t c, int out_bit);

/* kernel.h:245-391 QGTC_layer_hidden -> packed rows layout [ob][PAD8(M)][STEP128(N)*4] */
void qo_bitmm2bit(const uint32_t *X, size_t x_words, const uint32_t *Wt, size_t w_words,
                  int M, int K, int N, int a, int w, int ob, uint32_t *out);
/* kernel.h:651-810 QGTC_layer_hidden_col (intended semantics) -> cols layout
 * [ob][PAD128(N)][STEP128(M)*4] */
void qo_bitmm2bit_col(const uint32_t *X, size_t x_words, const uint32_t *Wt, size_t w_words,
                      int M, int K, int N, int a, int w, int ob, uint32_t *out);
/* kernel.h:816-932 / :938-1054 QGTC_layer_output_PAD8 / _PAD128 -> float [M,N] */
void qo_bitmm2int(const uint32_t *X, size_t x_words, const uint32_t *Wt, size_t w_words,
                  int M, int K, int N, int a, int w, int pad_128, float *out);

/* kernel.h:394-512 (base_cnt) and :516-648 (zerojump_cnt) tile counters, per call:
 *   *total   = STEP8(M)*STEP8(N)*STEP128(K)*a*w                    (kernel.h:452)
 *   *nonzero = number of those steps whose 8-row x 128-bit X tile is non-zero
 *              (kernel.h:574-592, intended semantics: the tile's own 8 rows) */
void qo_tile_counters(const uint32_t *X, size_t x_words, int M, int K, int N, int a, int w,
                      uint64_t *total, uint64_t *nonzero);

/* Number of OpenMP threads the oracle will use (1 when built without OpenMP). */
int qo_num_threads(void);
void qo_set_num_threads(int n);

#ifdef __cplusplus
}
#endif
#endif
