/*
 * qgtc_oracle.c — plain-C CPU restatement of the QGTC bit-GEMM hot path.
 * TEST INFRASTRUCTURE ONLY; see qgtc_oracle.h for who may use it and for the parity-pinning
 * status ("parity unpinned" beyond the unitest.py-derived known answers).
 *
 * Conventions (reference kernel.h:98,234 — `__brev(__ballot_sync(...))`):
 *   element i of a packed line lives in word i>>5, bit 31-(i&31)   (MSB first).
 * Every packed tensor is a flat array of 32-bit words.
 */
#include "qgtc_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* ---- utility.h:33-45 --------------------------------------------------------------- */
int qo_step8(int x) { return (x + 7) >> 3; }
int qo_step128(int x) { return (x + 127) >> 7; }
int qo_pad8(int x) { return qo_step8(x) << 3; }
int qo_pad128(int x) { return qo_step128(x) << 7; }

int qo_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* bench.py's cpu_baseline leg picks the thread count that gives the best rate (a 2 MB problem thrashes on 128 threads) */
void qo_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* QGTC_device.cu:115 — rows layout is allocated as {nbits*PAD8(H), STEP128(W)*4}. */
size_t qo_rows_words(int H, int W, int nbits) {
    return (size_t)nbits * (size_t)qo_pad8(H) * (size_t)qo_step128(W) * 4u;
}

/* QGTC_device.cu:97 {nbits*STEP128(H)*4, PAD128(W)}; :83 {nbits*STEP128(H)*4, PAD8(W)} when
 * output_layer. NOTE (reference defect): PackFcWeight128 strides planes by
 * STEP128(H)*PAD128(W)*4 words (kernel.h:88) even into the PAD8-sized allocation, i.e. it
 * writes out of bounds for nbits>1 unless PAD8(W)==PAD128(W). The oracle (and the HIP path)
 * keep every plane inside the allocation the reference reports: with output_layer the plane
 * stride is STEP128(H)*PAD8(W)*4 — the stride QGTC_layer_output_PAD8 reads with
 * (kernel.h:836) and the one the reference's own commented-out UnPackFcWeight128_OUTPUT
 * uses (kernel.h:155). For nbits==1, or PAD8(W)==PAD128(W), this is word-for-word what the
 * reference writes. */
static int cols_lines(int W, int output_layer) {
    return output_layer ? qo_pad8(W) : qo_pad128(W);
}
size_t qo_cols_words(int H, int W, int nbits, int output_layer) {
    return (size_t)nbits * (size_t)qo_step128(H) * 4u * (size_t)cols_lines(W, output_layer);
}

/* ---- quantisation: kernel.h:39-44 clip(), :49-71 Quantize_val ------------------------ */
int32_t qo_quantize_one(float x, int nbits) {
    /* kernel.h:58,66: ub = (1 << bitwidth), lb = 0, compared in float. 2^nbits is exact in
     * float for every supported nbits (<= 32); ldexpf avoids the int overflow of 1<<31/32. */
    const float lb = 0.0f;
    const float ub = ldexpf(1.0f, nbits);
    float y = x;
    if (x < lb) y = lb + 1;      /* kernel.h:41  negative -> 1            */
    else if (x > ub) y = ub - 1; /* kernel.h:42  above 2^b -> 2^b - 1     */
    /* kernel.h:68 __float2int_rn: round-half-to-even; NaN converts to 0. */
    if (isnan(y)) return 0;
    double r = nearbyint((double)y); /* default rounding mode = to-nearest-even */
    /* values up to 2^32 occur only for nbits>=31 (outside the reference's domain): keep the
     * low 32 bits so that plane p is still bit p of the rounded value. */
    return (int32_t)(uint32_t)(uint64_t)(int64_t)r;
}

void qo_quantize(const float *x, size_t n, int nbits, int32_t *q) {
    for (size_t i = 0; i < n; i++) q[i] = qo_quantize_one(x[i], nbits);
}

/* ---- packing ------------------------------------------------------------------------- */
/* kernel.h:204-242. out[p][r][c>>5] bit(31-(c&31)) = (q[r][c]>>p)&1, r<H, c<W; rest 0.
 * (:214 plane stride PAD8(H)*STEP128(W)*4; :237 row stride gdy*4 = STEP128(W)*4.) */
void qo_pack_rows(const int32_t *q, int H, int W, int nbits, uint32_t *out) {
    const size_t row_w = (size_t)qo_step128(W) * 4u;
    const size_t plane_w = (size_t)qo_pad8(H) * row_w;
    memset(out, 0, sizeof(uint32_t) * plane_w * (size_t)nbits);
    for (int p = 0; p < nbits; p++)
        for (int r = 0; r < H; r++)
            for (int c = 0; c < W; c++)
                if (((uint32_t)q[(size_t)r * W + c] >> p) & 1u)
                    out[p * plane_w + (size_t)r * row_w + (c >> 5)] |= 1u << (31 - (c & 31));
}

/* kernel.h:75-106. out[p][c][r>>5] bit(31-(r&31)) = (q[r][c]>>p)&1 (:96-101: line c at
 * (by*8+ly)*gdx*4, word bx*4+lx, lane = row within the 32-row group). */
void qo_pack_cols(const int32_t *q, int H, int W, int nbits, int output_layer, uint32_t *out) {
    const size_t line_w = (size_t)qo_step128(H) * 4u;
    const size_t plane_w = line_w * (size_t)cols_lines(W, output_layer);
    memset(out, 0, sizeof(uint32_t) * plane_w * (size_t)nbits);
    for (int p = 0; p < nbits; p++)
        for (int r = 0; r < H; r++)
            for (int c = 0; c < W; c++)
                if (((uint32_t)q[(size_t)r * W + c] >> p) & 1u)
                    out[p * plane_w + (size_t)c * line_w + (r >> 5)] |= 1u << (31 - (r & 31));
}

/* kernel.h:173-201: B[r][c] += ((word >> (31-lane)) & 1) << p into a zeroed tensor. */
void qo_unpack_rows(const uint32_t *bits, int nbits, int H, int W, int32_t *out) {
    const size_t row_w = (size_t)qo_step128(W) * 4u;
    const size_t plane_w = (size_t)qo_pad8(H) * row_w;
    for (int r = 0; r < H; r++)
        for (int c = 0; c < W; c++) {
            uint32_t v = 0;
            for (int p = 0; p < nbits; p++)
                v += ((bits[p * plane_w + (size_t)r * row_w + (c >> 5)] >> (31 - (c & 31))) & 1u)
                     << p;
            out[(size_t)r * W + c] = (int32_t)v;
        }
}

/* kernel.h:109-139 (same stride caveat as qo_cols_words for output_layer). */
void qo_unpack_cols(const uint32_t *bits, int nbits, int H, int W, int output_layer,
                    int32_t *out) {
    const size_t line_w = (size_t)qo_step128(H) * 4u;
    const size_t plane_w = line_w * (size_t)cols_lines(W, output_layer);
    for (int r = 0; r < H; r++)
        for (int c = 0; c < W; c++) {
            uint32_t v = 0;
            for (int p = 0; p < nbits; p++)
                v += ((bits[p * plane_w + (size_t)c * line_w + (r >> 5)] >> (31 - (r & 31))) & 1u)
                     << p;
            out[(size_t)r * W + c] = (int32_t)v;
        }
}

/* QGTC_device.cu:44-130 */
void qo_val2bit(const float *x, int H, int W, int nbits, int col_major, int output_layer,
                uint32_t *out) {
    /* fused quantise+pack, plane by plane, without the int32 temporary of :63 */
    if (col_major) {
        const size_t line_w = (size_t)qo_step128(H) * 4u;
        const size_t plane_w = line_w * (size_t)cols_lines(W, output_layer);
        memset(out, 0, sizeof(uint32_t) * plane_w * (size_t)nbits);
        for (int r = 0; r < H; r++)
            for (int c = 0; c < W; c++) {
                uint32_t q = (uint32_t)qo_quantize_one(x[(size_t)r * W + c], nbits);
                for (int p = 0; p < nbits; p++)
                    if ((q >> p) & 1u)
                        out[p * plane_w + (size_t)c * line_w + (r >> 5)] |= 1u << (31 - (r & 31));
            }
    } else {
        const size_t row_w = (size_t)qo_step128(W) * 4u;
        const size_t plane_w = (size_t)qo_pad8(H) * row_w;
        memset(out, 0, sizeof(uint32_t) * plane_w * (size_t)nbits);
        for (int r = 0; r < H; r++)
            for (int c = 0; c < W; c++) {
                uint32_t q = (uint32_t)qo_quantize_one(x[(size_t)r * W + c], nbits);
                for (int p = 0; p < nbits; p++)
                    if ((q >> p) & 1u)
                        out[p * plane_w + (size_t)r * row_w + (c >> 5)] |= 1u << (31 - (c & 31));
            }
    }
}

/* QGTC_device.cu:135-206 */
void qo_bit2val(const uint32_t *bits, int nbits, int H, int W, int col_major, int output_layer,
                int32_t *out) {
    if (col_major) qo_unpack_cols(bits, nbits, H, W, output_layer, out);
    else qo_unpack_rows(bits, nbits, H, W, out);
}

/* ---- the bit-GEMM -------------------------------------------------------------------- */
static inline uint32_t ld(const uint32_t *p, size_t n, size_t i) { return i < n ? p[i] : 0u; }

/* kernel.h:292-341. For plane pair (pa,pw) the bmma_sync(..., bmmaBitOpAND) chain over the
 * gdk=STEP128(K) k-steps (:301-308) is  sum_j popc(X[pa][m][j] & Wt[pw][n][j]);
 * :340 folds it in as  c += tmp << (pa+pw)  in 32-bit two's-complement arithmetic.
 * Offsets: act_offset = PAD8(M)*STEP128(K)*4 (:265), w_offset = STEP128(K)*w_lines*4
 * (:266 / :836 / :958), row and line stride gdk*4 words (:306-307). */
void qo_acc(const uint32_t *X, size_t x_words, const uint32_t *Wt, size_t w_words,
            int M, int K, int N, int a, int w, int w_lines, int32_t *acc) {
    const size_t kw = (size_t)qo_step128(K) * 4u;
    const size_t x_plane = (size_t)qo_pad8(M) * kw;
    const size_t w_plane = (size_t)w_lines * kw;
    const int fast = x_plane * (size_t)a <= x_words && w_plane * (size_t)(w - 1) +
                     (size_t)N * kw <= w_words;
#pragma omp parallel for schedule(static)
    for (int m = 0; m < M; m++) {
        for (int n = 0; n < N; n++) {
            uint32_t c = 0;
            for (int pw = 0; pw < w; pw++) {
                for (int pa = 0; pa < a; pa++) {
                    const size_t xo = (size_t)pa * x_plane + (size_t)m * kw;
                    const size_t wo = (size_t)pw * w_plane + (size_t)n * kw;
                    uint32_t t = 0;
                    if (fast) {
                        const uint32_t *xr = X + xo, *wr = Wt + wo;
                        size_t j = 0;
                        for (; j + 2 <= kw; j += 2) {
                            uint64_t xv, wv;
                            memcpy(&xv, xr + j, 8);
                            memcpy(&wv, wr + j, 8);
                            t += (uint32_t)__builtin_popcountll(xv & wv);
                        }
                        for (; j < kw; j++) t += (uint32_t)__builtin_popcount(xr[j] & wr[j]);
                    } else {
                        for (size_t j = 0; j < kw; j++)
                            t += (uint32_t)__builtin_popcount(ld(X, x_words, xo + j) &
                                                               ld(Wt, w_words, wo + j));
                    }
                    const int s = pa + pw; /* :295 b_opt */
                    c += (s < 32) ? (t << s) : 0u;
                }
            }
            acc[(size_t)m * N + n] = (int32_t)c;
        }
    }
}

/* kernel.h:31-37, called at :350 as quantize(c, out_bit, 1<<out_bit, 0):
 *   float val = c;  if (val > max) val = max-1;  if (val < min) val = min+1;
 *   ans = (int)((val-min) * (1<<bitwidth) / (max-min))
 * With min=0, max=2^ob the scale factor cancels exactly, so ans = (int)val: values
 * 0..2^ob pass through (2^ob itself is kept; its low ob bits are 0), larger ones become
 * 2^ob-1, negative ones (only reachable by int32 wrap-around) become 1. */
int32_t qo_requant(int32_t c, int out_bit) {
    const float maxv = ldexpf(1.0f, out_bit);
    float val = (float)c;
    if (val > maxv) val = maxv - 1.0f;
    if (val < 0.0f) val = 1.0f;
    float scaled = val * maxv / maxv;
    if (scaled >= 2147483648.0f) return INT32_MAX; /* cvt saturates; unreachable for ob<=30 */
    return (int32_t)scaled;
}

/* kernel.h:245-391. Output = pack_rows(requant(acc)) restricted to m<M, n<N (:367-372), shape
 * {ob*PAD8(M), STEP128(N)*4} (QGTC_device.cu:223). The byte stores at :386-387
 * (Cb[(bx*8+r)*gdm*16 + FLIPBITS(by,2)] = byte 3-.. of the brev'd ballot) put column
 * by*8+g of row r at bit 31-((by*8+g)&31) of word (by*8+g)>>5 — the pack_rows closed form
 * (checked by oracle/warp_emulation.py). */
void qo_bitmm2bit(const uint32_t *X, size_t x_words, const uint32_t *Wt, size_t w_words,
                  int M, int K, int N, int a, int w, int ob, uint32_t *out) {
    int32_t *acc = (int32_t *)malloc(sizeof(int32_t) * (size_t)M * (size_t)N + 4);
    qo_acc(X, x_words, Wt, w_words, M, K, N, a, w, qo_pad128(N), acc);
    for (size_t i = 0; i < (size_t)M * (size_t)N; i++) acc[i] = qo_requant(acc[i], ob);
    qo_pack_rows(acc, M, N, ob, out);
    free(acc);
}

/* kernel.h:651-810, intended semantics: the result re-packed in the cols
 * layout {ob*STEP128(M)*4, PAD128(N)} (QGTC_device.cu:456) so that it can be the right-hand
 * operand of the next product. (The reference's second half-tile read at :782 uses +4
 * instead of +32 and its bounds tests at :777-778 are not transposed; only all-equal inputs
 * hide that, so there is nothing well-defined to match beyond the intent.) */
void qo_bitmm2bit_col(const uint32_t *X, size_t x_words, const uint32_t *Wt, size_t w_words,
                      int M, int K, int N, int a, int w, int ob, uint32_t *out) {
    int32_t *acc = (int32_t *)malloc(sizeof(int32_t) * (size_t)M * (size_t)N + 4);
    qo_acc(X, x_words, Wt, w_words, M, K, N, a, w, qo_pad128(N), acc);
    for (size_t i = 0; i < (size_t)M * (size_t)N; i++) acc[i] = qo_requant(acc[i], ob);
    qo_pack_cols(acc, M, N, ob, 0, out);
    free(acc);
}

/* kernel.h:816-932 (PAD8: w_offset = STEP128(K)*PAD8(N)*4, :836) and :938-1054 (PAD128,
 * :958); no clamp; :925-926 store float(C) to the dense [M,N] output. */
void qo_bitmm2int(const uint32_t *X, size_t x_words, const uint32_t *Wt, size_t w_words,
                  int M, int K, int N, int a, int w, int pad_128, float *out) {
    int32_t *acc = (int32_t *)malloc(sizeof(int32_t) * (size_t)M * (size_t)N + 4);
    qo_acc(X, x_words, Wt, w_words, M, K, N, a, w, pad_128 ? qo_pad128(N) : qo_pad8(N), acc);
    for (size_t i = 0; i < (size_t)M * (size_t)N; i++) out[i] = (float)acc[i];
    free(acc);
}

/* kernel.h:452 atomicAdd(&counter_global,1) per (8x8 output tile, plane pair, k-step);
 * kernel.h:574-592 atomicAdd(&counter,1) only for steps whose 8x128-bit X tile has a set bit.
 * The reference probes rows with a wrong stride (:581 laneid*gdk*128 ints); the oracle states
 * the intended test — the tile's own 8 rows. */
void qo_tile_counters(const uint32_t *X, size_t x_words, int M, int K, int N, int a, int w,
                      uint64_t *total, uint64_t *nonzero) {
    const size_t gdx = (size_t)qo_step8(M), gdy = (size_t)qo_step8(N), gdk = (size_t)qo_step128(K);
    const size_t kw = gdk * 4u;
    const size_t x_plane = (size_t)qo_pad8(M) * kw;
    uint64_t nz = 0;
    for (int pa = 0; pa < a; pa++)
        for (size_t bx = 0; bx < gdx; bx++)
            for (size_t i = 0; i < gdk; i++) {
                uint32_t any = 0;
                for (size_t r = 0; r < 8; r++)
                    for (size_t q = 0; q < 4; q++)
                        any |= ld(X, x_words, (size_t)pa * x_plane + (bx * 8 + r) * kw + i * 4 + q);
                if (any) nz++;
            }
    *total = (uint64_t)(gdx * gdy * gdk) * (uint64_t)a * (uint64_t)w;
    *nonzero = nz * (uint64_t)gdy * (uint64_t)w;
}
