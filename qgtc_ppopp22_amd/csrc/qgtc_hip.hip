// qgtc_hip.hip — MI355X (gfx950 / CDNA4) kernels and C-ABI of the QGTC bit-GEMM hot path.
//
// What is here (reference file:line each piece replaces is in include/qgtc.h):
//   * fused quantise + bit-plane pack          (val2bit, rows and cols layouts)
//   * bit-plane unpack                         (bit2val)
//   * adjacency bit planes straight from an edge list (pack_edges), occupancy bitmaps
//   * multi-plane 1-bit GEMM: AND + popcount (v_and_b32 / v_bcnt_u32_b32) with shift-accumulate
//     into int32, in-workgroup split-K, zero-tile skipping / jumping, and a fused epilogue that
//     either re-quantises and re-packs (rows / cols layout) or converts to float32
//   * tile counters, the 200-rep profile loop, a grouped (batched) launch
//   * the int8 MFMA comparison GEMM.
//
// Design notes live in DESIGN.md; the short version of the bit-GEMM kernel is the comment block
// above `mm_tile` ("the bit-GEMM"): 32 x 32 output tile per workgroup, waves split K and stream
// their own operand slices (no barrier in the main loop), buffer loads with hardware range checks,
// 4 x 4 register micro-tile per lane, one LDS reduction, DPP-packed epilogue.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <type_traits>

#include "qgtc.h"

namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr int TM = 32, TN = 32;  // workgroup tile of the bit-GEMM

// ------------------------------------------------------------------------------------------
// shape algebra (reference utility.h:33-45)
// ------------------------------------------------------------------------------------------
__host__ __device__ constexpr int step8(int x) { return (x + 7) >> 3; }
__host__ __device__ constexpr int step128(int x) { return (x + 127) >> 7; }
__host__ __device__ constexpr int pad8(int x) { return step8(x) << 3; }
__host__ __device__ constexpr int pad128(int x) { return step128(x) << 7; }

thread_local char g_hip_err[256] = "";

int hip_fail(hipError_t e, const char *where) {
    snprintf(g_hip_err, sizeof(g_hip_err), "%s: %s", where, hipGetErrorString(e));
    return QGTC_EHIP;
}
#define HIP_TRY(expr)                                        \
    do {                                                     \
        hipError_t e_ = (expr);                              \
        if (e_ != hipSuccess) return hip_fail(e_, #expr);    \
    } while (0)

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
inline bool bits_ok(int b) { return b >= 1 && b <= 32; }

// bounds-safe word / granule loads: indices past the buffer read as zero
__device__ __forceinline__ uint32_t ldw(const uint32_t *__restrict__ p, unsigned long long n,
                                        unsigned long long i) {
    return i < n ? p[i] : 0u;
}
__device__ __forceinline__ uint4 ldg4(const uint32_t *__restrict__ p, unsigned long long n,
                                      unsigned long long i) {
    if (i + 4 <= n) return *reinterpret_cast<const uint4 *>(p + i);
    return make_uint4(ldw(p, n, i), ldw(p, n, i + 1), ldw(p, n, i + 2), ldw(p, n, i + 3));
}

// ------------------------------------------------------------------------------------------
// quantisation (reference kernel.h:39-44 clip, :68 __float2int_rn)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t quant1(float x, float ub, float ubm1) {
    float y = x;
    if (x < 0.0f) y = 1.0f;       // negative -> lb + 1
    else if (x > ub) y = ubm1;    // above 2^b -> 2^b - 1 (float arithmetic)
    if (y != y) return 0u;        // NaN converts to 0
    const float r = rintf(y);     // v_rndne_f32: round-half-to-even
    return r >= 4294967296.0f ? 0u : static_cast<uint32_t>(r);  // low 32 bits (nbits >= 31 only)
}

// OR over aligned groups of 8 lanes (every lane of the wave must be active)
__device__ __forceinline__ uint32_t or_reduce8(uint32_t x) {
    x |= static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), 0xB1, 0xf, 0xf, false));   // lane ^ 1
    x |= static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), 0x4E, 0xf, 0xf, false));   // lane ^ 2
    x |= static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), 0x141, 0xf, 0xf, false));  // 7 - lane
    return x;
}

// ------------------------------------------------------------------------------------------
// val2bit, rows layout, fast path (W % 4 == 0, 16-byte aligned input): HBM-streaming.
// A wave packs 256 columns of one row per unit: every lane loads one float4 (16 B/lane, 1 KiB per
// wave-instruction, fully coalesced), builds the nibble of its four columns per plane, and eight
// adjacent lanes OR their nibbles into one output word with DPP (lane 8k stores word k of the
// unit: 32 contiguous bytes per plane). UNROLL units are loaded before any is packed, so a wave
// keeps UNROLL KiB in flight. Every word of the padded output is written.
// ------------------------------------------------------------------------------------------
template <int UNROLL>
__global__ __launch_bounds__(256) void k_val2bit_rows_v4(const float *__restrict__ x, int H, int W,
                                                         int nbits, float ub, float ubm1,
                                                         uint32_t *__restrict__ out, int rows_pad,
                                                         int row_words) {
    const int lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
    const int chunks = (row_words + 7) >> 3;  // 256-column units per row
    const uint32_t units = static_cast<uint32_t>(rows_pad) * chunks;  // < 2^31 (host-checked)
    const size_t plane = static_cast<size_t>(rows_pad) * row_words;
    const uint32_t sh_n = 28 - 4 * (lane & 7);
    for (uint32_t u0 = wave * UNROLL; u0 < units; u0 += nwaves * UNROLL) {
        float4 v[UNROLL];
#pragma unroll
        for (int k = 0; k < UNROLL; k++) {
            const uint32_t u = u0 + k;
            const int r = static_cast<int>(u / chunks), ch = static_cast<int>(u % chunks);
            const int c = ch * 256 + lane * 4;
            v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (u < units && r < H && c < W)  // W % 4 == 0: the quad is entirely inside or outside
                v[k] = *reinterpret_cast<const float4 *>(x + static_cast<size_t>(r) * W + c);
        }
#pragma unroll
        for (int k = 0; k < UNROLL; k++) {
            const uint32_t u = u0 + k;
            if (u >= units) break;  // wave-uniform
            const int r = static_cast<int>(u / chunks), ch = static_cast<int>(u % chunks);
            const int c = ch * 256 + lane * 4;
            const bool in = r < H && c < W;
            const uint32_t q0 = in ? quant1(v[k].x, ub, ubm1) : 0u, q1 = in ? quant1(v[k].y, ub, ubm1) : 0u;
            const uint32_t q2 = in ? quant1(v[k].z, ub, ubm1) : 0u, q3 = in ? quant1(v[k].w, ub, ubm1) : 0u;
            const int wi = ch * 8 + (lane >> 3);
            uint32_t *dst = out + static_cast<size_t>(r) * row_words + wi;
            for (int p = 0; p < nbits; p++, dst += plane) {
                const uint32_t nib = (((q0 >> p) & 1u) << 3) | (((q1 >> p) & 1u) << 2) |
                                     (((q2 >> p) & 1u) << 1) | ((q3 >> p) & 1u);
                const uint32_t word = or_reduce8(nib << sh_n);
                if ((lane & 7) == 0 && wi < row_words) *dst = word;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// val2bit, rows layout: out[p][r][c>>5] bit(31-(c&31)) = bit p of quant(x[r][c])
// One wave per (row, 256-column chunk): 4 coalesced loads per lane, one 64-bit ballot per
// (plane, load), two bit-reversed words per ballot; lanes 0..7 store the chunk's 8 words.
// Every word of the padded output is written.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_val2bit_rows(const float *__restrict__ x, int H, int W,
                                                      int nbits, float ub, float ubm1,
                                                      uint32_t *__restrict__ out, int rows_pad,
                                                      int row_words) {
    const int lane = threadIdx.x & 63;
    const long wave = (static_cast<long>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
    const long nwaves = (static_cast<long>(gridDim.x) * blockDim.x) >> 6;
    const int chunks = (row_words + 7) >> 3;
    const long units = static_cast<long>(rows_pad) * chunks;
    const size_t plane = static_cast<size_t>(rows_pad) * row_words;
    for (long u = wave; u < units; u += nwaves) {
        const int r = static_cast<int>(u / chunks);
        const int ch = static_cast<int>(u % chunks);
        uint32_t q[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int c = ch * 256 + i * 64 + lane;
            q[i] = (r < H && c < W) ? quant1(x[static_cast<size_t>(r) * W + c], ub, ubm1) : 0u;
        }
        const int wi = ch * 8 + lane;  // word this lane stores (lanes 0..7)
        for (int p = 0; p < nbits; p++) {
            unsigned long long m[4];
#pragma unroll
            for (int i = 0; i < 4; i++) m[i] = __ballot((q[i] >> p) & 1u);
            const int sel = (lane >> 1) & 3;
            const unsigned long long mm = sel == 0 ? m[0] : sel == 1 ? m[1] : sel == 2 ? m[2] : m[3];
            const uint32_t half = (lane & 1) ? static_cast<uint32_t>(mm >> 32) : static_cast<uint32_t>(mm);
            if (lane < 8 && wi < row_words)
                out[p * plane + static_cast<size_t>(r) * row_words + wi] = __brev(half);
        }
    }
}

// ------------------------------------------------------------------------------------------
// val2bit, cols layout: out[p][c][r>>5] bit(31-(r&31)) = bit p of quant(x[r][c])
// One wave per (64-column chunk, 32-row group): lane = column, 32 coalesced row reads, each
// lane assembles its column's word per plane in registers. NB = compile-time bound on nbits.
// ------------------------------------------------------------------------------------------
template <int NB>
__global__ __launch_bounds__(256) void k_val2bit_cols(const float *__restrict__ x, int H, int W,
                                                      int nbits, float ub, float ubm1,
                                                      uint32_t *__restrict__ out, int lines,
                                                      int line_words) {
    const int lane = threadIdx.x & 63;
    const long wave = (static_cast<long>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
    const long nwaves = (static_cast<long>(gridDim.x) * blockDim.x) >> 6;
    const int cchunks = (lines + 63) >> 6;
    const long units = static_cast<long>(cchunks) * line_words;
    const size_t plane = static_cast<size_t>(lines) * line_words;
    for (long u = wave; u < units; u += nwaves) {
        const int cg = static_cast<int>(u % cchunks);
        const int rw = static_cast<int>(u / cchunks);
        const int c = cg * 64 + lane;
        uint32_t wd[NB];
#pragma unroll
        for (int p = 0; p < NB; p++) wd[p] = 0u;
#pragma unroll 8
        for (int rr = 0; rr < 32; rr++) {
            const int r = rw * 32 + rr;
            const uint32_t q =
                (r < H && c < W) ? quant1(x[static_cast<size_t>(r) * W + c], ub, ubm1) : 0u;
#pragma unroll
            for (int p = 0; p < NB; p++) wd[p] |= ((q >> p) & 1u) << (31 - rr);
        }
        if (c < lines) {
#pragma unroll
            for (int p = 0; p < NB; p++)
                if (p < nbits) out[p * plane + static_cast<size_t>(c) * line_words + rw] = wd[p];
        }
    }
}

// ------------------------------------------------------------------------------------------
// bit2val (reference kernel.h:109-139, :173-201): one thread per output element
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_bit2val(const uint32_t *__restrict__ bits,
                                                 unsigned long long words, int nbits, int H, int W,
                                                 int col_major, size_t plane, int line_words,
                                                 int32_t *__restrict__ out) {
    const size_t total = static_cast<size_t>(H) * W;
    for (size_t idx = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; idx < total;
         idx += static_cast<size_t>(gridDim.x) * blockDim.x) {
        const int r = static_cast<int>(idx / W), c = static_cast<int>(idx % W);
        const int line = col_major ? c : r, pos = col_major ? r : c;
        uint32_t v = 0;
        for (int p = 0; p < nbits; p++) {
            const uint32_t wd =
                ldw(bits, words, p * plane + static_cast<size_t>(line) * line_words + (pos >> 5));
            v += ((wd >> (31 - (pos & 31))) & 1u) << p;
        }
        out[idx] = static_cast<int32_t>(v);
    }
}

// ------------------------------------------------------------------------------------------
// Adjacency bit planes straight from an edge list (the packing sampler.py:80-101 does through a
// dense float n x n matrix: A[src][dst] += 1 per edge, then val2bit(A, nbits, rows layout)).
// One thread per DISTINCT (row, col) cell with its multiplicity: the cell's value is quantised
// exactly as quant1 would (count > 2^b -> 2^b - 1, so e.g. with b = 1 a doubled edge packs as 0)
// and its set planes are OR-ed into the zero-initialised output.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pack_edges(const int64_t *__restrict__ cells,
                                                    const int32_t *__restrict__ counts, size_t n_cells,
                                                    int H, int W, int nbits, float ub, float ubm1,
                                                    uint32_t *__restrict__ out, int rows_pad, int row_words) {
    const size_t plane = static_cast<size_t>(rows_pad) * row_words;
    for (size_t e = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; e < n_cells;
         e += static_cast<size_t>(gridDim.x) * blockDim.x) {
        const int64_t cell = cells[e];
        if (cell < 0) continue;
        const int64_t r = cell / W, c = cell % W;
        if (r >= H) continue;
        const uint32_t q = quant1(static_cast<float>(counts ? counts[e] : 1), ub, ubm1);
        const uint32_t bit = 1u << (31 - (c & 31));
        uint32_t *wd = out + static_cast<size_t>(r) * row_words + (c >> 5);
        for (int p = 0; p < nbits; p++)
            if ((q >> p) & 1u) atomicOr(wd + p * plane, bit);
    }
}

// ------------------------------------------------------------------------------------------
// int8 MFMA GEMM, the comparison path (analogue of the reference's cuBLASGemmEX benchmark,
// cublas_main.cu:123-172): C[M,N] (float32) = A[M,K] (int8, K contiguous) x B[K,N] given as
// Bt[N,K] (int8, K contiguous), int32 accumulation on v_mfma_i32_16x16x64_i8, exact.
// A workgroup owns a 16-row x 64-column tile; its waves split K (each wave streams its slice of
// the A rows and B lines straight into MFMA fragments, 16 B per lane per load) and are summed
// through LDS. Out-of-range rows / columns read as zero through buffer range checks.
// ------------------------------------------------------------------------------------------
typedef int i32x4 __attribute__((ext_vector_type(4)));
constexpr int I8_TM = 16, I8_TN = 64, I8_WAVES = 8;

__global__ __launch_bounds__(64 * I8_WAVES) void k_i8gemm(const int8_t *__restrict__ A,
                                                          const int8_t *__restrict__ Bt, int M, int K,
                                                          int N, float *__restrict__ C, int tiles_n) {
    __shared__ int red[I8_WAVES][4][4][64];  // [wave][column sub-tile][acc register][lane]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
    const int m0 = tm * I8_TM, n0 = tn * I8_TN;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<int8_t *>(A), 0, static_cast<int>(static_cast<uint32_t>(static_cast<size_t>(M) * K)), 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<int8_t *>(Bt), 0, static_cast<int>(static_cast<uint32_t>(static_cast<size_t>(N) * K)), 0x00020000);
    // fragment maps of mfma_i32_16x16x64_i8: lane l holds A[row l&15][k = 16*(l>>4) + 0..15] and
    // B[k = 16*(l>>4) + 0..15][col l&15]; C/D: col = l&15, row = 4*(l>>4) + reg
    const int fr = lane & 15, fk = (lane >> 4) * 16;
    const int ksteps = (K + 63) / 64;
    const int per = (ksteps + I8_WAVES - 1) / I8_WAVES;
    const int s0 = wv * per, s1 = min(s0 + per, ksteps);
    const bool row_ok = m0 + fr < M;
    uint32_t a_off = static_cast<uint32_t>(m0 + fr) * K + fk;
    uint32_t b_off[4];
    bool col_ok[4];
#pragma unroll
    for (int c = 0; c < 4; c++) {
        col_ok[c] = n0 + 16 * c + fr < N;
        b_off[c] = static_cast<uint32_t>(n0 + 16 * c + fr) * K + fk;
    }
    i32x4 acc[4];
#pragma unroll
    for (int c = 0; c < 4; c++) acc[c] = i32x4{0, 0, 0, 0};
    // Loads are unconditional: a k-step beyond this wave's slice (or beyond K) loads from offset
    // 0xffffffff, which the range check turns into zeros without touching memory, so the loop
    // has no branches around loads and hipcc can count vmcnt exactly. Four k-steps in flight.
    auto load = [&](int s, i32x4 &af, i32x4 (&bf)[4]) {
        const bool k_ok = s < s1 && s * 64 + fk < K;  // K % 16 == 0: 16-byte groups are all-in or all-out
        af = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                           ra, (row_ok && k_ok) ? a_off + s * 64 : 0xffffffffu, 0, 0));
#pragma unroll
        for (int c = 0; c < 4; c++)
            bf[c] = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                  rb, (col_ok[c] && k_ok) ? b_off[c] + s * 64 : 0xffffffffu, 0, 0));
    };
    auto mac = [&](const i32x4 &af, const i32x4 (&bf)[4]) {
#pragma unroll
        for (int c = 0; c < 4; c++) acc[c] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af, bf[c], acc[c], 0, 0, 0);
    };
    i32x4 a0, b0[4], a1, b1[4], a2, b2[4], a3, b3[4];
    load(s0, a0, b0);
    load(s0 + 1, a1, b1);
    load(s0 + 2, a2, b2);
    for (int s = s0; s < s1; s += 4) {
        load(s + 3, a3, b3);
        mac(a0, b0);
        load(s + 4, a0, b0);
        mac(a1, b1);
        load(s + 5, a1, b1);
        mac(a2, b2);
        load(s + 6, a2, b2);
        mac(a3, b3);
    }
#pragma unroll
    for (int c = 0; c < 4; c++)
#pragma unroll
        for (int r = 0; r < 4; r++) red[wv][c][r][lane] = acc[c][r];
    __syncthreads();
    // 16 x 64 outputs, 1024 (sub-tile, register, lane) slots over 512 threads
    for (int e = tid; e < 4 * 4 * 64; e += 64 * I8_WAVES) {
        const int l = e & 63, r = (e >> 6) & 3, c = e >> 8;
        int v = 0;
#pragma unroll
        for (int k = 0; k < I8_WAVES; k++) v += red[k][c][r][l];
        const int m = m0 + 4 * (l >> 4) + r, n = n0 + 16 * c + (l & 15);
        if (m < M && n < N) C[static_cast<size_t>(m) * N + n] = static_cast<float>(v);
    }
}

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

// ------------------------------------------------------------------------------------------
// the bit-GEMM
//
// Decomposition. A workgroup owns a 32 x 32 output tile for the whole K range, so no reduction
// ever crosses workgroups and the int32 sums are exact in any order. Its waves split K: wave v
// owns the k-quads (128-bit steps of a packed row) [v*per, v*per + per). Everything a wave
// multiplies is private to it: it loads its own slice of the X rows and W lines, stages it in
// its own LDS region and reads it back in the micro-tile pattern, so the main loop has NO
// workgroup barrier; the waves only meet once, to sum their 32 x 32 partial tiles through LDS.
//
// Per stage a wave holds QW k-quads of `ab` X planes and `wb` W planes:
//   global -> registers   raw buffer loads (hardware range check: any dword outside the stated
//                         extent reads as 0, so mis-sized / mis-laid operands can never fault),
//                         lane = (line, k-quad) with the k-quads of one packed row in adjacent
//                         lanes: every load instruction touches whole 16/32/64-byte runs;
//   registers -> LDS      ds_write_b128 into [plane][k-quad][line] (pitch RS granules: the 8
//                         lanes of one write group land in 8 different bank quads);
//   LDS -> registers      ds_read_b128: the 8 distinct granules a wave reads per instruction are
//                         contiguous, so reads are conflict-free and broadcast to 8 lanes each;
//   the loads of stage s+1 are in flight while stage s is multiplied.
// Each lane keeps a 4 x 4 register micro-tile (rows lm + 8i, columns ln + 8j) and spends, per
// k-quad and plane pair, 8 granule reads on 64 v_and_b32 + 64 v_bcnt_u32_b32 (accumulate form).
//
// Zero-tile skipping. While a stage is still in registers the wave ORs each X granule and
// ballots: one scalar bit per (X plane, k-quad) says whether the 32-row x 128-bit tile has any
// bit set. All-zero tiles are skipped with a scalar branch (no divergence, no extra VALU work).
// ------------------------------------------------------------------------------------------
#ifdef QGTC_STAMPS  // diagnostic build only (tools/kbench.hip): per-phase s_memtime stamps
// The stamps stay in scalar registers while the kernel runs (a store per stamp would put memory
// traffic and waits into the phases being timed); wave 0 of each workgroup writes them out at the end.
__device__ unsigned long long g_stamps[1024 * 16];
struct Stamps {
    unsigned long long t[16];
};
#define STAMP_DECL Stamps stamps_; for (int i_ = 0; i_ < 16; i_++) stamps_.t[i_] = 0ull
#define STAMP(slot) stamps_.t[slot] = __builtin_amdgcn_s_memtime()
#define STAMP_FLUSH()                                                                          \
    do {                                                                                       \
        if (threadIdx.x == 0)                                                                  \
            for (int i_ = 0; i_ < 16; i_++) g_stamps[blockIdx.x % 1024 * 16 + i_] = stamps_.t[i_]; \
    } while (0)
#define STAMP_ARG , Stamps &stamps_
#define STAMP_PASS , stamps_
#else
#define STAMP_DECL do { } while (0)
#define STAMP(slot) do { } while (0)
#define STAMP_FLUSH() do { } while (0)
#define STAMP_ARG
#define STAMP_PASS
#endif


struct MMShape {           // per-launch constants
    int a, w, ob;          // planes of X, planes of W, output planes
    int mode;              // 0 rows-layout bits, 1 cols-layout bits, 2 float32
    int ab, wb;            // planes staged at once (generic kernel; the fixed kernels stage all)
    int per;               // k-quads per wave (in-workgroup split-K slice)
    int waves;             // waves per workgroup (= blockDim.x / 64, passed so that no hidden argument is read)
    uint32_t inv_tiles_n;  // floor(2^32 / tiles_n), single launches only (tiles_n >= 2; else 0xffffffff)
    float maxv, maxm1;     // 2^ob and 2^ob - 1 as float (requant)
};

constexpr int MR = 4, MC = 4;        // per-lane micro-tile
constexpr int GPT = 8;               // granules (16 B) a lane may hold per stage
constexpr int SLAB_PITCH = 72;       // ints between the (i,j) planes of a wave's partial tile
constexpr int SLAB_BYTES = MR * MC * SLAB_PITCH * 4;
constexpr int MAX_WAVES = 8;

// granule pitch of one (plane, k-quad) line block in LDS
__host__ __device__ constexpr int lds_pitch(int qw) { return qw == 4 ? 34 : (qw == 2 ? 36 : 32); }
// slots (one 16-byte load per lane each) that `planes` plane tiles of QW k-quads need
__host__ __device__ constexpr int slots_for(int planes, int qw) { return (planes * qw + 1) / 2; }
// bytes of one wave's staging region
__host__ __device__ constexpr size_t region_bytes(int planes, int qw) {
    return static_cast<size_t>(planes) * qw * lds_pitch(qw) * 16;
}

// acc[i][j] += popcount(x[i] & w[j]) for two X words and four W words: 8 v_and_b32 into
// temporaries, then 8 v_bcnt_u32_b32 with the accumulator as the add operand. Written as one asm
// block because hipcc (a) turns __popc(a & b) + c into v_bcnt(..., 0) + v_add3 (2.5 instructions
// per pair instead of 2) and (b) likes to issue each v_bcnt right behind the v_and it depends on,
// which costs a dependent-issue bubble per pair; here every v_bcnt is 8 instructions behind.
__device__ __forceinline__ void and_popc_2x4(uint32_t &a00, uint32_t &a01, uint32_t &a02, uint32_t &a03,
                                             uint32_t &a10, uint32_t &a11, uint32_t &a12, uint32_t &a13,
                                             uint32_t x0, uint32_t x1, uint32_t w0, uint32_t w1,
                                             uint32_t w2, uint32_t w3) {
    uint32_t t0, t1, t2, t3, t4, t5, t6, t7;
    asm("v_and_b32 %8, %16, %18\n\tv_and_b32 %9, %16, %19\n\tv_and_b32 %10, %16, %20\n\tv_and_b32 %11, %16, %21\n\t"
        "v_and_b32 %12, %17, %18\n\tv_and_b32 %13, %17, %19\n\tv_and_b32 %14, %17, %20\n\tv_and_b32 %15, %17, %21\n\t"
        "v_bcnt_u32_b32 %0, %8, %0\n\tv_bcnt_u32_b32 %1, %9, %1\n\tv_bcnt_u32_b32 %2, %10, %2\n\tv_bcnt_u32_b32 %3, %11, %3\n\t"
        "v_bcnt_u32_b32 %4, %12, %4\n\tv_bcnt_u32_b32 %5, %13, %5\n\tv_bcnt_u32_b32 %6, %14, %6\n\tv_bcnt_u32_b32 %7, %15, %7"
        : "+v"(a00), "+v"(a01), "+v"(a02), "+v"(a03), "+v"(a10), "+v"(a11), "+v"(a12), "+v"(a13),
          "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7)
        : "v"(x0), "v"(x1), "v"(w0), "v"(w1), "v"(w2), "v"(w3));
}

// one k-quad of the 4 x 4 micro-tile: 64 AND + 64 BCNT
__device__ __forceinline__ void mac_quad(uint32_t (&acc)[MR][MC], const u32x4 (&xg)[MR],
                                         const u32x4 (&wg)[MC]) {
#define QGTC_MAC_WORD(c)                                                                            \
    and_popc_2x4(acc[0][0], acc[0][1], acc[0][2], acc[0][3], acc[1][0], acc[1][1], acc[1][2], acc[1][3], \
                 xg[0].c, xg[1].c, wg[0].c, wg[1].c, wg[2].c, wg[3].c);                             \
    and_popc_2x4(acc[2][0], acc[2][1], acc[2][2], acc[2][3], acc[3][0], acc[3][1], acc[3][2], acc[3][3], \
                 xg[2].c, xg[3].c, wg[0].c, wg[1].c, wg[2].c, wg[3].c);
    QGTC_MAC_WORD(x)
    QGTC_MAC_WORD(y)
    QGTC_MAC_WORD(z)
    QGTC_MAC_WORD(w)
#undef QGTC_MAC_WORD
}

__device__ __forceinline__ int requant(int c, float maxv, float maxm1) {
    // reference kernel.h:31-37 called as quantize(c, ob, 1<<ob, 0): float compare, then the
    // (val-min)*2^ob/(max-min) scaling, which is the identity for min=0, max=2^ob.
    float val = static_cast<float>(c);
    if (val > maxv) val = maxm1;
    if (val < 0.0f) val = 1.0f;
    return val >= 2147483648.0f ? 2147483647 : static_cast<int>(val);
}

// slot u of an operand, lane l  ->  (plane tile, line within the 32-line tile, k-quad of the chunk)
template <int QW>
__device__ __forceinline__ void slot_map(int u, int lane, int &pt, int &line, int &kk) {
    if (QW == 4) {
        pt = u >> 1;
        line = ((u & 1) << 4) + (lane >> 2);
        kk = lane & 3;
    } else if (QW == 2) {
        pt = u;
        line = lane >> 1;
        kk = lane & 1;
    } else {
        pt = 2 * u + (lane >> 5);
        line = lane & 31;
        kk = 0;
    }
}

// occupancy bits (bit kk = "k-quad kk of plane tile pt has a set bit") from the ballots of the
// slots that hold the tile
template <int QW>
__device__ __forceinline__ uint32_t tile_occupancy(const unsigned long long (&nzm)[GPT], int pt) {
    if (QW == 4) {
        const unsigned long long m = nzm[2 * pt] | nzm[2 * pt + 1];
        uint32_t o = 0;
#pragma unroll
        for (int kk = 0; kk < 4; kk++) o |= (m & (0x1111111111111111ull << kk)) ? (1u << kk) : 0u;
        return o;
    } else if (QW == 2) {
        const unsigned long long m = nzm[pt];
        return ((m & 0x5555555555555555ull) ? 1u : 0u) | ((m & 0xaaaaaaaaaaaaaaaaull) ? 2u : 0u);
    } else {
        const unsigned long long m = nzm[pt >> 1];
        return ((pt & 1) ? (m >> 32) : (m & 0xffffffffull)) ? 1u : 0u;
    }
}

// In-workgroup split-K reduction and the fused epilogue.
//
// Reduction: every wave stores its 32 x 32 partial tile as a slab [i*4+j][lane] (pitch 72 ints;
// the cols-layout epilogue stores it with the lane index transposed), so that four consecutive
// ints are four consecutive columns of a row (rows of a column). After the single barrier a
// thread sums one quad over the slabs with ds_read_b128. (LDS atomics were measured: 16
// ds_add_u32 per wave cost ~1300 cycles, four times the plain stores plus the wide reads.)
//
// Epilogue (MODE 0 rows-layout bits, 1 cols-layout bits, 2 float32): a thread requantises its
// quad and builds the quad's nibble of each output plane; eight adjacent lanes OR their nibbles
// into the 32-bit word of one row (column) of the tile with DPP. 256 threads finish a tile, so
// in workgroups of 4+ waves the upper waves leave right after the barrier. What runs here is
// latency-bound (a few waves, dependent instructions), so the code is kept short: every
// instruction behind the barrier costs the whole workgroup ~8 cycles.
// What a thread needs to finish its quad besides the sums. (Computing it at kernel start, under the
// first loads' latency, was measured: it shortens the tail by ~300 cycles but costs as much in the
// prologue and 5 VGPRs across the main loop.)
struct QuadPlan {
    uint32_t src;     // byte offset of the thread's elements inside a slab
    uint32_t *dst;    // first output word (or float) of the thread's elements
    int nvalid;       // leading elements that exist (rows layout / float: columns; cols layout: rows)
    uint32_t sh_n;    // shift of the thread's bits inside the 32-bit word; bit 31: this lane stores the word
};

// OR over aligned groups of 32/E lanes (E = 4: 8 lanes, E = 2: 16 lanes = one DPP row)
template <int E>
__device__ __forceinline__ uint32_t or_reduce_group(uint32_t x) {
    x = or_reduce8(x);
    if (E == 2)  // 15 - lane within the row of 16: joins the two 8-lane halves
        x |= static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), 0x140, 0xf, 0xf, false));
    return x;
}

// E consecutive elements per thread (E = 4: a quad, 256 threads finish a tile; E = 2: a pair, 512
// threads - used by 8-wave workgroups so that every SIMD has two waves to interleave in the
// latency-bound tail). Thread t = hi | a8 | lo | h with h the E-group inside the 8 columns (rows)
// of micro-tile block (i, j); rows layout / float: (hi,lo) = (i,j), cols layout: (j,i).
template <int MODE, int E>
__device__ __forceinline__ QuadPlan quad_plan(const qgtc_problem &pr, int t, int m0, int n0) {
    constexpr int HB = E == 4 ? 1 : 2;       // bits of h
    constexpr int G = 32 / E;                // threads per output word
    const int M = pr.M, N = pr.N;
    const int h = t & ((1 << HB) - 1), lo = (t >> HB) & 3, a8 = (t >> (HB + 2)) & 7, hi = (t >> (HB + 5)) & 3;
    const int i = MODE == 1 ? lo : hi, j = MODE == 1 ? hi : lo;
    QuadPlan q;
    q.src = static_cast<uint32_t>(((i * MC + j) * SLAB_PITCH + a8 * 8 + h * E) * 4);
    const int m = MODE == 1 ? m0 + 8 * i + E * h : m0 + a8 + 8 * i;
    const int n = MODE == 1 ? n0 + a8 + 8 * j : n0 + 8 * j + E * h;
    // valid elements: rows layout / float (m, n+e), cols layout (m+e, n)
    q.nvalid = MODE == 1 ? (n < N ? min(max(M - m, 0), E) : 0) : (m < M ? min(max(N - n, 0), E) : 0);
    // rows layout [ob][PAD8(M)][STEP128(N)*4] (reference kernel.h:357-389): word (m, n0/32);
    // cols layout [ob][PAD128(N)][STEP128(M)*4] (intended semantics of kernel.h:651-810): word (n, m0/32);
    // float32 [M,N] (reference kernel.h:915-930): row m, columns n .. n+E-1
    const size_t o0 = MODE == 2 ? static_cast<size_t>(m) * N + n
                    : MODE == 0 ? static_cast<size_t>(m) * (step128(N) * 4) + (n0 >> 5)
                                : static_cast<size_t>(n) * (step128(M) * 4) + (m0 >> 5);
    q.dst = static_cast<uint32_t *>(pr.out) + o0;
    const bool store = (t & (G - 1)) == 0 && (MODE == 0 ? m < pad8(M) : n < pad128(N));
    // element e of the word's 32 sits at bit 31 - e; this thread holds e = E*(t % G) .. +E-1
    q.sh_n = static_cast<uint32_t>(32 - E - E * (t & (G - 1))) | (store ? 0x80000000u : 0u);
    return q;
}

template <int MODE, int E, bool INT_RQ, bool ALL8>
__device__ __forceinline__ void quad_finish(const qgtc_problem &pr, const MMShape &sh, const QuadPlan &q,
                                            int extra, size_t oplane, const unsigned char *slabs, int nwv STAMP_ARG) {
    typedef int ivec __attribute__((ext_vector_type(E)));
    ivec part[MAX_WAVES];
#pragma unroll
    for (int k = 0; k < MAX_WAVES; k++)  // slabs that do not exist alias slab 0 and are masked
        part[k] = *reinterpret_cast<const ivec *>(slabs + q.src + ((ALL8 || k < nwv) ? k : 0) * SLAB_BYTES);
    int v[E];
#pragma unroll
    for (int e = 0; e < E; e++) v[e] = 0;
#pragma unroll
    for (int k = 0; k < MAX_WAVES; k++) {
        const bool on = ALL8 || k < nwv;
#pragma unroll
        for (int e = 0; e < E; e++) v[e] += on ? part[k][e] : 0;
    }
#ifdef QGTC_STAMPS
    asm volatile("" : "+v"(v[0]), "+v"(v[1]));
    STAMP(11);
#endif
    if (MODE == 2) {
        float *dst = reinterpret_cast<float *>(q.dst);
        if (q.nvalid == E && (pr.N & (E - 1)) == 0) {
            typedef float fvec __attribute__((ext_vector_type(E)));
            fvec f;
#pragma unroll
            for (int e = 0; e < E; e++) f[e] = static_cast<float>(v[e]);
            *reinterpret_cast<fvec *>(dst) = f;
        } else {
#pragma unroll
            for (int e = 0; e < E; e++)
                if (e < q.nvalid) dst[e] = static_cast<float>(v[e]);
        }
        return;
    }
    const int maxi = 1 << (sh.ob & 31);
    uint32_t qv[E];
#pragma unroll
    for (int e = 0; e < E; e++) {
        int c;
        if (INT_RQ) c = v[e] < 0 ? 1 : (v[e] > maxi ? maxi - 1 : v[e]);  // kernel.h:31-37
        else c = requant(v[e], sh.maxv, sh.maxm1);
        qv[e] = e < q.nvalid ? static_cast<uint32_t>(c) : 0u;
    }
#ifdef QGTC_STAMPS
    asm volatile("" : "+v"(qv[0]), "+v"(qv[1]));
    STAMP(12);
#endif
    const bool store = (q.sh_n >> 31) != 0u;
    const uint32_t sh_n = q.sh_n & 31u;
    uint32_t *out = q.dst;
    for (int p = 0; p < sh.ob; p++, out += oplane) {
        uint32_t bits = 0u;
#pragma unroll
        for (int e = 0; e < E; e++) bits |= ((qv[e] >> p) & 1u) << (E - 1 - e);
        const uint32_t word = or_reduce_group<E>(bits << sh_n);
#ifndef QGTC_ABL_NOSTORE
        if (store) {
            out[0] = word;
            for (int x = 1; x <= extra; x++) out[x] = 0u;  // row words past the last column tile
        }
#else
        asm volatile("" ::"v"(word));
#endif
    }
}

// Epilogue of a single-wave workgroup (the wave owns the whole K range, nothing to reduce): straight
// from the accumulators. Lane (lm, ln) holds rows lm + 8i and columns ln + 8j, so the 32 columns
// of a row live in the 8 lanes of one aligned group (4 each): a DPP OR assembles the row word.
// For the cols layout the 32 rows of a column live in the 8 lanes ln, ln+8, .., ln+56.
template <int MODE>
__device__ __forceinline__ void epi_direct(const qgtc_problem &pr, const MMShape &sh,
                                           const uint32_t (&tot)[MR][MC], int tm, int tn, int tiles_n) {
    const int lane = threadIdx.x & 63, lm = lane >> 3, ln = lane & 7;
    const int M = pr.M, N = pr.N, m0 = tm * TM, n0 = tn * TN;
    if (MODE == 2) {  // float32 [M,N] (reference kernel.h:915-930)
        float *out = static_cast<float *>(pr.out);
#pragma unroll
        for (int i = 0; i < MR; i++)
#pragma unroll
            for (int j = 0; j < MC; j++) {
                const int m = m0 + lm + 8 * i, n = n0 + ln + 8 * j;
                if (m < M && n < N) out[static_cast<size_t>(m) * N + n] = static_cast<float>(static_cast<int>(tot[i][j]));
            }
        return;
    }
    const bool int_rq = sh.ob <= 23;
    const int maxi = 1 << (sh.ob & 31);
    uint32_t qv[MR][MC];
#pragma unroll
    for (int i = 0; i < MR; i++)
#pragma unroll
        for (int j = 0; j < MC; j++) {
            const int c = static_cast<int>(tot[i][j]);
            const int r = int_rq ? (c < 0 ? 1 : (c > maxi ? maxi - 1 : c)) : requant(c, sh.maxv, sh.maxm1);
            qv[i][j] = (m0 + lm + 8 * i < M && n0 + ln + 8 * j < N) ? static_cast<uint32_t>(r) : 0u;
        }
    uint32_t *out = static_cast<uint32_t *>(pr.out);
    if (MODE == 0) {  // rows layout [ob][PAD8(M)][STEP128(N)*4] (reference kernel.h:357-389)
        const int rows_pad = pad8(M), row_words = step128(N) * 4;
        const size_t oplane = static_cast<size_t>(rows_pad) * row_words;
        const int extra = tn == tiles_n - 1 ? row_words - (n0 >> 5) - 1 : 0;
#pragma unroll
        for (int i = 0; i < MR; i++) {
            const int m = m0 + lm + 8 * i;
            uint32_t *dst = out + static_cast<size_t>(m) * row_words + (n0 >> 5);
            for (int p = 0; p < sh.ob; p++, dst += oplane) {
                // column ln + 8j sits at bit 31 - ln - 8j = (24 - 8j) + (7 - ln)
                const uint32_t x = (((qv[i][0] >> p) & 1u) << 24) | (((qv[i][1] >> p) & 1u) << 16) |
                                   (((qv[i][2] >> p) & 1u) << 8) | ((qv[i][3] >> p) & 1u);
                const uint32_t word = or_reduce8(x << (7 - ln));
                if (ln == 0 && m < rows_pad) {
                    dst[0] = word;
                    for (int e = 1; e <= extra; e++) dst[e] = 0u;
                }
            }
        }
    } else {  // cols layout [ob][PAD128(N)][STEP128(M)*4] (intended semantics of kernel.h:651-810)
        const int lines = pad128(N), line_words = step128(M) * 4;
        const size_t oplane = static_cast<size_t>(lines) * line_words;
#pragma unroll
        for (int j = 0; j < MC; j++) {
            const int n = n0 + ln + 8 * j;
            uint32_t *dst = out + static_cast<size_t>(n) * line_words + (m0 >> 5);
            for (int p = 0; p < sh.ob; p++, dst += oplane) {
                // row lm + 8i sits at bit 31 - lm - 8i
                uint32_t x = (((qv[0][j] >> p) & 1u) << 24) | (((qv[1][j] >> p) & 1u) << 16) |
                             (((qv[2][j] >> p) & 1u) << 8) | ((qv[3][j] >> p) & 1u);
                x <<= (7 - lm);
                x |= static_cast<uint32_t>(__shfl_xor(static_cast<int>(x), 8));
                x |= static_cast<uint32_t>(__shfl_xor(static_cast<int>(x), 16));
                x |= static_cast<uint32_t>(__shfl_xor(static_cast<int>(x), 32));
                if (lm == 0 && n < lines) dst[0] = x;
            }
        }
    }
}

template <int MODE>
__device__ __forceinline__ void epi_finish(const qgtc_problem &pr, const MMShape &sh,
                                           const uint32_t (&tot)[MR][MC], int tm, int tn,
                                           int tiles_m, int tiles_n, unsigned char *slabs STAMP_ARG) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nwv = sh.waves, NT = nwv * 64;
    const int lm = lane >> 3, ln = lane & 7;
    const int M = pr.M, N = pr.N, m0 = tm * TM, n0 = tn * TN;
    const bool last_n = tn == tiles_n - 1, last_m = tm == tiles_m - 1;
    const size_t oplane = MODE == 0 ? static_cast<size_t>(pad8(M)) * (step128(N) * 4)
                                    : static_cast<size_t>(pad128(N)) * (step128(M) * 4);
    if (nwv == 1) {
        epi_direct<MODE>(pr, sh, tot, tm, tn, tiles_n);
    } else {
        int *mine = reinterpret_cast<int *>(slabs + wv * SLAB_BYTES) + (MODE == 1 ? ln * 8 + lm : lane);
#pragma unroll
        for (int i = 0; i < MR; i++)
#pragma unroll
            for (int j = 0; j < MC; j++) mine[(i * MC + j) * SLAB_PITCH] = static_cast<int>(tot[i][j]);
    }
    STAMP(8);
    if (nwv > 1) __syncthreads();
    STAMP(9);
    const int extra = (MODE == 0 && last_n) ? step128(N) * 4 - (n0 >> 5) - 1 : 0;
    if (nwv == 1) {
    } else if (nwv == MAX_WAVES && sh.ob <= 23) {  // every slab exists: all 512 threads finish a pair each
        const QuadPlan q = quad_plan<MODE, 2>(pr, tid, m0, n0);
#ifdef QGTC_STAMPS
        asm volatile("" ::"v"(q.src), "v"(q.dst), "v"(q.nvalid), "v"(q.sh_n));
        STAMP(10);
#endif
        quad_finish<MODE, 2, true, true>(pr, sh, q, extra, oplane, slabs, nwv STAMP_PASS);
    } else {
        for (int t = tid; t < 256; t += NT) {
            const QuadPlan q = quad_plan<MODE, 4>(pr, t, m0, n0);
            // float(c) > 2^ob  <=>  c > 2^ob for every int c >= 0: integer requantisation when ob <= 23
            if (sh.ob > 23) quad_finish<MODE, 4, false, false>(pr, sh, q, extra, oplane, slabs, nwv STAMP_PASS);
            else quad_finish<MODE, 4, true, false>(pr, sh, q, extra, oplane, slabs, nwv STAMP_PASS);
        }
    }
    STAMP(14);
    if (MODE == 1) {
        // zero what no tile computes: words past the last row tile, lines past the last column tile
        uint32_t *out = static_cast<uint32_t *>(pr.out);
        const int lines = pad128(N), line_words = step128(M) * 4;
        const int w_core0 = m0 >> 5, w_core1 = min(line_words, w_core0 + 1);
        if (last_m && w_core1 < line_words) {
            for (int e = tid; e < sh.ob * TN; e += NT) {
                const int line = n0 + (e & (TN - 1)), p = e / TN;
                if (line < lines)
                    for (int wi = w_core1; wi < line_words; wi++)
                        out[p * oplane + static_cast<size_t>(line) * line_words + wi] = 0u;
            }
        }
        if (last_n && n0 + TN < lines) {
            const int nl = lines - (n0 + TN), w_end = last_m ? line_words : w_core1;
            for (int e = tid; e < sh.ob * nl; e += NT) {
                const int line = n0 + TN + e % nl, p = e / nl;
                for (int wi = w_core0; wi < w_end; wi++)
                    out[p * oplane + static_cast<size_t>(line) * line_words + wi] = 0u;
            }
        }
    }
}

// One output tile (tm, tn) of one problem. All threads of the workgroup call this.
// NA, NW > 0: compile-time plane counts (== sh.a, sh.w), QW k-quads per stage;
// NA == NW == 0: generic kernel, runtime plane blocks sh.ab x sh.wb, QW = 1.
template <int QW, int NA, int NW, bool ZS, bool OCC>
__device__ __forceinline__ void mm_tile(const qgtc_problem &pr, const MMShape &sh, int tm, int tn,
                                        int tiles_m, int tiles_n, unsigned char *smem) {
    constexpr bool GEN = NA == 0;
    static_assert(!GEN || QW == 1, "the generic kernel stages one k-quad at a time");
    constexpr int RS = lds_pitch(QW);
    STAMP_DECL;
    STAMP(0);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nwv = sh.waves;
    const int lm = lane >> 3, ln = lane & 7;

    const int M = pr.M, K = pr.K, N = pr.N;
    const int kq = step128(K);                   // k-quads per packed row
    const uint32_t kw = static_cast<uint32_t>(kq) * 4u;
    const uint32_t x_plane = static_cast<uint32_t>(pad8(M)) * kw;  // < 2^30 words (host-checked)
    const uint32_t w_plane = static_cast<uint32_t>(pr.w_lines) * kw;
    const int m0 = tm * TM, n0 = tn * TN;
    const int ab = GEN ? sh.ab : NA, wb = GEN ? sh.wb : NW;
    const int nsx = GEN ? slots_for(ab, 1) : slots_for(NA, QW);
    const int nsw = GEN ? slots_for(wb, 1) : slots_for(NW, QW);
    const int ks = wv * sh.per, ke = min(ks + sh.per, kq);  // this wave's k-quads

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint32_t *>(pr.X), 0, static_cast<int>(static_cast<uint32_t>(pr.x_words) * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint32_t *>(pr.W), 0, static_cast<int>(static_cast<uint32_t>(pr.w_words) * 4u), 0x00020000);

    // this wave's staging region: X plane tiles, then W plane tiles
    u32x4 *region = reinterpret_cast<u32x4 *>(smem) + wv * ((ab + wb) * QW * RS);
    const int wreg = ab * QW * RS;  // first granule of the W tiles

    // Per-lane slot table (slot u < nsx: X, else W). Only the stage origin (plane block, first
    // k-quad) changes from stage to stage and it is wave-uniform.
    uint32_t s_off[GPT];   // byte offset of the granule from the stage origin
    uint32_t s_lds[GPT];   // granule index inside the region
    uint32_t v_in = 0u;    // bit u: the lane's granule of slot u lies inside the region
    uint32_t v_ok = 0u;    // bit u: ... and its line exists (row < M / column < N)
    int kkl = 0;           // the lane's k-quad within the chunk (same for every slot)
#pragma unroll
    for (int u = 0; u < GPT; u++) {
        s_off[u] = 0u;
        s_lds[u] = 0u;
        if (u >= nsx + nsw) continue;
        const bool is_x = u < nsx;
        int pt, line, kk;
        slot_map<QW>(is_x ? u : u - nsx, lane, pt, line, kk);
        kkl = kk;
        const bool in = pt < (is_x ? ab : wb);
        const uint32_t gline = static_cast<uint32_t>((is_x ? m0 : n0) + line);
        const bool ok = in && static_cast<int>(gline) < (is_x ? M : N);
        s_off[u] = (static_cast<uint32_t>(pt) * (is_x ? x_plane : w_plane) + gline * kw) * 4u;
        s_lds[u] = static_cast<uint32_t>((is_x ? 0 : wreg) + (pt * QW + kk) * RS + line);
        v_in |= in ? (1u << u) : 0u;
        v_ok |= ok ? (1u << u) : 0u;
    }

    // One stage = up to QW k-quads of one (X plane block, W plane block). The k-quads a wave visits
    // are either all of its slice [ks, ke) in order, or - when the caller supplies the occupancy
    // bitmap of the left operand (pr.occ: one bit per 32-row tile and k-quad) - only those whose
    // X tile has a bit set: zero tiles are then neither loaded nor multiplied ("zero-tile jumping").
    struct Stage {
        int pa0, pw0;
        int i0, i1, i2, i3;  // k-quad of slot kk = 0..3 (named fields: an indexed array lands in scratch)
        int nk;              // slots in use
        bool valid;
    };
    const uint64_t *occ_row = (OCC && pr.occ) ? pr.occ + static_cast<size_t>(tm) * pr.occ_words : nullptr;
    int k_next = ks;                // dense mode: next k-quad
    int k_word = 0;                 // bitmap mode: current 64-k-quad word
    unsigned long long k_mask = 0;  // bitmap mode: unvisited k-quads of the current word, inside [ks, ke)
    auto k_word_mask = [&](int wi) -> unsigned long long {
        unsigned long long m = occ_row[wi];
        const int lo = ks - wi * 64, hi = ke - wi * 64;  // keep bits [lo, hi)
        if (lo > 0) m &= ~0ull << lo;
        if (hi < 64) m &= hi > 0 ? ~0ull >> (64 - hi) : 0ull;
        return m;
    };
    auto k_reset = [&]() {
        k_next = ks;
        if (occ_row) {
            k_word = ks >> 6;
            k_mask = ks < ke ? k_word_mask(k_word) : 0ull;
        }
    };
    // the next k-quads of the slice as a stage (by value: a Stage passed by reference through
    // the lambdas ends up in scratch); .valid = false when the wave's slice is exhausted
    auto k_take = [&](int pa0, int pw0) -> Stage {
        Stage st{pa0, pw0, 0, 0, 0, 0, 0, false};
        if (!occ_row) {
            if (k_next >= ke) return st;
            st.i0 = k_next;
            st.i1 = k_next + 1;
            st.i2 = k_next + 2;
            st.i3 = k_next + 3;
            st.nk = min(QW, ke - k_next);
            st.valid = true;
            k_next += QW;
            return st;
        }
        while (k_mask == 0ull) {
            k_word++;
            if (k_word * 64 >= ke) return st;
            k_mask = k_word_mask(k_word);
        }
        st.valid = true;
        // pop the lowest unvisited k-quads of the word (plain locals: a lambda capturing `st` by
        // reference keeps the struct in scratch)
        int nk = 0, q0 = 0, q1 = 0, q2 = 0, q3 = 0;
        unsigned long long m = k_mask;
#define QGTC_POP(dst)                                \
    if (m != 0ull) {                                 \
        dst = k_word * 64 + __builtin_ctzll(m);      \
        m &= m - 1ull;                               \
        nk++;                                        \
    }
        QGTC_POP(q0)
        if (QW > 1) { QGTC_POP(q1) }
        if (QW > 2) { QGTC_POP(q2) QGTC_POP(q3) }
#undef QGTC_POP
        k_mask = m;
        st.i0 = q0;
        st.i1 = q1;
        st.i2 = q2;
        st.i3 = q3;
        st.nk = nk;
        return st;
    };
    auto first_stage = [&]() -> Stage {
        k_reset();
        return k_take(0, 0);  // an empty slice (or an all-zero row tile) has no stage at all
    };
    auto next_stage = [&](const Stage &prev) -> Stage {
        Stage st = k_take(prev.pa0, prev.pw0);
        if (st.valid) return st;
        // the k range is exhausted: next plane block (generic kernel only), restart the k iteration
        int pa0 = prev.pa0, pw0 = prev.pw0 + wb;
        if (pw0 >= sh.w) {
            pw0 = 0;
            pa0 += ab;
            if (pa0 >= sh.a) return st;  // invalid
        }
        k_reset();
        return k_take(pa0, pw0);
    };

    // issue the loads of one stage into registers; lanes whose granule does not exist (row or
    // column out of range, plane or k-quad beyond this stage) load from offset 0xffffffff, which
    // the range check turns into zeros
    u32x4 pre[GPT];
    auto issue = [&](const int pa0, const int pw0, const int i0, const int i1, const int i2, const int i3,
                     const int nk_) {
        // (scalars by value: selecting among the fields of a Stage passed by reference makes hipcc
        // spill the struct and load the field through a computed scratch address)
        const int na = min(ab, sh.a - pa0), nw = min(wb, sh.w - pw0);
        const uint32_t xo = static_cast<uint32_t>(pa0) * x_plane * 4u;
        const uint32_t wo = static_cast<uint32_t>(pw0) * w_plane * 4u;
        // the lane's k-quad, as byte offset inside the packed row
        uint32_t ko = static_cast<uint32_t>(i0) * 16u;
        if (QW > 1) ko = kkl == 1 ? static_cast<uint32_t>(i1) * 16u : ko;
        if (QW > 2) {
            ko = kkl == 2 ? static_cast<uint32_t>(i2) * 16u : ko;
            ko = kkl == 3 ? static_cast<uint32_t>(i3) * 16u : ko;
        }
        const bool kk_ok = kkl < nk_;
#pragma unroll
        for (int u = 0; u < GPT; u++) {
            if (u >= nsx + nsw) break;
            const bool is_x = u < nsx;
            bool ok = ((v_ok >> u) & 1u) && kk_ok;
            if (GEN) {
                const int pt = 2 * (is_x ? u : u - nsx) + (lane >> 5);
                ok = ok && pt < (is_x ? na : nw);
            }
            const uint32_t off = ok ? s_off[u] + (is_x ? xo : wo) + ko : 0xffffffffu;
            pre[u] = __builtin_amdgcn_raw_buffer_load_b128(is_x ? rx : rw, off, 0, 0);
        }
    };

    Stage cur = first_stage();
    STAMP(1);
    if (cur.valid) issue(cur.pa0, cur.pw0, cur.i0, cur.i1, cur.i2, cur.i3, cur.nk);
    STAMP(2);

    uint32_t tot[MR][MC];  // unsigned: the reference's int32 accumulation wraps on overflow
#pragma unroll
    for (int i = 0; i < MR; i++)
#pragma unroll
        for (int j = 0; j < MC; j++) tot[i][j] = 0u;

    // fixed kernels keep one accumulator set per shift (pa + pw) for the whole K slice
    constexpr int NS = GEN ? 1 : NA + NW - 1;
    uint32_t acc[NS][MR][MC];
#pragma unroll
    for (int s = 0; s < NS; s++)
#pragma unroll
        for (int i = 0; i < MR; i++)
#pragma unroll
            for (int j = 0; j < MC; j++) acc[s][i][j] = 0u;

    const u32x4 *xrd = region + lm;         // + (pa*QW + kk)*RS + 8*i
    const u32x4 *wrd = region + wreg + ln;  // + (pw*QW + kk)*RS + 8*j
    auto read_x = [&](int tile_kk, u32x4 (&xr)[MR]) {
#pragma unroll
        for (int i = 0; i < MR; i++) xr[i] = xrd[tile_kk * RS + 8 * i];
    };
    auto read_w = [&](int tile_kk, u32x4 (&wr)[MC]) {
#pragma unroll
        for (int j = 0; j < MC; j++) wr[j] = wrd[tile_kk * RS + 8 * j];
    };

    for (int it = 0; cur.valid; it++) {
        // ---- registers -> LDS, and the occupancy ballots of the X tiles ----
        unsigned long long nzm[GPT];
#pragma unroll
        for (int u = 0; u < GPT; u++) {
            nzm[u] = 0ull;
            if (u >= nsx + nsw) continue;
            if ((v_in >> u) & 1u) region[s_lds[u]] = pre[u];
            if (ZS && u < nsx) nzm[u] = __ballot(((pre[u].x | pre[u].y) | (pre[u].z | pre[u].w)) != 0u);
        }
        if (it == 0) STAMP(3);
        const Stage now = cur;
        cur = next_stage(now);
        // the next stage's loads fly while this one is multiplied
        if (cur.valid) issue(cur.pa0, cur.pw0, cur.i0, cur.i1, cur.i2, cur.i3, cur.nk);
        // the wave reads what its other lanes wrote: LDS is in order per wave, the fence only
        // keeps the compiler from moving the reads above the writes
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (it == 0) STAMP(4);

        const int nk = now.nk;
        if constexpr (!GEN) {
            uint32_t occ[NA];
#pragma unroll
            for (int pa = 0; pa < NA; pa++) occ[pa] = ZS ? tile_occupancy<QW>(nzm, pa) : ((1u << nk) - 1u);
            constexpr int NOUT = QW * NA;  // (k-quad, X plane) pairs
            if constexpr (NOUT * NW <= 8) {
                // flat software pipeline over every (k-quad, X plane, W plane) step of the stage:
                // the granules of step t+1 are read from LDS while step t is multiplied
                constexpr int T = NOUT * NW;
                // X granules: double-buffered only when they change every step (NW == 1); with several W
                // planes per X tile one set is enough (the next tile's X is read behind its last
                // multiply) and 16 VGPRs fewer buy the (1,2) kernel a fourth wave per SIMD
                constexpr int XB = NW == 1 ? 2 : 1;
                u32x4 xg[XB][MR], wg[2][MC];
                read_x(0, xg[0]);
                read_w(0, wg[0]);
#pragma unroll
                for (int t = 0; t < T; t++) {
                    const int o = t / NW, pw = t % NW, kk = o / NA, pa = o % NA;
#ifdef QGTC_ABL_NOLDS  // timing-only build: every step multiplies the first step's granules
                    if (t == 0) {
                        read_w(1, wg[1]);
                        read_x(1, xg[1]);
                    }
#else
                    if (t + 1 < T) {
                        const int o1 = (t + 1) / NW, pw1 = (t + 1) % NW, kk1 = o1 / NA, pa1 = o1 % NA;
                        read_w(pw1 * QW + kk1, wg[(t + 1) & 1]);
                        if (XB == 2 && pw1 == 0) read_x(pa1 * QW + kk1, xg[o1 & 1]);
                    }
#endif
#ifdef QGTC_ABL_NOMAC  // timing-only build: keep the LDS reads, skip the multiply
                    asm volatile("" ::"v"(xg[o % XB][0].x), "v"(wg[t & 1][0].x), "v"(xg[o % XB][3].w), "v"(wg[t & 1][3].w));
#else
                    if ((occ[pa] >> kk) & 1u) mac_quad(acc[pa + pw], xg[o % XB], wg[t & 1]);
#endif
                    if (XB == 1 && t + 1 < T && (t + 1) % NW == 0) {
                        const int o1 = (t + 1) / NW;
                        read_x((o1 % NA) * QW + o1 / NA, xg[0]);
                    }
                }
            } else {
                // k-quads in a loop, the (X plane, W plane) steps of one k-quad unrolled
#pragma unroll 1
                for (int kk = 0; kk < nk; kk++) {
                    u32x4 xg[MR], wg[2][MC];
#pragma unroll
                    for (int pa = 0; pa < NA; pa++) {
                        if (!((occ[pa] >> kk) & 1u)) continue;
                        read_x(pa * QW + kk, xg);
                        read_w(kk, wg[0]);
#pragma unroll
                        for (int pw = 0; pw < NW; pw++) {
                            if (pw + 1 < NW) read_w((pw + 1) * QW + kk, wg[(pw + 1) & 1]);
                            mac_quad(acc[pa + pw], xg, wg[pw & 1]);
                        }
                    }
                }
            }
        } else {
            const int na = min(ab, sh.a - now.pa0), nw = min(wb, sh.w - now.pw0);
            uint32_t occ = 0u;  // bit pa: X plane tile pa of the stage has a set bit
#pragma unroll
            for (int u = 0; u < GPT / 2; u++)
                occ |= ((nzm[u] & 0xffffffffull) ? (1u << (2 * u)) : 0u) | ((nzm[u] >> 32) ? (2u << (2 * u)) : 0u);
            for (int pa = 0; pa < na; pa++) {
                if (ZS && !((occ >> pa) & 1u)) continue;
                u32x4 xg[MR];
                read_x(pa, xg);
                for (int pw = 0; pw < nw; pw++) {
                    u32x4 wg[MC];
                    read_w(pw, wg);
#pragma unroll
                    for (int i = 0; i < MR; i++)
#pragma unroll
                        for (int j = 0; j < MC; j++) acc[0][i][j] = 0u;
                    mac_quad(acc[0], xg, wg);
                    const int s = now.pa0 + pa + now.pw0 + pw;  // reference kernel.h:295,340
                    if (s < 32) {
#pragma unroll
                        for (int i = 0; i < MR; i++)
#pragma unroll
                            for (int j = 0; j < MC; j++) tot[i][j] += acc[0][i][j] << s;
                    }
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (it == 0) STAMP(5);
    }
    if constexpr (!GEN) {
#pragma unroll
        for (int s = 0; s < NS; s++)
#pragma unroll
            for (int i = 0; i < MR; i++)
#pragma unroll
                for (int j = 0; j < MC; j++) tot[i][j] += acc[s][i][j] << s;
    }
    STAMP(7);
#ifdef QGTC_ABL_NOEPI  // timing-only build: keep the sums alive, skip the reduction and epilogue
#pragma unroll
    for (int i = 0; i < MR; i++)
#pragma unroll
        for (int j = 0; j < MC; j++) asm volatile("" ::"v"(tot[i][j]));
    return;
#endif
    unsigned char *slabs = smem + nwv * ((ab + wb) * QW * RS * 16);
    if (sh.mode == 0) epi_finish<0>(pr, sh, tot, tm, tn, tiles_m, tiles_n, slabs STAMP_PASS);
    else if (sh.mode == 1) epi_finish<1>(pr, sh, tot, tm, tn, tiles_m, tiles_n, slabs STAMP_PASS);
    else epi_finish<2>(pr, sh, tot, tm, tn, tiles_m, tiles_n, slabs STAMP_PASS);
    STAMP(15);
    STAMP_FLUSH();
}

// Workgroups are dealt round-robin over the 8 XCDs (each with its own L2), so blocks b and b+8
// share an L2. Map block ids to tiles so that each XCD owns a contiguous range of tile ids: the
// column tiles of one row tile (which read the same X rows) then hit the same L2. Bijective for
// any grid size; placement only affects speed, never results.
__device__ __forceinline__ int xcd_remap(int bid, int nblocks) {
    constexpr int NX = 8;
    const int q = nblocks / NX, r = nblocks % NX;
    const int xcd = bid % NX, idx = bid / NX;
    return xcd * q + min(xcd, r) + idx;
}

template <int QW, int NA, int NW, bool ZS>
__global__ __launch_bounds__(64 * MAX_WAVES) void k_bitmm(qgtc_problem pr, MMShape sh, int tiles_m,
                                                          int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // pull every kernel argument into SGPRs with ONE scalar-load round trip (hipcc otherwise loads
    // them lazily in four dependent rounds, ~200 cycles each, ahead of the first global load)
    asm volatile("" ::"s"(pr.X), "s"(pr.W), "s"(pr.out), "s"(pr.x_words), "s"(pr.w_words), "s"(pr.M), "s"(pr.K),
                 "s"(pr.N), "s"(pr.w_lines), "s"(sh.a), "s"(sh.w), "s"(sh.ob), "s"(sh.mode), "s"(sh.per),
                 "s"(sh.inv_tiles_n), "s"(sh.waves), "s"(tiles_m), "s"(tiles_n));
    const int tile = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    // tile / tiles_n by multiply-high with floor(2^32 / tiles_n) (from the host) + one correction
    int tm = static_cast<int>(__umulhi(static_cast<uint32_t>(tile), sh.inv_tiles_n));
    int tn = tile - tm * tiles_n;
    if (tn >= tiles_n) {
        tn -= tiles_n;
        tm++;
    }
    mm_tile<QW, NA, NW, ZS, false>(pr, sh, tm, tn, tiles_m, tiles_n, smem);
}

// grouped launch: blockIdx.y = problem, blockIdx.x = tile (surplus tiles exit at once)
template <int QW, int NA, int NW, bool ZS, bool OCC>
__global__ __launch_bounds__(64 * MAX_WAVES) void k_bitmm_batched(const qgtc_problem *__restrict__ prs,
                                                                  MMShape sh) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const qgtc_problem pr = prs[blockIdx.y];
    const int tiles_m = (pr.M + TM - 1) / TM, tiles_n = (pr.N + TN - 1) / TN;
    const int tile = blockIdx.x;
    if (tile >= tiles_m * tiles_n) return;
    mm_tile<QW, NA, NW, ZS, OCC>(pr, sh, tile / tiles_n, tile % tiles_n, tiles_m, tiles_n, smem);
}

// ------------------------------------------------------------------------------------------
// The bit-GEMM on the matrix cores (opt-in engine, QGTC_ENGINE_MFMA): same operands, same
// results, but the bit planes of both operands are expanded on the fly to int8 VALUES (plane p
// contributes bit p of the byte, so a planes x w planes collapse into ONE product) and multiplied
// with v_mfma_i32_32x32x32_i8, int32 accumulation, exact. gfx950 has no 1-bit MFMA; expanding
// costs ~0.8 VALU operations per operand byte, which only pays when an expanded byte is reused by
// several MFMA tiles: a workgroup owns a 128 x 128 output tile (four waves, 64 x 64 each), so this
// engine is for wide N (>= 128) and/or several planes; the popcount kernels stay the default and
// remain the faster path at N = 64. Needs a, w <= 7 (non-negative int8).
//
// Per k-quad (128 bits of K) one thread of the expander waves expands one row of X or one column of W:
// every 32-bit word is bit-reversed (element i at bit i), each nibble is spread to four bytes with
// one 24-bit multiply ((n * 0x204081) & 0x01010101) and planes are merged with shift-or; the 128
// bytes go to LDS ([line][144-byte pitch]: the 16-byte MFMA fragment reads of 32 lines are
// conflict-free). The packed words of the next k-quad are loaded (range-checked buffer loads)
// while the current one is multiplied. The finished 128 x 128 int32 tile goes through LDS to the
// same three epilogues (rows-layout bits, cols-layout bits, float32); with 128-wide tiles every
// output word belongs to exactly one workgroup, so there is no padding to zero-fill separately.
// ------------------------------------------------------------------------------------------
typedef int i32x16 __attribute__((ext_vector_type(16)));
constexpr int MF_T = 128;          // tile edge
constexpr int MF_PITCH = 144;      // bytes between the expanded lines of one operand
constexpr int MF_CPITCH = 132;     // ints between the rows (cols layout: columns) of the result tile in LDS
constexpr int MF_STAGE = 2 * MF_T * MF_PITCH;                 // one staging buffer: X lines, then W lines
constexpr int MF_LDS = (MF_T * MF_CPITCH * 4 > 2 * MF_STAGE) ? MF_T * MF_CPITCH * 4 : 2 * MF_STAGE;

// 32 packed elements (MSB-first) of `planes` planes -> 32 bytes (8 dwords), byte = sum_p bit_p << p
template <int MAXP>
__device__ __forceinline__ void expand_word(const uint32_t (&wd)[MAXP], int planes, uint32_t (&out)[8]) {
#pragma unroll
    for (int d = 0; d < 8; d++) out[d] = 0u;
#pragma unroll
    for (int p = 0; p < MAXP; p++) {
        if (p >= planes) break;
        const uint32_t r = __brev(wd[p]);  // element i of the word at bit i
#pragma unroll
        for (int d = 0; d < 8; d++) {
            const uint32_t nib = (r >> (4 * d)) & 15u;
            const uint32_t bytes = __umul24(nib, 0x204081u) & 0x01010101u;  // bit e of the nibble -> byte e
            out[d] |= bytes << p;
        }
    }
}

// EXPW expander waves: 8 (two threads per line) when a CU holds one workgroup - a lone expander wave per
// SIMD is latency-bound - or 4 (one thread per line, fewer registers per workgroup) when the grid is large
// enough for two workgroups per CU to overlap each other.
template <int MAXP, int EXPW>
__global__ __launch_bounds__(64 * (4 + EXPW)) void k_bitmm_mfma(qgtc_problem pr, MMShape sh, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
    const int M = pr.M, K = pr.K, N = pr.N;
    const int m0 = tm * MF_T, n0 = tn * MF_T;
    const int kq = step128(K);
    // Waves 0-3 multiply, waves 4.. expand: waves v, v+4 (and v+8) share a SIMD, so the matrix pipe
    // (multiplying k-quad q) and the vector pipe (expanding k-quad q+1) of every SIMD run side by
    // side. Two staging buffers, one barrier per k-quad.
    const bool expander = wv >= 4;
#ifdef QGTC_STAMPS
    unsigned long long st_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define MF_STAMP(i) st_[i] = __builtin_amdgcn_s_memtime()
#else
#define MF_STAMP(i) do { } while (0)
#endif
    MF_STAMP(0);

    i32x16 acc[2][2];
    const int mw = wv & 3, wr = mw >> 1, wc = mw & 1;   // multiplier wave (wr, wc): a 64 x 64 quarter, 2 x 2 MFMA tiles
    const int fl = lane & 31, fh = lane >> 5;           // fragment line, k half (16 bytes each)
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0;

    if (expander) {
        const uint32_t kw = static_cast<uint32_t>(kq) * 4u;
        const uint32_t x_plane = static_cast<uint32_t>(pad8(M)) * kw, w_plane = static_cast<uint32_t>(pr.w_lines) * kw;
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<uint32_t *>(pr.X), 0, static_cast<int>(static_cast<uint32_t>(pr.x_words) * 4u), 0x00020000);
        const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<uint32_t *>(pr.W), 0, static_cast<int>(static_cast<uint32_t>(pr.w_words) * 4u), 0x00020000);
        // this thread's expansion unit: one line (row of X / column of W) of the tile; the wave-level
        // split (waves 4,5: X rows, waves 6,7: W columns) keeps the descriptor choice wave-uniform
        // two threads per line (each expands two of the four words of a k-quad): eight expander waves, so
        // that every SIMD has two of them to interleave - one expander wave per SIMD is latency-bound
        constexpr int TPL = EXPW / 4;  // threads per line
        const int u = (tid - 256) / TPL, hw = TPL == 2 ? (tid & 1) : 0;
        const bool is_x = wv < 4 + EXPW / 2;
        const int line = is_x ? u : u - MF_T;
        const int gline = (is_x ? m0 : n0) + line;
        const bool line_ok = gline < (is_x ? M : N);
        const int planes = is_x ? sh.a : sh.w;
        const uint32_t plane_words = is_x ? x_plane : w_plane;
        const uint32_t base = static_cast<uint32_t>(gline) * kw * 4u;  // byte offset of the line inside a plane
        unsigned char *my_stage = smem + (is_x ? 0 : MF_T * MF_PITCH) + line * MF_PITCH;
        // Packed words are loaded GQ k-quads at a time per line (GQ * 16 contiguous bytes per lane): one
        // k-quad per load instruction touches 64 different 128-byte lines for 16 bytes each and the L1
        // (32 KB) does not keep them until the next k-quad - measured: 1300 cycles per k-quad, all of
        // it L2 -> L1 traffic. Two register sets: group g+1 is in flight while group g is expanded.
        constexpr int GQ = MAXP <= 2 ? 4 : (MAXP <= 4 ? 2 : 1);
        u32x4 grp[2][GQ][MAXP];
        auto issue_group = [&](int g, u32x4 (&dst)[GQ][MAXP]) {
#pragma unroll
            for (int p = 0; p < MAXP; p++)
#pragma unroll
                for (int j = 0; j < GQ; j++) {
                    const int q = g * GQ + j;
                    const bool ok = line_ok && p < planes && q < kq;
                    const uint32_t off = ok ? static_cast<uint32_t>(p) * plane_words * 4u + base + static_cast<uint32_t>(q) * 16u : 0xffffffffu;
                    dst[j][p] = is_x ? __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0)
                                     : __builtin_amdgcn_raw_buffer_load_b128(rw, off, 0, 0);
                }
        };
        auto expand = [&](int q, const u32x4 (&src)[MAXP]) {  // packed words of k-quad q -> bytes in staging buffer q & 1
            unsigned char *stage = my_stage + (q & 1) * MF_STAGE;
#pragma unroll
            for (int cc = 0; cc < 4 / TPL; cc++) {
                const int c = (4 / TPL) * hw + cc;  // this thread's words of the k-quad
                uint32_t wd[MAXP], out[8];
#pragma unroll
                for (int p = 0; p < MAXP; p++) wd[p] = (TPL == 2 && hw) ? src[p][2 + cc] : src[p][cc];
#ifdef QGTC_MF_NOEXPAND  // timing-only build
#pragma unroll
                for (int d = 0; d < 8; d++) out[d] = wd[0];
#else
                expand_word<MAXP>(wd, planes, out);
#endif
#ifdef QGTC_MF_NOWRITE  // timing-only build
                asm volatile("" ::"v"(out[0]), "v"(out[1]), "v"(out[2]), "v"(out[3]), "v"(out[4]), "v"(out[5]), "v"(out[6]), "v"(out[7]));
#else
                *reinterpret_cast<u32x4 *>(stage + c * 32) = u32x4{out[0], out[1], out[2], out[3]};
                *reinterpret_cast<u32x4 *>(stage + c * 32 + 16) = u32x4{out[4], out[5], out[6], out[7]};
#endif
            }
        };
        issue_group(0, grp[0]);
        issue_group(1, grp[1]);
        MF_STAMP(1);
        expand(0, grp[0][0]);
        if (GQ == 1) issue_group(2, grp[0]);
        MF_STAMP(2);
        __syncthreads();
        MF_STAMP(3);
        // step J of a block of 2*GQ: k-quad q0+J is being multiplied; expand k-quad e = q0+J+1 (set (e/GQ)&1,
        // slot e%GQ); after the last slot of a set, refill the set with the group two ahead
#define QGTC_MF_STEP(J)                                                                   \
    if (J < 2 * GQ && q0 + J < kq) {                                                      \
        constexpr int E = (J + 1) % (2 * GQ);                                             \
        if (q0 + J + 1 < kq) {                                                            \
            expand(q0 + J + 1, grp[E / GQ][E % GQ]);                                      \
            if (E % GQ == GQ - 1) issue_group((q0 + J + 1) / GQ + 2, grp[E / GQ]);        \
        }                                                                                 \
        if (q0 + J == 8) MF_STAMP(6);                                                     \
        __syncthreads();                                                                  \
        if (q0 + J == 8) MF_STAMP(7);                                                     \
    }
        for (int q0 = 0; q0 < kq; q0 += 2 * GQ) {
            if (q0 == 2 * GQ) MF_STAMP(4);
            QGTC_MF_STEP(0)
            QGTC_MF_STEP(1)
            QGTC_MF_STEP(2)
            QGTC_MF_STEP(3)
            QGTC_MF_STEP(4)
            QGTC_MF_STEP(5)
            QGTC_MF_STEP(6)
            QGTC_MF_STEP(7)
        }
#undef QGTC_MF_STEP
        MF_STAMP(5);
    } else {
        __syncthreads();
        MF_STAMP(3);
        for (int q = 0; q < kq; q++) {
            if (q == 8) MF_STAMP(4);
            const unsigned char *xs = smem + (q & 1) * MF_STAGE + (64 * wr + fl) * MF_PITCH + 16 * fh;
            const unsigned char *ws = smem + (q & 1) * MF_STAGE + MF_T * MF_PITCH + (64 * wc + fl) * MF_PITCH + 16 * fh;
            // fragments of k sub-step s+1 are read from LDS while sub-step s is multiplied
            i32x4 af[2][2], bf[2][2];
#pragma unroll
            for (int i = 0; i < 2; i++) {
                af[0][i] = *reinterpret_cast<const i32x4 *>(xs + 32 * i * MF_PITCH);
                bf[0][i] = *reinterpret_cast<const i32x4 *>(ws + 32 * i * MF_PITCH);
            }
#pragma unroll
            for (int sub = 0; sub < 4; sub++) {
                if (sub + 1 < 4) {
#pragma unroll
                    for (int i = 0; i < 2; i++) {
                        af[(sub + 1) & 1][i] = *reinterpret_cast<const i32x4 *>(xs + 32 * i * MF_PITCH + 32 * (sub + 1));
                        bf[(sub + 1) & 1][i] = *reinterpret_cast<const i32x4 *>(ws + 32 * i * MF_PITCH + 32 * (sub + 1));
                    }
                }
#pragma unroll
                for (int i = 0; i < 2; i++)
#pragma unroll
                    for (int j = 0; j < 2; j++)
#ifdef QGTC_MF_NOMFMA  // timing-only build
                        asm volatile("" ::"v"(af[sub & 1][i]), "v"(bf[sub & 1][j]));
#else
                        acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[sub & 1][i], bf[sub & 1][j], acc[i][j], 0, 0, 0);
#endif
            }
            if (q == 8) MF_STAMP(1);
            __syncthreads();
            if (q == 8) MF_STAMP(2);
        }
        // ---- result tile to LDS (the staging buffers are free: the last barrier is behind every read):
        // MFMA C/D layout col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5). The cols-layout
        // epilogue wants four consecutive ROWS of a column in one 16-byte read: it gets the tile transposed.
        MF_STAMP(5);
        // (two loops, not a select per element: the addresses are then one lane base + immediates)
        int *ctw = reinterpret_cast<int *>(smem);
        if (sh.mode == 1) {
            int *basep = ctw + (64 * wc + fl) * MF_CPITCH + 64 * wr + 4 * fh;
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 2; j++)
#pragma unroll
                    for (int r = 0; r < 16; r++) basep[32 * j * MF_CPITCH + 32 * i + (r & 3) + 8 * (r >> 2)] = acc[i][j][r];
        } else {
            int *basep = ctw + (64 * wr + 4 * fh) * MF_CPITCH + 64 * wc + fl;
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 2; j++)
#pragma unroll
                    for (int r = 0; r < 16; r++) basep[(32 * i + (r & 3) + 8 * (r >> 2)) * MF_CPITCH + 32 * j] = acc[i][j][r];
        }
    }
    __syncthreads();

    if (!expander) MF_STAMP(6);
    // ---- epilogue: a thread takes four consecutive elements of a line of the tile (rows layout /
    // float: four columns of a row; cols layout: four rows of a column), 8 adjacent lanes make a word.
    // Thread t handles quad (t & 31) of lines (t >> 5) + k * NT/32: everything but the line is invariant.
    const int *ct = reinterpret_cast<const int *>(smem);
    const bool int_rq = sh.ob <= 23;  // float(c) > 2^ob  <=>  c > 2^ob for every int c >= 0
    const int maxi = 1 << (sh.ob & 31);
    constexpr int NT = 64 * (4 + EXPW), LSTEP = NT / 32;
    const int qd = tid & 31, ln0 = tid >> 5;
    auto finish = [&](auto mode_c) {
        constexpr int MODE = decltype(mode_c)::value;
        // along-the-line coordinate of the quad (global), the line's coordinate limit
        const int along = (MODE == 1 ? m0 : n0) + 4 * qd;
        const int nvalid_line = min(max((MODE == 1 ? M : N) - along, 0), 4);  // leading elements inside the matrix
        const int across0 = (MODE == 1 ? n0 : m0) + ln0, across_lim = MODE == 1 ? N : M;
        const int rows_pad = pad8(M), row_words = step128(N) * 4, line_words = step128(M) * 4;
        const size_t oplane = MODE == 0 ? static_cast<size_t>(rows_pad) * row_words : static_cast<size_t>(pad128(N)) * line_words;
        const size_t pitch = MODE == 2 ? static_cast<size_t>(N) : (MODE == 0 ? row_words : line_words);  // output units per line
        // rows layout [ob][PAD8(M)][STEP128(N)*4] (kernel.h:357-389): word (m, n / 32); cols layout
        // [ob][PAD128(N)][STEP128(M)*4] (kernel.h:651-810 as intended): word (n, m / 32); float32 [M,N]: (m, n)
        uint32_t *outp = static_cast<uint32_t *>(pr.out) + static_cast<size_t>(across0) * pitch + (MODE == 2 ? along : (along >> 5));
        const uint32_t sh_n = 28 - 4 * (tid & 7);
        const bool lead = (tid & 7) == 0;
        const int *src = ct + ln0 * MF_CPITCH + 4 * qd;
        for (int ln = ln0, across = across0; ln < MF_T; ln += LSTEP, across += LSTEP, outp += LSTEP * pitch, src += LSTEP * MF_CPITCH) {
            const int4 v4 = *reinterpret_cast<const int4 *>(src);
            const int v[4] = {v4.x, v4.y, v4.z, v4.w};
            const int nvalid = across < across_lim ? nvalid_line : 0;
            if (MODE == 2) {  // float32 [M,N] (reference kernel.h:915-930)
                float *dst = reinterpret_cast<float *>(outp);
                if (nvalid == 4 && (N & 3) == 0) {
                    *reinterpret_cast<float4 *>(dst) = make_float4(static_cast<float>(v[0]), static_cast<float>(v[1]),
                                                                   static_cast<float>(v[2]), static_cast<float>(v[3]));
                } else {
#pragma unroll
                    for (int e = 0; e < 4; e++)
                        if (e < nvalid) dst[e] = static_cast<float>(v[e]);
                }
                continue;
            }
            uint32_t qv[4];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int c = int_rq ? (v[e] < 0 ? 1 : (v[e] > maxi ? maxi - 1 : v[e])) : requant(v[e], sh.maxv, sh.maxm1);
                qv[e] = e < nvalid ? static_cast<uint32_t>(c) : 0u;
            }
            const bool store = lead && (MODE == 0 ? across < rows_pad : true);
            uint32_t *out = outp;
            for (int p = 0; p < sh.ob; p++, out += oplane) {
                const uint32_t nib = (((qv[0] >> p) & 1u) << 3) | (((qv[1] >> p) & 1u) << 2) |
                                     (((qv[2] >> p) & 1u) << 1) | ((qv[3] >> p) & 1u);
                const uint32_t word = or_reduce8(nib << sh_n);
                if (store) out[0] = word;
            }
        }
    };
    if (sh.mode == 0) finish(std::integral_constant<int, 0>{});
    else if (sh.mode == 1) finish(std::integral_constant<int, 1>{});
    else finish(std::integral_constant<int, 2>{});
#ifdef QGTC_STAMPS
    if (!expander) MF_STAMP(7);
    if ((tid == 0 || tid == 256) && blockIdx.x < 512)
        for (int i = 0; i < 8; i++) g_stamps[blockIdx.x * 16 + (tid ? 8 : 0) + i] = st_[i];
#endif
#undef MF_STAMP
}

// ------------------------------------------------------------------------------------------
// host-side launch plumbing
// ------------------------------------------------------------------------------------------
struct Plan {
    int waves;  // waves per workgroup (in-workgroup split-K factor)
    MMShape sh;
    size_t lds;
};

// Split K over the waves of a workgroup: `per` k-quads each. Split-K only buys parallelism: every
// extra wave repeats the prologue and adds a slab to the reduction, so a launch with many tiles
// (grouped cluster batches, wide N) runs one or two waves per tile and a launch with few tiles
// (the 4096 x 4096 x 64 micro-benchmark: 256 tiles) runs eight.
constexpr long kTargetWaves = 256 * 4 * 4;  // 4 waves on every SIMD of the chip
inline void plan_split(int K, int planes, int qw, long total_tiles, Plan *pl) {
    const int kq = step128(K);
    long want = (kTargetWaves + total_tiles - 1) / (total_tiles > 0 ? total_tiles : 1);
    if (want < 1) want = 1;
    if (want > MAX_WAVES) want = MAX_WAVES;
    const int per = (kq + static_cast<int>(want) - 1) / static_cast<int>(want);
    pl->sh.per = per;
    pl->waves = (kq + per - 1) / per;
    pl->sh.waves = pl->waves;
    // every wave's staging region, then every wave's partial-sum slab (separate, so that a wave
    // can store its slab while others are still multiplying); single-wave workgroups need no slab
    pl->lds = pl->waves * (region_bytes(planes, qw) + (pl->waves > 1 ? SLAB_BYTES : 0));
}

inline MMShape base_shape(int a, int w, int ob, int mode) {
    MMShape sh{};
    sh.a = a;
    sh.w = w;
    sh.ob = ob;
    sh.mode = mode;
    sh.ab = a;
    sh.wb = w;
    sh.maxv = std::ldexp(1.0f, ob);
    sh.maxm1 = sh.maxv - 1.0f;
    return sh;
}

template <int QW, int NA, int NW, bool ZS>
int launch_single(const qgtc_problem &pr, const Plan &pl, hipStream_t st) {
    const int tiles_m = (pr.M + TM - 1) / TM, tiles_n = (pr.N + TN - 1) / TN;
    static bool attr_set = false;
    if (!attr_set) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_bitmm<QW, NA, NW, ZS>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        attr_set = true;
    }
    MMShape sh = pl.sh;
    sh.inv_tiles_n = tiles_n > 1 ? static_cast<uint32_t>((1ull << 32) / tiles_n) : 0xffffffffu;
    hipLaunchKernelGGL((k_bitmm<QW, NA, NW, ZS>), dim3(tiles_m * tiles_n), dim3(64 * pl.waves), pl.lds,
                       st, pr, sh, tiles_m, tiles_n);
    HIP_TRY(hipGetLastError());
    return QGTC_OK;
}

template <int QW, int NA, int NW, bool ZS, bool OCC>
int launch_batched(const qgtc_problem *prs, int count, int max_M, int max_N, const Plan &pl,
                   hipStream_t st) {
    const int tiles = ((max_M + TM - 1) / TM) * ((max_N + TN - 1) / TN);
    static bool attr_set = false;
    if (!attr_set) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_bitmm_batched<QW, NA, NW, ZS, OCC>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        attr_set = true;
    }
    hipLaunchKernelGGL((k_bitmm_batched<QW, NA, NW, ZS, OCC>), dim3(tiles, count), dim3(64 * pl.waves),
                       pl.lds, st, prs, pl.sh);
    HIP_TRY(hipGetLastError());
    return QGTC_OK;
}

// Kernel selection: the plane combinations the reference's drivers and benchmarks use get a
// kernel with compile-time plane loops and per-shift accumulators; everything else (any a, w in
// 1..32) runs the generic kernel, which blocks the planes 8 x 8 at a time.
template <bool ZS, typename F>
int with_kernel(int a, int w, int K, int ob, int mode, long total_tiles, Plan *pl, F &&go) {
    pl->sh = base_shape(a, w, ob, mode);
#define QGTC_FIXED(QW_, NA_, NW_)                              \
    if (a == NA_ && w == NW_) {                                \
        plan_split(K, NA_ + NW_, QW_, total_tiles, pl);        \
        return go(std::integral_constant<int, QW_>{}, std::integral_constant<int, NA_>{}, \
                  std::integral_constant<int, NW_>{});         \
    }
    QGTC_FIXED(4, 1, 1)
    QGTC_FIXED(4, 1, 2)
    QGTC_FIXED(2, 1, 4)
    QGTC_FIXED(1, 1, 8)
    QGTC_FIXED(4, 2, 2)
    QGTC_FIXED(2, 4, 4)
#undef QGTC_FIXED
    pl->sh.ab = a < 8 ? a : 8;
    pl->sh.wb = w < 8 ? w : 8;
    plan_split(K, pl->sh.ab + pl->sh.wb, 1, total_tiles, pl);
    return go(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{},
              std::integral_constant<int, 0>{});
}

template <bool ZS>
int dispatch_single(const qgtc_problem &pr, int K, int a, int w, int ob, int mode, hipStream_t st) {
    Plan pl;
    const long tiles = static_cast<long>((pr.M + TM - 1) / TM) * ((pr.N + TN - 1) / TN);
    return with_kernel<ZS>(a, w, K, ob, mode, tiles, &pl, [&](auto qw, auto na, auto nw) {
        return launch_single<decltype(qw)::value, decltype(na)::value, decltype(nw)::value, ZS>(pr, pl, st);
    });
}

template <bool ZS, bool OCC>
int dispatch_batched(const qgtc_problem *prs, int count, int max_M, int max_N, int K_hint, int a,
                     int w, int ob, int mode, hipStream_t st) {
    Plan pl;
    const long tiles = static_cast<long>(count) * ((max_M + TM - 1) / TM) * ((max_N + TN - 1) / TN);
    return with_kernel<ZS>(a, w, K_hint, ob, mode, tiles, &pl, [&](auto qw, auto na, auto nw) {
        return launch_batched<decltype(qw)::value, decltype(na)::value, decltype(nw)::value, ZS, OCC>(
            prs, count, max_M, max_N, pl, st);
    });
}

// the MFMA engine handles plane counts whose values fit a non-negative int8
inline bool mfma_ok(int a, int w) { return a >= 1 && a <= 7 && w >= 1 && w <= 7; }

// QGTC_ENGINE_AUTO: pick the engine by a two-line cost model fitted to the round-1 measurements
// (DESIGN.md section 5.4b): popcount runs at ~0.95e15 bit-ops/s plus ~3 us of launch and tail; the
// matrix-core engine pays ~6 us fixed and ~0.46 us per k-quad and 128 x 128 tile round (a quarter
// more per extra plane to expand), rounds = tiles / 256 CUs. MFMA only when it wins by 10 %.
inline bool auto_prefers_mfma(int M, int K, int N, int a, int w) {
    if (!mfma_ok(a, w)) return false;
    const double tiles = static_cast<double>((M + MF_T - 1) / MF_T) * ((N + MF_T - 1) / MF_T);
    const double rounds = tiles <= 256.0 ? 1.0 : tiles / 256.0 * 0.9;
    const int maxp = a > w ? a : w;
    const double t_mfma = 6.0 + 0.46 * step128(K) * (1.0 + 0.25 * (maxp - 1)) * rounds;
    const double t_pop = 3.0 + 2.0 * M * static_cast<double>(K) * N * a * w / 0.95e15 * 1e6;
    return t_mfma < 0.9 * t_pop;
}

int launch_mfma(const qgtc_problem &pr, int a, int w, int ob, int mode, hipStream_t st) {
    MMShape sh = base_shape(a, w, ob, mode);
    const int tiles_m = (pr.M + MF_T - 1) / MF_T, tiles_n = (pr.N + MF_T - 1) / MF_T;
    const int maxp = a > w ? a : w;
    static bool attr_set = false;
    if (!attr_set) {
#define QGTC_MF_ATTR(P, E) \
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_bitmm_mfma<P, E>), hipFuncAttributeMaxDynamicSharedMemorySize, MF_LDS));
        QGTC_MF_ATTR(1, 4) QGTC_MF_ATTR(2, 4) QGTC_MF_ATTR(4, 4) QGTC_MF_ATTR(7, 4)
        QGTC_MF_ATTR(1, 8) QGTC_MF_ATTR(2, 8) QGTC_MF_ATTR(4, 8) QGTC_MF_ATTR(7, 8)
#undef QGTC_MF_ATTR
        attr_set = true;
    }
    const dim3 grid(tiles_m * tiles_n);
    // two workgroups per CU overlap each other from 512 tiles on; below that one 12-wave workgroup per CU
    const bool wide = tiles_m * tiles_n < 512;
#define QGTC_MF_LAUNCH(P)                                                                              \
    if (wide) hipLaunchKernelGGL((k_bitmm_mfma<P, 8>), grid, dim3(768), MF_LDS, st, pr, sh, tiles_n);   \
    else hipLaunchKernelGGL((k_bitmm_mfma<P, 4>), grid, dim3(512), MF_LDS, st, pr, sh, tiles_n);
    if (maxp <= 1) { QGTC_MF_LAUNCH(1) }
    else if (maxp <= 2) { QGTC_MF_LAUNCH(2) }
    else if (maxp <= 4) { QGTC_MF_LAUNCH(4) }
    else { QGTC_MF_LAUNCH(7) }
#undef QGTC_MF_LAUNCH
    HIP_TRY(hipGetLastError());
    return QGTC_OK;
}

int check_mm_args(const uint32_t *X, const uint32_t *W, const void *out, int M, int K, int N,
                  int a, int w) {
    if (!X || !W || !out) return QGTC_EINVAL;
    if (M <= 0 || K <= 0 || N <= 0) return QGTC_EINVAL;
    if (!bits_ok(a) || !bits_ok(w)) return QGTC_EINVAL;
    if (!aligned16(X) || !aligned16(W)) return QGTC_EALIGN;
    if (step128(K) * 4 >= (1 << 24)) return QGTC_EINVAL;  // 24-bit row-stride multiplies in the kernel
    return QGTC_OK;
}

// in-kernel byte offsets inside one operand are 32-bit
inline bool words_ok(size_t x_words, size_t w_words) {
    return x_words < (1ull << 30) && w_words < (1ull << 30);  // < 4 GiB per packed operand
}

int grid_for(size_t work_items, int per_block) {
    size_t blocks = (work_items + per_block - 1) / per_block;
    if (blocks < 1) blocks = 1;
    if (blocks > 2048) blocks = 2048;  // 256 CUs x 8 resident blocks, grid-stride the rest
    return static_cast<int>(blocks);
}

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

const char *qgtc_last_hip_error(void) { return g_hip_err; }

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
            constexpr int UNR = 4;
            hipLaunchKernelGGL(k_val2bit_rows_v4<UNR>, dim3(grid_for((units + UNR - 1) / UNR, 4)), dim3(256),
                               0, st, x, H, W, nbits, ub, ubm1, out, rows_pad, row_words);
        } else {
            hipLaunchKernelGGL(k_val2bit_rows, dim3(grid_for(units, 4)), dim3(256), 0, st, x, H, W, nbits,
                               ub, ubm1, out, rows_pad, row_words);
        }
    } else {
        if (out_words < qgtc_cols_words(H, W, nbits, output_layer)) return QGTC_ESIZE;
        const int lines = output_layer ? pad8(W) : pad128(W), line_words = step128(H) * 4;
        const size_t units = static_cast<size_t>((lines + 63) / 64) * line_words;
        const dim3 g(grid_for(units, 4)), b(256);
        if (nbits <= 1)
            hipLaunchKernelGGL(k_val2bit_cols<1>, g, b, 0, st, x, H, W, nbits, ub, ubm1, out, lines, line_words);
        else if (nbits <= 2)
            hipLaunchKernelGGL(k_val2bit_cols<2>, g, b, 0, st, x, H, W, nbits, ub, ubm1, out, lines, line_words);
        else if (nbits <= 4)
            hipLaunchKernelGGL(k_val2bit_cols<4>, g, b, 0, st, x, H, W, nbits, ub, ubm1, out, lines, line_words);
        else if (nbits <= 8)
            hipLaunchKernelGGL(k_val2bit_cols<8>, g, b, 0, st, x, H, W, nbits, ub, ubm1, out, lines, line_words);
        else
            hipLaunchKernelGGL(k_val2bit_cols<32>, g, b, 0, st, x, H, W, nbits, ub, ubm1, out, lines, line_words);
    }
    HIP_TRY(hipGetLastError());
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
    hipLaunchKernelGGL(k_bit2val, dim3(grid_for(total, 256)), dim3(256), 0, st, bits,
                       static_cast<unsigned long long>(bits_words), nbits, H, W, col_major, plane,
                       line_words, out);
    HIP_TRY(hipGetLastError());
    return QGTC_OK;
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
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (((flags & QGTC_ENGINE_MFMA) && mfma_ok(bit1, bit2)) || ((flags & QGTC_ENGINE_AUTO) && auto_prefers_mfma(M, K, N, bit1, bit2)))
        return launch_mfma(pr, bit1, bit2, output_bit, cols ? 1 : 0, st);
    if (flags & QGTC_NO_ZERO_SKIP)
        return dispatch_single<false>(pr, K, bit1, bit2, output_bit, cols ? 1 : 0, st);
    return dispatch_single<true>(pr, K, bit1, bit2, output_bit, cols ? 1 : 0, st);
}

int qgtc_bitmm2int(const uint32_t *X, size_t x_words, const uint32_t *W, size_t w_words, int M,
                   int K, int N, int bit1, int bit2, int pad_128, float *out, size_t out_elems,
                   unsigned flags, void *stream) {
    int rc = check_mm_args(X, W, out, M, K, N, bit1, bit2);
    if (rc != QGTC_OK) return rc;
    if (!words_ok(x_words, w_words)) return QGTC_EINVAL;
    if (out_elems < static_cast<size_t>(M) * N) return QGTC_ESIZE;
    qgtc_problem pr{X, W, out, x_words, w_words, M, K, N, pad_128 ? pad128(N) : pad8(N), 0, nullptr};
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (((flags & QGTC_ENGINE_MFMA) && mfma_ok(bit1, bit2)) || ((flags & QGTC_ENGINE_AUTO) && auto_prefers_mfma(M, K, N, bit1, bit2)))
        return launch_mfma(pr, bit1, bit2, 1, 2, st);
    if (flags & QGTC_NO_ZERO_SKIP) return dispatch_single<false>(pr, K, bit1, bit2, 1, 2, st);
    return dispatch_single<true>(pr, K, bit1, bit2, 1, 2, st);
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
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    HIP_TRY(hipEventRecord(e0, st));
    for (int i = 0; i < reps && rc == QGTC_OK; i++)
        rc = qgtc_bitmm2bit(X, x_words, W, w_words, M, K, N, bit1, bit2, output_bit, out, out_words,
                            flags, stream);
    hipError_t e = hipEventRecord(e1, st);
    if (e == hipSuccess) e = hipEventSynchronize(e1);
    float ms = 0.0f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
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

int qgtc_bitmm_batched(const qgtc_problem *problems, int count, int max_M, int max_K, int max_N,
                       int bit1, int bit2, int output_bit, int mode, unsigned flags,
                       void *stream) {
    if (!problems || count <= 0 || max_M <= 0 || max_K <= 0 || max_N <= 0) return QGTC_EINVAL;
    if (!bits_ok(bit1) || !bits_ok(bit2) || mode < 0 || mode > 2) return QGTC_EINVAL;
    if (mode != 2 && !bits_ok(output_bit)) return QGTC_EINVAL;
    if (count > 65535) return QGTC_EINVAL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    // K is per problem; the split-K factor and chunk size are chosen for the longest K. Any
    // choice is correct; this only affects speed.
    const int k_hint = max_K;
    const int ob_ = mode == 2 ? 1 : output_bit;
    if (flags & QGTC_NO_ZERO_SKIP)
        return dispatch_batched<false, false>(problems, count, max_M, max_N, k_hint, bit1, bit2, ob_, mode, st);
    if (flags & QGTC_ZERO_JUMP)  // the descriptors carry occupancy bitmaps (qgtc_tile_occupancy)
        return dispatch_batched<true, true>(problems, count, max_M, max_N, k_hint, bit1, bit2, ob_, mode, st);
    return dispatch_batched<true, false>(problems, count, max_M, max_N, k_hint, bit1, bit2, ob_, mode, st);
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
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    HIP_TRY(hipEventRecord(e0, st));
    for (int i = 0; i < reps && rc == QGTC_OK; i++) rc = qgtc_i8gemm(A, Bt, M, K, N, C, c_elems, stream);
    hipError_t e = hipEventRecord(e1, st);
    if (e == hipSuccess) e = hipEventSynchronize(e1);
    float ms = 0.0f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (rc != QGTC_OK) return rc;
    if (e != hipSuccess) return hip_fail(e, "profile events");
    *elapsed_ms = ms;
    return QGTC_OK;
}

}  // extern "C"
