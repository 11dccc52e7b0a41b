// qgtc_hip.hip — MI355X (gfx950 / CDNA4) kernels and C-ABI of the QGTC bit-GEMM hot path.
//
// What is here (reference file:line each piece replaces is in include/qgtc.h):
//   * fused quantise + bit-plane pack          (val2bit, rows and cols layouts)
//   * bit-plane unpack                         (bit2val)
//   * multi-plane 1-bit GEMM: AND + popcount (v_and_b32 / v_bcnt_u32_b32) with shift-accumulate
//     into int32, in-workgroup split-K, zero-tile skipping, and a fused epilogue that either
//     re-quantises and re-packs (rows / cols layout) or converts to float32
//   * tile counters, the 200-rep profile loop, and a grouped (batched) launch.
//
// Design notes live in DESIGN.md; the short version of the GEMM kernel:
//   - a workgroup owns a TM x TN output tile for the whole K range (no inter-workgroup
//     reduction, so results are exact and order-independent); its WK waves split each staged
//     K chunk among themselves and are summed through LDS at the end;
//   - both operands are staged global -> LDS in 16-byte granules (= 128 bits of one packed
//     row), laid out [plane][k-quad][row] so that the 8 distinct granules a wave reads per
//     ds_read_b128 are contiguous (conflict-free) and the staging ds_write_b128 of 8
//     consecutive k-quads hits 8 distinct 4-bank groups (row count padded to odd);
//   - each lane keeps an R x C register micro-tile: per k-quad it reads R + C granules and
//     issues R*C*4 v_and_b32 + v_bcnt_u32_b32 pairs;
//   - while staging X the wave ballots "granule != 0" and ORs a per-k-quad occupancy bitmap
//     into LDS; compute waves skip k-quads whose TM x 128-bit X tile is all zero (wave-uniform
//     scalar branch, no divergence).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>

#include "qgtc.h"

namespace {

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
// ------------------------------------------------------------------------------------------
#ifdef QGTC_STAMPS  // diagnostic build only (tools/kbench.hip): per-phase s_memtime stamps
__device__ unsigned long long g_stamps[1024 * 32];
#define STAMP(slot)                                                                      \
    do {                                                                                 \
        if (threadIdx.x == 0 && (slot) < 32) g_stamps[blockIdx.x % 1024 * 32 + (slot)] = clock64(); \
    } while (0)
#else
#define STAMP(slot) do { } while (0)
#endif

struct MMShape {           // per-launch constants
    int a, w, ob;          // planes of X, planes of W, output planes
    int mode;              // 0 rows-layout bits, 1 cols-layout bits, 2 float32
    int qc;                // k-quads (128-bit steps) staged per chunk: power of two, WK..64
    int ab, wb;            // plane blocking (planes staged at once)
    float maxv, maxm1;     // 2^ob and 2^ob - 1 as float (requant)
};

// acc += popcount(x & w): v_and_b32 + v_bcnt_u32_b32 with the accumulator as the add operand
// (hipcc otherwise emits v_bcnt(...,0) + v_add3_u32, 2.5 instructions per pair instead of 2).
__device__ __forceinline__ void and_popc_acc(uint32_t &acc, uint32_t x, uint32_t w) {
    const uint32_t t = x & w;
    asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(acc) : "v"(t));
}

__device__ __forceinline__ int requant(int c, float maxv, float maxm1) {
    // reference kernel.h:31-37 called as quantize(c, ob, 1<<ob, 0): float compare, then the
    // (val-min)*2^ob/(max-min) scaling, which is the identity for min=0, max=2^ob.
    float val = static_cast<float>(c);
    if (val > maxv) val = maxm1;
    if (val < 0.0f) val = 1.0f;
    return val >= 2147483648.0f ? 2147483647 : static_cast<int>(val);
}

template <int R, int C, int WK>
struct MMCfg {
    static constexpr int LM = 8, LN = 8;
    static constexpr int TM = LM * R, TN = LN * C;
    static constexpr int NT = 64 * WK;
    static constexpr int XR = TM + 1, WR = TN + 1;  // odd row counts: conflict-free staging writes
    static constexpr int TNP = TN + 8;              // reduction row pitch (words)
    static constexpr int RED_BYTES = WK * TM * TNP * 4;
    static constexpr int GPT = 8;                   // granules a thread may prefetch per stage
};

// bytes of one LDS staging buffer (there are two)
template <int R, int C, int WK>
__host__ __device__ constexpr size_t mm_stage_bytes(int ab, int wb, int qc) {
    using Cfg = MMCfg<R, C, WK>;
    return static_cast<size_t>(ab * Cfg::XR + wb * Cfg::WR) * qc * 16;
}

template <int R, int C, int WK>
__host__ __device__ constexpr size_t mm_lds_bytes(int ab, int wb, int qc) {
    using Cfg = MMCfg<R, C, WK>;
    size_t stage2 = 2 * mm_stage_bytes<R, C, WK>(ab, wb, qc);
    size_t body = stage2 > static_cast<size_t>(Cfg::RED_BYTES) ? stage2 : Cfg::RED_BYTES;
    return body;
}

// Position in the (X plane block, W plane block, K chunk) iteration space.
struct MMCursor {
    int pa0, pw0, q0;
    bool valid;
};

// One output tile (tm, tn) of one problem. All threads of the workgroup call this.
//
// Software pipeline, one barrier per stage: while the waves compute stage s out of LDS buffer
// s&1, the global loads of stage s+1 are in flight into registers (GPT granules per thread);
// they are written to buffer (s+1)&1 at the top of the next iteration.
//
// Zero-tile skipping (ZS): before a wave spends R*C*4 AND+popcount pairs on a k-quad it ORs the
// R granules each lane just read from LDS and ballots: if the whole TM x 128-bit X tile is zero
// the wave skips the k-quad (wave-uniform scalar branch). For products with several W planes
// the test is hoisted out of the plane loop (one pre-pass over the wave's k-quads per X plane).
template <int R, int C, int WK, bool ZS>
__device__ __forceinline__ void mm_tile(const qgtc_problem &pr, const MMShape &sh, int tm, int tn,
                                        int tiles_m, int tiles_n, unsigned char *smem) {
    using Cfg = MMCfg<R, C, WK>;
    constexpr int TM = Cfg::TM, TN = Cfg::TN, NT = Cfg::NT, XR = Cfg::XR, WR = Cfg::WR,
                  TNP = Cfg::TNP, GPT = Cfg::GPT;
    STAMP(26);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wk = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lm = lane >> 3, ln = lane & 7;

    const int M = pr.M, K = pr.K, N = pr.N;
    const int kq = step128(K);                   // k-quads per packed row
    const uint32_t kw = static_cast<uint32_t>(kq) * 4u;
    const uint32_t x_plane = static_cast<uint32_t>(pad8(M)) * kw;  // < 2^32 (checked on the host)
    const uint32_t w_plane = static_cast<uint32_t>(pr.w_lines) * kw;
    const int m0 = tm * TM, n0 = tn * TN;
#ifdef QGTC_STAMPS
    if (M > 0) STAMP(27);  // after the first use of a kernel argument
#endif
    const int qc = sh.qc, lqc = 31 - __clz(qc);
    const int qw = qc / WK;                      // k-quads per wave per chunk

    const uint32_t stage_granules = static_cast<uint32_t>(sh.ab * XR + sh.wb * WR) * qc;
    uint4 *stage_base = reinterpret_cast<uint4 *>(smem);
    int *red = reinterpret_cast<int *>(smem);

    auto advance = [&](MMCursor &c) {
        c.q0 += qc;
        if (c.q0 >= kq) {
            c.q0 = 0;
            c.pw0 += sh.wb;
            if (c.pw0 >= sh.w) {
                c.pw0 = 0;
                c.pa0 += sh.ab;
                if (c.pa0 >= sh.a) c.valid = false;
            }
        }
    };

    // Per-thread slot table, computed once per tile. A stage holds ab*TM*qc X granules and
    // wb*TN*qc W granules; X granules use slots [0, nxs), W granules slots [nxs, nslots): slot u
    // of this thread is granule g = tid + u*NT (resp. tid + (u-nxs)*NT) of its operand, decoded
    // as (plane, row, k-quad) with the k-quad fastest, so consecutive lanes load consecutive 16
    // bytes of a packed row. Only the stage origin (plane block, first k-quad) changes from
    // stage to stage, and it is wave-uniform: loads are SGPR base + 32-bit VGPR offset.
    const int xg_full = sh.ab * TM * qc, wg_full = sh.wb * TN * qc;
    const int nxs = (xg_full + NT - 1) / NT;
    const int nslots = nxs + (wg_full + NT - 1) / NT;  // wave-uniform, <= GPT (plan_for)
    uint32_t s_boff[GPT];  // BYTE offset of the granule from the stage origin (< 4 GiB, host-checked)
    int32_t s_lim[GPT];    // largest stage origin (in words) for which the granule is in bounds
    uint32_t s_meta[GPT];  // bit0 valid slot, bit2 row in range, bits 8..15 plane, bits 16.. k-quad
    {
        // cheap decode: 24-bit multiplies (v_mul_u32_u24 is full rate, v_mul_lo_u32 is not) and
        // no divergent branches. q and the row-within-slot are the same for every slot.
        const uint32_t q = tid & (qc - 1), t0 = tid >> lqc, tstep = NT >> lqc;
        const uint32_t xpl_lo = x_plane & 0xffffffu, xpl_hi = x_plane >> 24;
        const uint32_t wpl_lo = w_plane & 0xffffffu, wpl_hi = w_plane >> 24;
#pragma unroll
        for (int u = 0; u < GPT; u++) {
            s_boff[u] = 0u;
            s_lim[u] = -8;
            s_meta[u] = 0u;
            if (u >= nslots) continue;
            const bool is_x = u < nxs;
            const uint32_t t = t0 + (is_x ? u : u - nxs) * tstep;    // flattened (plane,row) index
            const uint32_t row = is_x ? (t & (TM - 1)) : (t & (TN - 1));
            const uint32_t pl = is_x ? (t / TM) : (t / TN);
            const bool in_stage = is_x ? (t < static_cast<uint32_t>(sh.ab * TM)) : (t < static_cast<uint32_t>(sh.wb * TN));
            const uint32_t grow = (is_x ? m0 : n0) + row;
            const uint32_t ploff = is_x ? (__umul24(pl, xpl_lo) + (__umul24(pl, xpl_hi) << 24))
                                        : (__umul24(pl, wpl_lo) + (__umul24(pl, wpl_hi) << 24));
            const uint32_t goff = ploff + __umul24(grow, kw) + q * 4u;  // kw < 2^24 (host-checked)
            const int32_t words = static_cast<int32_t>(is_x ? pr.x_words : pr.w_words);
            const bool row_ok = static_cast<int>(grow) < (is_x ? M : N);
            s_boff[u] = goff * 4u;
            s_lim[u] = in_stage ? words - 4 - static_cast<int32_t>(goff) : -8;
            s_meta[u] = (in_stage ? 1u : 0u) | (row_ok ? 4u : 0u) | (pl << 8) | (q << 16);
        }
    }

    // issue the global loads of one stage into registers; out-of-range rows / planes / k-quads /
    // words read as zero.
    uint4 pre[GPT];
    auto issue = [&](const MMCursor &c) {
        const int na = min(sh.ab, sh.a - c.pa0), nw = min(sh.wb, sh.w - c.pw0);
        const bool full = na == sh.ab && nw == sh.wb && c.q0 + qc <= kq;  // wave-uniform
        const int32_t xo = static_cast<int32_t>(c.pa0 * x_plane + c.q0 * 4u);
        const int32_t wo = static_cast<int32_t>(c.pw0 * w_plane + c.q0 * 4u);
        const char *xb = reinterpret_cast<const char *>(pr.X + xo);   // uniform stage origins
        const char *wbp = reinterpret_cast<const char *>(pr.W + wo);
        // Fast path only inside the load loop (exec-masked loads, no merge with other values, so
        // the compiler leaves all of them in flight); granules that straddle the end of a
        // mis-sized buffer are fixed up afterwards in a wave-uniform, rarely taken branch.
        uint32_t partial = 0u;
#pragma unroll
        for (int u = 0; u < GPT; u++) {
            if (u >= nslots) break;
            const bool is_x = u < nxs;
            const uint32_t meta = s_meta[u];
            bool ok = (meta & 5u) == 5u;
            if (!full) {
                const int pl = (meta >> 8) & 0xff, q = meta >> 16;
                ok = ok && c.q0 + q < kq && pl < (is_x ? na : nw);
            }
            const int32_t org = is_x ? xo : wo, lim = s_lim[u];
            pre[u] = make_uint4(0u, 0u, 0u, 0u);
            if (ok && org <= lim) {
                // SGPR base + zero-extended 32-bit VGPR byte offset (global_load saddr form)
                pre[u] = is_x ? *reinterpret_cast<const uint4 *>(xb + s_boff[u])
                              : *reinterpret_cast<const uint4 *>(wbp + s_boff[u]);
            }
            if (ok && org > lim && org < lim + 4) partial |= 1u << u;
        }
        if (__builtin_expect(__ballot(partial != 0u) != 0ull, 0)) {
#pragma unroll
            for (int u = 0; u < GPT; u++) {
                if (u >= nslots) break;
                const bool is_x = u < nxs;
                if ((partial >> u) & 1u)  // the buffer ends inside this granule
                    pre[u] = ldg4(is_x ? pr.X : pr.W, is_x ? pr.x_words : pr.w_words,
                                  static_cast<unsigned long long>(is_x ? xo : wo) + (s_boff[u] >> 2));
            }
        }
    };

    MMCursor cur{0, 0, 0, true};
    STAMP(0);
    issue(cur);
    STAMP(1);

    // LDS granule index of each slot ([plane][k-quad][row], X block then W block); computed
    // while the first loads are in flight
    uint32_t s_loff[GPT];
#pragma unroll
    for (int u = 0; u < GPT; u++) {
        const uint32_t meta = s_meta[u];
        const uint32_t pl = (meta >> 8) & 0xff, q = meta >> 16;
        const bool is_x = u < nxs;
        const uint32_t t = (tid >> lqc) + (is_x ? u : u - nxs) * (NT >> lqc);
        s_loff[u] = is_x ? (pl * qc + q) * XR + (t & (TM - 1))
                         : sh.ab * qc * XR + (pl * qc + q) * WR + (t & (TN - 1));
    }

    uint32_t tot[R][C];  // unsigned: the reference's int32 accumulation wraps on overflow
#pragma unroll
    for (int i = 0; i < R; i++)
#pragma unroll
        for (int j = 0; j < C; j++) tot[i][j] = 0u;

    for (int it = 0; cur.valid; it++) {
        const int buf = it & 1;
        {   // write the prefetched granules into LDS buffer `buf`
            uint4 *dst = stage_base + buf * stage_granules;
#pragma unroll
            for (int u = 0; u < GPT; u++) {
                if (u >= nslots) break;
                if (s_meta[u] & 1u) dst[s_loff[u]] = pre[u];
            }
        }
        STAMP(2 + it * 4);
        __syncthreads();
        STAMP(3 + it * 4);

        MMCursor nxt = cur;
        advance(nxt);
        if (nxt.valid) issue(nxt);  // loads fly while this stage is computed
        STAMP(4 + it * 4);

        // ---- compute: this wave's k-quads [wk*qw, wk*qw+qw) of the stage ----
        const int na = min(sh.ab, sh.a - cur.pa0), nw = min(sh.wb, sh.w - cur.pw0);
        const uint4 *Xs = stage_base + buf * stage_granules;
        const uint4 *Ws = Xs + sh.ab * qc * XR;
        const int qlo = wk * qw;
        const int qhi = min(qlo + qw, min(qc, kq - cur.q0));  // k-quads of this chunk that exist
        for (int pa = 0; pa < na; pa++) {
            // occupancy of this wave's k-quads of X plane pa (bit q - qlo); hoisted out of the
            // W-plane loop when there are several W planes, tested inline otherwise
            unsigned long long occ = ~0ull;
            const bool prepass = ZS && nw > 2;
            if (prepass) {
                occ = 0ull;
                for (int q = qlo; q < qhi; q++) {
                    const uint4 *xp = Xs + (pa * qc + q) * XR + lm;
                    uint32_t any = 0u;
#pragma unroll
                    for (int i = 0; i < R; i++) {
                        const uint4 g = xp[i * Cfg::LM];
                        any |= (g.x | g.y) | (g.z | g.w);
                    }
                    if (__ballot(any != 0u)) occ |= 1ull << (q - qlo);
                }
                if (occ == 0ull) continue;  // this wave's slice of the X plane is all zero
            }
            for (int pw = 0; pw < nw; pw++) {
                uint32_t part[R][C];
#pragma unroll
                for (int i = 0; i < R; i++)
#pragma unroll
                    for (int j = 0; j < C; j++) part[i][j] = 0u;
                bool have = false;  // wave-uniform: at least one k-quad was multiplied
                // software pipeline over the wave's k-quads: the granules of k-quad q+1 are read
                // from LDS while k-quad q is being multiplied
                uint4 xg[R], wg[C], xn[R], wn[C];
                auto lds_read = [&](int q, uint4 (&xr)[R], uint4 (&wr)[C]) {
                    const uint4 *xp = Xs + (pa * qc + q) * XR + lm;
                    const uint4 *wp = Ws + (pw * qc + q) * WR + ln;
#pragma unroll
                    for (int i = 0; i < R; i++) xr[i] = xp[i * Cfg::LM];
#pragma unroll
                    for (int j = 0; j < C; j++) wr[j] = wp[j * Cfg::LN];
                };
                // multiply one k-quad held in registers (or skip it when its X tile is zero)
                auto mac = [&](int q, const uint4 (&xr)[R], const uint4 (&wr)[C]) {
                    bool skip = prepass && !((occ >> (q - qlo)) & 1ull);
                    if (ZS && !prepass) {
                        uint32_t any = 0u;
#pragma unroll
                        for (int i = 0; i < R; i++) any |= (xr[i].x | xr[i].y) | (xr[i].z | xr[i].w);
                        skip = __ballot(any != 0u) == 0ull;  // zero X tile
                    }
                    if (skip) return;
                    // word-major order: consecutive v_bcnt hit different accumulators
                    have = true;
#pragma unroll
                    for (int i = 0; i < R; i++)
#pragma unroll
                        for (int j = 0; j < C; j++) and_popc_acc(part[i][j], xr[i].x, wr[j].x);
#pragma unroll
                    for (int i = 0; i < R; i++)
#pragma unroll
                        for (int j = 0; j < C; j++) and_popc_acc(part[i][j], xr[i].y, wr[j].y);
#pragma unroll
                    for (int i = 0; i < R; i++)
#pragma unroll
                        for (int j = 0; j < C; j++) and_popc_acc(part[i][j], xr[i].z, wr[j].z);
#pragma unroll
                    for (int i = 0; i < R; i++)
#pragma unroll
                        for (int j = 0; j < C; j++) and_popc_acc(part[i][j], xr[i].w, wr[j].w);
                };
                // two register sets in ping-pong (no register copies)
                if (qlo < qhi) lds_read(qlo, xg, wg);
                for (int q = qlo; q < qhi; q += 2) {
                    if (q + 1 < qhi) lds_read(q + 1, xn, wn);
                    mac(q, xg, wg);
                    if (q + 1 >= qhi) break;
                    if (q + 2 < qhi) lds_read(q + 2, xg, wg);
                    mac(q + 1, xn, wn);
                }
                const int s = cur.pa0 + pa + cur.pw0 + pw;  // reference kernel.h:295,340
                if (have && s < 32) {
#pragma unroll
                    for (int i = 0; i < R; i++)
#pragma unroll
                        for (int j = 0; j < C; j++) tot[i][j] += part[i][j] << s;
                }
            }
        }
        STAMP(5 + it * 4);
        cur = nxt;
    }
    __syncthreads();  // every wave is done with the staging buffers: reuse them for the reduction
    STAMP(28);

    // ---- in-workgroup split-K reduction through LDS ----
#pragma unroll
    for (int i = 0; i < R; i++)
#pragma unroll
        for (int j = 0; j < C; j++)
            red[(wk * TM + lm + i * Cfg::LM) * TNP + ln + j * Cfg::LN] = static_cast<int>(tot[i][j]);
    __syncthreads();
    STAMP(29);

    const bool last_n = tn == tiles_n - 1, last_m = tm == tiles_m - 1;
    if (sh.mode == 2) {
        // float32 [M,N] (reference kernel.h:915-930)
        float *out = static_cast<float *>(pr.out);
        for (int e = tid; e < TM * TN; e += NT) {
            const int row = e / TN, col = e % TN;
            int v = 0;
#pragma unroll
            for (int k = 0; k < WK; k++) v += red[(k * TM + row) * TNP + col];
            const int m = m0 + row, n = n0 + col;
            if (m < M && n < N) out[static_cast<size_t>(m) * N + n] = static_cast<float>(v);
        }
    } else if (sh.mode == 0) {
        // rows layout [ob][PAD8(M)][STEP128(N)*4] (reference kernel.h:357-389)
        uint32_t *out = static_cast<uint32_t *>(pr.out);
        const int rows_pad = pad8(M), row_words = step128(N) * 4;
        const size_t oplane = static_cast<size_t>(rows_pad) * row_words;
        for (int e = tid; e < TM * TN; e += NT) {
            const int row = e / TN, col = e % TN;  // each 32-lane half: 32 columns of one row
            int v = 0;
#pragma unroll
            for (int k = 0; k < WK; k++) v += red[(k * TM + row) * TNP + col];
            const int m = m0 + row, n = n0 + col;
            const int qv = (m < M && n < N) ? requant(v, sh.maxv, sh.maxm1) : 0;
            for (int p = 0; p < sh.ob; p++) {
                const unsigned long long mk = __ballot((qv >> p) & 1);
                if ((lane & 31) == 0 && m < rows_pad) {
                    const uint32_t half = lane ? static_cast<uint32_t>(mk >> 32) : static_cast<uint32_t>(mk);
                    out[p * oplane + static_cast<size_t>(m) * row_words + (n >> 5)] = __brev(half);
                }
            }
        }
        if (last_n) {  // zero the row words beyond the last column tile
            const int w0 = tiles_n * (TN / 32), nz = row_words - w0;
            for (int e = tid; e < sh.ob * TM; e += NT) {
                const int row = e & (TM - 1), p = e / TM;
                if (m0 + row < rows_pad)
                    for (int wi = 0; wi < nz; wi++)
                        out[p * oplane + static_cast<size_t>(m0 + row) * row_words + w0 + wi] = 0u;
            }
        }
    } else {
        // cols layout [ob][PAD128(N)][STEP128(M)*4] (intended semantics of kernel.h:651-810)
        uint32_t *out = static_cast<uint32_t *>(pr.out);
        const int lines = pad128(N), line_words = step128(M) * 4;
        const size_t oplane = static_cast<size_t>(lines) * line_words;
        for (int e = tid; e < TM * TN; e += NT) {
            const int row = e % TM, col = e / TM;  // each 32-lane half: 32 rows of one column
            int v = 0;
#pragma unroll
            for (int k = 0; k < WK; k++) v += red[(k * TM + row) * TNP + col];
            const int m = m0 + row, n = n0 + col;
            const int qv = (m < M && n < N) ? requant(v, sh.maxv, sh.maxm1) : 0;
            for (int p = 0; p < sh.ob; p++) {
                const unsigned long long mk = __ballot((qv >> p) & 1);
                if ((lane & 31) == 0 && n < lines && (m >> 5) < line_words) {
                    const uint32_t half = lane ? static_cast<uint32_t>(mk >> 32) : static_cast<uint32_t>(mk);
                    out[p * oplane + static_cast<size_t>(n) * line_words + (m >> 5)] = __brev(half);
                }
            }
        }
        // zero what no tile computes: lines past the last column tile, words past the last row tile
        const int l_end = last_n ? lines : min(lines, n0 + TN);
        const int w_core0 = m0 >> 5, w_core1 = min(line_words, (m0 + TM) >> 5);
        const int w_end = last_m ? line_words : w_core1;
        const int nl = l_end - n0, nwd = w_end - w_core0;
        for (int e = tid; e < sh.ob * nl * nwd; e += NT) {
            const int wi = w_core0 + e % nwd, line = n0 + (e / nwd) % nl, p = e / (nwd * nl);
            const bool core = line < n0 + TN && wi < w_core1;
            if (!core) out[p * oplane + static_cast<size_t>(line) * line_words + wi] = 0u;
        }
    }
    STAMP(30);
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

template <int R, int C, int WK, bool ZS>
__global__ __launch_bounds__(64 * WK) void k_bitmm(qgtc_problem pr, MMShape sh, int tiles_m,
                                                   int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    mm_tile<R, C, WK, ZS>(pr, sh, tile / tiles_n, tile % tiles_n, tiles_m, tiles_n, smem);
}

// grouped launch: blockIdx.y = problem, blockIdx.x = tile (surplus tiles exit at once)
template <int R, int C, int WK, bool ZS>
__global__ __launch_bounds__(64 * WK) void k_bitmm_batched(const qgtc_problem *__restrict__ prs,
                                                           MMShape sh) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    using Cfg = MMCfg<R, C, WK>;
    const qgtc_problem pr = prs[blockIdx.y];
    const int tiles_m = (pr.M + Cfg::TM - 1) / Cfg::TM, tiles_n = (pr.N + Cfg::TN - 1) / Cfg::TN;
    const int tile = blockIdx.x;
    if (tile >= tiles_m * tiles_n) return;
    mm_tile<R, C, WK, ZS>(pr, sh, tile / tiles_n, tile % tiles_n, tiles_m, tiles_n, smem);
}

// ------------------------------------------------------------------------------------------
// host-side launch plumbing
// ------------------------------------------------------------------------------------------
constexpr size_t kLdsBudget = 80 * 1024;  // keeps two workgroups per CU resident (160 KiB LDS)

struct Plan {
    int wk;  // waves per workgroup (split-K factor)
    MMShape sh;
    size_t lds;
};

inline int pow2_floor(int x) {
    int p = 1;
    while (p * 2 <= x) p *= 2;
    return p;
}

template <int R, int C, int WK>
bool plan_for(int K, int a, int w, int ob, int mode, Plan *pl) {
    using Cfg = MMCfg<R, C, WK>;
    const int kq = step128(K);
    MMShape sh{};
    sh.a = a;
    sh.w = w;
    sh.ob = ob;
    sh.mode = mode;
    sh.maxv = std::ldexp(1.0f, ob);
    sh.maxm1 = sh.maxv - 1.0f;
    // k-quads per stage: a power of two >= WK (every wave gets >= 1), as small as the pipeline
    // allows (more stages = earlier first compute); planes are blocked until one stage fits the
    // per-thread prefetch registers (GPT granules) and two stages fit the LDS budget.
    int qc = WK > 8 ? WK : 8;
    while (qc > WK && qc / 2 >= kq) qc /= 2;
    int ab = a, wb = w;
    auto fits = [&](int ab_, int wb_, int qc_) {
        const int xs = (ab_ * Cfg::TM * qc_ + Cfg::NT - 1) / Cfg::NT;
        const int ws = (wb_ * Cfg::TN * qc_ + Cfg::NT - 1) / Cfg::NT;
        return xs + ws <= Cfg::GPT && mm_lds_bytes<R, C, WK>(ab_, wb_, qc_) <= kLdsBudget;
    };
    while (!fits(ab, wb, qc)) {
        if (wb >= ab && wb > 1) wb = (wb + 1) / 2;
        else if (ab > 1) ab = (ab + 1) / 2;
        else if (qc > WK) qc /= 2;
        else return false;
    }
    // with room to spare, stage more K per barrier (fewer barriers) up to 64 k-quads
    while (qc < 64 && qc < kq && fits(ab, wb, qc * 2)) qc *= 2;
    sh.qc = qc;
    sh.ab = ab;
    sh.wb = wb;
    pl->wk = WK;
    pl->sh = sh;
    pl->lds = mm_lds_bytes<R, C, WK>(ab, wb, qc);
    return true;
}

template <int R, int C, int WK, bool ZS>
int launch_single(const qgtc_problem &pr, const Plan &pl, hipStream_t st) {
    using Cfg = MMCfg<R, C, WK>;
    const int tiles_m = (pr.M + Cfg::TM - 1) / Cfg::TM, tiles_n = (pr.N + Cfg::TN - 1) / Cfg::TN;
    static bool attr_set = false;
    if (!attr_set) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_bitmm<R, C, WK, ZS>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize,
                                    static_cast<int>(kLdsBudget)));
        attr_set = true;
    }
    hipLaunchKernelGGL((k_bitmm<R, C, WK, ZS>), dim3(tiles_m * tiles_n), dim3(Cfg::NT), pl.lds, st,
                       pr, pl.sh, tiles_m, tiles_n);
    HIP_TRY(hipGetLastError());
    return QGTC_OK;
}

template <int R, int C, int WK, bool ZS>
int launch_batched(const qgtc_problem *prs, int count, int max_M, int max_N, const Plan &pl,
                   hipStream_t st) {
    using Cfg = MMCfg<R, C, WK>;
    const int tiles = ((max_M + Cfg::TM - 1) / Cfg::TM) * ((max_N + Cfg::TN - 1) / Cfg::TN);
    static bool attr_set = false;
    if (!attr_set) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_bitmm_batched<R, C, WK, ZS>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize,
                                    static_cast<int>(kLdsBudget)));
        attr_set = true;
    }
    hipLaunchKernelGGL((k_bitmm_batched<R, C, WK, ZS>), dim3(tiles, count), dim3(Cfg::NT), pl.lds,
                       st, prs, pl.sh);
    HIP_TRY(hipGetLastError());
    return QGTC_OK;
}

// Pick the split-K factor from the K extent: short K (feature/weight products, K <= 512 bits)
// has too few k-quads to feed 8 waves.
inline int choose_wk(int K) {
    const int kq = step128(K);
    if (kq >= 8) return 8;
    if (kq >= 4) return 4;
    if (kq >= 2) return 2;
    return 1;
}

template <bool ZS>
int dispatch_single(const qgtc_problem &pr, int K, int a, int w, int ob, int mode, hipStream_t st) {
    Plan pl;
    switch (choose_wk(K)) {
        case 8:
            if (!plan_for<4, 4, 8>(K, a, w, ob, mode, &pl)) return QGTC_EINVAL;
            return launch_single<4, 4, 8, ZS>(pr, pl, st);
        case 4:
            if (!plan_for<4, 4, 4>(K, a, w, ob, mode, &pl)) return QGTC_EINVAL;
            return launch_single<4, 4, 4, ZS>(pr, pl, st);
        case 2:
            if (!plan_for<4, 4, 2>(K, a, w, ob, mode, &pl)) return QGTC_EINVAL;
            return launch_single<4, 4, 2, ZS>(pr, pl, st);
        default:
            if (!plan_for<4, 4, 1>(K, a, w, ob, mode, &pl)) return QGTC_EINVAL;
            return launch_single<4, 4, 1, ZS>(pr, pl, st);
    }
}

template <bool ZS>
int dispatch_batched(const qgtc_problem *prs, int count, int max_M, int max_N, int K_hint, int a,
                     int w, int ob, int mode, hipStream_t st) {
    Plan pl;
    switch (choose_wk(K_hint)) {
        case 8:
            if (!plan_for<4, 4, 8>(K_hint, a, w, ob, mode, &pl)) return QGTC_EINVAL;
            return launch_batched<4, 4, 8, ZS>(prs, count, max_M, max_N, pl, st);
        case 4:
            if (!plan_for<4, 4, 4>(K_hint, a, w, ob, mode, &pl)) return QGTC_EINVAL;
            return launch_batched<4, 4, 4, ZS>(prs, count, max_M, max_N, pl, st);
        case 2:
            if (!plan_for<4, 4, 2>(K_hint, a, w, ob, mode, &pl)) return QGTC_EINVAL;
            return launch_batched<4, 4, 2, ZS>(prs, count, max_M, max_N, pl, st);
        default:
            if (!plan_for<4, 4, 1>(K_hint, a, w, ob, mode, &pl)) return QGTC_EINVAL;
            return launch_batched<4, 4, 1, ZS>(prs, count, max_M, max_N, pl, st);
    }
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
        hipLaunchKernelGGL(k_val2bit_rows, dim3(grid_for(units, 4)), dim3(256), 0, st, x, H, W, nbits,
                           ub, ubm1, out, rows_pad, row_words);
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
    qgtc_problem pr{X, W, out, x_words, w_words, M, K, N, pad128(N)};
    hipStream_t st = static_cast<hipStream_t>(stream);
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
    qgtc_problem pr{X, W, out, x_words, w_words, M, K, N, pad_128 ? pad128(N) : pad8(N)};
    hipStream_t st = static_cast<hipStream_t>(stream);
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
    if (flags & QGTC_NO_ZERO_SKIP)
        return dispatch_batched<false>(problems, count, max_M, max_N, k_hint, bit1, bit2,
                                       mode == 2 ? 1 : output_bit, mode, st);
    return dispatch_batched<true>(problems, count, max_M, max_N, k_hint, bit1, bit2,
                                  mode == 2 ? 1 : output_bit, mode, st);
}

}  // extern "C"
