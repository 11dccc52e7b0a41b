// pack_kernels.hip.h — part of libqgtc_hip.so (included by qgtc_hip.hip, one translation unit).
// Fused quantise + bit-plane pack (val2bit, rows and cols layouts), unpack (bit2val) and adjacency
// planes straight from an edge list (pack_edges). HBM-streaming kernels.
#pragma once

namespace {

// ------------------------------------------------------------------------------------------
// val2bit, rows layout, fast path (W % 4 == 0, 16-byte aligned input): HBM-streaming.
// A wave packs 256 columns of one row per unit: every lane loads one float4 (16 B/lane, 1 KiB per
// wave-instruction, fully coalesced), builds the nibble of its four columns per plane, and eight
// adjacent lanes OR their nibbles into one output word with DPP (lane 8k stores word k of the
// unit: 32 contiguous bytes per plane). UNROLL units are loaded before any is packed, so a wave
// keeps UNROLL KiB in flight. Every word of the padded output is written.
// ------------------------------------------------------------------------------------------
template <int UNROLL>
__device__ __forceinline__ void val2bit_rows_v4_body(const float *__restrict__ x, int H, int W, int nbits, float ub, float ubm1,
                                                     uint32_t *__restrict__ out, int rows_pad, int row_words, uint32_t wave,
                                                     uint32_t nwaves) {
    const int lane = threadIdx.x & 63;
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

template <int UNROLL>
__global__ __launch_bounds__(256) void k_val2bit_rows_v4(const float *__restrict__ x, int H, int W,
                                                         int nbits, float ub, float ubm1,
                                                         uint32_t *__restrict__ out, int rows_pad,
                                                         int row_words, int nwaves) {
    // (nwaves = 4 x the grid, handed over: gridDim and blockDim are HIDDEN kernel arguments - a kernel that reads them has a 312-byte
    // argument segment for hipLaunchKernel to write instead of 60; bitmm_fp4_one.hip.h, tools/kernarg_probe.hip)
    val2bit_rows_v4_body<UNROLL>(x, H, W, nbits, ub, ubm1, out, rows_pad, row_words, (blockIdx.x * 256u + threadIdx.x) >> 6, nwaves);
}

// ------------------------------------------------------------------------------------------
// val2bit, rows layout: out[p][r][c>>5] bit(31-(c&31)) = bit p of quant(x[r][c])
// One wave per (row, 256-column chunk): 4 coalesced loads per lane, one 64-bit ballot per
// (plane, load), two bit-reversed words per ballot; lanes 0..7 store the chunk's 8 words.
// Every word of the padded output is written.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void val2bit_rows_body(const float *__restrict__ x, int H, int W, int nbits, float ub, float ubm1,
                                                  uint32_t *__restrict__ out, int rows_pad, int row_words, long wave, long nwaves) {
    const int lane = threadIdx.x & 63;
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

__global__ __launch_bounds__(256) void k_val2bit_rows(const float *__restrict__ x, int H, int W,
                                                      int nbits, float ub, float ubm1,
                                                      uint32_t *__restrict__ out, int rows_pad,
                                                      int row_words, int nwaves) {
    val2bit_rows_body(x, H, W, nbits, ub, ubm1, out, rows_pad, row_words, (static_cast<long>(blockIdx.x) * 256 + threadIdx.x) >> 6, static_cast<long>(nwaves));
}

// ------------------------------------------------------------------------------------------
// val2bit, cols layout: out[p][c][r>>5] bit(31-(r&31)) = bit p of quant(x[r][c])
// One wave per (64-column chunk, 32-row group): lane = column, 32 coalesced row reads in flight, each
// lane assembles its column's word per plane in registers. NB = compile-time bound on nbits.
// ------------------------------------------------------------------------------------------
template <int NB>
__device__ __forceinline__ void val2bit_cols_body(const float *__restrict__ x, int H, int W, int nbits, float ub, float ubm1,
                                                  uint32_t *__restrict__ out, int lines, int line_words, long wave, long nwaves) {
    const int lane = threadIdx.x & 63;
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
        // all 32 row reads are issued before any is packed (one memory latency per unit, not four)
        float v[32];
#pragma unroll
        for (int rr = 0; rr < 32; rr++) {
            const int r = rw * 32 + rr;
            v[rr] = (r < H && c < W) ? x[static_cast<size_t>(r) * W + c] : 0.0f;   // quantises to 0
        }
#pragma unroll
        for (int rr = 0; rr < 32; rr++) {
            const uint32_t q = quant1(v[rr], ub, ubm1);
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

// The same unit with the ROWS layout written beside the cols layout (the data loader wants both of X: sampler.py:99's bit_X and the
// left operand of the layout-correct chain's first X . W): the 32 x 64 quantised values are in registers once - a ballot over the lanes
// (= 64 columns) of row rr's plane p IS the two rows-layout words of that row, kept by lane rr. One read of X
// instead of two (round 4: 14.6 + 19.6 us for the ogbn-arxiv-sized iterator's 46.5 MB of features).
template <int NB>
__device__ __forceinline__ void val2bit_cols_rows_body(const float *__restrict__ x, int H, int W, int nbits, float ub, float ubm1,
                                                       uint32_t *__restrict__ out, int lines, int line_words, uint32_t *__restrict__ rows_out,
                                                       int rows_pad, int row_words, long wave, long nwaves) {
    const int lane = threadIdx.x & 63;
    const int cchunks = (lines + 63) >> 6;
    const long units = static_cast<long>(cchunks) * line_words;
    const size_t plane = static_cast<size_t>(lines) * line_words, rplane = static_cast<size_t>(rows_pad) * row_words;
    for (long u = wave; u < units; u += nwaves) {
        const int cg = static_cast<int>(u % cchunks);
        const int rw = static_cast<int>(u / cchunks);
        const int c = cg * 64 + lane;
        float v[32];
#pragma unroll
        for (int rr = 0; rr < 32; rr++) {
            const int r = rw * 32 + rr;
            v[rr] = (r < H && c < W) ? x[static_cast<size_t>(r) * W + c] : 0.0f;   // quantises to 0
        }
        uint32_t q[32];
#pragma unroll
        for (int rr = 0; rr < 32; rr++) q[rr] = quant1(v[rr], ub, ubm1);
        const int r_mine = rw * 32 + (lane & 31);
#pragma unroll
        for (int p = 0; p < NB; p++) {
            if (p >= nbits) break;   // (launch-uniform)
            uint32_t wd = 0u, lo = 0u, hi = 0u;
#pragma unroll
            for (int rr = 0; rr < 32; rr++) {
                const uint32_t bit = (q[rr] >> p) & 1u;
                wd |= bit << (31 - rr);
                const unsigned long long m = __ballot(bit != 0u);
                // lane rr keeps the ballot (a select on a constant lane mask; an inline v_writelane_b32 here returned a wrong word in one
                // lane of 64 - inline asm is opaque to hipcc's hazard recogniser)
                lo = lane == rr ? static_cast<uint32_t>(m) : lo;
                hi = lane == rr ? static_cast<uint32_t>(m >> 32) : hi;
            }
            if (c < lines) out[p * plane + static_cast<size_t>(c) * line_words + rw] = wd;
            // lane rr < 32: columns 64 cg .. 64 cg + 63 of row 32 rw + rr = words 2 cg, 2 cg + 1 (element i of a word at bit 31 - i)
            if (lane < 32 && r_mine < rows_pad && 2 * cg < row_words) {
                uint32_t *dst = rows_out + p * rplane + static_cast<size_t>(r_mine) * row_words + 2 * cg;
                if (2 * cg + 1 < row_words) *reinterpret_cast<u32x2 *>(dst) = u32x2{__brev(lo), __brev(hi)};
                else *dst = __brev(lo);
            }
        }
    }
}

template <int NB>
__global__ __launch_bounds__(256) void k_val2bit_cols(const float *__restrict__ x, int H, int W,
                                                      int nbits, float ub, float ubm1,
                                                      uint32_t *__restrict__ out, int lines,
                                                      int line_words, int nwaves) {
    val2bit_cols_body<NB>(x, H, W, nbits, ub, ubm1, out, lines, line_words, (static_cast<long>(blockIdx.x) * 256 + threadIdx.x) >> 6, static_cast<long>(nwaves));
}

// ------------------------------------------------------------------------------------------
// bit2val (reference kernel.h:109-139, :173-201): one thread per output element
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_bit2val(const uint32_t *__restrict__ bits,
                                                 unsigned long long words, int nbits, int H, int W,
                                                 int col_major, size_t plane, int line_words,
                                                 int32_t *__restrict__ out, unsigned nthreads) {
    const size_t total = static_cast<size_t>(H) * W;
    for (size_t idx = static_cast<size_t>(blockIdx.x) * 256u + threadIdx.x; idx < total; idx += nthreads) {
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
// 1-bit adjacency straight from the RAW edge list (duplicates allowed, no sort): three bitmaps count
// every cell's multiplicity in unary - t1: seen once, t2: twice, t3: three times or more. atomicOr
// returns the word as it was, so of the threads that hit the same cell exactly one finds its bit clear
// in t1, exactly one of the others finds it clear in t2, the rest reach t3. The 1-bit quantiser maps the
// multiplicities 1, 2, >= 3 to the bits 1, 0, 1 (kernel.h:39-44: 2.0 rounds to 2 = 2^1, whose plane 0 is 0;
// anything above 2 clamps to 1), so plane 0 = t1 & (~t2 | t3). t1 is the output buffer itself.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_edge_list_count(const int64_t *__restrict__ src, const int64_t *__restrict__ dst,
                                                         size_t n_edges, int H, int W, uint32_t *__restrict__ t1,
                                                         uint32_t *__restrict__ t2, uint32_t *__restrict__ t3,
                                                         int row_words, int *__restrict__ bad) {
    for (size_t e = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; e < n_edges;
         e += static_cast<size_t>(gridDim.x) * blockDim.x) {
        const int64_t r = src[e], c = dst[e];
        if (r < 0 || r >= H || c < 0 || c >= W) {
            if (bad) *bad = 1;
            continue;
        }
        const uint32_t bit = 1u << (31 - (c & 31));
        const size_t wi = static_cast<size_t>(r) * row_words + (c >> 5);
        if (atomicOr(t1 + wi, bit) & bit)
            if (atomicOr(t2 + wi, bit) & bit) atomicOr(t3 + wi, bit);
    }
}

__global__ __launch_bounds__(256) void k_edge_list_finish(uint32_t *__restrict__ t1, const uint32_t *__restrict__ t2,
                                                          const uint32_t *__restrict__ t3, size_t words) {
    for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < words;
         i += static_cast<size_t>(gridDim.x) * blockDim.x)
        t1[i] &= ~t2[i] | t3[i];
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

}  // namespace
