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
        int rk[UNROLL], chk[UNROLL];   // (row, 256-column chunk) of unit u0 + k: ONE division per pass, the rest by carry (a division by a
        rk[0] = static_cast<int>(u0 / chunks);   // run-time value is a dozen quarter-rate multiplies; it was done twice per unit)
        chk[0] = static_cast<int>(u0 - static_cast<uint32_t>(rk[0]) * chunks);
#pragma unroll
        for (int k = 1; k < UNROLL; k++) {
            const bool wrap = chk[k - 1] + 1 == chunks;
            chk[k] = wrap ? 0 : chk[k - 1] + 1;
            rk[k] = wrap ? rk[k - 1] + 1 : rk[k - 1];
        }
#pragma unroll
        for (int k = 0; k < UNROLL; k++) {
            const uint32_t u = u0 + k;
            const int r = rk[k], c = chk[k] * 256 + lane * 4;
            v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (u < units && r < H && c < W)  // W % 4 == 0: the quad is entirely inside or outside
                v[k] = *reinterpret_cast<const float4 *>(x + static_cast<size_t>(r) * W + c);
        }
#pragma unroll
        for (int k = 0; k < UNROLL; k++) {
            const uint32_t u = u0 + k;
            if (u >= units) break;  // wave-uniform
            const int r = rk[k], ch = chk[k];
            // (outside the matrix the quad was left at 0.0f, which quantises to 0: no select, and no branch around the quantisation)
            const uint32_t q0 = quant1(v[k].x, ub, ubm1), q1 = quant1(v[k].y, ub, ubm1);
            const uint32_t q2 = quant1(v[k].z, ub, ubm1), q3 = quant1(v[k].w, ub, ubm1);
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

// unit u -> (u % n, u / n): units are counted in 64 bits (a 2^31-row matrix is legal), but a 64-bit division by a run-time value is ~100
// instructions per unit - taken only when the count does not fit 32 bits (launch-uniform)
__device__ __forceinline__ void split_unit(long u, int n, bool small, int &rem, int &quo) {
    if (small) {
        const uint32_t u32 = static_cast<uint32_t>(u), q = u32 / static_cast<uint32_t>(n);
        quo = static_cast<int>(q);
        rem = static_cast<int>(u32 - q * static_cast<uint32_t>(n));
    } else {
        quo = static_cast<int>(u / n);
        rem = static_cast<int>(u % n);
    }
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
    const bool small = units < (1l << 31);
    for (long u = wave; u < units; u += nwaves) {
        int cg, rw;
        split_unit(u, cchunks, small, cg, rw);
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

// 32 x 32 bit-matrix transpose over the 32 lanes of a half-wave (both halves at once): afterwards bit b of lane i's word is what bit i of
// lane b's word was. Five butterfly stages (Hacker's Delight 7-3 across lanes): exchange with lane i ^ j (ds_swizzle: the LDS crossbar, no
// memory), keep the diagonal blocks, take the partner's off-diagonal block moved by j bits - a ROTATE does the move for both kinds of lane,
// as the bits that would wrap are masked off. `li` = lane & 31; the masks and rotate counts depend on the lane only (hoisted out of loops).
template <int J>
__device__ __forceinline__ uint32_t transpose32_stage(uint32_t x, int li) {
    constexpr uint32_t M = J == 16 ? 0x0000ffffu : J == 8 ? 0x00ff00ffu : J == 4 ? 0x0f0f0f0fu : J == 2 ? 0x33333333u : 0x55555555u;   // bits b with (b & J) == 0
    const uint32_t p = static_cast<uint32_t>(__builtin_amdgcn_ds_swizzle(static_cast<int>(x), (J << 10) | 0x1f));   // lane ^ J within its 32
    const bool hi = (li & J) != 0;
    const uint32_t keep = hi ? ~M : M;
    const uint32_t moved = p & keep;
    return (x & keep) | __builtin_amdgcn_alignbit(moved, moved, hi ? J : 32 - J);   // hi lanes: >> J, the others: << J
}
__device__ __forceinline__ uint32_t transpose32_lanes(uint32_t x, int li) {
    x = transpose32_stage<16>(x, li);
    x = transpose32_stage<8>(x, li);
    x = transpose32_stage<4>(x, li);
    x = transpose32_stage<2>(x, li);
    return transpose32_stage<1>(x, li);
}

// The same unit with the ROWS layout written beside the cols layout (the data loader wants both of X: sampler.py:99's bit_X and the
// left operand of the layout-correct chain's first X . W): the 32 x 64 quantised values are in registers once, and the cols-layout word a
// lane has assembled for its column (row rr at bit 31 - rr) is one row of a 32 x 32 bit matrix whose TRANSPOSE is the rows layout: after
// transpose32_lanes lane i of a half holds, bit b = column b of the half's 32, the word of row 31 - i - bit-reversed, a rows-layout word.
// 22 operations a plane. (Round 5's first form took a ballot per row and plane and a select on a lane mask into the lane that keeps it: 160
// operations a plane and 128 scalar registers of ballots - rocprofv3: 1371 VALU instructions a unit, v_readlane / v_writelane spills of
// scalar registers, 8.2 M wave instructions = 13 of the launch's 21 us.) One read of X instead of two (round 4: 14.6 + 19.6 us for the
// ogbn-arxiv-sized iterator's 46.5 MB of features). The loads are buffer loads: rows past H and columns past W come back as 0.0f - which
// quantises to 0 - from the range check instead of from a branch around every load. x_bytes = H * W * 4 < 2^31 (the caller's check).
template <int NB>
__device__ __forceinline__ void val2bit_cols_rows_body(const float *__restrict__ x, int H, int W, int nbits, float ub, float ubm1,
                                                       uint32_t *__restrict__ out, int lines, int line_words, uint32_t *__restrict__ rows_out,
                                                       int rows_pad, int row_words, long wave, long nwaves) {
    const int lane = threadIdx.x & 63, li = lane & 31, half = lane >> 5;
    const int cchunks = (lines + 63) >> 6;
    const long units = static_cast<long>(cchunks) * line_words;
    const size_t plane = static_cast<size_t>(lines) * line_words, rplane = static_cast<size_t>(rows_pad) * row_words;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(x), 0, static_cast<int>(static_cast<uint32_t>(H) * static_cast<uint32_t>(W) * 4u), 0x00020000);
    const uint32_t row_bytes = static_cast<uint32_t>(W) * 4u;
    const bool small = units < (1l << 31);
    for (long u = wave; u < units; u += nwaves) {
        int cg, rw;
        split_unit(u, cchunks, small, cg, rw);
        const int c = cg * 64 + lane;
        // (a column past W: an offset no row's bytes bring back into range)
        const uint32_t lane_off = c < W ? static_cast<uint32_t>(c) * 4u : 0x80000000u;
        float v[32];
#pragma unroll
        for (int rr = 0; rr < 32; rr++)
            v[rr] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, lane_off + static_cast<uint32_t>(rw * 32 + rr) * row_bytes, 0, 0));
        uint32_t wd[NB];
#pragma unroll
        for (int p = 0; p < NB; p++) wd[p] = 0u;
#pragma unroll
        for (int rr = 0; rr < 32; rr++) {
            const uint32_t q = quant1(v[rr], ub, ubm1);
#pragma unroll
            for (int p = 0; p < NB; p++) wd[p] |= ((q >> p) & 1u) << (31 - rr);
        }
        const int r_mine = rw * 32 + 31 - li, w_mine = 2 * cg + half;
#pragma unroll
        for (int p = 0; p < NB; p++) {
            if (p >= nbits) break;   // (launch-uniform)
            if (c < lines) out[p * plane + static_cast<size_t>(c) * line_words + rw] = wd[p];
            const uint32_t t = __brev(transpose32_lanes(wd[p], li));
            if (r_mine < rows_pad && w_mine < row_words) rows_out[p * rplane + static_cast<size_t>(r_mine) * row_words + w_mine] = t;
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
