// fp4_rbw_common.hip.h — part of libqgtc_hip.so (qgtc_fp4.hip, qgtc_chainx.hip).
// What the chain entries' kernel families share (bitmm_fp4_rbw.hip.h: one width of 1 .. 4 bits, N <= 128; bitmm_fp4_rbx.hip.h: 5 .. 8
// bits and up to 256 columns): the register order of the second product's K index, FP4 operand tuples, the workgroup-id remap, and the
// float32 row stores.
#pragma once

namespace {

// lane (fl, fh) of the first product holds, per column block j, the values of columns 32 j + t + 8 gq + 4 fh in register
// 4 gq + t. requant_pack16 puts them a byte each: P[t] byte 3 - gq. Two dwords per block: nibble 2 b + u of dword A (t = u)
// and of dword B (t = 2 + u) is the value of byte b = 3 - gq. K index of the second product (MFMA m, half fh, dword d,
// nibble i): column 32 (2 m + (d >> 1)) + (2 (d & 1) + (i & 1)) + 8 (3 - (i >> 1)) + 4 fh.
__host__ __device__ constexpr int rbw_column(int m, int fh, int d, int i) {
    return 32 * (2 * m + (d >> 1)) + (2 * (d & 1) + (i & 1)) + 8 * (3 - (i >> 1)) + 4 * fh;
}

// An FP4 MFMA operand from its four dwords: the instruction takes a 256-bit register tuple but reads only the first 128
// bits of an FP4 operand - the upper half is left UNDEFINED (zeros there cost four v_mov per operand, a third of this
// kernel's VALU instructions in its first build)
__device__ __forceinline__ i32x8 fp4_op(const u32x4 &v) {
    const i32x4 t = __builtin_bit_cast(i32x4, v);
    return __builtin_shufflevector(t, t, 0, 1, 2, 3, -1, -1, -1, -1);
}
__device__ __forceinline__ i32x8 fp4_op(uint32_t a, uint32_t b, uint32_t c, uint32_t d) { return fp4_op(u32x4{a, b, c, d}); }
__device__ __forceinline__ i32x8 fp4_op(const i32x8 &v) { return __builtin_shufflevector(v, v, 0, 1, 2, 3, -1, -1, -1, -1); }
__device__ __forceinline__ f32x16 f32x16_zero() {
    f32x16 z;
#pragma unroll
    for (int r = 0; r < 16; r++) z[r] = 0.0f;
    return z;
}


struct RbwShape {
    int per;      // != 0: all workgroups of a batch on one XCD
    int a;        // planes of the packed left operand (k_rbw_xw)
    int tiles;    // k_rbw_chain: the adjacency is in the tile format of k_rows_to_tiles (else the rows layout)
};

// (gx, gy = the grid, handed over as kernel arguments: gridDim lives in the HIDDEN kernel arguments, which are not preloaded into
// scalar registers with the wave - reading it was one more dependent scalar-load round trip ahead of the descriptor's)
__device__ __forceinline__ void rbw_ids(int per, int gx, int gy, int &grp, int &batch) {
    grp = static_cast<int>(blockIdx.x);
    batch = static_cast<int>(blockIdx.y);
    if (per) {
        const int v = xcd_consecutive(batch * gx + grp, gx * gy);
        batch = v / gx;
        grp = v - batch * gx;
    }
}

// every field of a descriptor in scalar registers NOW: left alone, the compiler loads M, tests the early exit and fetches the rest
// behind the branch - two dependent round trips where one does
__device__ __forceinline__ void rbw_pin(const qgtc_problem &pr) {
    asm volatile("" ::"s"(pr.X), "s"(pr.W), "s"(pr.out), "s"(pr.x_words), "s"(pr.w_words), "s"(pr.K), "s"(pr.N), "s"(pr.occ), "s"(pr.occ_words));
}

// 16 re-quantised values (low OB bits of each byte of P) -> the two code dwords of a column block (see rbw_column)
template <int OB>
__device__ __forceinline__ void rbw_nibbles(const uint32_t (&P)[4], uint32_t &A, uint32_t &B) {
    constexpr uint32_t mask = ((1u << OB) - 1u) * 0x01010101u;   // (the low OB bits of a byte are the value: requant_pack16)
    A = (P[0] & mask) | ((P[1] & mask) << 4);
    B = (P[2] & mask) | ((P[3] & mask) << 4);
}


// float32 rows from a swapped product's accumulators: lane (fl, fh) holds of ITS row the columns col0 + 8 g + 4 fh + t in register
// 4 g + t. Four consecutive columns are ONE 16-byte store where they exist (rows of 4 N bytes are only 4-byte aligned - N = 10 classes;
// buffer stores take any dword alignment), a row's tail is an 8- and / or a 4-byte store. The first form stored every element on its
// own whenever N % 4 != 0: ten 4-byte store instructions a row for the 10-class output layer, 0.9 us of its 4.45 us launch
// (timing-only build, tools/rbw_bench).
// BRANCH-FREE per lane: a lane that has nothing to store (row past M, columns past N) gets the offset 0xffffffff and the range check
// drops it; the only branches are on wave-uniform column counts. With a per-lane `if (m < M)` around the stores hipcc sank the second
// product's LDS reads and MFMAs INTO the branch - and an MFMA takes its operands from ALL lanes: the rows past M then supplied garbage
// as W' lines (wrong columns 32 jn + fl for every fl past the last valid row of the block).
__device__ __forceinline__ void rbw_store_f32_row(__amdgpu_buffer_rsrc_t ro, uint32_t row_off /* bytes, 0xffffffff = no row */, const f32x16 &acc,
                                                  int col0, int fh, int N) {
#pragma unroll
    for (int g = 0; g < 4; g++) {
        const int rem0 = N - (col0 + 8 * g);        // wave-uniform: columns left from the half fh = 0's first one
        if (rem0 <= 0) break;
        const int rem = rem0 - 4 * fh;              // per lane
        const uint32_t off = row_off == 0xffffffffu ? 0xffffffffu : row_off + static_cast<uint32_t>(col0 + 8 * g + 4 * fh) * 4u;
        const u32x4 v = {__float_as_uint(acc[4 * g]), __float_as_uint(acc[4 * g + 1]), __float_as_uint(acc[4 * g + 2]), __float_as_uint(acc[4 * g + 3])};
        if (rem0 >= 8) {
            __builtin_amdgcn_raw_buffer_store_b128(v, ro, off, 0, 0);
        } else {
            const bool row = off != 0xffffffffu;   // (off + 8 below must not wrap a missing row's offset back into the buffer)
            __builtin_amdgcn_raw_buffer_store_b128(v, ro, rem >= 4 ? off : 0xffffffffu, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b64(u32x2{v.x, v.y}, ro, (rem == 2 || rem == 3) ? off : 0xffffffffu, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32(rem == 3 ? v.z : v.x, ro, row && rem == 3 ? off + 8u : (rem == 1 ? off : 0xffffffffu), 0, 0);
        }
    }
}


}  // namespace
