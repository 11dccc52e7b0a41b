// fp4_rowblock.hip.h — part of libqgtc_hip.so.
// What the row-block kernels (bitmm_fp4_rows / _chain / _rbw / _wide .hip.h) share: a lane's packed words -> the E2M1 operand of one
// base-4 digit, and the re-quantise + pack epilogue of a 32 x 32 accumulator tile.
#pragma once

namespace {

// the lane's packed word (bits 32 g .. 32 g + 31 of its line) of up to NP planes -> E2M1 operand registers of digit d
template <int NP>
__device__ __forceinline__ i32x8 strip_operand(const uint32_t (&pl)[NP], int digit) {
    uint32_t wd[2], e[4];
    wd[0] = pl[2 * digit];
    wd[1] = 2 * digit + 1 < NP ? pl[(2 * digit + 1) % NP] : 0u;
    expand_word_fp4<2>(wd, 2, e);
    return i32x8{static_cast<int>(e[0]), static_cast<int>(e[1]), static_cast<int>(e[2]), static_cast<int>(e[3]), 0, 0, 0, 0};
}

// OB: output planes at compile time (1, 2, 4, 8: the widths the reference publishes; 0 = any, runtime loop)
// Re-quantise the 16 sums a lane holds of a 32 x 32 tile (kernel.h:31-37,350: c > 2^ob ? 2^ob - 1 : c) and pack them a
// byte each: P[t] byte 3 - gq = value of register 4 gq + t (OB = 1, 2: its low OB bits are, the rest of the byte is not
// meaningful). For OB = 1 .. 7 the clamp happens on the float (an exact integer) and v_cvt_pk_u8_f32 converts AND inserts
// the byte: 2 - 3 operations per value where convert + compare + select + shift/mask/or took 4.5 (these kernels are bound
// by exactly this VALU work). OB = 0 (any width at run time) and 8 keep the
// integer route (OB = 8 must map the sum 256 to the byte 0, which a saturating conversion cannot).
template <int OB>
__device__ __forceinline__ void requant_pack16(const f32x16 &acc, int ob, uint32_t (&P)[4], uint32_t (&qv)[16]) {
    const int maxi = 1 << ob;   // (host: ob <= 23, so the reference's float compare c > 2^ob is this integer compare)
    const uint32_t ones = static_cast<uint32_t>(maxi - 1);
    if constexpr (OB >= 1 && OB <= 7) {
        const float lim = static_cast<float>(maxi), onesf = static_cast<float>(ones);
#pragma unroll
        for (int t = 0; t < 4; t++) {
            uint32_t pk = 0u;
#pragma unroll
            for (int gq = 0; gq < 4; gq++) {
                float f = acc[4 * gq + t];
                // Only the low OB bits of a byte are read below. OB = 1 / 2: ONE multiplication instead of compare + select -
                // by m = 85 / 53: m = 1 (mod 2^OB), so m c = c (mod 2^OB) for c <= 2^OB (c = 2^OB itself packs as 0,
                // kernel.h:350), and m (2^OB + 1) >= 255, so every larger sum saturates to 255 = 2^OB - 1 (mod 2^OB).
                // (OB = 3 .. 7 have no such m - e.g. OB = 4: 17 m >= 255 and 16 m <= 255 leave m = 15 only, which is not 1 mod 16.)
                if constexpr (OB == 1) f *= 85.0f;
                else if constexpr (OB == 2) f *= 53.0f;
                else f = f > lim ? onesf : f;
                pk = __builtin_amdgcn_cvt_pk_u8_f32(f, 3 - gq, pk);
            }
            P[t] = pk;
        }
    } else {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int c = static_cast<int>(acc[r]);   // exact: the sums are integers below 2^24
            qv[r] = c > maxi ? ones : static_cast<uint32_t>(c);
        }
#pragma unroll
        for (int t = 0; t < 4; t++) P[t] = ((qv[t] & 255u) << 24) | ((qv[4 + t] & 255u) << 16) | ((qv[8 + t] & 255u) << 8) | (qv[12 + t] & 255u);
    }
}

}  // namespace
