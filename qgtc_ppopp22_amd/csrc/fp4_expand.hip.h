// fp4_expand.hip.h — part of libqgtc_hip.so (qgtc_wide.hip).
// Packed words -> E2M1 operand registers of the block-scaled FP4 MFMAs, IN PLACE: a lane holds four packed words of a
// line (128 elements of K) and MFMA s = 0..3 takes the bits s, s + 4, .. of all four. Which elements of K an instruction
// covers is free as long as both operands agree.
//   * one plane: the nibble code 1 << s stands for 0.5, 1 or 2 and the E8M0 scale 2^(1 - s) turns it back into 1
//     (s = 3 needs one shift: code 8 is -0): 5 VALU operations per packed word for its four MFMAs;
//   * two planes: the 2-bit code of (plane 0, plane 1) is built once per word pair for the even and the odd bit
//     positions (6 operations), then one AND (or shift + AND) per MFMA: 12 per word pair;
//   * more planes: base-4 digits (planes 2d, 2d + 1), the E8M0 scale carries 4^d.
// (Tried in the row-block kernel of the epochs too, bitmm_fp4_rows.hip.h: 16 VALU operations fewer per pair of k-quads,
// but all four words' codes stay live across the four MFMAs - 75 instead of 54 VGPRs, six waves per SIMD instead of
// eight - and the stage was no faster: 10.7 against 10.3 us. Not kept there.)
#pragma once

namespace {

// Operands of more than two planes are split into base-4 DIGITS (planes 2d, 2d + 1): one MFMA per pair of digits, the
// E8M0 scales carry 4^(dl + dr).
constexpr int wd_digits(int np) { return np == 1 ? 1 : np / 2; }

// what a lane keeps of one fragment chunk (four packed words per plane) and digit between the four MFMAs that use it
template <int NP>
struct WdPrep {
    uint32_t a[4], b[4];
};
template <int NP>
__device__ __forceinline__ void wd_prep(const u32x4 (&pl)[NP], int digit, WdPrep<NP> &pp) {
    if constexpr (NP == 1) {
#pragma unroll
        for (int t = 0; t < 4; t++) pp.a[t] = pl[0][t];
    } else {   // the 2-bit code v of (plane 2d, plane 2d + 1) at the even (a) and the odd (b) bit positions
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const uint32_t w0 = pl[2 * digit][t], w1 = pl[2 * digit + 1][t];
            pp.a[t] = (w0 & 0x55555555u) | ((w1 & 0x55555555u) << 1);
            pp.b[t] = ((w0 >> 1) & 0x55555555u) | (w1 & 0xaaaaaaaau);
        }
    }
}
// the four operand registers of MFMA s (bits s, s + 4, .. of the four words) and the E8M0 scale that makes the code count
// as the integer it stands for
template <int NP>
__device__ __forceinline__ i32x8 wd_operand(const WdPrep<NP> &pp, int s) {
    uint32_t o[4];
#pragma unroll
    for (int t = 0; t < 4; t++) {
        if constexpr (NP == 1) o[t] = s < 3 ? (pp.a[t] & (0x11111111u << s)) : ((pp.a[t] >> 3) & 0x11111111u);
        else o[t] = s == 0 ? (pp.a[t] & 0x33333333u) : s == 1 ? (pp.b[t] & 0x33333333u) : s == 2 ? ((pp.a[t] >> 2) & 0x33333333u) : ((pp.b[t] >> 2) & 0x33333333u);
    }
    return i32x8{static_cast<int>(o[0]), static_cast<int>(o[1]), static_cast<int>(o[2]), static_cast<int>(o[3]), 0, 0, 0, 0};
}
template <int NP>
__device__ __forceinline__ constexpr int wd_scale(int s, int digit) { return NP == 1 ? (s < 3 ? 128 - s : 128) : 128 + 2 * digit; }

}  // namespace
