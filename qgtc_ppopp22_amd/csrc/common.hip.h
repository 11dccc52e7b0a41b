// common.hip.h — part of libqgtc_hip.so (included by qgtc_hip.hip, one translation unit).
// Shared device helpers of the QGTC kernels: vector types, the reference's shape algebra, the
// quantiser and the 8-lane DPP OR used by every packer.
#pragma once

extern thread_local char qgtc_g_hip_err[256];   // text of the last HIP error of this thread (qgtc_last_hip_error)

namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
// column blocks (32 columns) a pre-expanded weight table of the chain entries has: three are kept as four - the kernels of every width
// but the 2-bit family run three blocks as four (launch_fp4.hip.h) and read the fourth block's (all-zero) codes -, five to seven as
// eight (bitmm_fp4_rbx.hip.h: up to 256 columns)
__host__ __device__ constexpr int weight_table_blocks(int N) { return (N + 31) / 32 == 3 ? 4 : ((N + 31) / 32 > 4 ? 8 : (N + 31) / 32); }
// 64-column slices of K per column block in a table: the two halves of a k-quad (order 0) / the MFMAs of the second product (order 1:
// two up to 128 columns of K, four up to 256)
__host__ __device__ constexpr int weight_table_slices(int K, int order) { return (order == 1 && K > 128) ? 4 : 2; }
// base-4 digits of a value in the chain entries' formats: 1 (1 / 2 bits), 2 (3 / 4 bits), 4 (5 .. 8 bits - the kernels of that class run
// four digits whatever the width is; the top digits of 5- and 6-bit values are zero)
__host__ __device__ constexpr int chain_digits(int bits) { return bits <= 2 ? 1 : (bits <= 4 ? 2 : 4); }

constexpr int TM = 32, TN = 32;  // workgroup tile of the bit-GEMM

// ------------------------------------------------------------------------------------------
// shape algebra (reference utility.h:33-45)
// ------------------------------------------------------------------------------------------
__host__ __device__ constexpr int step8(int x) { return (x + 7) >> 3; }
__host__ __device__ constexpr int step128(int x) { return (x + 127) >> 7; }
__host__ __device__ constexpr int pad8(int x) { return step8(x) << 3; }
__host__ __device__ constexpr int pad128(int x) { return step128(x) << 7; }

// (one buffer for the whole library: defined in qgtc_hip.hip, declared before this namespace opens)

int hip_fail(hipError_t e, const char *where) {
    snprintf(qgtc_g_hip_err, sizeof(qgtc_g_hip_err), "%s: %s", where, hipGetErrorString(e));
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
// Selects only: written with `if`s and an early return hipcc compiled every element's quantisation into exec-masked branches (37
// s_and_saveexec in k_val2bit_rows_v4), and rocprofv3 counted 111 VALU instructions per float4 - the pack kernels were bound by instruction
// issue, not by HBM (7.3 M wave instructions = 11.8 of the 12.9 us of a 4096 x 4096 pack).
__device__ __forceinline__ uint32_t quant1(float x, float ub, float ubm1) {
    float y = x > ub ? ubm1 : x;      // above 2^b -> 2^b - 1 (float arithmetic); a NaN compares false and stays
    y = x < 0.0f ? 1.0f : y;          // negative -> lb + 1   (-0.0 is not negative: it rounds to -0.0 and converts to 0)
    const float r = rintf(y);         // v_rndne_f32: round-half-to-even
    // NaN converts to 0, and so does 2^32 (reached at nbits = 32 only; the low 32 bits of the reference's conversion): both fail `r < 2^32`,
    // and the conversion's own result is not looked at then
    return r < 4294967296.0f ? static_cast<uint32_t>(r) : 0u;
}

// Workgroups whose ids are congruent mod 8 run on the same XCD (round-robin placement): give those CONSECUTIVE virtual
// ids, so that neighbours in the work list share one L2. Bijective for every n (MI355X guide: the simple remap is not).
__device__ __forceinline__ int xcd_consecutive(int id, int n) {
    const int q = n >> 3, r = n & 7, x = id & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (id >> 3);
}

constexpr int AUX_SC1 = 16;   // cache-policy operand of the raw buffer builtins on gfx940+: bit 4 = sc1 (agent scope)

// A packed output word. PUB = the fused layer's first stage: the word is read by OTHER workgroups (possibly on another
// XCD, whose L2 is not coherent with ours) later in the same launch, so it is stored with agent scope (write-through,
// `sc1`) instead of being left dirty in this XCD's L2 (MI355X_MICROARCH.md: publish with write-through stores, then a
// drained flag).
template <bool PUB>
__device__ __forceinline__ void st_word(uint32_t *p, uint32_t v) {
    if (PUB) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}

// OR over aligned groups of 8 lanes (every lane of the wave must be active)
__device__ __forceinline__ uint32_t or_reduce8(uint32_t x) {
    x |= static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), 0xB1, 0xf, 0xf, false));   // lane ^ 1
    x |= static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), 0x4E, 0xf, 0xf, false));   // lane ^ 2
    x |= static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), 0x141, 0xf, 0xf, false));  // 7 - lane
    return x;
}

}  // namespace
