// tools/mfma_overlap.hip — round 6: what a SIMD gets out of v_mfma_scale_f32_32x32x64_f8f6f4 when the wave that issues the MFMAs also expands
// its operands (the long-K kernel, bitmm_fp4_stream.hip.h). Sections, in the order they print:
//   * roles: four multiplying waves (one a SIMD, the pinned step, fragments read from LDS, one barrier a group) + four fetching waves (LDS-DMA);
//   * the pinned step of the kernel alone in a loop, 4 x 2 and 2 x 2 fragments, with variants (flat VALU load, two accumulators, MFMAs only,
//     a second wave that only waits at a barrier);
//   * bare MFMAs on different operand DATA: cycles per MFMA do not move, the clock the chip holds does (2.14 GHz on small integers, 1.68 on
//     random bits);
//   * the all-waves-do-both forms of the step (k_like: as hipcc ordered it, software-pipelined by hand, sched_group_barrier, the pinned snake);
//   * per MFMA, V v_and_b32 behind it (k): accumulators in VGPRs / AGPRs, operands fed by the VALU results or constant, one / two waves a
//     SIMD, s_setprio around the MFMA; the VALU results written OVER the operand of the MFMA in flight (k_inplace).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/mfma_overlap tools/mfma_overlap.hip      (STEP_ONLY=1 / ROLES_ONLY=1: the first sections only)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <utility>
#include <type_traits>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s\n", hipGetErrorString(e)); return 1; } } while (0)

// INPLACE: the V VALU operations OVERWRITE the registers the MFMA just issued reads its A operand from (what a register allocator does
// with an operand that is dead once its MFMA is issued): a write-after-read on an MFMA in flight
template <int V>
__global__ __launch_bounds__(512) void k_inplace(int iters, unsigned long long *cycles, int *sink, unsigned seed) {
    i32x4 x[4], w[2];
    for (int i = 0; i < 4; i++) x[i] = i32x4{(int)(seed * 3 + threadIdx.x + i), (int)(seed + i), 3 + i, 4 + i};
    for (int i = 0; i < 2; i++) w[i] = i32x4{5 + i, 6 + i, 7 + i, (int)threadIdx.x};
    f32x16 d[4] = {};
    i32x4 a4 = x[0] & 0x11111111;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const i32x8 a8 = __builtin_shufflevector(a4, a4, 0, 1, 2, 3, -1, -1, -1, -1);
            const i32x4 b4 = w[u & 1];
            const i32x8 b8 = __builtin_shufflevector(b4, b4, 0, 1, 2, 3, -1, -1, -1, -1);
            d[u & 3] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, d[u & 3], 4, 4, 0, 128, 0, 127);
#pragma unroll
            for (int q = 0; q < V; q++) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a4[q & 3]) : "v"(x[(u + 1) & 3][q & 3]));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
    float s = 0;
    for (int u = 0; u < 4; u++) s += d[u][1];
    if (s == 12345.0f && a4[0] == 77) *sink = 1;
}
template <int V>
int run_inplace(int waves, unsigned long long *d, int *sink) {
    unsigned long long h[16];
    const int iters = 2000;
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL((k_inplace<V>), dim3(256), dim3(64 * waves), 0, 0, iters, d, sink, 1u);
        if (hipDeviceSynchronize() != hipSuccess) return 1;
    }
    if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return 1;
    unsigned long long mx = 0;
    for (int i = 0; i < waves; i++) mx = h[i] > mx ? h[i] : mx;
    printf("%d wave(s)/SIMD  %2d VALU per MFMA written OVER the A operand of the MFMA in flight: %6.1f cycles per MFMA and SIMD\n", waves / 4, V, (double)mx / (8.0 * iters) / (waves / 4));
    return 0;
}

// The long-K kernel's multiply step as it is written there (two X fragments, CF W fragments of 32 lines, four MFMAs s = 0..3 per pair with the
// in-place expansion, the fourth with a shift), operands from registers, nothing else in the loop. FORM 0: as in the kernel; 1: every operand
// of a step expanded AHEAD of its MFMAs with the MFMAs kept together; 2: the next MFMA's operand expanded right behind each MFMA pair
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int FORM, int CF>
__global__ __launch_bounds__(512) void k_like(int iters, unsigned long long *cycles, int *sink, unsigned seed) {
    constexpr int RF = 2;
    u32x4 xr[RF], wr[CF];
    for (int i = 0; i < RF; i++) xr[i] = u32x4{seed * 3 + threadIdx.x + i, seed + i, 3u + i, 4u + i};
    for (int i = 0; i < CF; i++) wr[i] = u32x4{5u + i, 6u + i, 7u + i, threadIdx.x};
    f32x16 acc[RF][CF] = {};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    auto op = [](const u32x4 &v, int s) {
        const unsigned mask = s < 3 ? 0x11111111u << s : 0x11111111u;
        const u32x4 t = s < 3 ? v : v >> 3;
        const i32x4 a4 = {(int)(t.x & mask), (int)(t.y & mask), (int)(t.z & mask), (int)(t.w & mask)};
        return a4;
    };
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < RF; i++) asm volatile("" : "+v"(xr[i]));   // (fresh words every step, as from LDS)
#pragma unroll
        for (int j = 0; j < CF; j++) asm volatile("" : "+v"(wr[j]));
        if (FORM == 0 || FORM == 3) {
#pragma unroll
            for (int s = 0; s < 4; s++) {
                const int sc = s < 3 ? 128 - s : 128;
                i32x8 b8[CF];
#pragma unroll
                for (int j = 0; j < CF; j++) { const i32x4 b4 = op(wr[j], s); b8[j] = __builtin_shufflevector(b4, b4, 0, 1, 2, 3, -1, -1, -1, -1); }
#pragma unroll
                for (int i = 0; i < RF; i++) {
                    const i32x4 a4 = op(xr[i], s);
                    const i32x8 a8 = __builtin_shufflevector(a4, a4, 0, 1, 2, 3, -1, -1, -1, -1);
#pragma unroll
                    for (int j = 0; j < CF; j++) acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8[j], acc[i][j], 4, 4, 0, sc, 0, sc);
                }
            }
            if (FORM == 3) {   // the same step, scheduled as one MFMA, then five VALU operations (the NEXT operands), 4 RF CF times
#pragma unroll
                for (int n = 0; n < 4 * RF * CF; n++) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // one MFMA
                    __builtin_amdgcn_sched_group_barrier(0x002, (4 * (RF + CF) * 5 / 4 + RF * CF - 1) / (RF * CF), 0);   // VALU
                }
            }
        } else if (FORM == 4) {
            // one MFMA, then ONE operand expansion (four ANDs) and one or two of the step's sixteen shifts, pinned with sched_barrier: the
            // MFMAs of a bit s go (a0,b0) (a0,b1) (a1,b1) (a1,b0); the gaps behind them make a1(s), a0(s+1), b0(s+1), b1(s+1) in the OTHER
            // register set, so no expansion writes what an MFMA in flight reads
            static_assert(FORM != 4 || CF == 2, "the snake is written for 2 x 2 fragments");
            i32x4 A[2][2], B[2][2];
            u32x4 xs[2], ws[2];
            int shifts = 0;
            auto shift_one = [&](int n) {   // shift n of the sixteen: xs0, ws0, ws1, xs1 - the order the s = 3 expansions need them
                const int f = n >> 2, e = n & 3;
                if (f == 0) xs[0][e] = xr[0][e] >> 3; else if (f == 1) ws[0][e] = wr[0][e] >> 3; else if (f == 2) ws[1][e] = wr[1][e] >> 3; else xs[1][e] = xr[1][e] >> 3;
            };
            auto expa = [&](int i, int s2) { const unsigned m = s2 < 3 ? 0x11111111u << s2 : 0x11111111u; const u32x4 v = s2 < 3 ? xr[i] : xs[i];
                                             A[s2 & 1][i] = i32x4{(int)(v.x & m), (int)(v.y & m), (int)(v.z & m), (int)(v.w & m)}; };
            auto expb = [&](int j, int s2) { const unsigned m = s2 < 3 ? 0x11111111u << s2 : 0x11111111u; const u32x4 v = s2 < 3 ? wr[j] : ws[j];
                                             B[s2 & 1][j] = i32x4{(int)(v.x & m), (int)(v.y & m), (int)(v.z & m), (int)(v.w & m)}; };
            expa(0, 0); expb(0, 0); expb(1, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int n = 0; n < 16; n++) {
                const int s2 = n >> 2, q = n & 3, sc = s2 < 3 ? 128 - s2 : 128;
                const int i = q >> 1, j = (q == 1 || q == 2) ? 1 : 0;
                const i32x8 a8 = __builtin_shufflevector(A[s2 & 1][i], A[s2 & 1][i], 0, 1, 2, 3, -1, -1, -1, -1);
                const i32x8 b8 = __builtin_shufflevector(B[s2 & 1][j], B[s2 & 1][j], 0, 1, 2, 3, -1, -1, -1, -1);
                acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, acc[i][j], 4, 4, 0, sc, 0, sc);
                if (q == 0) expa(1, s2);
                else if (s2 < 3) { if (q == 1) expa(0, s2 + 1); else if (q == 2) expb(0, s2 + 1); else expb(1, s2 + 1); }
                const int target = n >= 10 ? 16 : (16 * (n + 1) + 10) / 11;
#pragma unroll
                for (int z = 0; z < 2; z++) if (shifts < target) shift_one(shifts++);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            // software-pipelined by hand: the operands of MFMA group n + 1 are expanded right behind the MFMAs of group n and kept apart from them
            // (sched_barrier: hipcc keeps the order; the fake use keeps group n's operand registers alive past the expansion)
            i32x4 a_cur = op(xr[0], 0), b_cur[CF];
#pragma unroll
            for (int j = 0; j < CF; j++) b_cur[j] = op(wr[j], 0);
#pragma unroll
            for (int n = 0; n < 4 * RF; n++) {   // group n = (s = n / RF, i = n % RF): CF MFMAs
                const int s = n / RF, i = n % RF, sc = s < 3 ? 128 - s : 128;
                const i32x8 a8 = __builtin_shufflevector(a_cur, a_cur, 0, 1, 2, 3, -1, -1, -1, -1);
#pragma unroll
                for (int j = 0; j < CF; j++) {
                    const i32x8 b8 = __builtin_shufflevector(b_cur[j], b_cur[j], 0, 1, 2, 3, -1, -1, -1, -1);
                    acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, acc[i][j], 4, 4, 0, sc, 0, sc);
                }
                if (FORM == 2) __builtin_amdgcn_sched_barrier(0);
                if (n + 1 < 4 * RF) {
                    const int s1 = (n + 1) / RF, i1 = (n + 1) % RF;
                    const i32x4 a_next = op(xr[i1], s1);
                    if (i1 == 0) {   // a new s: the W operands too
                        i32x4 b_next[CF];
#pragma unroll
                        for (int j = 0; j < CF; j++) b_next[j] = op(wr[j], s1);
                        if (FORM == 2) {
#pragma unroll
                            for (int j = 0; j < CF; j++) asm volatile("" ::"v"(b_cur[j]));
                        }
#pragma unroll
                        for (int j = 0; j < CF; j++) b_cur[j] = b_next[j];
                    }
                    if (FORM == 2) asm volatile("" ::"v"(a_cur));
                    a_cur = a_next;
                }
                if (FORM == 2) __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
    float sm = 0;
    for (int i = 0; i < RF; i++) for (int j = 0; j < CF; j++) sm += acc[i][j][1];
    if (sm == 12345.0f) *sink = 1;
}
template <int FORM, int CF>
int run_like(int waves, unsigned long long *d, int *sink) {
    unsigned long long h[16];
    const int iters = 1000;
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL((k_like<FORM, CF>), dim3(256), dim3(64 * waves), 0, 0, iters, d, sink, 1u);
        if (hipDeviceSynchronize() != hipSuccess) return 1;
    }
    if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return 1;
    unsigned long long mx = 0;
    for (int i = 0; i < waves; i++) mx = h[i] > mx ? h[i] : mx;
    printf("the kernel's step, form %d, 2 x %d fragments, %d wave(s)/SIMD: %6.1f cycles per MFMA and SIMD\n", FORM, CF, waves / 4, (double)mx / (8.0 * CF * iters) / (waves / 4));
    return 0;
}

// What the operands' DATA does to a bare MFMA loop (no VALU work at all): cycles per MFMA and the clock the chip holds (s_memtime / s_memrealtime).
// PAT 0: small integers (mostly zero nibbles); 1: random bits (every E2M1 code); 2: random bits masked to one bit a nibble (what the in-place expansion feeds)
template <int PAT>
__global__ __launch_bounds__(512) void k_data(int iters, unsigned long long *cycles, int *sink, unsigned seed) {
    i32x4 x[4], w[2];
    auto rnd = [&](unsigned k) { unsigned h = (threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u) ^ (k * 2246822519u) ^ seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16; return h; };
    for (int i = 0; i < 4; i++) for (int e = 0; e < 4; e++) x[i][e] = PAT == 0 ? (int)(i + e + 1) : PAT == 1 ? (int)rnd(i * 4 + e) : (int)(rnd(i * 4 + e) & 0x22222222u);
    for (int i = 0; i < 2; i++) for (int e = 0; e < 4; e++) w[i][e] = PAT == 0 ? (int)(i + e + 5) : PAT == 1 ? (int)rnd(100 + i * 4 + e) : (int)(rnd(100 + i * 4 + e) & 0x22222222u);
    f32x16 d[4][2] = {};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const i32x8 a8 = __builtin_shufflevector(x[u >> 1], x[u >> 1], 0, 1, 2, 3, -1, -1, -1, -1);
            const i32x8 b8 = __builtin_shufflevector(w[u & 1], w[u & 1], 0, 1, 2, 3, -1, -1, -1, -1);
            d[u >> 1][u & 1] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, d[u >> 1][u & 1], 4, 4, 0, 127, 0, 127);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0) { cycles[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0; if (threadIdx.x == 0) cycles[blockIdx.x * 16 + 15] = r1 - r0; }
    float sm = 0;
    for (int u = 0; u < 8; u++) sm += d[u >> 1][u & 1][1];
    if (sm == 12345.0f) *sink = 1;
}
template <int PAT>
int run_data(int waves, unsigned long long *d, int *sink) {
    unsigned long long h[16];
    const int iters = 4000;
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL((k_data<PAT>), dim3(256), dim3(64 * waves), 0, 0, iters, d, sink, 12345u);
        if (hipDeviceSynchronize() != hipSuccess) return 1;
    }
    if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return 1;
    unsigned long long mx = 0;
    for (int i = 0; i < waves; i++) mx = h[i] > mx ? h[i] : mx;
    printf("bare MFMAs, %d wave(s)/SIMD, operands %s: %6.1f cycles per MFMA and SIMD at %.3f GHz\n", waves / 4,
           PAT == 0 ? "small integers" : PAT == 1 ? "random bits" : "random bits, one bit a nibble", (double)mx / (8.0 * iters) / (waves / 4), (double)h[0] / (double)h[15] * 0.1);
    return 0;
}

// The long-K kernel's step for RF x CF fragments in the order bitmm_fp4_stream.hip.h pins (tables of tools/stream_schedule.py), alone in a loop:
// one wave a SIMD, fragments constant. MODE 0: as in the kernel; 1: the same MFMAs with exactly four ANDs behind each (no shifts, flat);
// 2: the kernel's order on two accumulators only; 3: MFMAs only
template <int RF, int CF> constexpr int t_exp_by(int n) {
    if constexpr (RF == 2 && CF == 2) { constexpr int t[] = {4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19}; return t[n]; }
    else { constexpr int t[] = {4, 4, 5, 5, 6, 7, 8, 9, 10, 10, 11, 11, 12, 13, 14, 15, 16, 16, 17, 17, 18, 19, 20, 21, 22, 22, 23, 23, 24, 25, 26, 27}; return t[n]; }
}
template <int RF, int CF> constexpr int t_shift_by(int n) {
    if constexpr (RF == 2 && CF == 2) { constexpr int t[] = {0, 0, 0, 0, 0, 2, 4, 6, 8, 10, 12, 16, 16, 16, 16, 16}; return t[n]; }
    else { constexpr int t[] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 2, 4, 6, 8, 14, 16, 20, 22, 24, 24, 24, 24, 24, 24, 24, 24, 24, 24, 24}; return t[n]; }
}
template <class F, int... I> __device__ __forceinline__ void t_for_impl(F &&f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F> __device__ __forceinline__ void t_for(F &&f) { t_for_impl(f, std::make_integer_sequence<int, N>{}); }
template <int RF, int CF, int MODE>
__global__ __launch_bounds__(512) void k_step(int iters, unsigned long long *cycles, int *sink, unsigned seed) {
    if (threadIdx.x >= 256) { __builtin_amdgcn_s_barrier(); __builtin_amdgcn_s_barrier(); return; }   // (launched with 512 threads: four more waves that only wait at the barriers)
    u32x4 xr[RF], wr[CF], x0[RF], w0[CF];
    for (int i = 0; i < RF; i++) x0[i] = u32x4{seed * 3 + threadIdx.x + i, seed * 77 + i, 0x9e3779b9u * (threadIdx.x + i), 0x85ebca6bu + i};
    for (int i = 0; i < CF; i++) w0[i] = u32x4{0xc2b2ae35u * (threadIdx.x + 1), seed + 6u + i, 0x27d4eb2fu + i, threadIdx.x};
    f32x16 acc[RF][CF] = {};
    constexpr int MN = 4 * RF * CF, OPB = RF + CF, EN = 4 * OPB, EP = 3;
    i32x4 A[4], B[2][CF];
    auto expand = [&](auto e_) {
        constexpr int e = decltype(e_)::value % EN, s = e / OPB, o = e % OPB;
        constexpr unsigned mask = s < 3 ? 0x11111111u << s : 0x11111111u;
        constexpr bool is_a = o == 0 || o > CF;
        constexpr int f = o == 0 ? 0 : (o > CF ? o - CF : o - 1);
        u32x4 v;
        if constexpr (is_a) v = xr[f]; else v = wr[f];
        const i32x4 r = {(int)(v.x & mask), (int)(v.y & mask), (int)(v.z & mask), (int)(v.w & mask)};
        if constexpr (is_a) { A[(s * RF + f) & 3] = r; asm volatile("" : "+v"(A[(s * RF + f) & 3])); }
        else { B[s & 1][f] = r; asm volatile("" : "+v"(B[s & 1][f])); }
    };
    auto shift = [&](auto z_) {
        constexpr int z = decltype(z_)::value, o = z >> 2, el = z & 3;
        constexpr bool is_a = o == 0 || o > CF;
        constexpr int f = o == 0 ? 0 : (o > CF ? o - CF : o - 1);
        if constexpr (is_a) { xr[f][el] >>= 3; if constexpr (el == 3) asm volatile("" : "+v"(xr[f])); }
        else { wr[f][el] >>= 3; if constexpr (el == 3) asm volatile("" : "+v"(wr[f])); }
    };
    for (int i = 0; i < RF; i++) xr[i] = x0[i];
    for (int i = 0; i < CF; i++) wr[i] = w0[i];
    t_for<EP>([&](auto e_) { expand(e_); });
    __builtin_amdgcn_s_barrier();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        for (int i = 0; i < RF; i++) { xr[i] = x0[i]; asm volatile("" : "+v"(xr[i])); }
        for (int i = 0; i < CF; i++) { wr[i] = w0[i]; asm volatile("" : "+v"(wr[i])); }
        __builtin_amdgcn_sched_barrier(0);
        t_for<MN>([&](auto n_) {
            constexpr int n = decltype(n_)::value, s = n / (RF * CF), q = n % (RF * CF);
            constexpr int i = q / CF, j = (i & 1) ? CF - 1 - q % CF : q % CF;
            constexpr int sc = s < 3 ? 128 - s : 128;
            const i32x4 a4 = A[(s * RF + i) & 3], b4 = B[s & 1][j];
            const i32x8 a8 = __builtin_shufflevector(a4, a4, 0, 1, 2, 3, -1, -1, -1, -1);
            const i32x8 b8 = __builtin_shufflevector(b4, b4, 0, 1, 2, 3, -1, -1, -1, -1);
            constexpr int ai = MODE == 2 ? (i & 1) : i, aj = MODE == 2 ? 0 : j;
            acc[ai][aj] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, acc[ai][aj], 4, 4, 0, sc, 0, sc);
            if constexpr (MODE == 0 || MODE == 2) {
                constexpr int e0 = n == 0 ? EP : t_exp_by<RF, CF>(n == 0 ? 0 : n - 1), e1 = t_exp_by<RF, CF>(n);
                t_for<e1 - e0>([&](auto k_) { expand(std::integral_constant<int, e0 + decltype(k_)::value>{}); });
                constexpr int z0 = n == 0 ? 0 : t_shift_by<RF, CF>(n == 0 ? 0 : n - 1), z1 = t_shift_by<RF, CF>(n);
                t_for<z1 - z0>([&](auto k_) { shift(std::integral_constant<int, z0 + decltype(k_)::value>{}); });
            } else if constexpr (MODE == 1) {
                // four ANDs behind every MFMA: the operand MFMA n + 2 reads (the real order needs fewer: this is the flat version of its VALU load)
                constexpr int n2 = (n + 2) % MN, s2 = n2 / (RF * CF), q2 = n2 % (RF * CF), i2 = q2 / CF, j2 = (i2 & 1) ? CF - 1 - q2 % CF : q2 % CF;
                constexpr unsigned mask = 0x11111111u << (s2 % 3);
                if constexpr (n % 2 == 0) { const u32x4 v = xr[i2]; A[(s2 * RF + i2) & 3] = i32x4{(int)(v.x & mask), (int)(v.y & mask), (int)(v.z & mask), (int)(v.w & mask)}; asm volatile("" : "+v"(A[(s2 * RF + i2) & 3])); }
                else { const u32x4 v = wr[j2]; B[s2 & 1][j2] = i32x4{(int)(v.x & mask), (int)(v.y & mask), (int)(v.z & mask), (int)(v.w & mask)}; asm volatile("" : "+v"(B[s2 & 1][j2])); }
            }
            __builtin_amdgcn_sched_barrier(0);
        });
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (blockDim.x == 512) __builtin_amdgcn_s_barrier();
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
    float sm = 0;
    for (int i = 0; i < RF; i++) for (int j = 0; j < CF; j++) sm += acc[i][j][1];
    if (sm == 12345.0f) *sink = 1;
}
template <int RF, int CF, int MODE>
int run_step(unsigned long long *d, int *sink, int threads = 256) {
    unsigned long long h[16];
    const int iters = 400;
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL((k_step<RF, CF, MODE>), dim3(256), dim3(threads), 0, 0, iters, d, sink, 7u);
        if (hipDeviceSynchronize() != hipSuccess) return 1;
    }
    if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return 1;
    unsigned long long mx = 0;
    for (int i = 0; i < 4; i++) mx = h[i] > mx ? h[i] : mx;
    printf("the pinned step, %d x %d fragments, one wave a SIMD%s, %s: %6.1f cycles per MFMA\n", RF, CF, threads == 512 ? " + one waiting at a barrier" : "",
           MODE == 0 ? "as in the kernel" : MODE == 1 ? "four ANDs behind every MFMA, no shifts" : MODE == 2 ? "as in the kernel on two accumulators" : "MFMAs only", (double)mx / ((double)(4 * RF * CF) * iters));
    return 0;
}

// Producer / consumer roles: waves 0-3 (one a SIMD) multiply - per group of K one barrier, then four steps of 16 MFMAs in the pinned order
// (one MFMA, one expansion, a shift or two), the fragments of step u + 1 read from LDS under the MFMAs of step u - while waves 4-7 do
// nothing but fetch: twelve LDS-DMA pieces a wave and group (48 KB a workgroup and group: the long-K kernel's 128 x 256-byte X tile +
// 64 x 256 bytes of W) out of a buffer larger than the caches. DMA = 0: the producers only meet the barrier.
typedef int rsrc4 __attribute__((ext_vector_type(4)));
template <int DMA>
__global__ __launch_bounds__(512) void k_roles(int groups, const unsigned *src, unsigned src_bytes, unsigned long long *cycles, int *sink) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int STAGE = 48 * 1024;
    f32x16 acc[2][2] = {};
    unsigned long long t0 = 0, t1 = 0;
    if (wv >= 4) {   // producer
        const rsrc4 rs = {(int)(uintptr_t)src, (int)(((uintptr_t)src >> 32) & 0xffffu), (int)src_bytes, 0x00020000};
        const unsigned lds0 = (unsigned)(uintptr_t)lds;
        const unsigned voff = lane * 16u;
        auto issue = [&](int g) {
            if (!DMA) return;
#pragma unroll
            for (int j = 0; j < 12; j++) {
                const unsigned dst = lds0 + (g % 3) * STAGE + ((wv - 4) * 12 + j) * 1024u;
                const unsigned soff = (blockIdx.x * (unsigned)groups + g) * (unsigned)STAGE + ((wv - 4) * 12 + j) * 1024u;
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "s"(dst), "v"(voff), "s"(rs), "s"(soff) : "memory");
            }
        };
        issue(0); issue(1);
        for (int g = 0; g < groups; g++) {
            if (DMA) { if (g + 1 < groups) asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
            __builtin_amdgcn_s_barrier();
            if (g + 2 < groups) issue(g + 2);
        }
    } else {
        auto rd = [&](int g, int u, u32x4 (&xr)[2], u32x4 (&wr)[2]) {
            const unsigned char *st = lds + (g % 3) * STAGE + lane * 16 + u * 4096 + wv * 8192;
            xr[0] = *reinterpret_cast<const u32x4 *>(st); xr[1] = *reinterpret_cast<const u32x4 *>(st + 1024);
            wr[0] = *reinterpret_cast<const u32x4 *>(st + 2048); wr[1] = *reinterpret_cast<const u32x4 *>(st + 3072);
        };
        i32x4 A[2][2], B[2][2];
        u32x4 xs[2], ws[2];
        auto step = [&](const u32x4 (&xr)[2], const u32x4 (&wr)[2], const u32x4 (&xn)[2], const u32x4 (&wn)[2]) {
            int shifts = 0;
            auto shift_one = [&](int n) {
                const int f = n >> 2, e = n & 3;
                if (f == 0) xs[0][e] = xr[0][e] >> 3; else if (f == 1) ws[0][e] = wr[0][e] >> 3; else if (f == 2) ws[1][e] = wr[1][e] >> 3; else xs[1][e] = xr[1][e] >> 3;
            };
            auto expa = [&](const u32x4 (&x)[2], int i, int s2) { const unsigned m = s2 < 3 ? 0x11111111u << s2 : 0x11111111u; const u32x4 v = s2 < 3 ? x[i] : xs[i];
                                             A[s2 & 1][i] = i32x4{(int)(v.x & m), (int)(v.y & m), (int)(v.z & m), (int)(v.w & m)}; };
            auto expb = [&](const u32x4 (&w)[2], int j, int s2) { const unsigned m = s2 < 3 ? 0x11111111u << s2 : 0x11111111u; const u32x4 v = s2 < 3 ? w[j] : ws[j];
                                             B[s2 & 1][j] = i32x4{(int)(v.x & m), (int)(v.y & m), (int)(v.z & m), (int)(v.w & m)}; };
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int n = 0; n < 16; n++) {
                const int s2 = n >> 2, q = n & 3, sc = s2 < 3 ? 128 - s2 : 128;
                const int i = q >> 1, j = (q == 1 || q == 2) ? 1 : 0;
                const i32x8 a8 = __builtin_shufflevector(A[s2 & 1][i], A[s2 & 1][i], 0, 1, 2, 3, -1, -1, -1, -1);
                const i32x8 b8 = __builtin_shufflevector(B[s2 & 1][j], B[s2 & 1][j], 0, 1, 2, 3, -1, -1, -1, -1);
                acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, acc[i][j], 4, 4, 0, sc, 0, sc);
                asm volatile("" : "+v"(acc[i][j]));   // (pins the MFMA ahead of the gap's VALU operations: volatile asms keep their order)
                if (q == 0) { expa(xr, 1, s2); asm volatile("" : "+v"(A[s2 & 1][1])); }
                else if (s2 < 3) { if (q == 1) { expa(xr, 0, s2 + 1); asm volatile("" : "+v"(A[(s2 + 1) & 1][0])); } else if (q == 2) { expb(wr, 0, s2 + 1); asm volatile("" : "+v"(B[(s2 + 1) & 1][0])); } else { expb(wr, 1, s2 + 1); asm volatile("" : "+v"(B[(s2 + 1) & 1][1])); } }
                else { if (q == 1) { expa(xn, 0, 0); asm volatile("" : "+v"(A[0][0])); } else if (q == 2) { expb(wn, 0, 0); asm volatile("" : "+v"(B[0][0])); } else { expb(wn, 1, 0); asm volatile("" : "+v"(B[0][1])); } }   // the next step's first operands
                const int target = n >= 10 ? 16 : (16 * (n + 1) + 10) / 11;
#pragma unroll
                for (int z = 0; z < 2; z++) if (shifts < target) shift_one(shifts++);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        u32x4 xa[2], wa[2], xb[2], wb[2];
        __builtin_amdgcn_s_barrier();   // group 0
        t0 = __builtin_amdgcn_s_memtime();
        rd(0, 0, xa, wa);
        A[0][0] = i32x4{(int)(xa[0].x & 0x11111111u), (int)(xa[0].y & 0x11111111u), (int)(xa[0].z & 0x11111111u), (int)(xa[0].w & 0x11111111u)};
        B[0][0] = i32x4{(int)(wa[0].x & 0x11111111u), (int)(wa[0].y & 0x11111111u), (int)(wa[0].z & 0x11111111u), (int)(wa[0].w & 0x11111111u)};
        B[0][1] = i32x4{(int)(wa[1].x & 0x11111111u), (int)(wa[1].y & 0x11111111u), (int)(wa[1].z & 0x11111111u), (int)(wa[1].w & 0x11111111u)};
        for (int g = 0; g < groups; g++) {
            rd(g, 1, xb, wb); step(xa, wa, xb, wb);
            rd(g, 2, xa, wa); step(xb, wb, xa, wa);
            rd(g, 3, xb, wb); step(xa, wa, xb, wb);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (g + 1 < groups) { __builtin_amdgcn_s_barrier(); rd(g + 1, 0, xa, wa); }
            step(xb, wb, xa, wa);
        }
        asm volatile("" : "+v"(acc[0][0]));
        t1 = __builtin_amdgcn_s_memtime();
    }
    if (lane == 0) cycles[blockIdx.x * 16 + wv] = t1 - t0;
    float sm = 0;
    for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++) sm += acc[i][j][1];
    if (sm == 12345.0f) *sink = 1;
}
template <int DMA>
int run_roles(unsigned long long *d, int *sink, const unsigned *src, unsigned src_bytes) {
    unsigned long long h[16];
    const int groups = 16;
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(&k_roles<DMA>), hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024) != hipSuccess) return 1;
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL((k_roles<DMA>), dim3(256), dim3(512), 144 * 1024, 0, groups, src, src_bytes, d, sink);
        if (hipDeviceSynchronize() != hipSuccess) { printf("roles: launch failed\n"); return 1; }
    }
    if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return 1;
    unsigned long long mx = 0;
    for (int i = 0; i < 4; i++) mx = h[i] > mx ? h[i] : mx;
    printf("roles (4 multiplying waves, 4 fetching%s), %d groups of 64 MFMAs a wave: %6.1f cycles per MFMA and SIMD, %llu cycles a group\n", DMA ? "" : " - DMAs off", groups,
           (double)mx / (64.0 * groups), mx / groups);
    return 0;
}

template <int V, int AGPR, int DEP, int PRIO>
__global__ __launch_bounds__(512) void k(int iters, unsigned long long *cycles, int *sink, unsigned seed) {
    i32x4 x[4], w[2];
    for (int i = 0; i < 4; i++) x[i] = i32x4{(int)(seed * 3 + threadIdx.x + i), (int)(seed + i), 3 + i, 4 + i};
    for (int i = 0; i < 2; i++) w[i] = i32x4{5 + i, 6 + i, 7 + i, (int)threadIdx.x};
    f32x16 d[4] = {};
    i32x4 a4 = x[0] & 0x11111111, e4 = x[1];
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const i32x8 a8 = __builtin_shufflevector(a4, a4, 0, 1, 2, 3, -1, -1, -1, -1);
            const i32x4 b4 = w[u & 1];
            const i32x8 b8 = __builtin_shufflevector(b4, b4, 0, 1, 2, 3, -1, -1, -1, -1);
            if (AGPR) asm volatile("" : "+a"(d[u & 3]));
            if (PRIO) __builtin_amdgcn_s_setprio(1);
            d[u & 3] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, d[u & 3], 4, 4, 0, 128, 0, 127);
            if (PRIO) __builtin_amdgcn_s_setprio(0);
            if (AGPR) asm volatile("" : "+a"(d[u & 3]));
            // V VALU operations behind the MFMA; with DEP four of them make the next MFMA's A operand
            i32x4 n4 = x[(u + 1) & 3] & (0x11111111 << ((u >> 1) & 1));
            if (V >= 4) { asm volatile("" : "+v"(n4)); if (DEP) a4 = n4; else e4 ^= n4; }
#pragma unroll
            for (int q = 4; q < V; q++) { e4[q & 3] = (e4[q & 3] & 0x33333333) ; asm volatile("" : "+v"(e4)); }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
    float s = 0;
    for (int u = 0; u < 4; u++) s += d[u][1];
    if (s == 12345.0f && e4[0] + e4[1] + e4[2] + e4[3] + a4[0] == 77) *sink = 1;
}

template <int V, int AGPR, int DEP, int PRIO>
int run(int waves, unsigned long long *d, int *sink) {
    unsigned long long h[16];
    const int iters = 2000;
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL((k<V, AGPR, DEP, PRIO>), dim3(256), dim3(64 * waves), 0, 0, iters, d, sink, 1u);
        CK(hipDeviceSynchronize());
    }
    CK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
    unsigned long long mx = 0;
    for (int i = 0; i < waves; i++) mx = h[i] > mx ? h[i] : mx;
    printf("%d wave(s)/SIMD  %2d VALU per MFMA  acc in %s  A operand %s  %s: %6.1f cycles per MFMA and SIMD (the first wave alone: %.1f per MFMA)\n", waves / 4, V,
           AGPR ? "AGPRs" : "VGPRs", DEP ? "from the VALU results (other registers)" : "constant", PRIO ? "setprio 1 around the MFMA" : "no setprio",
           (double)mx / (8.0 * iters) / (waves / 4), (double)h[0] / (8.0 * iters));
    return 0;
}

int main() {
    unsigned long long *d; int *sink;
    CK(hipMalloc(&d, 256 * 16 * 8)); CK(hipMalloc(&sink, 4));
    {
        const unsigned src_bytes = 256u * 16u * 48u * 1024u;   // 192 MiB: every workgroup its own 768 KB
        unsigned *src;
        CK(hipMalloc(&src, src_bytes)); CK(hipMemset(src, 0x5a, src_bytes));
        run_roles<1>(d, sink, src, src_bytes); run_roles<0>(d, sink, src, src_bytes);
        CK(hipFree(src));
        run_step<4, 2, 0>(d, sink, 512); run_step<4, 2, 3>(d, sink, 512);
        run_step<4, 2, 0>(d, sink); run_step<4, 2, 1>(d, sink); run_step<4, 2, 2>(d, sink); run_step<4, 2, 3>(d, sink);
        run_step<2, 2, 0>(d, sink); run_step<2, 2, 1>(d, sink); run_step<2, 2, 3>(d, sink);
        if (getenv("STEP_ONLY")) return 0;
        for (int waves : {4, 8}) { run_data<0>(waves, d, sink); run_data<1>(waves, d, sink); run_data<2>(waves, d, sink); }
        if (getenv("ROLES_ONLY")) return 0;
    }
    for (int waves : {4, 8}) {
        run_like<0, 2>(waves, d, sink); run_like<1, 2>(waves, d, sink); run_like<2, 2>(waves, d, sink); run_like<3, 2>(waves, d, sink); run_like<4, 2>(waves, d, sink); run_like<0, 4>(waves, d, sink); run_like<2, 4>(waves, d, sink); run_like<3, 4>(waves, d, sink);
        if (getenv("LIKE_ONLY")) continue;
        run<0, 0, 0, 0>(waves, d, sink); run<4, 0, 1, 0>(waves, d, sink); run<5, 0, 1, 0>(waves, d, sink); run<8, 0, 1, 0>(waves, d, sink);
        run<5, 0, 0, 0>(waves, d, sink); run<0, 1, 0, 0>(waves, d, sink); run<5, 1, 1, 0>(waves, d, sink); run<8, 1, 1, 0>(waves, d, sink);
        run<5, 0, 1, 1>(waves, d, sink); run<5, 1, 1, 1>(waves, d, sink);
        run_inplace<0>(waves, d, sink); run_inplace<4>(waves, d, sink); run_inplace<5>(waves, d, sink); run_inplace<8>(waves, d, sink);
    }
    return 0;
}
