// tools/mfma_overlap.hip — round 6: does VALU work overlap the FP4 block-scaled MFMAs, and what decides it? (the long-K and wide kernels run
// 55 cycles per v_mfma_scale_f32_32x32x64_f8f6f4 at 5 VALU operations per MFMA where the pipe needs 32-34: additive, two waves a SIMD or not)
//   A wave issues, per MFMA, V v_and_b32 (4 of them produce the NEXT MFMA's A operand when DEP = 1, like the in-place expansion), then the MFMA.
//   Variants: accumulators in VGPRs / AGPRs, MFMA operands fed by VALU results or constant, one / two waves a SIMD, s_setprio around the MFMA.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/mfma_overlap tools/mfma_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
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
        if (FORM == 0) {
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
    for (int waves : {4, 8}) {
        run_like<0, 2>(waves, d, sink); run_like<1, 2>(waves, d, sink); run_like<2, 2>(waves, d, sink); run_like<0, 4>(waves, d, sink); run_like<2, 4>(waves, d, sink);
        if (getenv("LIKE_ONLY")) continue;
        run<0, 0, 0, 0>(waves, d, sink); run<4, 0, 1, 0>(waves, d, sink); run<5, 0, 1, 0>(waves, d, sink); run<8, 0, 1, 0>(waves, d, sink);
        run<5, 0, 0, 0>(waves, d, sink); run<0, 1, 0, 0>(waves, d, sink); run<5, 1, 1, 0>(waves, d, sink); run<8, 1, 1, 0>(waves, d, sink);
        run<5, 0, 1, 1>(waves, d, sink); run<5, 1, 1, 1>(waves, d, sink);
        run_inplace<0>(waves, d, sink); run_inplace<4>(waves, d, sink); run_inplace<5>(waves, d, sink); run_inplace<8>(waves, d, sink);
    }
    return 0;
}
