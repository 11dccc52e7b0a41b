// tools/mfma_rates.hip — issue rate of the two MFMA forms the matrix-core engine uses and whether VALU work of
// ANOTHER wave on the same SIMD overlaps with them.
//   hipcc --offload-arch=gfx950 -O2 -o tools/mfma_rates tools/mfma_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s\n", hipGetErrorString(e)); return 1; } } while (0)

// mode 0: every wave runs MFMA i8; 1: every wave runs MFMA fp4; 2: every wave VALU;
// 3: wave 0 of a pair i8 + wave 1 VALU; 4: wave 0 fp4 + wave 1 VALU   (blockDim = 64 * waves, one block per CU)
__global__ void k(int mode, int iters, unsigned long long *cycles, int *sink) {
    const int wv = threadIdx.x >> 6;
    const bool valu = mode == 2 || (mode >= 3 && (wv & 4));   // waves 4..7 share SIMDs 0..3 with waves 0..3
    const bool fp4 = mode == 1 || mode == 4;
    i32x4 a = {1, 2, 3, 4}, b = {5, 6, 7, 8};
    i32x8 a8 = {1, 2, 3, 4, 0, 0, 0, 0}, b8 = {5, 6, 7, 8, 0, 0, 0, 0};
    i32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
    f32x16 f0 = {}, f1 = {}, f2 = {}, f3 = {};
    unsigned v0 = threadIdx.x, v1 = 3, v2 = 5, v3 = 7;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (valu) {
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                v0 = (v0 >> 1) & 0x11111111u; v1 = (v1 >> 2) & 0x11111111u; v2 = (v2 >> 3) & 0x11111111u; v3 = (v3 >> 1) ^ v0;
                asm volatile("" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
            }
        }
    } else if (fp4) {
        for (int i = 0; i < iters; i++) {
            f0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, f0, 4, 4, 0, 128, 0, 128);
            f1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, f1, 4, 4, 0, 128, 0, 128);
            f2 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, f2, 4, 4, 0, 128, 0, 128);
            f3 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, f3, 4, 4, 0, 128, 0, 128);
        }
    } else {
        for (int i = 0; i < iters; i++) {
            c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c3, 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * 16 + wv] = t1 - t0;
    int s = c0[0] + c1[1] + c2[2] + c3[3] + (int)(f0[0] + f1[1] + f2[2] + f3[3]) + (int)(v0 + v1 + v2 + v3);
    if (s == 0x7fffffff) *sink = s;
}

int main() {
    unsigned long long *d; int *sink;
    CK(hipMalloc(&d, 256 * 16 * 8)); CK(hipMalloc(&sink, 4));
    unsigned long long h[16];
    const int iters = 2000;
    const char *names[] = {"4 waves/CU: i8 32x32x32", "4 waves/CU: fp4 32x32x64", "4 waves/CU: VALU (32 ops/iter)",
                           "8 waves/CU: waves 0-3 i8, waves 4-7 VALU", "8 waves/CU: waves 0-3 fp4, waves 4-7 VALU"};
    for (int mode = 0; mode < 5; mode++) {
        const int waves = mode >= 3 ? 8 : 4;
        for (int rep = 0; rep < 2; rep++) {
            hipLaunchKernelGGL(k, dim3(256), dim3(64 * waves), 0, 0, mode, iters, d, sink);
            CK(hipDeviceSynchronize());
        }
        CK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
        printf("%-46s", names[mode]);
        if (mode < 2) printf(" %.1f ticks per MFMA\n", (double)h[0] / (4.0 * iters));
        else if (mode == 2) printf(" %.2f ticks per VALU op\n", (double)h[0] / (32.0 * iters));
        else printf(" MFMA wave: %.1f ticks per MFMA, VALU wave: %.2f ticks per op\n", (double)h[0] / (4.0 * iters), (double)h[4] / (32.0 * iters));
    }
    return 0;
}
