// tools/mfma_rates2.hip — what ONE wave per SIMD gets out of the FP4 block-scaled MFMAs when it also does the operand
// expansion itself (the register-tile kernel's situation): issue interval of v_mfma_scale_f32_16x16x128_f8f6f4 and
// v_mfma_scale_f32_32x32x64_f8f6f4, alone and with V independent VALU operations of the SAME wave between two MFMAs.
//   hipcc --offload-arch=gfx950 -O2 -o tools/mfma_rates2 tools/mfma_rates2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s\n", hipGetErrorString(e)); return 1; } } while (0)

template <int FORM, int V>   // FORM 0: 16x16x128, 1: 32x32x64; V VALU ops (v_and_b32 with a literal) after every MFMA
__global__ void k(int iters, unsigned long long *cycles, int *sink, unsigned seed) {
    i32x8 a8 = {(int)seed, 2, 3, 4, 0, 0, 0, 0}, b8 = {5, 6, 7, (int)threadIdx.x, 0, 0, 0, 0};
    f32x4 c[8] = {};
    f32x16 d[4] = {};
    unsigned v[8];
    for (int i = 0; i < 8; i++) v[i] = threadIdx.x * 7u + i;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (FORM == 0) c[u] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8, b8, c[u], 4, 4, 0, 128, 0, 127);
            else d[u & 3] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, d[u & 3], 4, 4, 0, 128, 0, 127);
#pragma unroll
            for (int x = 0; x < V; x++) {
                v[x & 7] = (v[x & 7] & 0x11111111u) + 3u;
                asm volatile("" : "+v"(v[x & 7]));
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
    float s = 0;
    for (int u = 0; u < 8; u++) s += c[u][0];
    for (int u = 0; u < 4; u++) s += d[u][1];
    unsigned t = 0;
    for (int i = 0; i < 8; i++) t += v[i];
    if (s == 12345.0f && t == 77u) *sink = 1;
}

template <int FORM, int V>
int run(const char *name, int waves, unsigned long long *d, int *sink) {
    unsigned long long h[16];
    const int iters = 1000;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms = 0;
    for (int rep = 0; rep < 2; rep++) {
        CK(hipEventRecord(e0, nullptr));
        hipLaunchKernelGGL((k<FORM, V>), dim3(256), dim3(64 * waves), 0, 0, iters, d, sink, 1u);
        CK(hipEventRecord(e1, nullptr));
        CK(hipDeviceSynchronize());
        CK(hipEventElapsedTime(&ms, e0, e1));
    }
    CK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
    const double per = (double)h[0] / (8.0 * iters);
    const double flops = FORM == 0 ? 2.0 * 16 * 16 * 128 : 2.0 * 32 * 32 * 64;
    printf("%-28s %d wave(s)/SIMD, %2d VALU per MFMA: %6.1f ticks per MFMA  (%.0f flop/tick/SIMD)  %.1f us = %.2f ticks/ns, %.2f PFLOP/s chip\n", name, waves / 4, V, per,
           flops * (waves / 4) / per, ms * 1e3, (double)h[0] / (ms * 1e6), flops * 8.0 * iters * waves * 256 / (ms * 1e-3) / 1e15);
    return 0;
}

int main() {
    unsigned long long *d; int *sink;
    CK(hipMalloc(&d, 256 * 16 * 8)); CK(hipMalloc(&sink, 4));
    run<0, 0>("fp4 16x16x128", 4, d, sink); run<0, 2>("fp4 16x16x128", 4, d, sink); run<0, 4>("fp4 16x16x128", 4, d, sink);
    run<0, 8>("fp4 16x16x128", 4, d, sink); run<0, 0>("fp4 16x16x128", 8, d, sink); run<0, 4>("fp4 16x16x128", 8, d, sink);
    run<1, 0>("fp4 32x32x64", 4, d, sink); run<1, 4>("fp4 32x32x64", 4, d, sink); run<1, 8>("fp4 32x32x64", 4, d, sink);
    run<1, 16>("fp4 32x32x64", 4, d, sink); run<1, 0>("fp4 32x32x64", 8, d, sink); run<1, 8>("fp4 32x32x64", 8, d, sink);
    return 0;
}
