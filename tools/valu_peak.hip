// tools/valu_peak.hip — measures the v_and_b32 + v_bcnt_u32_b32 issue rate on the box (the VALU
// roofline of the popcount GEMM): every CU runs W waves/SIMD of a register-only loop.
//   hipcc --offload-arch=gfx950 -O3 -o tools/valu_peak tools/valu_peak.hip && tools/valu_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>
__global__ void k_valu(uint32_t *out, int iters) {
    uint32_t a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;
    uint32_t x = threadIdx.x * 2654435761u, w = blockIdx.x * 40503u + 77;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (MODE == 0) {  // and + bcnt(acc)
                uint32_t t0, t1, t2, t3, t4, t5, t6, t7;
                asm volatile(
                    "v_and_b32 %8, %16, %17\n v_and_b32 %9, %16, %18\n v_and_b32 %10, %16, %19\n v_and_b32 %11, %16, %20\n"
                    "v_and_b32 %12, %16, %21\n v_and_b32 %13, %16, %22\n v_and_b32 %14, %16, %23\n v_and_b32 %15, %16, %24\n"
                    "v_bcnt_u32_b32 %0, %8, %0\n v_bcnt_u32_b32 %1, %9, %1\n v_bcnt_u32_b32 %2, %10, %2\n v_bcnt_u32_b32 %3, %11, %3\n"
                    "v_bcnt_u32_b32 %4, %12, %4\n v_bcnt_u32_b32 %5, %13, %5\n v_bcnt_u32_b32 %6, %14, %6\n v_bcnt_u32_b32 %7, %15, %7\n"
                    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7),
                      "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7)
                    : "v"(x), "v"(w), "v"(w + 1), "v"(w + 2), "v"(w + 3), "v"(w + 4), "v"(w + 5), "v"(w + 6), "v"(w + 7));
            } else {  // fma f32 for comparison (16 instrs)
                float f0 = __uint_as_float(a0), f1 = __uint_as_float(a1), f2 = __uint_as_float(a2), f3 = __uint_as_float(a3);
                float f4 = __uint_as_float(a4), f5 = __uint_as_float(a5), f6 = __uint_as_float(a6), f7 = __uint_as_float(a7);
                float fx = __uint_as_float(x);
                asm volatile(
                    "v_fma_f32 %0, %8, %0, %0\n v_fma_f32 %1, %8, %1, %1\n v_fma_f32 %2, %8, %2, %2\n v_fma_f32 %3, %8, %3, %3\n"
                    "v_fma_f32 %4, %8, %4, %4\n v_fma_f32 %5, %8, %5, %5\n v_fma_f32 %6, %8, %6, %6\n v_fma_f32 %7, %8, %7, %7\n"
                    "v_fma_f32 %0, %8, %0, %0\n v_fma_f32 %1, %8, %1, %1\n v_fma_f32 %2, %8, %2, %2\n v_fma_f32 %3, %8, %3, %3\n"
                    "v_fma_f32 %4, %8, %4, %4\n v_fma_f32 %5, %8, %5, %5\n v_fma_f32 %6, %8, %6, %6\n v_fma_f32 %7, %8, %7, %7\n"
                    : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(fx));
                a0 = __float_as_uint(f0); a1 = __float_as_uint(f1); a2 = __float_as_uint(f2); a3 = __float_as_uint(f3);
                a4 = __float_as_uint(f4); a5 = __float_as_uint(f5); a6 = __float_as_uint(f6); a7 = __float_as_uint(f7);
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

int main() {
    uint32_t *out;
    hipMalloc(&out, 256 * 8 * 1024 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 20000;
    for (int mode = 0; mode < 2; mode++)
        for (int wps : {1, 2, 4, 8}) {            // waves per SIMD
            const int threads = 256, blocks = 256 * wps;  // 4 waves per block = 1 per SIMD
            for (int rep = 0; rep < 2; rep++) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(k_valu<0>, dim3(blocks), dim3(threads), 0, 0, out, iters);
                else hipLaunchKernelGGL(k_valu<1>, dim3(blocks), dim3(threads), 0, 0, out, iters);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
            }
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double instr = (double)blocks * (threads / 64) * iters * 8 * 16;   // wave-instructions
            const double rate = instr / (ms * 1e-3);                                  // wave-instr/s chip-wide
            printf("%s waves/SIMD=%d: %.3f ms  %.3e wave-instr/s  = %.2f cycles/instr/SIMD @2.4GHz  (%.3e lane-instr/s)\n",
                   mode == 0 ? "and+bcnt" : "fma_f32 ", wps, ms, rate, 2.4e9 * 1024 / rate, rate * 64);
        }
    return 0;
}
