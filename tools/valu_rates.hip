// tools/valu_rates.hip — per-instruction issue rates of the integer VALU ops the popcount GEMM can be
// built from (one opcode per loop; 2 and 8 waves per SIMD on every CU).
//   hipcc --offload-arch=gfx950 -O3 -o tools/valu_rates tools/valu_rates.hip && tools/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(s) s s s s s s s s
template <int MODE>
__global__ void k_rate(uint32_t *out, int iters) {
    uint32_t a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;
    uint32_t x = threadIdx.x * 2654435761u, w = blockIdx.x * 40503u + 77;
    for (int i = 0; i < iters; i++) {
#define BODY(op3)                                                                                   \
    asm volatile(REP8(op3 " %0, %8, %0\n" op3 " %1, %9, %1\n" op3 " %2, %8, %2\n" op3 " %3, %9, %3\n"  \
                      op3 " %4, %8, %4\n" op3 " %5, %9, %5\n" op3 " %6, %8, %6\n" op3 " %7, %9, %7\n") \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)    \
                 : "v"(x), "v"(w));
        if (MODE == 0) { BODY("v_and_b32") }
        if (MODE == 1) { BODY("v_bcnt_u32_b32") }
        if (MODE == 2) { BODY("v_add_u32") }
        if (MODE == 3) { BODY("v_xor_b32") }
        if (MODE == 4) { BODY("v_add_f32") }
        if (MODE == 5) { BODY("v_lshlrev_b32") }
        if (MODE == 6) {
            asm volatile(REP8("v_and_or_b32 %0, %8, %9, %0\n v_and_or_b32 %1, %8, %9, %1\n v_and_or_b32 %2, %8, %9, %2\n v_and_or_b32 %3, %8, %9, %3\n"
                              "v_and_or_b32 %4, %8, %9, %4\n v_and_or_b32 %5, %8, %9, %5\n v_and_or_b32 %6, %8, %9, %6\n v_and_or_b32 %7, %8, %9, %7\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(w));
        }
        if (MODE == 7) {
            asm volatile(REP8("v_add3_u32 %0, %8, %9, %0\n v_add3_u32 %1, %8, %9, %1\n v_add3_u32 %2, %8, %9, %2\n v_add3_u32 %3, %8, %9, %3\n"
                              "v_add3_u32 %4, %8, %9, %4\n v_add3_u32 %5, %8, %9, %5\n v_add3_u32 %6, %8, %9, %6\n v_add3_u32 %7, %8, %9, %7\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(w));
        }
        if (MODE == 8) {
            asm volatile(REP8("v_fma_f32 %0, %8, %9, %0\n v_fma_f32 %1, %8, %9, %1\n v_fma_f32 %2, %8, %9, %2\n v_fma_f32 %3, %8, %9, %3\n"
                              "v_fma_f32 %4, %8, %9, %4\n v_fma_f32 %5, %8, %9, %5\n v_fma_f32 %6, %8, %9, %6\n v_fma_f32 %7, %8, %9, %7\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(w));
        }
        if (MODE == 9) {
            asm volatile(REP8("v_pk_add_u16 %0, %8, %0\n v_pk_add_u16 %1, %9, %1\n v_pk_add_u16 %2, %8, %2\n v_pk_add_u16 %3, %9, %3\n"
                              "v_pk_add_u16 %4, %8, %4\n v_pk_add_u16 %5, %9, %5\n v_pk_add_u16 %6, %8, %6\n v_pk_add_u16 %7, %9, %7\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(w));
        }
        if (MODE == 11) { BODY("v_mul_u32_u24") }
        if (MODE == 12) {
            asm volatile(REP8("v_bfe_u32 %0, %8, 4, 4\n v_bfe_u32 %1, %9, 8, 4\n v_bfe_u32 %2, %8, 12, 4\n v_bfe_u32 %3, %9, 16, 4\n"
                              "v_bfe_u32 %4, %8, 20, 4\n v_bfe_u32 %5, %9, 24, 4\n v_bfe_u32 %6, %8, 28, 4\n v_bfe_u32 %7, %9, 0, 4\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(w));
        }
        if (MODE == 13) {
            asm volatile(REP8("v_lshl_or_b32 %0, %8, 7, %0\n v_lshl_or_b32 %1, %9, 7, %1\n v_lshl_or_b32 %2, %8, 7, %2\n v_lshl_or_b32 %3, %9, 7, %3\n"
                              "v_lshl_or_b32 %4, %8, 7, %4\n v_lshl_or_b32 %5, %9, 7, %5\n v_lshl_or_b32 %6, %8, 7, %6\n v_lshl_or_b32 %7, %9, 7, %7\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(w));
        }
        if (MODE == 14) {
            asm volatile(REP8("v_perm_b32 %0, %8, %9, %0\n v_perm_b32 %1, %8, %9, %1\n v_perm_b32 %2, %8, %9, %2\n v_perm_b32 %3, %8, %9, %3\n"
                              "v_perm_b32 %4, %8, %9, %4\n v_perm_b32 %5, %8, %9, %5\n v_perm_b32 %6, %8, %9, %6\n v_perm_b32 %7, %8, %9, %7\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(w));
        }
        if (MODE == 15) { BODY("v_lshrrev_b32") }
        if (MODE == 16) {
            asm volatile(REP8("v_mul_lo_u32 %0, %8, %0\n v_mul_lo_u32 %1, %9, %1\n v_mul_lo_u32 %2, %8, %2\n v_mul_lo_u32 %3, %9, %3\n"
                              "v_mul_lo_u32 %4, %8, %4\n v_mul_lo_u32 %5, %9, %5\n v_mul_lo_u32 %6, %8, %6\n v_mul_lo_u32 %7, %9, %7\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(w));
        }
        if (MODE == 10) {
            asm volatile(REP8("v_dot4_u32_u8 %0, %8, %9, %0\n v_dot4_u32_u8 %1, %8, %9, %1\n v_dot4_u32_u8 %2, %8, %9, %2\n v_dot4_u32_u8 %3, %8, %9, %3\n"
                              "v_dot4_u32_u8 %4, %8, %9, %4\n v_dot4_u32_u8 %5, %8, %9, %5\n v_dot4_u32_u8 %6, %8, %9, %6\n v_dot4_u32_u8 %7, %8, %9, %7\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(w));
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int MODE>
void run(const char *name, uint32_t *out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 4000;
    for (int wps : {1, 2, 8}) {
        const int threads = 256, blocks = 256 * wps;
        float ms = 0;
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_rate<MODE>, dim3(blocks), dim3(threads), 0, 0, out, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        const double instr = (double)blocks * (threads / 64) * iters * 64;
        const double rate = instr / (ms * 1e-3);
        printf("%-16s waves/SIMD=%d: %.3f ms  %.2f cycles/instr/SIMD @2.4GHz  (%.3e lane-instr/s)\n", name, wps, ms,
               2.4e9 * 1024 / rate, rate * 64);
    }
}

int main() {
    uint32_t *out;
    hipMalloc(&out, 256 * 8 * 1024 * 4);
    run<0>("v_and_b32", out);
    run<1>("v_bcnt_u32_b32", out);
    run<2>("v_add_u32", out);
    run<3>("v_xor_b32", out);
    run<4>("v_add_f32", out);
    run<5>("v_lshlrev_b32", out);
    run<6>("v_and_or_b32", out);
    run<7>("v_add3_u32", out);
    run<8>("v_fma_f32", out);
    run<9>("v_pk_add_u16", out);
    run<10>("v_dot4_u32_u8", out);
    run<11>("v_mul_u32_u24", out);
    run<12>("v_bfe_u32", out);
    run<13>("v_lshl_or_b32", out);
    run<14>("v_perm_b32", out);
    run<15>("v_lshrrev_b32", out);
    run<16>("v_mul_lo_u32", out);
    return 0;
}
