// tools/vgpr_bank.hip — does v_and_b32 pay for two source VGPRs in the same register bank (index mod 4)?
// The MAC loop ANDs component c of an X granule with component c of a W granule; both granules are
// 4-aligned register tuples, so both sources of every AND sit in the same bank.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int SAME>
__global__ void k(uint32_t *out, int iters) {
    uint32_t r = threadIdx.x;
    for (int i = 0; i < iters; i++) {
        if (SAME) {
            asm volatile(
                "v_and_b32 v40, v4, v8\n v_and_b32 v41, v4, v12\n v_and_b32 v42, v4, v16\n v_and_b32 v43, v4, v20\n"
                "v_and_b32 v44, v24, v8\n v_and_b32 v45, v24, v12\n v_and_b32 v46, v24, v16\n v_and_b32 v47, v24, v20\n"
                "v_bcnt_u32_b32 v50, v40, v50\n v_bcnt_u32_b32 v51, v41, v51\n v_bcnt_u32_b32 v52, v42, v52\n v_bcnt_u32_b32 v53, v43, v53\n"
                "v_bcnt_u32_b32 v54, v44, v54\n v_bcnt_u32_b32 v55, v45, v55\n v_bcnt_u32_b32 v56, v46, v56\n v_bcnt_u32_b32 v57, v47, v57\n"
                ::: "v4", "v8", "v12", "v16", "v20", "v24", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47",
                    "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57");
        } else {
            asm volatile(
                "v_and_b32 v40, v4, v9\n v_and_b32 v41, v4, v13\n v_and_b32 v42, v4, v17\n v_and_b32 v43, v4, v21\n"
                "v_and_b32 v44, v24, v9\n v_and_b32 v45, v24, v13\n v_and_b32 v46, v24, v17\n v_and_b32 v47, v24, v21\n"
                "v_bcnt_u32_b32 v50, v40, v50\n v_bcnt_u32_b32 v51, v41, v51\n v_bcnt_u32_b32 v52, v42, v52\n v_bcnt_u32_b32 v53, v43, v53\n"
                "v_bcnt_u32_b32 v54, v44, v54\n v_bcnt_u32_b32 v55, v45, v55\n v_bcnt_u32_b32 v56, v46, v56\n v_bcnt_u32_b32 v57, v47, v57\n"
                ::: "v4", "v9", "v13", "v17", "v21", "v24", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47",
                    "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57");
        }
    }
    asm volatile("v_mov_b32 %0, v50" : "=v"(r));
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
int main() {
    uint32_t *out;
    hipMalloc(&out, 256 * 8 * 1024 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 20000;
    for (int same = 0; same < 2; same++)
        for (int wps : {2, 4}) {
            float ms = 0;
            for (int rep = 0; rep < 2; rep++) {
                hipEventRecord(e0);
                if (same) hipLaunchKernelGGL(k<1>, dim3(256 * wps), dim3(256), 0, 0, out, iters);
                else hipLaunchKernelGGL(k<0>, dim3(256 * wps), dim3(256), 0, 0, out, iters);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
            }
            const double instr = 256.0 * wps * 4 * iters * 16;
            printf("%s bank, %d waves/SIMD: %.2f cycles/instr/SIMD @2.4GHz\n", same ? "same" : "different", wps,
                   2.4e9 * 1024 / (instr / (ms * 1e-3)));
        }
    return 0;
}
