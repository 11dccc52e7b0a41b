// tools/fp6_probe.hip — can a 4-bit value travel as ONE matrix-core operand element? E2M3 (fp6) has codes 0 .. 15 = v / 8 exactly
// (subnormals m / 8, then 1 + m / 8), so a 4-bit value stored as the 6-bit field v means v / 8 - no split into two base-4 digits.
//   (1) layout: field i (bits 6 i .. 6 i + 5 of the lane's 192 bits, 6 registers) of A (fp6) meets nibble i of B (fp4);
//   (2) random 4-bit x 1-bit and 4-bit x 4-bit products equal the integer product with E8M0 scales 2^3 (fp6) / 2^1 (fp4);
//   (3) issue rate of the fp6 form against fp4 (same passes?).
//   hipcc --offload-arch=gfx950 -O2 -o tools/fp6_probe tools/fp6_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// mode 0: A fp6 x B fp4; mode 1: A fp6 x B fp6
__global__ void k(const int *a, const int *b, float *c, int mode) {
    const int lane = threadIdx.x;
    i32x8 av, bv;
    for (int i = 0; i < 8; i++) { av[i] = i < 6 ? a[lane * 8 + i] : 0; bv[i] = i < 6 ? b[lane * 8 + i] : 0; }
    f32x16 acc;
    for (int i = 0; i < 16; i++) acc[i] = 0.f;
    if (mode == 0) acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, acc, 2, 4, 0, 130, 0, 128);
    else acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, acc, 2, 2, 0, 130, 0, 130);
    for (int r = 0; r < 16; r++) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), col = lane & 31;
        c[row * 32 + col] = acc[r];
    }
}

template <int FA, int FB>
__global__ void rate(float *out, int iters) {
    i32x8 a, b;
    for (int i = 0; i < 8; i++) { a[i] = threadIdx.x + i; b[i] = threadIdx.x * 3 + i; }
    f32x16 acc[4];
    for (int j = 0; j < 4; j++) for (int i = 0; i < 16; i++) acc[j][i] = 0.f;
    for (int it = 0; it < iters; it++)
        for (int j = 0; j < 4; j++) acc[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc[j], FA, FB, 0, 127, 0, 127);
    float s = 0.f;
    for (int j = 0; j < 4; j++) s += acc[j][0];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    std::vector<int> ha(512), hb(512);
    std::vector<float> hc(1024);
    int *da, *db; float *dc;
    hipMalloc(&da, 2048); hipMalloc(&db, 2048); hipMalloc(&dc, 4096);
    srand(1);
    for (int mode = 0; mode < 2; mode++) {
        std::vector<int> xa(32 * 64), xb(32 * 64);   // [line][k], k = 32 * (lane >> 5) + element index
        for (auto &v : xa) v = rand() & 15;
        for (auto &v : xb) v = mode == 0 ? (rand() & 3) : (rand() & 15);
        for (auto &v : ha) v = 0;
        for (auto &v : hb) v = 0;
        for (int lane = 0; lane < 64; lane++)
            for (int e = 0; e < 32; e++) {
                const int kk = 32 * (lane >> 5) + e;
                const unsigned long long va = xa[(lane & 31) * 64 + kk], vb = xb[(lane & 31) * 64 + kk];
                // fp6: field e at bits 6 e .. 6 e + 5 of the lane's 192 bits
                const int bit = 6 * e;
                ha[lane * 8 + bit / 32] |= (int)((va << (bit % 32)) & 0xffffffffu);
                if (bit % 32 > 26) ha[lane * 8 + bit / 32 + 1] |= (int)(va >> (32 - bit % 32));
                if (mode == 0) hb[lane * 8 + e / 8] |= (int)(vb << (4 * (e % 8)));
                else {
                    hb[lane * 8 + bit / 32] |= (int)((vb << (bit % 32)) & 0xffffffffu);
                    if (bit % 32 > 26) hb[lane * 8 + bit / 32 + 1] |= (int)(vb >> (32 - bit % 32));
                }
            }
        hipMemcpy(da, ha.data(), 2048, hipMemcpyHostToDevice);
        hipMemcpy(db, hb.data(), 2048, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dc, mode);
        hipMemcpy(hc.data(), dc, 4096, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int r = 0; r < 32; r++)
            for (int c = 0; c < 32; c++) {
                int s = 0;
                for (int kk = 0; kk < 64; kk++) s += xa[r * 64 + kk] * xb[c * 64 + kk];
                if (hc[r * 32 + c] != (float)s) { if (bad < 5) printf("mode %d (%d,%d): got %g want %d\n", mode, r, c, hc[r * 32 + c], s); bad++; }
            }
        printf("mode %d (%s): %d mismatches of 1024 (C[0][0] = %g)\n", mode, mode == 0 ? "fp6 x fp4" : "fp6 x fp6", bad, hc[0]);
    }
    float *dout; hipMalloc(&dout, 256 * 1024 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto time = [&](auto kern, const char *name) {
        const int iters = 2000;
        hipLaunchKernelGGL(kern, dim3(1024), dim3(256), 0, 0, dout, iters);
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(1024), dim3(256), 0, 0, dout, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double mfmas = 1024.0 * 4 * iters * 4;   // per wave 4 per iteration, 4 waves a workgroup
        printf("%s: %.3f ms, %.1f cycles per MFMA per SIMD at 2.4 GHz\n", name, ms, ms * 1e-3 * 2.4e9 / (mfmas / 1024.0));
    };
    time(rate<4, 4>, "fp4 x fp4");
    time(rate<2, 4>, "fp6 x fp4");
    time(rate<2, 2>, "fp6 x fp6");
    time(rate<0, 0>, "fp8 x fp8");
    return 0;
}
