// tools/fp4_probe.hip — what v_mfma_scale_f32_32x32x64_f8f6f4 does with E2M1 (fp4) operands:
//   (1) codes 0..3 are 0, 0.5, 1.0, 1.5 (so a 2-bit value v stored as the nibble v means v / 2),
//   (2) E8M0 scale 128 on both operands (x2 each) makes the sum the integer product,
//   (3) nibble i of lane (l & 31, l >> 5) of A meets nibble i of the same lane position of B (the k order is
//       the same for both operands, which is all the bit-GEMM needs).
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/fp4_probe tools/fp4_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void k(const int *a, const int *b, float *c, int scale) {
    const int lane = threadIdx.x;
    i32x8 av, bv;
    for (int i = 0; i < 8; i++) { av[i] = i < 4 ? a[lane * 4 + i] : 0; bv[i] = i < 4 ? b[lane * 4 + i] : 0; }
    f32x16 acc;
    for (int i = 0; i < 16; i++) acc[i] = 0.f;
    if (scale == 127) acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, acc, 4, 4, 0, 127, 0, 127);
    else acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, acc, 4, 4, 0, 128, 0, 128);
    for (int r = 0; r < 16; r++) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), col = lane & 31;
        c[row * 32 + col] = acc[r];
    }
}

int main() {
    std::vector<int> ha(256), hb(256);
    std::vector<float> hc(1024);
    int *da, *db; float *dc;
    hipMalloc(&da, 1024); hipMalloc(&db, 1024); hipMalloc(&dc, 4096);
    auto run = [&](int scale) {
        hipMemcpy(da, ha.data(), 1024, hipMemcpyHostToDevice);
        hipMemcpy(db, hb.data(), 1024, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dc, scale);
        hipMemcpy(hc.data(), dc, 4096, hipMemcpyDeviceToHost);
    };
    // (1)+(2): random 2-bit values, compare with the integer product
    srand(1);
    std::vector<int> xa(32 * 64), xb(32 * 64);   // [line][k], k = 32 * (lane >> 5) + nibble index
    for (auto &v : xa) v = rand() & 3;
    for (auto &v : xb) v = rand() & 3;
    for (int lane = 0; lane < 64; lane++)
        for (int d = 0; d < 4; d++) {
            unsigned wa = 0, wb = 0;
            for (int n = 0; n < 8; n++) {
                const int kk = 32 * (lane >> 5) + 8 * d + n;
                wa |= (unsigned)xa[(lane & 31) * 64 + kk] << (4 * n);
                wb |= (unsigned)xb[(lane & 31) * 64 + kk] << (4 * n);
            }
            ha[lane * 4 + d] = (int)wa; hb[lane * 4 + d] = (int)wb;
        }
    for (int scale : {127, 128}) {
        run(scale);
        int bad = 0;
        for (int r = 0; r < 32; r++)
            for (int c = 0; c < 32; c++) {
                int s = 0;
                for (int kk = 0; kk < 64; kk++) s += xa[r * 64 + kk] * xb[c * 64 + kk];
                const float want = scale == 127 ? s / 4.0f : (float)s;
                if (hc[r * 32 + c] != want) { if (bad < 5) printf("scale %d (%d,%d): got %g want %g\n", scale, r, c, hc[r * 32 + c], want); bad++; }
            }
        printf("scale %d: %d mismatches of 1024 (C[0][0] = %g)\n", scale, bad, hc[0]);
    }
    return 0;
}
