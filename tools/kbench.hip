// tools/kbench.hip — standalone (no torch) kernel bench + phase-stamp dump for the bit-GEMM.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude [-DQGTC_STAMPS] -o /tmp/kbench tools/kbench.hip
//   /tmp/kbench M K N a w ob reps [density]
#include "../qgtc_ppopp22_amd/csrc/qgtc_hip.hip"

#include <cstdlib>
#include <random>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s\n", hipGetErrorString(e), #x); exit(1);} } while (0)

int main(int argc, char **argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 4096, K = argc > 2 ? atoi(argv[2]) : 4096;
    const int N = argc > 3 ? atoi(argv[3]) : 64, a = argc > 4 ? atoi(argv[4]) : 1;
    const int w = argc > 5 ? atoi(argv[5]) : 1, ob = argc > 6 ? atoi(argv[6]) : w;
    const int reps = argc > 7 ? atoi(argv[7]) : 200;
    const double density = argc > 8 ? atof(argv[8]) : 0.5;
    const size_t xw = qgtc_rows_words(M, K, a), ww = qgtc_cols_words(K, N, w, 0), ow = qgtc_rows_words(M, N, ob);
    std::mt19937 rng(3);
    std::vector<uint32_t> hx(xw), hw(ww);
    std::bernoulli_distribution bern(density);
    for (auto &v : hx) { uint32_t t = 0; for (int b = 0; b < 32; b++) t |= (uint32_t)bern(rng) << b; v = t; }
    for (auto &v : hw) v = rng();
    uint32_t *dx, *dw, *dout;
    CK(hipMalloc(&dx, xw * 4)); CK(hipMalloc(&dw, ww * 4)); CK(hipMalloc(&dout, ow * 4));
    CK(hipMemcpy(dx, hx.data(), xw * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dw, hw.data(), ww * 4, hipMemcpyHostToDevice));
    const unsigned flags = (getenv("NOZS") ? QGTC_NO_ZERO_SKIP : 0u) | (getenv("MFMA") ? QGTC_ENGINE_MFMA : 0u);
    float ms = 0, best = 1e30f;
    for (int r = 0; r < 5; r++) {
        int rc = qgtc_bitmm2bit_profile(dx, xw, dw, ww, M, K, N, a, w, ob, dout, ow, flags, reps, &ms, nullptr);
        if (rc) { printf("rc=%d %s %s\n", rc, qgtc_strerror(rc), qgtc_last_hip_error()); return 1; }
        if (ms < best) best = ms;
    }
    const double us = best * 1e3 / reps;
    printf("%dx%dx%d a=%d w=%d: %.2f us/launch  eff %.1f TOPS  valu-frac %.3f\n", M, K, N, a, w, us,
           2.0 * M * K * N / us / 1e6, 2.0 * M * K * N * a * w / (us * 1e-6) / 2.516e15);
#ifdef QGTC_STAMPS
    std::vector<unsigned long long> st(1024 * 16);
    CK(hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_stamps), st.size() * 8));
    for (int b : {0, 1, 100, 255, 600, 900, 1023}) {
        printf("block %3d:", b);
        unsigned long long t0 = st[b * 16];
        for (int s = 0; s < 16; s++) {
            unsigned long long t = st[b * 16 + s];
            if (t >= t0 && t - t0 < 100000000ull) printf(" [%d]%llu", s, t - t0);
            else if (s >= 8 && t && st[b * 16 + 8]) printf(" [%d]%lld", s, (long long)(t - st[b * 16 + 8]));
        }
        printf("\n");
    }
#endif
    return 0;
}
