// tools/rbw_bench.hip — standalone (no torch) timing of the chain entries (bitmm_fp4_rbw.hip.h) on cluster-batch-like operands:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -mllvm -amdgpu-kernarg-preload-count=16 [-DQGTC_RBW_STAMPS] -o tools/rbw_bench tools/rbw_bench.hip
//   tools/rbw_bench [count=75] [n=1213] [N1=128] [N2=128] [mode2=1] [extra_prob=0.03]
// 200 launches between two events (best of 5), the occupied-tile statistics of the synthetic adjacency, and with
// -DQGTC_RBW_STAMPS the s_memtime stamps of wave 0 of the first 1024 workgroups.
#define QGTC_SINGLE_TU 1
#include "../qgtc_ppopp22_amd/csrc/qgtc_hip.hip"

#include <cstdlib>
#include <random>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s\n", hipGetErrorString(e), #x); exit(1);} } while (0)

int main(int argc, char **argv) {
    const int count = argc > 1 ? atoi(argv[1]) : 75, n = argc > 2 ? atoi(argv[2]) : 1213;
    const int N1 = argc > 3 ? atoi(argv[3]) : 128, N2 = argc > 4 ? atoi(argv[4]) : 128, mode2 = argc > 5 ? atoi(argv[5]) : 1;
    const double extra = argc > 6 ? atof(argv[6]) : 0.03;
    std::mt19937 rng(3);
    const size_t aw = qgtc_rows_words(n, n, 1), tw = qgtc_chain_words(n, N1), t2w = mode2 == 1 ? qgtc_chain_words(n, N2) : (size_t)n * (mode2 == 0 ? N1 : N2);
    const int rw = (n + 127) / 128 * 4;
    std::vector<uint32_t> ha(aw, 0u);
    std::bernoulli_distribution far(extra);
    for (int r = 0; r < n; r++) {
        const int q = r / 128;
        for (int k = 0; k < 6; k++) {   // ~6 neighbours inside the row's own k-quad (the planted block)
            const int c = std::min(n - 1, q * 128 + (int)(rng() % 128));
            ha[(size_t)r * rw + c / 32] |= 1u << (31 - c % 32);
        }
        if (far(rng)) {
            const int c = rng() % n;
            ha[(size_t)r * rw + c / 32] |= 1u << (31 - c % 32);
        }
    }
    std::vector<uint32_t> ht(tw);
    for (auto &v : ht) v = rng() & 0x33333333u;
    uint32_t *dA, *dT, *dT2, *dW, *dWc;
    uint64_t *docc;
    const size_t occw = qgtc_occupancy_words(n, n);
    CK(hipMalloc(&dA, aw * 4 * count)); CK(hipMalloc(&dT, tw * 4 * count)); CK(hipMalloc(&dT2, t2w * 4 * count)); CK(hipMalloc(&docc, occw * 8 * count));
    for (int b = 0; b < count; b++) {
        CK(hipMemcpy(dA + aw * b, ha.data(), aw * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dT + tw * b, ht.data(), tw * 4, hipMemcpyHostToDevice));
        if (int rc = qgtc_tile_occupancy(dA + aw * b, aw, n, n, 1, docc + occw * b, occw, nullptr)) { printf("occ rc=%d\n", rc); return 1; }
    }
    std::vector<uint64_t> hocc(occw);
    CK(hipMemcpy(hocc.data(), docc, occw * 8, hipMemcpyDeviceToHost));
    size_t set = 0;
    for (auto v : hocc) set += __builtin_popcountll(v);
    printf("adjacency: %zu of %zu 32-row x 128-bit tiles occupied (%.3f), %.2f k-quads per row block\n", set, occw * ((n + 127) / 128), (double)set / (occw * ((n + 127) / 128)), (double)set / occw);
    const size_t ww = qgtc_cols_words(N1, N2, 2, 0), wcw = qgtc_weight_codes_words(N2, 2);
    std::vector<uint32_t> hw(ww);
    for (auto &v : hw) v = rng();
    CK(hipMalloc(&dW, ww * 4)); CK(hipMalloc(&dWc, wcw * 4));
    CK(hipMemcpy(dW, hw.data(), ww * 4, hipMemcpyHostToDevice));
    qgtc_expand_job ej{dW, dWc, ww, N1, N2, 2, (N2 + 127) / 128 * 128, 1, 0};
    if (int rc = qgtc_expand_weights(&ej, 1, nullptr)) { printf("expand rc=%d\n", rc); return 1; }
    std::vector<qgtc_problem> h1(count), h2(count);
    for (int b = 0; b < count; b++) {
        h1[b] = qgtc_problem{dA + aw * b, dT + tw * b, mode2 == 0 ? (void *)(dT2 + t2w * b) : nullptr, aw, tw, n, n, N1, 128, 1, getenv("NOOCC") ? nullptr : docc + occw * b};
        h2[b] = qgtc_problem{nullptr, dW, dT2 + t2w * b, 0, ww, n, N1, N2, 128, 0, nullptr};
    }
    qgtc_problem *d1, *d2;
    CK(hipMalloc(&d1, count * sizeof(qgtc_problem))); CK(hipMalloc(&d2, count * sizeof(qgtc_problem)));
    CK(hipMemcpy(d1, h1.data(), count * sizeof(qgtc_problem), hipMemcpyHostToDevice));
    CK(hipMemcpy(d2, h2.data(), count * sizeof(qgtc_problem), hipMemcpyHostToDevice));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto go = [&]() { return qgtc_chain_aggregate(d1, mode2 == 0 ? nullptr : d2, count, n, n, N1, N2, 2, 2, 2, mode2, dWc, 0, st); };
    if (int rc = go()) { printf("rc=%d %s\n", rc, qgtc_strerror(rc)); return 1; }
    CK(hipStreamSynchronize(st));
    float best = 1e9f;
    for (int rep = 0; rep < 5; rep++) {
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < 200; i++) go();
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms);
    }
    printf("chain aggregate count=%d n=%d N1=%d N2=%d mode2=%d: %.2f us per launch (200 eager launches, best of 5)\n", count, n, N1, N2, mode2, best * 1e3 / 200);
#ifdef QGTC_RBW_STAMPS
    {
        go();
        CK(hipStreamSynchronize(st));
        std::vector<unsigned long long> hs(1024 * 16);
        CK(hipMemcpyFromSymbol(hs.data(), HIP_SYMBOL(g_stamps), hs.size() * 8));
        double sum[12] = {0};
        int cnt = 0;
        unsigned long long first = ~0ull, last = 0;
        for (int s = 0; s < 1024; s++) {
            const unsigned long long *p = &hs[s * 16];
            if (!p[0] || !p[9]) continue;
            cnt++;
            for (int i = 1; i < 10; i++) sum[i] += (double)(p[i] - p[0]);
            first = std::min(first, p[0]);
            last = std::max(last, p[9]);
        }
        printf("stamps (mean ticks from the wave's start over %d waves; 100 MHz ticks = 10 ns): desc %.0f occ %.0f loads-issued %.0f data %.0f product1 %.0f epi1 %.0f product2 %.0f epi2 %.0f end %.0f | first start -> last end %.2f us\n",
               cnt, sum[1] / cnt, sum[2] / cnt, sum[3] / cnt, sum[4] / cnt, sum[5] / cnt, sum[6] / cnt, sum[7] / cnt, sum[8] / cnt, sum[9] / cnt, (last - first) * 0.01);
    }
#endif
    return 0;
}
