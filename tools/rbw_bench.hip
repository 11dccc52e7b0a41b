// tools/rbw_bench.hip — standalone (no torch) timing of the chain entries (bitmm_fp4_rbw.hip.h) on cluster-batch-like operands:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -mllvm -amdgpu-kernarg-preload-count=16 [-DQGTC_RBW_STAMPS] -o tools/rbw_bench tools/rbw_bench.hip
//   [BITS=4] [TILES=1] [SPLIT=parts] tools/rbw_bench [count=75] [n=1213] [N1=128] [N2=128] [mode2=1] [extra_prob=0.03]
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
    const int bits = getenv("BITS") ? atoi(getenv("BITS")) : 2;   // 2 (N1, N2 <= 128) or 4 (<= 64): the widths of the two epochs
    std::mt19937 rng(3);
    const size_t aw = qgtc_rows_words(n, n, 1), tw = qgtc_chain_words(n, N1, bits), t2w = mode2 == 1 ? qgtc_chain_words(n, N2, bits) : (size_t)n * (mode2 == 0 ? N1 : N2);
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
    for (auto &v : ht) v = rng() & (bits == 2 ? 0x33333333u : 0xffffffffu);
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
    const size_t ww = qgtc_cols_words(N1, N2, bits, 0), wcw = qgtc_weight_codes_words(N1, N2, bits, 1);
    std::vector<uint32_t> hw(ww);
    for (auto &v : hw) v = rng();
    CK(hipMalloc(&dW, ww * 4)); CK(hipMalloc(&dWc, wcw * 4));
    CK(hipMemcpy(dW, hw.data(), ww * 4, hipMemcpyHostToDevice));
    qgtc_expand_job ej{dW, dWc, ww, N1, N2, bits, (N2 + 127) / 128 * 128, 1, static_cast<uint32_t>(wcw)};
    if (int rc = qgtc_expand_weights(&ej, 1, nullptr)) { printf("expand rc=%d\n", rc); return 1; }
    const bool tiles = getenv("TILES") != nullptr;   // the adjacency as 512-byte tiles (qgtc_adj_tiles_from_rows + QGTC_CHAIN_ADJ_TILES)
    const size_t atw = qgtc_adj_tiles_words(n, n);
    uint32_t *dAT = nullptr;
    if (tiles) {
        CK(hipMalloc(&dAT, atw * 4 * count));
        for (int b = 0; b < count; b++)
            if (int rc = qgtc_adj_tiles_from_rows(dA + aw * b, aw, n, n, dAT + atw * b, atw, nullptr)) { printf("tiles rc=%d\n", rc); return 1; }
        CK(hipDeviceSynchronize());
    }
    std::vector<qgtc_problem> h1(count), h2(count);
    for (int b = 0; b < count; b++) {
        h1[b] = qgtc_problem{tiles ? dAT + atw * b : dA + aw * b, dT + tw * b, mode2 == 0 ? (void *)(dT2 + t2w * b) : nullptr, tiles ? atw : aw, tw, n, n, N1, 128, 1, getenv("NOOCC") ? nullptr : docc + occw * b};
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
    auto go = [&]() { return qgtc_chain_aggregate(d1, mode2 == 0 ? nullptr : d2, count, n, n, N1, N2, bits, bits, bits, mode2, dWc, tiles ? QGTC_CHAIN_ADJ_TILES : 0u, st); };
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
    if (getenv("SPLIT")) {   // the same launch as SPLIT parts on as many streams, 200 rounds captured into ONE graph (no host in the loop):
                             // do the parts hide each other's launch boundaries and latency chains?
        const int parts = std::max(2, atoi(getenv("SPLIT")));
        std::vector<hipStream_t> ss(parts);
        std::vector<hipEvent_t> done(parts);
        for (auto &x : ss) CK(hipStreamCreate(&x));
        for (auto &x : done) CK(hipEventCreate(&x));
        hipEvent_t fork;
        CK(hipEventCreate(&fork));
        auto part = [&](int p, hipStream_t s_) {
            const int b0 = count * p / parts, b1 = count * (p + 1) / parts;
            return qgtc_chain_aggregate(d1 + b0, mode2 == 0 ? nullptr : d2 + b0, b1 - b0, n, n, N1, N2, bits, bits, bits, mode2, dWc, tiles ? QGTC_CHAIN_ADJ_TILES : 0u, s_);
        };
        hipGraph_t graph;
        hipGraphExec_t exec;
        CK(hipStreamBeginCapture(ss[0], hipStreamCaptureModeGlobal));
        CK(hipEventRecord(fork, ss[0]));
        for (int p = 1; p < parts; p++) CK(hipStreamWaitEvent(ss[p], fork, 0));
        for (int i = 0; i < 200; i++)
            for (int p = 0; p < parts; p++)
                if (int rc = part(p, ss[p])) { printf("rc=%d\n", rc); return 1; }
        for (int p = 1; p < parts; p++) { CK(hipEventRecord(done[p], ss[p])); CK(hipStreamWaitEvent(ss[0], done[p], 0)); }
        CK(hipStreamEndCapture(ss[0], &graph));
        CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        float bs = 1e9f;
        for (int rep = 0; rep < 5; rep++) {
            CK(hipEventRecord(e0, ss[0]));
            CK(hipGraphLaunch(exec, ss[0]));
            CK(hipEventRecord(e1, ss[0]));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            bs = std::min(bs, ms);
        }
        // and the unsplit launch replayed the same way, for the like-for-like figure
        hipGraph_t g1;
        hipGraphExec_t x1;
        CK(hipStreamBeginCapture(ss[0], hipStreamCaptureModeGlobal));
        for (int i = 0; i < 200; i++)
            if (int rc = qgtc_chain_aggregate(d1, mode2 == 0 ? nullptr : d2, count, n, n, N1, N2, bits, bits, bits, mode2, dWc, tiles ? QGTC_CHAIN_ADJ_TILES : 0u, ss[0])) { printf("rc=%d\n", rc); return 1; }
        CK(hipStreamEndCapture(ss[0], &g1));
        CK(hipGraphInstantiate(&x1, g1, nullptr, nullptr, 0));
        float b1 = 1e9f;
        for (int rep = 0; rep < 5; rep++) {
            CK(hipEventRecord(e0, ss[0]));
            CK(hipGraphLaunch(x1, ss[0]));
            CK(hipEventRecord(e1, ss[0]));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            b1 = std::min(b1, ms);
        }
        printf("as %d parts on %d streams (graph replay): %.2f us per round of the whole launch's work; unsplit, replayed the same way: %.2f us\n", parts, parts, bs * 1e3 / 200, b1 * 1e3 / 200);
    }
    if (getenv("EPOCH")) {   // the Cluster-GCN epoch (four chain-entry launches) on the same synthetic batches, F = H = 128, C = 10
        const int F = 128, H = 128, C = 10;
        const size_t xw_ = qgtc_rows_words(n, F, 2), th = qgtc_chain_words(n, H, bits), tc = qgtc_chain_words(n, C, bits);
        uint32_t *dX, *dT1, *dT2, *dT3, *dW1, *dW2, *dW3, *c1, *c2, *c3;
        float *dout;
        CK(hipMalloc(&dX, xw_ * 4 * count)); CK(hipMalloc(&dT1, th * 4 * count)); CK(hipMalloc(&dT2, th * 4 * count)); CK(hipMalloc(&dT3, tc * 4 * count));
        CK(hipMalloc(&dout, (size_t)n * C * 4 * count));
        std::vector<uint32_t> hx(xw_);
        for (auto &v : hx) v = rng();
        for (int b = 0; b < count; b++) CK(hipMemcpy(dX + xw_ * b, hx.data(), xw_ * 4, hipMemcpyHostToDevice));
        const size_t w1 = qgtc_cols_words(F, H, 2, 0), w2 = qgtc_cols_words(H, H, 2, 0), w3 = qgtc_cols_words(H, C, 2, 0);
        std::vector<uint32_t> hw1(w1), hw2(w2), hw3(w3);
        for (auto &v : hw1) v = rng();
        for (auto &v : hw2) v = rng();
        for (auto &v : hw3) v = rng();
        CK(hipMalloc(&dW1, w1 * 4)); CK(hipMalloc(&dW2, w2 * 4)); CK(hipMalloc(&dW3, w3 * 4));
        CK(hipMemcpy(dW1, hw1.data(), w1 * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dW2, hw2.data(), w2 * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dW3, hw3.data(), w3 * 4, hipMemcpyHostToDevice));
        const size_t k1 = qgtc_weight_codes_words(F, H, 2, 0), k2 = qgtc_weight_codes_words(H, H, 2, 1), k3 = qgtc_weight_codes_words(H, C, 2, 1);
        CK(hipMalloc(&c1, k1 * 4)); CK(hipMalloc(&c2, k2 * 4)); CK(hipMalloc(&c3, k3 * 4));
        qgtc_expand_job ej3[3] = {{dW1, c1, w1, F, H, 2, 128, 0, (uint32_t)k1}, {dW2, c2, w2, H, H, 2, 128, 1, (uint32_t)k2}, {dW3, c3, w3, H, C, 2, 128, 1, (uint32_t)k3}};
        if (int rc = qgtc_expand_weights(ej3, 3, nullptr)) { printf("expand rc=%d\n", rc); return 1; }
        std::vector<qgtc_problem> hs(6 * count);
        for (int b = 0; b < count; b++) {
            const qgtc_problem A1{dA + aw * b, dT1 + th * b, nullptr, aw, th, n, n, H, 128, 1, getenv("NOOCC") ? nullptr : docc + occw * b};
            qgtc_problem A2 = A1, A3 = A1;
            A2.W = dT2 + th * b;
            A3.W = dT3 + tc * b; A3.w_words = tc; A3.N = C; A3.out = dout + (size_t)n * C * b;
            hs[0 * count + b] = qgtc_problem{dX + xw_ * b, dW1, dT1 + th * b, xw_, w1, n, F, H, 128, 0, nullptr};
            hs[1 * count + b] = A1;
            hs[2 * count + b] = qgtc_problem{nullptr, dW2, dT2 + th * b, 0, w2, n, H, H, 128, 0, nullptr};
            hs[3 * count + b] = A2;
            hs[4 * count + b] = qgtc_problem{nullptr, dW3, dT3 + tc * b, 0, w3, n, H, C, 128, 0, nullptr};
            hs[5 * count + b] = A3;
        }
        qgtc_problem *ds;
        CK(hipMalloc(&ds, hs.size() * sizeof(qgtc_problem)));
        CK(hipMemcpy(ds, hs.data(), hs.size() * sizeof(qgtc_problem), hipMemcpyHostToDevice));
        const qgtc_problem *st6[6];
        for (int i = 0; i < 6; i++) st6[i] = ds + (size_t)i * count;
        auto four = [&]() {
            int rc = qgtc_chain_transform(st6[0], count, n, F, H, 2, 2, c1, 0, st);
            if (!rc) rc = qgtc_chain_aggregate(st6[1], st6[2], count, n, n, H, H, 2, 2, 2, 1, c2, 0, st);
            if (!rc) rc = qgtc_chain_aggregate(st6[3], st6[4], count, n, n, H, C, 2, 2, 2, 1, c3, 0, st);
            if (!rc) rc = qgtc_chain_aggregate(st6[5], nullptr, count, n, n, C, 0, 2, 0, 0, 0, nullptr, 0, st);
            return rc;
        };
        {   // the transform launch (X . W1) on its own
            float bx = 1e9f;
            for (int rep = 0; rep < 5; rep++) {
                CK(hipEventRecord(e0, st));
                for (int i = 0; i < 200; i++) qgtc_chain_transform(st6[0], count, n, F, H, 2, 2, c1, 0, st);
                CK(hipEventRecord(e1, st));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                bx = std::min(bx, ms);
            }
            printf("chain transform X.W1 alone: %.2f us per launch (200 eager launches, best of 5)\n", bx * 1e3 / 200);
        }
        for (int variant = 0; variant < 1; variant++) {
            if (int rc = four()) { printf("epoch variant %d rc=%d %s\n", variant, rc, qgtc_strerror(rc)); continue; }
            CK(hipStreamSynchronize(st));
            float best2 = 1e9f;
            for (int rep = 0; rep < 5; rep++) {
                CK(hipEventRecord(e0, st));
                for (int i = 0; i < 100; i++) four();
                CK(hipEventRecord(e1, st));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                best2 = std::min(best2, ms);
            }
            printf("Cluster-GCN epoch (four launches): %.2f us per epoch (100 epochs from a C loop, best of 5)\n", best2 * 1e3 / 100);
        }
    }
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
        {   // when the recorded waves started and how long they ran (the dispatch ramp and the per-wave chain)
            double s0 = 0, d = 0;
            unsigned long long s0max = 0, dmax = 0;
            for (int s = 0; s < 1024; s++) {
                const unsigned long long *p = &hs[s * 16];
                if (!p[0] || !p[9]) continue;
                s0 += (double)(p[0] - first); d += (double)(p[9] - p[0]);
                s0max = std::max(s0max, p[0] - first); dmax = std::max(dmax, p[9] - p[0]);
            }
            printf("wave starts after the first: mean %.2f us, last %.2f us; a wave runs: mean %.2f us, longest %.2f us\n", s0 / cnt * 0.01, s0max * 0.01, d / cnt * 0.01, dmax * 0.01);
        }
        printf("stamps (mean ticks from the wave's start over %d waves; 100 MHz ticks = 10 ns): desc %.0f occ %.0f loads-issued %.0f data %.0f product1 %.0f epi1 %.0f product2 %.0f epi2 %.0f end %.0f | first start -> last end %.2f us\n",
               cnt, sum[1] / cnt, sum[2] / cnt, sum[3] / cnt, sum[4] / cnt, sum[5] / cnt, sum[6] / cnt, sum[7] / cnt, sum[8] / cnt, sum[9] / cnt, (last - first) * 0.01);
    }
#endif
    return 0;
}
