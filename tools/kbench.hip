// tools/kbench.hip — standalone (no torch) kernel bench + phase-stamp dump for the bit-GEMM.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude [-DQGTC_STAMPS] -o /tmp/kbench tools/kbench.hip
//   /tmp/kbench M K N a w ob reps [density]
//   env: NOZS=1 (no zero-tile skipping), MFMA=1 (matrix-core engine), AUTO=1 (the default engine's rules), XPLANES=n / WPLANES=n (planes n .. are all zero),
//        GROUPED=count (count copies of the problem in one grouped launch; X gets dense 64 x 64
//        diagonal blocks + `density` elsewhere, like a cluster batch), JUMP=1 (with occupancy bitmaps),
//        MODE=0|1|2 (grouped only: rows bits, cols bits, float)
#define QGTC_SINGLE_TU 1   // pull the second translation unit (FP4 narrow-operand kernels) into this one
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
    // XPLANES / WPLANES: only that many low planes carry bits (the rest all zero) - what wide --bit_width operands look like (features
    // quantised at 32 bits are 0 .. 4: three of 32 planes; all-ones weights: one)
    if (const char *e = getenv("XPLANES")) std::fill(hx.begin() + (size_t)atoi(e) * (xw / a), hx.end(), 0u);
    if (const char *e = getenv("WPLANES")) std::fill(hw.begin() + (size_t)atoi(e) * (ww / w), hw.end(), 0u);
    uint32_t *dx, *dw, *dout;
    CK(hipMalloc(&dx, xw * 4)); CK(hipMalloc(&dw, ww * 4)); CK(hipMalloc(&dout, ow * 4));
    CK(hipMemcpy(dx, hx.data(), xw * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dw, hw.data(), ww * 4, hipMemcpyHostToDevice));
    const unsigned flags = (getenv("NOZS") ? QGTC_NO_ZERO_SKIP : 0u) | (getenv("MFMA") ? QGTC_ENGINE_MFMA : 0u) | (getenv("AUTO") ? QGTC_ENGINE_AUTO : 0u);
    float ms = 0, best = 1e30f;
    if (const char *g = getenv("GROUPED")) {
        const int count = atoi(g), mode = getenv("MODE") ? atoi(getenv("MODE")) : 0;
        const bool jump = getenv("JUMP") != nullptr;
        // cluster-batch-like left operand: dense diagonal blocks, `density` elsewhere
        std::fill(hx.begin(), hx.end(), 0u);
        const int rw_ = (K + 127) / 128 * 4, rp = (M + 7) / 8 * 8;
        for (int p = 0; p < a; p++)
            for (int r = 0; r < M; r++)
                for (int c = 0; c < K; c++) {
                    const bool in_blk = r / 64 == c / 64;
                    if (in_blk ? (rng() & 1) : bern(rng)) hx[(size_t)p * rp * rw_ + (size_t)r * rw_ + c / 32] |= 1u << (31 - c % 32);
                }
        CK(hipMemcpy(dx, hx.data(), xw * 4, hipMemcpyHostToDevice));
        const size_t out_bytes = mode == 2 ? (size_t)M * N * 4 : (mode == 1 ? qgtc_cols_words(M, N, ob, 0) : ow) * 4;
        std::vector<qgtc_problem> hp(count);
        uint64_t *docc = nullptr;
        const size_t occw = qgtc_occupancy_words(M, K);
        CK(hipMalloc(&docc, occw * 8));
        if (int rc = qgtc_tile_occupancy(dx, xw, M, K, a, docc, occw, nullptr)) { printf("occ rc=%d\n", rc); return 1; }
        std::vector<uint64_t> hocc(occw);
        CK(hipMemcpy(hocc.data(), docc, occw * 8, hipMemcpyDeviceToHost));
        size_t set = 0;
        for (auto v : hocc) set += __builtin_popcountll(v);
        size_t set128 = 0, all128 = 0;
        const int ow64 = ((K + 127) / 128 + 63) / 64, rts = (M + 31) / 32;
        for (int t = 0; t < (M + 127) / 128; t++)
            for (int wi = 0; wi < ow64; wi++) {
                uint64_t m = 0;
                for (int r = 0; r < 4; r++) if (4 * t + r < rts) m |= hocc[(size_t)(4 * t + r) * ow64 + wi];
                set128 += __builtin_popcountll(m);
            }
        all128 = (size_t)((M + 127) / 128) * ((K + 127) / 128);
        printf("occupied: %.3f of the 32-row tiles, %.3f of the 128-row tiles\n",
               (double)set / ((double)rts * ((K + 127) / 128)), (double)set128 / all128);
        for (int i = 0; i < count; i++) {
            void *o;
            CK(hipMalloc(&o, out_bytes));
            hp[i] = qgtc_problem{dx, dw, o, xw, ww, M, K, N, (N + 127) / 128 * 128, jump ? ow64 : 0, jump ? docc : nullptr};
        }
        qgtc_problem *dp;
        CK(hipMalloc(&dp, count * sizeof(qgtc_problem)));
        CK(hipMemcpy(dp, hp.data(), count * sizeof(qgtc_problem), hipMemcpyHostToDevice));
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        const unsigned gf = flags | (jump ? QGTC_ZERO_JUMP : 0u);
        for (int r = 0; r < 5; r++) {
            CK(hipEventRecord(e0, nullptr));
            for (int i = 0; i < reps; i++)
                if (int rc = qgtc_bitmm_batched(dp, count, M, K, N, a, w, ob, mode, gf, nullptr)) { printf("rc=%d\n", rc); return 1; }
            CK(hipEventRecord(e1, nullptr));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        printf("grouped x%d %dx%dx%d a=%d w=%d mode=%d%s%s: %.2f us per grouped launch\n", count, M, K, N, a, w, mode,
               jump ? " jump" : "", (flags & QGTC_ENGINE_MFMA) ? " mfma" : "", best * 1e3 / reps);
    } else
    for (int r = 0; r < 5; r++) {
        int rc = qgtc_bitmm2bit_profile(dx, xw, dw, ww, M, K, N, a, w, ob, dout, ow, flags, reps, &ms, nullptr);
        if (rc) { printf("rc=%d %s %s\n", rc, qgtc_strerror(rc), qgtc_last_hip_error()); return 1; }
        if (ms < best) best = ms;
    }
    const double us = best * 1e3 / reps;
    if (!getenv("GROUPED"))
    printf("%dx%dx%d a=%d w=%d: %.2f us/launch  eff %.1f TOPS  valu-frac %.3f\n", M, K, N, a, w, us,
           2.0 * M * K * N / us / 1e6, 2.0 * M * K * N * a * w / (us * 1e-6) / 2.516e15);
#ifdef QGTC_STEP_TRACE
    {
        std::vector<unsigned long long> tr(1024 * 16);
        CK(hipMemcpyFromSymbol(tr.data(), HIP_SYMBOL(g_stamps), tr.size() * 8));
        const unsigned long long t0 = tr[8192 + 1];
        for (int q = 0; q < 40 && tr[8192 + 2 * q]; q++)
            printf("step %2d  mult: mfma issued %6lld barrier passed %6lld | X exp: done %6lld passed %6lld | W exp: done %6lld passed %6lld\n", q,
                   (long long)(tr[8192 + 2 * q] - t0), (long long)(tr[8192 + 2 * q + 1] - t0),
                   (long long)(tr[8192 + 128 + 2 * q] - t0), (long long)(tr[8192 + 128 + 2 * q + 1] - t0),
                   (long long)(tr[8192 + 256 + 2 * q] - t0), (long long)(tr[8192 + 256 + 2 * q + 1] - t0));
    }
#endif
#ifdef QGTC_STAMPS
    std::vector<unsigned long long> st(1024 * 16);
    CK(hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_stamps), st.size() * 8));
    std::vector<int> slots = {0, 1, 100, 255, 600, 900, 1023};
    if (getenv("SLOTS16")) slots = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15};   // (kernels that record a slot per wave)
    for (int b : slots) {
        printf("block %3d:", b);
        unsigned long long t0 = st[b * 16];
        for (int s = 0; s < 16; s++) {
            unsigned long long t = st[b * 16 + s];
            if (s == 14 && st[b * 16 + 13] && t > st[b * 16 + 13]) printf(" [loop: %.2f us, %.3f GHz]", (t - st[b * 16 + 13]) * 0.01, (double)(st[b * 16 + 11] - t0) / ((t - st[b * 16 + 13]) * 10.0));
            else if (s == 13) continue;
            else if (t >= t0 && t - t0 < 100000000ull) printf(" [%d]%llu", s, t - t0);
            else if (s >= 8 && t && st[b * 16 + 8]) printf(" [%d]%lld", s, (long long)(t - st[b * 16 + 8]));
        }
        printf("\n");
    }
#endif
    return 0;
}
