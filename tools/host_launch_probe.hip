// tools/host_launch_probe.hip — what the HOST pays per launch of an 11-argument kernel (the headline kernel's signature) and what
// 20 back-to-back launches + one synchronise take on the wall clock (the driver's `bench.py --steps 20`), four ways:
//   A  hipLaunchKernelGGL (what the library does)            B  hipModuleLaunchKernel with a cached hipFunction_t and a packed argument buffer
//   C  hipExtModuleLaunchKernel (same arguments)             D  one hipGraphLaunch of 20 captured kernel nodes
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/host_launch_probe tools/host_launch_probe.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
#include <algorithm>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s\n", hipGetErrorString(e), #x); return 1; } } while (0)

__global__ __launch_bounds__(512) void k_probe(const uint32_t *X, const uint32_t *W, void *out, uint32_t xb, uint32_t wb, uint32_t ob, int M, int K, int N, int w_lines,
                                               uint32_t cfg) {
    // ~2.9 us of "work" per workgroup (256 x 512 threads, like the headline launch)
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < static_cast<unsigned long long>(cfg)) __builtin_amdgcn_s_sleep(1);
    if (threadIdx.x == 9999 && out) static_cast<uint32_t *>(out)[0] = X[0] + W[0] + xb + wb + ob + M + K + N + w_lines;
}

struct Args { const uint32_t *X; const uint32_t *W; void *out; uint32_t xb, wb, ob; int M, K, N, w_lines; uint32_t cfg; };

int main(int argc, char **argv) {
    const uint32_t ticks = argc > 1 ? atoi(argv[1]) : 130;   // s_memtime ticks (100 MHz): 130 = 1.3 us of body + ~1.6 us launch gap
    uint32_t *d;
    CK(hipMalloc(&d, 4096));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipFunction_t fn;
    CK(hipGetFuncBySymbol(&fn, reinterpret_cast<const void *>(&k_probe)));
    Args a{d, d, d, 1, 2, 3, 4096, 4096, 64, 128, ticks};
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto us = [](auto t0, auto t1) { return std::chrono::duration<double, std::micro>(t1 - t0).count(); };
    auto launch_a = [&]() { hipLaunchKernelGGL(k_probe, dim3(256), dim3(512), 0, st, a.X, a.W, a.out, a.xb, a.wb, a.ob, a.M, a.K, a.N, a.w_lines, a.cfg); };
    size_t sz = sizeof(Args);
    void *extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &a, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
    auto launch_b = [&]() { (void)hipModuleLaunchKernel(fn, 256, 1, 1, 512, 1, 1, 0, st, nullptr, extra); };
    auto launch_c = [&]() { (void)hipExtModuleLaunchKernel(fn, 256 * 512, 1, 1, 512, 1, 1, 0, st, nullptr, extra, nullptr, nullptr, 0); };
    hipGraph_t graph;
    hipGraphExec_t exec;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int i = 0; i < 20; i++) launch_a();
    CK(hipStreamEndCapture(st, &graph));
    CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    for (int i = 0; i < 200; i++) { launch_a(); launch_b(); launch_c(); }
    CK(hipGraphLaunch(exec, st));
    CK(hipStreamSynchronize(st));
    CK(hipGetLastError());
    const char *names[4] = {"A hipLaunchKernelGGL      ", "B hipModuleLaunchKernel  ", "C hipExtModuleLaunchKernel", "D hipGraphLaunch(20 nodes)"};
    for (int round = 0; round < 2; round++)
        for (int m = 0; m < 4; m++) {
            std::vector<double> issue, wall;
            for (int rep = 0; rep < 200; rep++) {
                CK(hipStreamSynchronize(st));
                auto t0 = now();
                if (m == 3) { (void)hipGraphLaunch(exec, st); }
                else for (int i = 0; i < 20; i++) { if (m == 0) launch_a(); else if (m == 1) launch_b(); else launch_c(); }
                auto t1 = now();
                CK(hipStreamSynchronize(st));
                auto t2 = now();
                issue.push_back(us(t0, t1));
                wall.push_back(us(t0, t2));
            }
            std::sort(issue.begin(), issue.end());
            std::sort(wall.begin(), wall.end());
            printf("%s  host issue of 20: median %.1f us (%.2f per launch)   issue + synchronise: median %.1f us, min %.1f\n", names[m], issue[100], issue[100] / 20, wall[100], wall[0]);
        }
    return 0;
}
