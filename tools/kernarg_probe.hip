// Host cost of hipLaunchKernel against the size of the kernel-argument segment: a kernel that reads gridDim carries the 256 bytes of
// implicit arguments behind its own (code object v5), one that does not is 56 bytes here. Prints microseconds of host time per launch
// (200 launches issued without waiting) and the device-side time per launch of the same 200.
//   hipcc --offload-arch=gfx950 -O3 -o kernarg_probe kernarg_probe.hip && ./kernarg_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>

__global__ __launch_bounds__(256) void k_lean(const uint32_t *a, const uint32_t *b, uint32_t *o, uint32_t x, uint32_t y, uint32_t z, int M, int K, int N, int L, uint32_t cfg) {
    if (threadIdx.x == 0 && blockIdx.x == 0 && cfg == 77u) o[0] = a[0] + b[0] + x + y + z + M + K + N + L;
}
__global__ __launch_bounds__(256) void k_implicit(const uint32_t *a, const uint32_t *b, uint32_t *o, uint32_t x, uint32_t y, uint32_t z, int M, int K, int N, int L, uint32_t cfg) {
    if (threadIdx.x == 0 && blockIdx.x == gridDim.x - 1 && cfg == 77u) o[0] = a[0] + b[0] + x + y + z + M + K + N + L;
}

template <class F>
static void run(const char *name, F launch) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 2000; i++) launch();
    hipDeviceSynchronize();
    double best_host = 1e9, best_dev = 1e9;
    for (int rep = 0; rep < 9; rep++) {
        hipEventRecord(e0, 0);
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < 200; i++) launch();
        auto t1 = std::chrono::steady_clock::now();
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double h = std::chrono::duration<double, std::micro>(t1 - t0).count() / 200.0;
        if (h < best_host) best_host = h;
        if (ms * 1000.0 / 200.0 < best_dev) best_dev = ms * 1000.0 / 200.0;
    }
    printf("%-28s host %.3f us / launch, device %.3f us / launch\n", name, best_host, best_dev);
}

int main() {
    uint32_t *a, *b, *o;
    hipMalloc(&a, 4096);
    hipMalloc(&b, 4096);
    hipMalloc(&o, 4096);
    for (int grid : {1, 2048}) {
        printf("grid %d\n", grid);
        run("lean (56-byte kernarg)", [&] { hipLaunchKernelGGL(k_lean, dim3(grid), dim3(256), 0, 0, a, b, o, 1u, 2u, 3u, 4, 5, 6, 7, 0u); });
        run("implicit (312-byte kernarg)", [&] { hipLaunchKernelGGL(k_implicit, dim3(grid), dim3(256), 0, 0, a, b, o, 1u, 2u, 3u, 4, 5, 6, 7, 0u); });
        run("lean (56-byte kernarg)", [&] { hipLaunchKernelGGL(k_lean, dim3(grid), dim3(256), 0, 0, a, b, o, 1u, 2u, 3u, 4, 5, 6, 7, 0u); });
        // the same kernel through the module API: function handle resolved once, arguments handed over as ONE packed buffer
        hipFunction_t fn;
        if (hipGetFuncBySymbol(&fn, reinterpret_cast<const void *>(k_lean)) == hipSuccess) {
            struct __attribute__((packed)) Args { const uint32_t *a, *b; uint32_t *o; uint32_t x, y, z; int M, K, N, L; uint32_t cfg; } args{a, b, o, 1u, 2u, 3u, 4, 5, 6, 7, 0u};
            size_t size = sizeof(args);
            void *extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
            run("lean, hipModuleLaunchKernel", [&] { hipModuleLaunchKernel(fn, grid, 1, 1, 256, 1, 1, 0, 0, nullptr, extra); });
            void *params[] = {&args.a, &args.b, &args.o, &args.x, &args.y, &args.z, &args.M, &args.K, &args.N, &args.L, &args.cfg};
            run("lean, module API, param array", [&] { hipModuleLaunchKernel(fn, grid, 1, 1, 256, 1, 1, 0, 0, params, nullptr); });
        } else {
            printf("hipGetFuncBySymbol failed\n");
        }
    }
    return 0;
}
