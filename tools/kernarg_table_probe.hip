// tools/kernarg_table_probe.hip — how long a wave waits for a per-workgroup table line read (a) from a device table in global memory and
// (b) from a table passed BY VALUE in the kernel-argument segment (read through the segment pointer, dynamic index), at kernel
// start with cold caches: 200 launches, 1426 workgroups, the mean s_memtime ticks of the one dependent scalar load.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/kernarg_table_probe tools/kernarg_table_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s\n", hipGetErrorString(e), #x); return 1; } } while (0)

struct Table { unsigned v[75 * 8]; };   // 2400 bytes: 75 lines of 32 bytes

__global__ __launch_bounds__(256) void k_global(const unsigned *__restrict__ tab, unsigned long long *__restrict__ out, unsigned *__restrict__ sink) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const int line = blockIdx.x % 75;
    const unsigned a = tab[line * 8], b = tab[line * 8 + 7];
    asm volatile("" ::"s"(a), "s"(b));
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[blockIdx.x] = t1 - t0; sink[blockIdx.x] = a + b; }
}

__global__ __launch_bounds__(256) void k_kernarg(Table tab, unsigned long long *__restrict__ out, unsigned *__restrict__ sink) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const int line = blockIdx.x % 75;
    const __attribute__((address_space(4))) unsigned *p = (const __attribute__((address_space(4))) unsigned *)__builtin_amdgcn_kernarg_segment_ptr();   // (the by-value table is the first argument)
    const unsigned a = p[line * 8], b = p[line * 8 + 7];
    asm volatile("" ::"s"(a), "s"(b));
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[blockIdx.x] = t1 - t0; sink[blockIdx.x] = a + b + tab.v[0]; }
}

int main() {
    const int G = 1426;
    Table h;
    for (int i = 0; i < 600; i++) h.v[i] = i * 7;
    unsigned *dtab, *sink;
    unsigned long long *dout;
    CK(hipMalloc(&dtab, sizeof(h))); CK(hipMalloc(&sink, G * 4)); CK(hipMalloc(&dout, G * 8));
    CK(hipMemcpy(dtab, h.v, sizeof(h), hipMemcpyHostToDevice));
    std::vector<unsigned long long> ho(G);
    for (int which = 0; which < 2; which++) {
        double sum = 0, mx = 0;
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int rep = 0; rep < 20; rep++) {
            h.v[0] = rep;
            if (which == 0) hipLaunchKernelGGL(k_global, dim3(G), dim3(256), 0, 0, dtab, dout, sink);
            else hipLaunchKernelGGL(k_kernarg, dim3(G), dim3(256), 0, 0, h, dout, sink);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(ho.data(), dout, G * 8, hipMemcpyDeviceToHost));
            for (auto v : ho) { sum += (double)v; mx = std::max(mx, (double)v); }
        }
        CK(hipEventRecord(e0, 0));
        for (int rep = 0; rep < 200; rep++) {
            if (which == 0) hipLaunchKernelGGL(k_global, dim3(G), dim3(256), 0, 0, dtab, dout, sink);
            else hipLaunchKernelGGL(k_kernarg, dim3(G), dim3(256), 0, 0, h, dout, sink);
        }
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%s: the dependent table read costs a wave %.0f ticks on average (max %.0f); %.2f us per launch of %d workgroups\n",
               which == 0 ? "device table in global memory" : "table by value in the kernel arguments", sum / (20.0 * G), mx, ms * 1e3 / 200, G);
    }
    return 0;
}
