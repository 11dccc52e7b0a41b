// tools/launch_floor.hip — per-launch cost of back-to-back trivial kernels on one stream (the floor
// under any "N launches between two events" number), by grid shape, eager and as a hipGraph.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_empty(int *p) { if (p && threadIdx.x == 9999) p[0] = 1; }
// every workgroup writes `n` dwords (one lane each, stride `stride` dwords) — the shape of the bit-GEMM's output
__global__ void k_write(uint32_t *o, int n, int stride) {
    if (threadIdx.x < n) o[(blockIdx.x * n + threadIdx.x) * stride] = threadIdx.x;
}
__global__ void k_touch(const uint4 *x, uint4 *o, int n) {
    const int g = blockIdx.x * 2048 + threadIdx.x;
    uint4 a = x[g], b = x[g + 512], c = x[g + 1024], d = x[g + 1536];
    uint4 s = make_uint4(a.x ^ b.x ^ c.x ^ d.x, a.y ^ b.y ^ c.y ^ d.y, a.z ^ b.z ^ c.z ^ d.z, a.w ^ b.w ^ c.w ^ d.w);
    if ((s.x | s.y | s.z | s.w) == 0x12345u) o[blockIdx.x] = s;
}
int main() {
    uint4 *x, *o;
    hipMalloc(&x, 256 * 2048 * 16 + 4096);
    hipMalloc(&o, 4096 * 16);
    hipMemset(x, 1, 256 * 2048 * 16);
    hipStream_t st;
    hipStreamCreate(&st);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int grids[][2] = {{1, 64}, {256, 64}, {256, 512}, {1024, 256}, {4096, 256}};
    for (auto &g : grids) {
        float best = 1e9;
        for (int rep = 0; rep < 4; rep++) {
            hipEventRecord(e0, st);
            for (int i = 0; i < 200; i++) hipLaunchKernelGGL(k_empty, dim3(g[0]), dim3(g[1]), 0, st, nullptr);
            hipEventRecord(e1, st);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        printf("eager empty %4d x %3d: %.2f us/launch\n", g[0], g[1], best * 1e3 / 200);
    }
    {
        float best = 1e9;
        for (int rep = 0; rep < 4; rep++) {
            hipEventRecord(e0, st);
            for (int i = 0; i < 200; i++) hipLaunchKernelGGL(k_touch, dim3(256), dim3(512), 0, st, x, o, 0);
            hipEventRecord(e1, st);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        printf("eager read-32KiB-per-WG 256 x 512: %.2f us/launch\n", best * 1e3 / 200);
    }
    for (int cfg = 0; cfg < 4; cfg++) {
        const int n = cfg == 0 ? 1 : 32, stride = cfg <= 1 ? 1 : (cfg == 2 ? 4 : 64);
        float best = 1e9;
        for (int rep = 0; rep < 4; rep++) {
            hipEventRecord(e0, st);
            for (int i = 0; i < 200; i++) hipLaunchKernelGGL(k_write, dim3(256), dim3(512), 0, st, (uint32_t *)x, n, stride);
            hipEventRecord(e1, st);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        printf("eager write %d dwords/WG stride %d, 256 x 512: %.2f us/launch\n", n, stride, best * 1e3 / 200);
    }
    // the same 200 launches captured in a graph
    for (auto &g : grids) {
        hipGraph_t graph;
        hipGraphExec_t exec;
        hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
        for (int i = 0; i < 200; i++) hipLaunchKernelGGL(k_empty, dim3(g[0]), dim3(g[1]), 0, st, nullptr);
        hipStreamEndCapture(st, &graph);
        hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        float best = 1e9;
        for (int rep = 0; rep < 4; rep++) {
            hipEventRecord(e0, st);
            hipGraphLaunch(exec, st);
            hipEventRecord(e1, st);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        printf("graph empty %4d x %3d: %.2f us/launch\n", g[0], g[1], best * 1e3 / 200);
        hipGraphExecDestroy(exec);
        hipGraphDestroy(graph);
    }
    return 0;
}
