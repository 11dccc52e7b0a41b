// tools/dispatch_floor.hip — GPU-side cost of a grouped launch that is NOT work: how long the chip needs to start
// (and retire) G workgroups of 256 threads when each of them (a) returns at once, (b) reads a 64-byte descriptor with one
// scalar load, (c) follows the descriptor to a second scalar load (the occupancy word) and then to one 16-byte vector load
// per lane (the operand) - the dependent round trips of the row-block kernels. 200 launches replayed from a captured
// graph (an eager launch costs the HOST >= 2.45 us, which hides anything shorter), best of 4.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
struct Item { const u32x4 *data; const unsigned long long *occ; uint32_t n, pad[11]; };   // 64 bytes

__global__ __launch_bounds__(256) void k_none(const Item *it, u32x4 *sink) { if (it && threadIdx.x == 9999) sink[0] = u32x4{0, 0, 0, 0}; }
__global__ __launch_bounds__(256) void k_desc(const Item *it, u32x4 *sink) {
    const Item d = it[blockIdx.x];
    if (d.n == 0xdeadbeefu) sink[blockIdx.x] = u32x4{d.n, 0, 0, 0};
}
__global__ __launch_bounds__(256) void k_chain(const Item *it, u32x4 *sink) {
    const Item d = it[blockIdx.x];
    const unsigned long long o = d.occ[blockIdx.x & 1023];
    const u32x4 v = d.data[(threadIdx.x + (unsigned)(o & 7)) & 1023];
    if (v.x == 0xdeadbeefu) sink[blockIdx.x] = v;
}
// one round trip: the item already holds the occupancy word
__global__ __launch_bounds__(256) void k_chain1(const Item *it, u32x4 *sink) {
    const Item d = it[blockIdx.x];
    const u32x4 v = d.data[(threadIdx.x + (d.n & 7)) & 1023];
    if (v.x == 0xdeadbeefu) sink[blockIdx.x] = v;
}
// persistent form: gridDim.x workgroups walk `total` items
__global__ __launch_bounds__(256) void k_chain1_loop(const Item *it, u32x4 *sink, int total) {
    for (int i = blockIdx.x; i < total; i += gridDim.x) {
        const Item d = it[i];
        const u32x4 v = d.data[(threadIdx.x + (d.n & 7)) & 1023];
        if (v.x == 0xdeadbeefu) sink[i] = v;
    }
}

template <typename F>
static float graph_us(hipStream_t st, F &&launch) {
    hipGraph_t graph;
    hipGraphExec_t exec;
    hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
    for (int i = 0; i < 200; i++) launch();
    hipStreamEndCapture(st, &graph);
    hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 5; rep++) {
        hipEventRecord(e0, st);
        hipGraphLaunch(exec, st);
        hipEventRecord(e1, st);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    hipGraphExecDestroy(exec);
    hipGraphDestroy(graph);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    return best * 1e3f / 200.0f;
}

int main() {
    const int maxg = 16384;
    u32x4 *data, *sink;
    unsigned long long *occ;
    Item *items;
    hipMalloc(&data, 1024 * 16);
    hipMalloc(&sink, maxg * 16);
    hipMalloc(&occ, 1024 * 8);
    hipMalloc(&items, maxg * sizeof(Item));
    hipMemset(data, 0, 1024 * 16);
    hipMemset(occ, 0, 1024 * 8);
    std::vector<Item> h(maxg);
    for (int i = 0; i < maxg; i++) h[i] = Item{data, occ, (uint32_t)i, {}};
    hipMemcpy(items, h.data(), maxg * sizeof(Item), hipMemcpyHostToDevice);
    hipStream_t st;
    hipStreamCreate(&st);
    const int grids[] = {256, 1024, 1426, 2048, 2852, 4096, 8192};
    printf("%-8s %10s %10s %10s %10s\n", "grid", "return", "1 s_load", "3 trips", "2 trips");
    for (int g : grids) {
        const float a = graph_us(st, [&] { hipLaunchKernelGGL(k_none, dim3(g), dim3(256), 0, st, items, sink); });
        const float b = graph_us(st, [&] { hipLaunchKernelGGL(k_desc, dim3(g), dim3(256), 0, st, items, sink); });
        const float c = graph_us(st, [&] { hipLaunchKernelGGL(k_chain, dim3(g), dim3(256), 0, st, items, sink); });
        const float d = graph_us(st, [&] { hipLaunchKernelGGL(k_chain1, dim3(g), dim3(256), 0, st, items, sink); });
        printf("%-8d %10.2f %10.2f %10.2f %10.2f   us per launch (256 threads per workgroup)\n", g, a, b, c, d);
    }
    printf("persistent: 2852 items walked by G workgroups (2 trips per item)\n");
    for (int g : {256, 512, 713, 951, 1024, 1426, 2048, 2852}) {
        const float d = graph_us(st, [&] { hipLaunchKernelGGL(k_chain1_loop, dim3(g), dim3(256), 0, st, items, sink, 2852); });
        printf("  G = %-5d %8.2f us per launch\n", g, d);
    }
    return 0;
}
