// atomic_probe.hip - what the data loader's edge kernel pays per edge (tools/README.md): E random bits OR-ed into a bitmap of B bytes with
// device-scope atomics - with the old word returned (duplicate detection), fire-and-forget, two bitmaps per edge, and a plain dense
// pass over the bitmap for scale.   hipcc --offload-arch=gfx950 -O3 -o atomic_probe atomic_probe.hip ; ./atomic_probe [E] [MB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k_ret(const unsigned *idx, unsigned *bm, unsigned *dup, int E) {
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < E; e += gridDim.x * blockDim.x) {
        const unsigned i = idx[e], bit = 1u << (i & 31);
        if (atomicOr(bm + (i >> 5), bit) & bit) atomicAdd(dup, 1u);
    }
}
__global__ void k_noret(const unsigned *idx, unsigned *bm, int E) {
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < E; e += gridDim.x * blockDim.x) {
        const unsigned i = idx[e];
        __hip_atomic_fetch_or(bm + (i >> 5), 1u << (i & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
__global__ void k_two(const unsigned *idx, unsigned *bm, unsigned *bm2, unsigned *dup, int E) {
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < E; e += gridDim.x * blockDim.x) {
        const unsigned i = idx[e], bit = 1u << (i & 31);
        __hip_atomic_fetch_or(bm2 + ((i >> 5) ^ 0x155), bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (atomicOr(bm + (i >> 5), bit) & bit) atomicAdd(dup, 1u);
    }
}
__global__ void k_wg(const unsigned *idx, unsigned *bm, unsigned *dup, int E) {   // workgroup-scope atomics (executed in the XCD's L2): NOT coherent across XCDs - timing only
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < E; e += gridDim.x * blockDim.x) {
        const unsigned i = idx[e], bit = 1u << (i & 31);
        if (__hip_atomic_fetch_or(bm + (i >> 5), bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) & bit) atomicAdd(dup, 1u);
    }
}
__global__ void k_dense(const uint4 *in, uint4 *out, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint4 v = in[i];
        v.x |= v.y;
        out[i] = v;
    }
}
int main(int argc, char **argv) {
    const int E = argc > 1 ? atoi(argv[1]) : 536728;
    const size_t MB = argc > 2 ? atoi(argv[2]) : 14;
    const size_t words = MB * 1024 * 1024 / 4;
    std::vector<unsigned> h(E);
    unsigned long long s = 88172645463325252ull;
    for (int e = 0; e < E; e++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[e] = (unsigned)(s % (words * 32)); }
    unsigned *idx, *bm, *bm2, *dup;
    CK(hipMalloc(&idx, E * 4)); CK(hipMalloc(&bm, words * 4)); CK(hipMalloc(&bm2, words * 4)); CK(hipMalloc(&dup, 4));
    CK(hipMemcpy(idx, h.data(), E * 4, hipMemcpyHostToDevice));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto run = [&](const char *name, auto launch) {
        float best = 1e9f;
        for (int r = 0; r < 6; r++) {
            CK(hipMemsetAsync(bm, 0, words * 4)); CK(hipMemsetAsync(bm2, 0, words * 4)); CK(hipMemsetAsync(dup, 0, 4));
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(a)); launch(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b)); if (r && ms < best) best = ms;
        }
        unsigned d; CK(hipMemcpy(&d, dup, 4, hipMemcpyDeviceToHost));
        printf("%-42s %8.2f us  (dups %u)\n", name, best * 1e3f, d);
    };
    const int T = 256;
    for (int per = 1; per <= 4; per *= 2) {
        const int G = (E + T * per - 1) / (T * per);
        printf("-- %d edge(s) a thread, %d workgroups\n", per, G);
        run("atomicOr, old word returned", [&] { hipLaunchKernelGGL(k_ret, dim3(G), dim3(T), 0, 0, idx, bm, dup, E); });
        run("atomic OR, fire and forget", [&] { hipLaunchKernelGGL(k_noret, dim3(G), dim3(T), 0, 0, idx, bm, E); });
        run("two bitmaps (one returned)", [&] { hipLaunchKernelGGL(k_two, dim3(G), dim3(T), 0, 0, idx, bm, bm2, dup, E); });
        run("workgroup-scope atomicOr (timing only)", [&] { hipLaunchKernelGGL(k_wg, dim3(G), dim3(T), 0, 0, idx, bm, dup, E); });
    }
    run("dense pass over the bitmap (read + write)", [&] { hipLaunchKernelGGL(k_dense, dim3(2048), dim3(256), 0, 0, (const uint4 *)bm, (uint4 *)bm2, words / 4); });
    run("memset of the bitmap", [&] { CK(hipMemsetAsync(bm2, 0, words * 4)); });
    return 0;
}
