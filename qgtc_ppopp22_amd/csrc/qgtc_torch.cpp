// qgtc_torch.cpp — the `QGTC` PyTorch-ROCm extension: a thin pybind11 binding over the C-ABI of
// libqgtc_hip.so (include/qgtc.h). It replaces the reference's QGTC_host.cpp (bindings,
// :259-271) and the host halves of QGTC_device.cu (output allocation + shape rules), keeping the
// eight exported names and their positional signatures.
//
// Differences from the reference host code, all deliberate:
//   * outputs are allocated on the *input's* device with torch::empty (every word is written by
//     the kernels) instead of a CPU torch::zeros + .to(kCUDA) (QGTC_device.cu:63,115,223,507);
//   * kernels are launched on the current HIP stream and do not synchronise, except the
//     profile / counter variants which (like the reference) block;
//   * errors raise RuntimeError instead of printf + exit(-1) (QGTC_device.cu:67-71);
//   * dtype / dim / size checks are added (the reference reads raw data<int>() pointers);
//   * bit2val does not printf the tensor (QGTC_device.cu:169,187,201-202).
#include <torch/extension.h>

#include <ATen/hip/EmptyTensor.h>
#include <c10/core/DeviceGuard.h>
#include <c10/hip/HIPFunctions.h>
#include <c10/hip/HIPStream.h>
#include <torch/csrc/autograd/python_variable.h>

#include <array>
#include <pybind11/stl.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <memory>
#include <mutex>
#include <vector>

#include "qgtc.h"

namespace {

inline int S8(int x) { return (x + 7) >> 3; }
inline int S128(int x) { return (x + 127) >> 7; }
inline int P8(int x) { return S8(x) << 3; }
inline int P128(int x) { return S128(x) << 7; }

// Same wording as the reference's CHECK_INPUT (QGTC_host.cpp:99-101).
#define CHECK_CUDA(x) TORCH_CHECK(x.is_cuda(), #x " must be a CUDA tensor")
#define CHECK_CONTIGUOUS(x) TORCH_CHECK(x.is_contiguous(), #x " must be contiguous")
#define CHECK_INPUT(x) \
    CHECK_CUDA(x);     \
    CHECK_CONTIGUOUS(x)

void check_rc(int rc, const char *op) {
    if (rc == QGTC_OK) return;
    if (rc == QGTC_EHIP)
        TORCH_CHECK(false, "QGTC.", op, ": ", qgtc_strerror(rc), " (", qgtc_last_hip_error(), ")");
    TORCH_CHECK(false, "QGTC.", op, ": ", qgtc_strerror(rc));
}

void *current_stream(const torch::Tensor &t) {
    return static_cast<void *>(c10::hip::getCurrentHIPStream(t.get_device()).stream());
}

const uint32_t *words(const torch::Tensor &t) {
    return reinterpret_cast<const uint32_t *>(t.data_ptr<int32_t>());
}
uint32_t *words_mut(torch::Tensor &t) { return reinterpret_cast<uint32_t *>(t.data_ptr<int32_t>()); }

void check_bits_tensor(const torch::Tensor &t, const char *name) {
    TORCH_CHECK(t.scalar_type() == torch::kInt32, name, " must be an int32 bit tensor");
}

// Process-wide switches of the binding (the reference's callers are single-threaded and hold the GIL for the whole
// call, QGTC_host.cpp; the C-ABI itself is stateless - every call carries its flags). std::atomic: a second Python
// thread may flip them, a launch then simply sees the old or the new value.
std::atomic<bool> g_zero_skip{true};
// 0 popcount (AND + v_bcnt kernels), 1 mfma (matrix cores wherever the plane counts allow it), 2 auto: per call the
// kernel family that measured fastest on MI355X (launch_common.hip.h) - the DEFAULT, and what bench.py's headline runs.
// QGTC_ENGINE=popcount|mfma|auto or set_engine() choose another; every engine returns the same words.
std::atomic<int> g_engine{2};
unsigned mm_flags() {
    return (g_zero_skip ? 0u : QGTC_NO_ZERO_SKIP) | (g_engine == 1 ? QGTC_ENGINE_MFMA : 0u) |
           (g_engine == 2 ? QGTC_ENGINE_AUTO : 0u);
}

// process-cumulative tile counters, like the reference's __device__ globals (kernel.h:13-14)
std::atomic<unsigned long long> g_counter{0}, g_counter_global{0};
std::atomic<double> g_last_profile_ms{0.0};

// Side streams for the multi-stream launchers, one pool PER DEVICE (a pool filled on device 0 must never be used
// under device 1's guard).
constexpr int kMaxDevices = 64;
std::mutex g_pool_mutex;
std::array<std::vector<c10::hip::HIPStream>, kMaxDevices> g_stream_pools;
std::vector<c10::hip::HIPStream> side_streams(int device, int n) {
    TORCH_CHECK(device >= 0 && device < kMaxDevices, "device index out of range");
    std::lock_guard<std::mutex> lock(g_pool_mutex);
    auto &pool = g_stream_pools[device];
    while (static_cast<int>(pool.size()) < n) pool.push_back(c10::hip::getStreamFromPool(false, device));
    return std::vector<c10::hip::HIPStream>(pool.begin(), pool.begin() + n);
}

// hipEvent that is destroyed on every path out of the scope (TORCH_CHECK throws)
struct ScopedEvent {
    hipEvent_t ev = nullptr;
    ScopedEvent() { TORCH_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess, "hipEventCreate failed"); }
    ~ScopedEvent() { if (ev) (void)hipEventDestroy(ev); }
    ScopedEvent(const ScopedEvent &) = delete;
    ScopedEvent &operator=(const ScopedEvent &) = delete;
};

// ---------------------------------------------------------------------------------------------
torch::Tensor val2bit(torch::Tensor input, const int nbits, const bool col_major,
                      const bool output_layer) {
    CHECK_INPUT(input);
    TORCH_CHECK(input.scalar_type() == torch::kFloat32, "input must be a float32 tensor");
    TORCH_CHECK(input.dim() == 2, "input must be 2-D");
    const int H = input.size(0), W = input.size(1);
    c10::DeviceGuard guard(input.device());
    const auto opts = torch::TensorOptions().dtype(torch::kInt32).device(input.device());
    torch::Tensor out;
    if (col_major)  // QGTC_device.cu:83,97
        out = torch::empty({static_cast<int64_t>(nbits) * S128(H) * 4, output_layer ? P8(W) : P128(W)}, opts);
    else            // QGTC_device.cu:115
        out = torch::empty({static_cast<int64_t>(nbits) * P8(H), S128(W) * 4}, opts);
    check_rc(qgtc_val2bit(input.data_ptr<float>(), H, W, nbits, col_major, output_layer,
                          words_mut(out), out.numel(), current_stream(input)),
             "val2bit");
    return out;
}

torch::Tensor bit2val(torch::Tensor input, const int nbits, const int height, const int width,
                      const bool col_major, const bool output_layer) {
    CHECK_INPUT(input);
    check_bits_tensor(input, "input");
    c10::DeviceGuard guard(input.device());
    TORCH_CHECK(height > 0 && width > 0, "height and width must be positive");
    auto out = torch::empty({height, width},
                            torch::TensorOptions().dtype(torch::kInt32).device(input.device()));
    check_rc(qgtc_bit2val(words(input), input.numel(), nbits, height, width, col_major,
                          output_layer, out.data_ptr<int32_t>(), current_stream(input)),
             "bit2val");
    return out;
}

torch::Tensor mm2bit_impl(const torch::Tensor &bit_X1, const torch::Tensor &bit_X2, int M, int K,
                          int N, int bit1, int bit2, int ob, bool cols, const char *op) {
    CHECK_INPUT(bit_X1);
    CHECK_INPUT(bit_X2);
    check_bits_tensor(bit_X1, "bit_X1");
    check_bits_tensor(bit_X2, "bit_X2");
    TORCH_CHECK(bit_X1.device() == bit_X2.device(), "bit_X1 and bit_X2 must be on the same device");
    TORCH_CHECK(M > 0 && K > 0 && N > 0 && ob >= 1 && ob <= 32, "bad dimensions / output_bit");
    c10::DeviceGuard guard(bit_X1.device());
    const auto opts = torch::TensorOptions().dtype(torch::kInt32).device(bit_X1.device());
    torch::Tensor out = cols
        ? torch::empty({static_cast<int64_t>(ob) * S128(M) * 4, P128(N)}, opts)   // QGTC_device.cu:456
        : torch::empty({static_cast<int64_t>(ob) * P8(M), S128(N) * 4}, opts);    // QGTC_device.cu:223
    check_rc(qgtc_bitmm2bit(words(bit_X1), bit_X1.numel(), words(bit_X2), bit_X2.numel(), M, K, N,
                            bit1, bit2, ob, words_mut(out), out.numel(),
                            mm_flags() | (cols ? QGTC_OUT_COLS : 0u), current_stream(bit_X1)),
             op);
    return out;
}

torch::Tensor bitMM2Bit(torch::Tensor bit_X1, torch::Tensor bit_X2, const int X1_height,
                        const int X1_width, const int X2_width, const int bit1, const int bit2,
                        const int output_bit) {
    return mm2bit_impl(bit_X1, bit_X2, X1_height, X1_width, X2_width, bit1, bit2, output_bit, false,
                       "bitMM2Bit");
}

torch::Tensor bitMM2Bit_col(torch::Tensor bit_X1, torch::Tensor bit_X2, const int X1_height,
                            const int X1_width, const int X2_width, const int bit1, const int bit2,
                            const int output_bit) {
    return mm2bit_impl(bit_X1, bit_X2, X1_height, X1_width, X2_width, bit1, bit2, output_bit, true,
                       "bitMM2Bit_col");
}

// Times `reps` launches (blocking) and returns (packed result, elapsed ms).
std::pair<torch::Tensor, double> profile_impl(const torch::Tensor &bit_X1, const torch::Tensor &bit_X2,
                                              int M, int K, int N, int bit1, int bit2, int ob,
                                              int reps) {
    CHECK_INPUT(bit_X1);
    CHECK_INPUT(bit_X2);
    check_bits_tensor(bit_X1, "bit_X1");
    check_bits_tensor(bit_X2, "bit_X2");
    TORCH_CHECK(M > 0 && K > 0 && N > 0 && ob >= 1 && ob <= 32, "bad dimensions / output_bit");
    c10::DeviceGuard guard(bit_X1.device());
    auto out = torch::empty({static_cast<int64_t>(ob) * P8(M), S128(N) * 4},
                            torch::TensorOptions().dtype(torch::kInt32).device(bit_X1.device()));
    float ms = 0.0f;
    check_rc(qgtc_bitmm2bit_profile(words(bit_X1), bit_X1.numel(), words(bit_X2), bit_X2.numel(), M,
                                    K, N, bit1, bit2, ob, words_mut(out), out.numel(), mm_flags(),
                                    reps, &ms, current_stream(bit_X1)),
             "bitMM2Bit_profile");
    g_last_profile_ms = ms;
    return {out, ms};
}

// Enqueue `reps` launches writing into a caller-provided packed output (no allocation, no
// synchronisation): the lean launch path used by bench.py and by steady-state serving loops.
void bitMM2Bit_enqueue(torch::Tensor out, torch::Tensor bit_X1, torch::Tensor bit_X2, int M, int K,
                       int N, int bit1, int bit2, int ob, int reps, bool cols) {
    CHECK_INPUT(out);
    CHECK_INPUT(bit_X1);
    CHECK_INPUT(bit_X2);
    check_bits_tensor(out, "out");
    check_bits_tensor(bit_X1, "bit_X1");
    check_bits_tensor(bit_X2, "bit_X2");
    TORCH_CHECK(reps > 0, "reps must be positive");
    TORCH_CHECK(out.device() == bit_X1.device() && bit_X2.device() == bit_X1.device(), "out, bit_X1 and bit_X2 must be on the same device");
    c10::DeviceGuard guard(bit_X1.device());
    void *st = current_stream(bit_X1);
    for (int i = 0; i < reps; i++)
        check_rc(qgtc_bitmm2bit(words(bit_X1), bit_X1.numel(), words(bit_X2), bit_X2.numel(), M, K, N,
                                bit1, bit2, ob, words_mut(out), out.numel(), mm_flags() | (cols ? QGTC_OUT_COLS : 0u), st),
                 "bitMM2Bit_enqueue");
}

// Enqueue `reps` INDEPENDENT bitMM2Bit launches round-robin on `outs.size()` HIP streams (launch i
// writes outs[i % n]); the current stream waits for all of them. Cluster batches are independent,
// so a serving loop may overlap their kernels: the head and tail of one launch run under the
// multiply phase of another instead of being serialised by stream order.
void bitMM2Bit_enqueue_streams(std::vector<torch::Tensor> outs, torch::Tensor bit_X1, torch::Tensor bit_X2,
                               int M, int K, int N, int bit1, int bit2, int ob, int reps) {
    CHECK_INPUT(bit_X1);
    CHECK_INPUT(bit_X2);
    check_bits_tensor(bit_X1, "bit_X1");
    check_bits_tensor(bit_X2, "bit_X2");
    TORCH_CHECK(!outs.empty() && reps > 0, "need at least one output buffer and one launch");
    c10::DeviceGuard guard(bit_X1.device());
    const int n = static_cast<int>(outs.size());
    TORCH_CHECK(bit_X2.device() == bit_X1.device(), "bit_X1 and bit_X2 must be on the same device");
    for (auto &o : outs) {
        CHECK_INPUT(o);
        check_bits_tensor(o, "out");
        TORCH_CHECK(o.device() == bit_X1.device(), "every output must be on the operands' device");
    }
    auto cur = c10::hip::getCurrentHIPStream(bit_X1.get_device());
    const auto pool = side_streams(bit_X1.get_device(), n);
    ScopedEvent sev;
    hipEvent_t ev = sev.ev;
    TORCH_CHECK(hipEventRecord(ev, cur.stream()) == hipSuccess, "hipEventRecord failed");
    for (int s = 0; s < n; s++) TORCH_CHECK(hipStreamWaitEvent(pool[s].stream(), ev, 0) == hipSuccess, "wait failed");
    for (int i = 0; i < reps; i++) {
        torch::Tensor &o = outs[i % n];
        check_rc(qgtc_bitmm2bit(words(bit_X1), bit_X1.numel(), words(bit_X2), bit_X2.numel(), M, K, N, bit1,
                                bit2, ob, words_mut(o), o.numel(), mm_flags(),
                                static_cast<void *>(pool[i % n].stream())),
                 "bitMM2Bit_enqueue_streams");
    }
    for (int s = 0; s < n; s++) {
        TORCH_CHECK(hipEventRecord(ev, pool[s].stream()) == hipSuccess, "hipEventRecord failed");
        TORCH_CHECK(hipStreamWaitEvent(cur.stream(), ev, 0) == hipSuccess, "wait failed");
    }
}

torch::Tensor bitMM2Bit_profile(torch::Tensor bit_X1, torch::Tensor bit_X2, const int X1_height,
                                const int X1_width, const int X2_width, const int bit1,
                                const int bit2, const int output_bit) {
    constexpr int PROF = 200;  // QGTC_device.cu:409
    auto r = profile_impl(bit_X1, bit_X2, X1_height, X1_width, X2_width, bit1, bit2, output_bit, PROF);
    // line format of QGTC_device.cu:420-422 (2_7c / 5_9 consumers read it from stdout)
    const float ops = 2.0f * X1_height * X1_width * X2_width * PROF;
    printf("X1_height %d, X1_width: %d, X2_width: %d, TFLOPs: %.3f\n", X1_height, X1_width, X2_width,
           ops / (r.second / 1e3) / 1e12);
    fflush(stdout);
    return r.first;
}

std::vector<unsigned long long> tile_counters(const torch::Tensor &bit_X1, int M, int K, int N,
                                              int bit1, int bit2) {
    c10::DeviceGuard guard(bit_X1.device());
    auto buf = torch::empty({2}, torch::TensorOptions().dtype(torch::kInt64).device(bit_X1.device()));
    check_rc(qgtc_tile_counters(words(bit_X1), bit_X1.numel(), M, K, N, bit1, bit2,
                                reinterpret_cast<uint64_t *>(buf.data_ptr<int64_t>()),
                                current_stream(bit_X1)),
             "tile_counters");
    auto host = buf.cpu();  // synchronises, as the reference's cudaEventSynchronize does
    return {static_cast<unsigned long long>(host.data_ptr<int64_t>()[0]),
            static_cast<unsigned long long>(host.data_ptr<int64_t>()[1])};
}

torch::Tensor bitMM2Bit_base_cnt(torch::Tensor bit_X1, torch::Tensor bit_X2, const int X1_height,
                                 const int X1_width, const int X2_width, const int bit1,
                                 const int bit2, const int output_bit) {
    auto out = mm2bit_impl(bit_X1, bit_X2, X1_height, X1_width, X2_width, bit1, bit2, output_bit,
                           false, "bitMM2Bit_base_cnt");
    auto c = tile_counters(bit_X1, X1_height, X1_width, X2_width, bit1, bit2);
    const unsigned long long total = g_counter_global.fetch_add(c[0]) + c[0];
    printf("counter_global: %d\n", static_cast<int>(total));  // kernel.h:27 (%d of a 64-bit)
    fflush(stdout);
    return out;
}

torch::Tensor bitMM2Bit_zerojump_cnt(torch::Tensor bit_X1, torch::Tensor bit_X2,
                                     const int X1_height, const int X1_width, const int X2_width,
                                     const int bit1, const int bit2, const int output_bit) {
    auto out = mm2bit_impl(bit_X1, bit_X2, X1_height, X1_width, X2_width, bit1, bit2, output_bit,
                           false, "bitMM2Bit_zerojump_cnt");
    auto c = tile_counters(bit_X1, X1_height, X1_width, X2_width, bit1, bit2);
    const unsigned long long total = g_counter.fetch_add(c[1]) + c[1];
    printf("counter: %d\n", static_cast<int>(total));  // kernel.h:19
    fflush(stdout);
    return out;
}

torch::Tensor bitMM2Int(torch::Tensor bit_X1, torch::Tensor bit_X2, const int X1_height,
                        const int X1_width, const int X2_width, const int bit1, const int bit2,
                        const bool pad_128) {
    CHECK_INPUT(bit_X1);
    CHECK_INPUT(bit_X2);
    check_bits_tensor(bit_X1, "bit_X1");
    check_bits_tensor(bit_X2, "bit_X2");
    TORCH_CHECK(bit_X1.device() == bit_X2.device(), "bit_X1 and bit_X2 must be on the same device");
    TORCH_CHECK(X1_height > 0 && X1_width > 0 && X2_width > 0, "bad dimensions");
    c10::DeviceGuard guard(bit_X1.device());
    auto out = torch::empty({X1_height, X2_width},
                            torch::TensorOptions().dtype(torch::kFloat32).device(bit_X1.device()));
    check_rc(qgtc_bitmm2int(words(bit_X1), bit_X1.numel(), words(bit_X2), bit_X2.numel(), X1_height,
                            X1_width, X2_width, bit1, bit2, pad_128, out.data_ptr<float>(),
                            out.numel(), mm_flags(), current_stream(bit_X1)),
             "bitMM2Int");
    return out;
}

// ---------------------------------------------------------------------------------------------
// Adjacency bit planes from an edge list (additive; the dense val2bit route stays). The
// multiplicity of duplicate edges is what the reference's to_dense() would sum (sampler.py:87-89).
torch::Tensor pack_edges(torch::Tensor src, torch::Tensor dst, const int height, const int width,
                         const int nbits, const bool validate) {
    CHECK_INPUT(src);
    CHECK_INPUT(dst);
    TORCH_CHECK(src.scalar_type() == torch::kInt64 && dst.scalar_type() == torch::kInt64,
                "src and dst must be int64 index tensors");
    TORCH_CHECK(src.dim() == 1 && src.sizes() == dst.sizes(), "src and dst must be 1-D and equally long");
    TORCH_CHECK(height > 0 && width > 0, "height and width must be positive");
    c10::DeviceGuard guard(src.device());
    const auto i32 = torch::TensorOptions().dtype(torch::kInt32).device(src.device());
    if (nbits == 1) {
        // the adjacency case: raw edge list, no sort / unique, no host round trip unless validate asks for one. The result owns exactly
        // its own words; the two scratch bitmaps are a separate allocation that goes back to the caching allocator when this call
        // returns (round 4 returned a view of one [out | scratch] allocation: every adjacency a caller kept pinned three times its size).
        const int64_t words = static_cast<int64_t>(P8(height)) * S128(width) * 4;
        auto out = torch::empty({static_cast<int64_t>(P8(height)), S128(width) * 4}, i32);
        auto scratch = torch::empty({2 * words}, i32);
        torch::Tensor bad;
        if (validate) bad = torch::empty({1}, i32);
        check_rc(qgtc_pack_edge_list(src.numel() ? src.data_ptr<int64_t>() : nullptr, src.numel() ? dst.data_ptr<int64_t>() : nullptr,
                                     src.numel(), height, width, words_mut(out), words, words_mut(scratch),
                                     2 * words, validate ? bad.data_ptr<int>() : nullptr, current_stream(src)),
                 "pack_edges");
        if (validate) TORCH_CHECK(bad.item<int>() == 0, "edge index out of range");
        return out;
    }
    auto out = torch::empty({static_cast<int64_t>(nbits) * P8(height), S128(width) * 4}, i32);
    torch::Tensor cells, counts;
    if (src.numel() > 0) {
        TORCH_CHECK(src.min().item<int64_t>() >= 0 && src.max().item<int64_t>() < height &&
                    dst.min().item<int64_t>() >= 0 && dst.max().item<int64_t>() < width, "edge index out of range");
        auto uq = at::_unique2(src * width + dst, /*sorted=*/true, /*return_inverse=*/false, /*return_counts=*/true);
        cells = std::get<0>(uq).contiguous();
        counts = std::get<2>(uq).to(torch::kInt32).contiguous();
    }
    check_rc(qgtc_pack_edges(cells.defined() ? cells.data_ptr<int64_t>() : nullptr,
                             counts.defined() ? counts.data_ptr<int32_t>() : nullptr,
                             cells.defined() ? cells.numel() : 0, height, width, nbits, words_mut(out),
                             out.numel(), current_stream(src)),
             "pack_edges");
    return out;
}

// int8 MFMA GEMM (comparison path, cuBLASGemmEX analogue): float32 [M,N] = A[M,K] x Bt[N,K]^T
torch::Tensor i8gemm(torch::Tensor A, torch::Tensor Bt) {
    CHECK_INPUT(A);
    CHECK_INPUT(Bt);
    TORCH_CHECK(A.scalar_type() == torch::kInt8 && Bt.scalar_type() == torch::kInt8, "A and Bt must be int8");
    TORCH_CHECK(A.dim() == 2 && Bt.dim() == 2 && A.size(1) == Bt.size(1), "A is [M,K], Bt is [N,K]");
    c10::DeviceGuard guard(A.device());
    const int M = A.size(0), K = A.size(1), N = Bt.size(0);
    auto out = torch::empty({M, N}, torch::TensorOptions().dtype(torch::kFloat32).device(A.device()));
    check_rc(qgtc_i8gemm(A.data_ptr<int8_t>(), Bt.data_ptr<int8_t>(), M, K, N, out.data_ptr<float>(),
                         out.numel(), current_stream(A)),
             "i8gemm");
    return out;
}

double i8gemm_profile(torch::Tensor A, torch::Tensor Bt, int reps, bool print) {
    CHECK_INPUT(A);
    CHECK_INPUT(Bt);
    TORCH_CHECK(A.scalar_type() == torch::kInt8 && Bt.scalar_type() == torch::kInt8, "A and Bt must be int8");
    TORCH_CHECK(A.dim() == 2 && Bt.dim() == 2 && A.size(1) == Bt.size(1), "A is [M,K], Bt is [N,K]");
    c10::DeviceGuard guard(A.device());
    const int M = A.size(0), K = A.size(1), N = Bt.size(0);
    auto out = torch::empty({M, N}, torch::TensorOptions().dtype(torch::kFloat32).device(A.device()));
    float ms = 0.0f;
    check_rc(qgtc_i8gemm_profile(A.data_ptr<int8_t>(), Bt.data_ptr<int8_t>(), M, K, N,
                                 out.data_ptr<float>(), out.numel(), reps, &ms, current_stream(A)),
             "i8gemm_profile");
    if (print) {  // line format of cublas_main.cu:170
        printf("M: %d, K: %d, N: %d, TFLOPS: %.2f\n", M, K, N,
               static_cast<double>(reps) * (static_cast<double>(M) * K * N * 2) / (ms / 1000.) / 1e12);
        fflush(stdout);
    }
    return ms;
}

// ---------------------------------------------------------------------------------------------
// Grouped launch over many cluster batches (additive API; the per-batch calls above stay).
// A BatchedGemm owns the device array of problem descriptors so that an epoch can re-launch it
// without any host work besides the launch itself.
// ---------------------------------------------------------------------------------------------
struct BatchedGemm {
    torch::Tensor descs;  // uint8 device tensor holding qgtc_problem[count]
    std::vector<torch::Tensor> keep;  // operands + pools kept alive
    std::vector<torch::Tensor> outs;  // views into one pooled allocation per launch
    std::vector<torch::Tensor> occs;  // occupancy bitmaps of the left operands (zero_jump), views into one pool
    std::vector<qgtc_problem> host_descs;  // the same descriptors on the host, for run_per_problem()
    torch::Tensor stats;  // device: [occupied, all] 32-row x 128-bit tiles of the left operands (zero_jump)
    int count = 0, max_M = 0, max_K = 0, max_N = 0, bit1 = 1, bit2 = 1, ob = 1, mode = 0;
    bool jump_asked = false;
    static constexpr double kJumpBelow = 0.25;  // measured: at 19 % occupied tiles jumping gains 10 %, at 43 % it loses 15 %

    // Xs[i]: rows-layout left operand of problem i; Ws: one shared right operand (len 1) or one
    // per problem; dims[i] = (M, K, N). mode 0/1/2 as qgtc_bitmm_batched; pad_128 only for mode 2.
    // zero_jump: build the occupancy bitmap of every left operand once (one grouped launch) so that
    // run() neither loads nor multiplies all-zero 32-row x 128-bit X tiles. Jumping pays when most X
    // tiles are empty (block-diagonal cluster adjacency); whether it does is decided on the device
    // (qgtc_tile_occupancy_batched clears the descriptors' bitmaps above a quarter occupied), so building
    // the plan never waits for the GPU; .zero_jump / .occupied_fraction read the outcome back lazily.
    // The bitmap describes Xs[i] as it is NOW: pass false when the left operands are outputs of an
    // earlier stage that change between runs.
    BatchedGemm(std::vector<torch::Tensor> Xs, std::vector<torch::Tensor> Ws,
                std::vector<std::tuple<int, int, int>> dims, int bit1_, int bit2_, int ob_,
                int mode_, bool pad_128, bool zero_jump, std::vector<torch::Tensor> reuse_occs)
        : bit1(bit1_), bit2(bit2_), ob(ob_), mode(mode_), jump_asked(zero_jump) {
        count = static_cast<int>(Xs.size());
        TORCH_CHECK(count > 0, "empty batch");
        TORCH_CHECK(Ws.size() == 1 || static_cast<int>(Ws.size()) == count, "Ws must have 1 or len(Xs) tensors");
        TORCH_CHECK(static_cast<int>(dims.size()) == count, "dims must have len(Xs) entries");
        TORCH_CHECK(mode >= 0 && mode <= 2, "mode must be 0, 1 or 2");
        TORCH_CHECK(reuse_occs.empty() || static_cast<int>(reuse_occs.size()) == count, "occs must have len(Xs) tensors");
        const auto dev = Xs[0].device();
        c10::DeviceGuard guard(dev);
        std::vector<qgtc_problem> h(count);
        // one allocation for all outputs (and one for all bitmaps): every view starts 16-byte aligned
        std::vector<int64_t> out_off(count + 1, 0), occ_off(count + 1, 0);
        auto round4 = [](int64_t n) { return (n + 3) & ~int64_t(3); };
        for (int i = 0; i < count; i++) {
            const int M = std::get<0>(dims[i]), K = std::get<1>(dims[i]), N = std::get<2>(dims[i]);
            TORCH_CHECK(M > 0 && K > 0 && N > 0, "bad dimensions");
            const int64_t n_out = mode == 2 ? static_cast<int64_t>(M) * N
                                : mode == 1 ? static_cast<int64_t>(ob) * S128(M) * 4 * P128(N)
                                            : static_cast<int64_t>(ob) * P8(M) * S128(N) * 4;
            out_off[i + 1] = out_off[i] + round4(n_out);
            occ_off[i + 1] = occ_off[i] + static_cast<int64_t>(qgtc_occupancy_words(M, K));
        }
        torch::Tensor out_pool = torch::empty({out_off[count]}, torch::TensorOptions().dtype(mode == 2 ? torch::kFloat32 : torch::kInt32).device(dev));
        torch::Tensor occ_pool;
        if (zero_jump && reuse_occs.empty())
            occ_pool = torch::empty({occ_off[count]}, torch::TensorOptions().dtype(torch::kInt64).device(dev));
        keep.push_back(out_pool);
        for (int i = 0; i < count; i++) {
            const torch::Tensor &X = Xs[i];
            const torch::Tensor &W = Ws.size() == 1 ? Ws[0] : Ws[i];
            CHECK_INPUT(X);
            CHECK_INPUT(W);
            check_bits_tensor(X, "X");
            check_bits_tensor(W, "W");
            TORCH_CHECK(X.device() == dev && W.device() == dev, "all operands must share a device");
            const int M = std::get<0>(dims[i]), K = std::get<1>(dims[i]), N = std::get<2>(dims[i]);
            TORCH_CHECK(X.numel() < (1LL << 30) && W.numel() < (1LL << 30), "packed operand too large (>= 4 GiB)");
            TORCH_CHECK((reinterpret_cast<uintptr_t>(X.data_ptr()) & 15) == 0 &&
                        (reinterpret_cast<uintptr_t>(W.data_ptr()) & 15) == 0, "packed operands must be 16-byte aligned");
            torch::Tensor flat = out_pool.narrow(0, out_off[i], mode == 2 ? static_cast<int64_t>(M) * N
                                                 : mode == 1 ? static_cast<int64_t>(ob) * S128(M) * 4 * P128(N)
                                                             : static_cast<int64_t>(ob) * P8(M) * S128(N) * 4);
            torch::Tensor out = mode == 2 ? flat.view({M, N})
                              : mode == 1 ? flat.view({static_cast<int64_t>(ob) * S128(M) * 4, P128(N)})
                                          : flat.view({static_cast<int64_t>(ob) * P8(M), S128(N) * 4});
            h[i].X = words(X);
            h[i].W = words(W);
            h[i].out = out.data_ptr();
            h[i].x_words = X.numel();
            h[i].w_words = W.numel();
            h[i].M = M;
            h[i].K = K;
            h[i].N = N;
            h[i].w_lines = (mode == 2 && !pad_128) ? P8(N) : P128(N);
            h[i].occ = nullptr;
            h[i].occ_words = 0;
            if (zero_jump) {
                const int64_t nw = occ_off[i + 1] - occ_off[i];
                torch::Tensor occ;
                if (!reuse_occs.empty()) {  // bitmaps of the same left operands from an earlier BatchedGemm
                    occ = reuse_occs[i];
                    TORCH_CHECK(occ.is_cuda() && occ.is_contiguous() && occ.scalar_type() == torch::kInt64 &&
                                occ.numel() >= nw && occ.device() == dev, "bad occupancy bitmap");
                } else {
                    occ = occ_pool.narrow(0, occ_off[i], nw);
                }
                h[i].occ = reinterpret_cast<const uint64_t *>(occ.data_ptr<int64_t>());
                h[i].occ_words = (S128(K) + 63) / 64;
                occs.push_back(occ);
            }
            max_M = std::max(max_M, M);
            max_K = std::max(max_K, K);
            max_N = std::max(max_N, N);
            keep.push_back(X);
            keep.push_back(W);
            outs.push_back(out);
        }
        host_descs = h;
        auto host = torch::empty({static_cast<int64_t>(count * sizeof(qgtc_problem))},
                                 torch::TensorOptions().dtype(torch::kUInt8));
        std::memcpy(host.data_ptr(), h.data(), count * sizeof(qgtc_problem));
        descs = host.to(dev);
        if (zero_jump) {
            stats = torch::empty({2}, torch::TensorOptions().dtype(torch::kInt64).device(dev));
            qgtc_problem *dp = reinterpret_cast<qgtc_problem *>(descs.data_ptr());
            uint64_t *sp = reinterpret_cast<uint64_t *>(stats.data_ptr<int64_t>());
            int rc;
            if (reuse_occs.empty()) {
                rc = qgtc_tile_occupancy_batched(dp, count, max_M, max_K, bit1, static_cast<float>(kJumpBelow), sp, current_stream(descs));
            } else {  // the bitmaps exist: only count and decide
                rc = qgtc_tile_occupancy_decide(dp, count, static_cast<float>(kJumpBelow), sp, current_stream(descs));
            }
            check_rc(rc, "BatchedGemm (tile occupancy)");
        }
    }

    // outcome of the on-device decision (waits for it)
    double occupied_fraction() const {
        if (!jump_asked) return 1.0;
        auto s = stats.cpu();
        const double set = static_cast<double>(s.data_ptr<int64_t>()[0]), all = static_cast<double>(s.data_ptr<int64_t>()[1]);
        return all > 0.0 ? set / all : 1.0;
    }
    bool zero_jump() const { return jump_asked && occupied_fraction() <= kJumpBelow; }

    // The reference's launch structure (one launch per cluster batch and operator) with the
    // independent batches spread over `n_streams` HIP streams: batch i always runs on stream
    // i % n_streams, so the stages of one batch stay ordered without any cross-stream dependency
    // while kernels of different batches overlap. The current stream waits for all of them.
    void run_per_problem(int n_streams) {
        TORCH_CHECK(n_streams >= 1 && n_streams <= 32, "n_streams must be in 1..32");
        c10::DeviceGuard guard(descs.device());
        const int dev = descs.get_device();
        auto cur = c10::hip::getCurrentHIPStream(dev);
        const auto pool = side_streams(dev, n_streams);
        ScopedEvent sev;
        hipEvent_t ev = sev.ev;
        TORCH_CHECK(hipEventRecord(ev, cur.stream()) == hipSuccess, "hipEventRecord failed");
        for (int s = 0; s < n_streams; s++)
            TORCH_CHECK(hipStreamWaitEvent(pool[s].stream(), ev, 0) == hipSuccess, "wait failed");
        for (int i = 0; i < count; i++) {
            const qgtc_problem &p = host_descs[i];
            void *st = static_cast<void *>(pool[i % n_streams].stream());
            int rc;
            if (mode == 2)
                rc = qgtc_bitmm2int(p.X, p.x_words, p.W, p.w_words, p.M, p.K, p.N, bit1, bit2,
                                    p.w_lines == P128(p.N), static_cast<float *>(p.out), outs[i].numel(), mm_flags(), st);
            else
                rc = qgtc_bitmm2bit(p.X, p.x_words, p.W, p.w_words, p.M, p.K, p.N, bit1, bit2, ob,
                                    static_cast<uint32_t *>(p.out), outs[i].numel(),
                                    mm_flags() | (mode == 1 ? QGTC_OUT_COLS : 0u), st);
            check_rc(rc, "BatchedGemm.run_per_problem");
        }
        for (int s = 0; s < n_streams; s++) {
            TORCH_CHECK(hipEventRecord(ev, pool[s].stream()) == hipSuccess, "hipEventRecord failed");
            TORCH_CHECK(hipStreamWaitEvent(cur.stream(), ev, 0) == hipSuccess, "wait failed");
        }
    }

    void run() {
        c10::DeviceGuard guard(descs.device());
        check_rc(qgtc_bitmm_batched(reinterpret_cast<const qgtc_problem *>(descs.data_ptr()), count,
                                    max_M, max_K, max_N, bit1, bit2, ob, mode,
                                    mm_flags() | (jump_asked ? QGTC_ZERO_JUMP : 0u),
                                    current_stream(descs)),
                 "BatchedGemm.run");
    }
};

// One quantised GNN layer for all cluster batches in ONE call (qgtc_gcn_layer_batched): stage1 = the grouped
// bitMM2Bit_col X.W (mode 1), stage2 = the grouped A.(XW) whose right operands ARE stage1's outputs.
struct FusedLayer {
    std::shared_ptr<BatchedGemm> s1, s2;

    FusedLayer(std::shared_ptr<BatchedGemm> stage1, std::shared_ptr<BatchedGemm> stage2)
        : s1(std::move(stage1)), s2(std::move(stage2)) {
        TORCH_CHECK(s1 && s2, "FusedLayer needs two BatchedGemm plans");
        TORCH_CHECK(s1->count == s2->count, "both stages must cover the same cluster batches");
        TORCH_CHECK(s1->mode == 1, "stage 1 must produce cols-layout bits (mode 1: it is stage 2's right operand)");
        TORCH_CHECK(s2->mode == 0 || s2->mode == 2, "stage 2 must produce rows-layout bits (mode 0) or float32 (mode 2)");
        TORCH_CHECK(s1->ob == s2->bit2, "stage 1's output bits must be stage 2's right-operand planes");
        TORCH_CHECK(s1->descs.device() == s2->descs.device(), "both stages must live on one device");
        for (int i = 0; i < s1->count; i++) {
            const qgtc_problem &a = s1->host_descs[i], &b = s2->host_descs[i];
            TORCH_CHECK(b.W == static_cast<const uint32_t *>(a.out), "stage 2's right operand ", i, " must be stage 1's output ", i);
            TORCH_CHECK(a.M == b.M && a.M == b.K && a.N == b.N, "batch ", i, ": stage shapes do not chain (n x f_in x f_out, then n x n x f_out)");
            TORCH_CHECK(b.w_lines == P128(b.N), "stage 2 reads a cols-layout operand with PAD128 lines");
        }
    }

    void run() {
        c10::DeviceGuard guard(s1->descs.device());
        const int max_M = std::max(s1->max_M, s2->max_M), max_N = std::max(s1->max_N, s2->max_N);
        const unsigned flags = mm_flags() | (s2->jump_asked ? QGTC_ZERO_JUMP : 0u);
        check_rc(qgtc_gcn_layer_batched(reinterpret_cast<const qgtc_problem *>(s1->descs.data_ptr()),
                                        reinterpret_cast<const qgtc_problem *>(s2->descs.data_ptr()), s1->count,
                                        max_M, s1->max_K, s2->max_K, max_N, s1->bit1, s1->bit2, s1->ob, s2->bit1, s2->ob, s2->mode,
                                        flags, current_stream(s1->descs)),
                 "FusedLayer.run");
    }
};

// An aggregation stage and the next layer's feature transform in one call (qgtc_gcn_chain_batched): stage_a produces
// rows-layout bits (mode 0) that are stage_xw's left operands; stage_xw produces cols-layout bits (mode 1).
struct ChainedPair {
    std::shared_ptr<BatchedGemm> sa, sx;
    bool discard = false;   // the aggregate itself is not wanted (QGTC_CHAIN_DISCARD: a hint)

    ChainedPair(std::shared_ptr<BatchedGemm> stage_a, std::shared_ptr<BatchedGemm> stage_xw, bool discard_)
        : sa(std::move(stage_a)), sx(std::move(stage_xw)), discard(discard_) {
        TORCH_CHECK(sa && sx, "ChainedPair needs two BatchedGemm plans");
        TORCH_CHECK(sa->count == sx->count, "both stages must cover the same cluster batches");
        TORCH_CHECK(sa->mode == 0, "the aggregation stage must produce rows-layout bits (mode 0)");
        TORCH_CHECK(sx->mode == 1 || sx->mode == 2, "the feature-transform stage must produce cols-layout bits (mode 1) or float32 (mode 2)");
        TORCH_CHECK(sa->ob == sx->bit1, "the first stage's output bits must be the second stage's left-operand planes");
        TORCH_CHECK(sa->descs.device() == sx->descs.device(), "both stages must live on one device");
        for (int i = 0; i < sa->count; i++) {
            const qgtc_problem &a = sa->host_descs[i], &b = sx->host_descs[i];
            TORCH_CHECK(b.X == static_cast<const uint32_t *>(a.out), "the second stage's left operand ", i, " must be the first stage's output ", i);
            TORCH_CHECK(a.M == b.M && a.N == b.K, "batch ", i, ": stage shapes do not chain (n x n x f, then n x f x f')");
        }
    }

    void run() {
        c10::DeviceGuard guard(sa->descs.device());
        check_rc(qgtc_gcn_chain_batched(reinterpret_cast<const qgtc_problem *>(sa->descs.data_ptr()),
                                        reinterpret_cast<const qgtc_problem *>(sx->descs.data_ptr()), sa->count,
                                        std::max(sa->max_M, sx->max_M), sa->max_K, sa->max_N, sx->max_N, sa->bit1, sa->bit2, sa->ob,
                                        sx->bit2, sx->ob, sx->mode, mm_flags() | (sa->jump_asked ? QGTC_ZERO_JUMP : 0u) | (discard ? QGTC_CHAIN_DISCARD : 0u),
                                        current_stream(sa->descs)),
                 "ChainedPair.run");
    }
};

// One quantised GNN layer on ONE subgraph in one call: requant(A . requant(X . W)) (QGTC_conv.py:14-22). bit_A: rows
// layout, 1.. planes, [n, n]; bit_X: rows layout [n, f_in]; bit_W: cols layout [f_in, f_out]. Returns the packed
// activations (rows layout, act_bit planes) or, with output = true, float32 [n, f_out]. Two fully asynchronous launches
// of the tuned single-problem kernels: no descriptors in device memory, no host synchronisation.
torch::Tensor gcn_layer(torch::Tensor bit_A, torch::Tensor bit_X, torch::Tensor bit_W, int n, int f_in, int f_out,
                        int a_bit, int act_bit, int w_bit, bool output) {
    CHECK_INPUT(bit_A);
    check_bits_tensor(bit_A, "bit_A");
    TORCH_CHECK(bit_A.device() == bit_X.device() && bit_A.device() == bit_W.device(), "all operands must share a device");
    TORCH_CHECK(n > 0 && f_in > 0 && f_out > 0, "bad dimensions");
    torch::Tensor T1 = mm2bit_impl(bit_X, bit_W, n, f_in, f_out, act_bit, w_bit, act_bit, true, "gcn_layer");
    if (output) return bitMM2Int(bit_A, T1, n, n, f_out, a_bit, act_bit, true);
    return mm2bit_impl(bit_A, T1, n, n, f_out, a_bit, act_bit, act_bit, false, "gcn_layer");
}


// val2bit of up to QGTC_MAX_PACK_JOBS matrices in ONE launch (qgtc_val2bit_batched): what an epoch does to its three weight
// matrices inside the clock (main_qgtc.py:100-110). Every output is exactly val2bit(inputs[i], nbits, col_major[i], output_layer[i]).
std::vector<torch::Tensor> val2bit_many(std::vector<torch::Tensor> inputs, int nbits, std::vector<bool> col_major,
                                        std::vector<bool> output_layer) {
    const int n = static_cast<int>(inputs.size());
    TORCH_CHECK(n >= 1 && n <= QGTC_MAX_PACK_JOBS, "val2bit_many takes 1..", QGTC_MAX_PACK_JOBS, " matrices");
    TORCH_CHECK(static_cast<int>(col_major.size()) == n && static_cast<int>(output_layer.size()) == n, "one col_major / output_layer flag per matrix");
    const auto dev = inputs[0].device();
    c10::DeviceGuard guard(dev);
    const auto opts = torch::TensorOptions().dtype(torch::kInt32).device(dev);
    qgtc_pack_job jobs[QGTC_MAX_PACK_JOBS];
    std::vector<torch::Tensor> outs;
    for (int i = 0; i < n; i++) {
        const torch::Tensor &x = inputs[i];
        CHECK_INPUT(x);
        TORCH_CHECK(x.scalar_type() == torch::kFloat32 && x.dim() == 2 && x.device() == dev, "inputs must be 2-D float32 tensors on one device");
        const int H = x.size(0), W = x.size(1);
        torch::Tensor out = col_major[i] ? torch::empty({static_cast<int64_t>(nbits) * S128(H) * 4, output_layer[i] ? P8(W) : P128(W)}, opts)   // QGTC_device.cu:83,97
                                         : torch::empty({static_cast<int64_t>(nbits) * P8(H), S128(W) * 4}, opts);                                // QGTC_device.cu:115
        jobs[i] = qgtc_pack_job{x.data_ptr<float>(), words_mut(out), static_cast<uint64_t>(out.numel()), H, W, nbits, col_major[i] ? 1 : 0,
                                output_layer[i] ? 1 : 0, 0};
        outs.push_back(out);
    }
    check_rc(qgtc_val2bit_batched(jobs, n, current_stream(inputs[0])), "val2bit_many");
    return outs;
}

// ---------------------------------------------------------------------------------------------
// EpochPlan: a grouped epoch whose descriptors are filled ON THE DEVICE (qgtc_epoch_plan_fill).
//   * the constructor is the DATA LOADER's part (beside ClusterIter's packing, sampler.py:92-105, outside the epoch clock):
//     one qgtc_batch per cluster batch on the device, the adjacency's occupancy bitmaps and the jump-or-not decision;
//   * bind() is what the epoch clock sees of the plan (main_qgtc.py:96 starts it before the weights exist): one pool
//     allocation, one descriptor allocation, ONE launch;
//   * run() issues the epoch's launches; outs(stage) makes the per-batch views on demand (never inside the clock).
// ---------------------------------------------------------------------------------------------
struct EpochPlan {
    std::vector<torch::Tensor> keep;    // packed batches, bitmaps
    torch::Tensor batches;              // device: qgtc_batch[count]
    torch::Tensor stats;                // device: [occupied, all] tiles of the adjacencies
    std::vector<int32_t> nodes;
    int count = 0, max_n = 0, a_bits = 1;
    bool jumping = false;
    bool x_chain = false;   // the batches' X also exists in the chain format (QGTC_SRC_XC)
    bool a_tiles = false;   // the batches' adjacency also exists in the tile format of qgtc_chain_aggregate (QGTC_SRC_AT)
    double occupied = 1.0;
    // bound state
    std::vector<qgtc_stage> stages;
    std::vector<torch::Tensor> weights;
    torch::Tensor pool, descs;
    struct Launch {
        int kind;        // 0 grouped GEMM, 1 chained pair (qgtc_gcn_chain_batched), 2 layer (qgtc_gcn_layer_batched, two launches),
                         // 3 qgtc_chain_transform (s1), 4 qgtc_chain_aggregate (s1, and s2 unless it is the float32 aggregation: s2 < 0)
        int s1, s2;
        unsigned extra;  // QGTC_CHAIN_* flags
        int codes;       // kinds 3 / 4: index of the pre-expanded weight (weight_codes)
    };
    std::vector<torch::Tensor> weight_codes;   // qgtc_expand_weights outputs, made by bind()
    std::vector<torch::Tensor> As, Xs, Xrs;    // per-batch views of the packed batches (the loader route: views into pools), made on first use
    torch::Tensor pool_a, pool_x, pool_xr;     // the loader route's pools
    std::vector<int64_t> off_a, off_x, off_xr;
    int feat_cols = 0, feat_bits = 0;
    void make_views() {   // 3 x count narrow + view calls: ~1 us each on the host - not inside a pack that an epoch clock may see
        if (!As.empty() || !pool_a.defined()) return;
        for (int i = 0; i < count; i++) {
            const int n = nodes[i];
            As.push_back(pool_a.narrow(0, off_a[i], off_a[i + 1] - off_a[i]).view({P8(n), S128(n) * 4}));                                              // QGTC_device.cu:115
            Xs.push_back(pool_x.narrow(0, off_x[i], off_x[i + 1] - off_x[i]).view({static_cast<int64_t>(feat_bits) * S128(n) * 4, P128(feat_cols)}));   // QGTC_device.cu:97
            if (pool_xr.defined())
                Xrs.push_back(pool_xr.narrow(0, off_xr[i], static_cast<int64_t>(qgtc_rows_words(n, feat_cols, feat_bits))).view({static_cast<int64_t>(feat_bits) * P8(n), S128(feat_cols) * 4}));
        }
    }
    std::vector<qgtc_batch> host_batches;      // the per-batch table as uploaded (format_of)
    EpochPlan() = default;

    // one batch's operand in a loader format, as a flat non-owning tensor (int32 words; the bitmap as int64): for inspection / tests.
    // which: QGTC_SRC_A / _X / _XR / _XC / _AT, or -1 = the occupancy bitmap. Valid while the plan lives.
    torch::Tensor format_of(int i, int which) const {
        TORCH_CHECK(i >= 0 && i < count && static_cast<int>(host_batches.size()) == count, "no such batch");
        const qgtc_batch &b = host_batches[i];
        const auto dev = batches.device();
        if (which < 0) {
            TORCH_CHECK(b.occ != nullptr, "no occupancy bitmap");
            return torch::from_blob(const_cast<uint64_t *>(b.occ), {static_cast<int64_t>(qgtc_occupancy_words(b.n, b.n))},
                                    torch::TensorOptions().dtype(torch::kInt64).device(dev));
        }
        const qgtc_operand &o = which == QGTC_SRC_A ? b.A : (which == QGTC_SRC_X ? b.X : (which == QGTC_SRC_XR ? b.XR : (which == QGTC_SRC_XC ? b.XC : b.AT)));
        TORCH_CHECK(o.ptr != nullptr, "the plan does not hold this format");
        return torch::from_blob(const_cast<uint32_t *>(o.ptr), {static_cast<int64_t>(o.words)}, torch::TensorOptions().dtype(torch::kInt32).device(dev));
    }
    std::vector<Launch> launches;
    std::vector<uint64_t> offsets;      // lazily: word offset of every (stage, batch) output in the pool
    bool fill_checked = false;          // the plan-fill kernel's violation record has been read for this bind (outs(): after the clock)
    static constexpr double kJumpBelow = BatchedGemm::kJumpBelow;

    // x_chain_bits > 0: every X (cols layout [n, x_cols], x_chain_bits planes) is also kept in the chain format of the
    // chain entries (qgtc_chain_from_cols) - for epochs whose FIRST product is an aggregation A . X (Batched-GIN)
    // a_tiles_: every one-plane adjacency is also kept as 512-byte tiles (qgtc_adj_tiles_from_rows) for the aggregation launches of
    // the chain entries - one pooled allocation, converted here, beside the packing
    EpochPlan(std::vector<torch::Tensor> As, std::vector<torch::Tensor> Xs, std::vector<torch::Tensor> Xrs, std::vector<int> ns,
              int a_bits_, bool zero_jump, int x_chain_bits, int x_cols, bool a_tiles_) : a_bits(a_bits_) {
        count = static_cast<int>(As.size());
        x_chain = x_chain_bits > 0;
        a_tiles = a_tiles_ && a_bits_ == 1;
        TORCH_CHECK(count > 0 && count <= 65535, "1..65535 cluster batches");
        TORCH_CHECK(static_cast<int>(Xs.size()) == count && static_cast<int>(ns.size()) == count, "one X and one node count per batch");
        TORCH_CHECK(Xrs.empty() || static_cast<int>(Xrs.size()) == count, "Xrs: none, or one per batch");
        const auto dev = As[0].device();
        c10::DeviceGuard guard(dev);
        std::vector<qgtc_batch> h(count);
        std::vector<int64_t> occ_off(count + 1, 0);
        for (int i = 0; i < count; i++) {
            TORCH_CHECK(ns[i] > 0, "bad node count");
            occ_off[i + 1] = occ_off[i] + static_cast<int64_t>(qgtc_occupancy_words(ns[i], ns[i]));
        }
        torch::Tensor occ_pool;
        if (zero_jump) occ_pool = torch::empty({occ_off[count]}, torch::TensorOptions().dtype(torch::kInt64).device(dev));
        auto operand = [&](const torch::Tensor &t, const char *name) {
            CHECK_INPUT(t);
            check_bits_tensor(t, name);
            TORCH_CHECK(t.device() == dev, "all batches must live on one device");
            TORCH_CHECK(t.numel() < (1LL << 30), "packed operand too large (>= 4 GiB)");
            TORCH_CHECK((reinterpret_cast<uintptr_t>(t.data_ptr()) & 15) == 0, "packed operands must be 16-byte aligned");
            keep.push_back(t);
            return qgtc_operand{words(t), static_cast<uint64_t>(t.numel())};
        };
        std::vector<qgtc_problem> tmp(count);   // the adjacencies as left operands, for the bitmap launch
        torch::Tensor tile_pool;
        std::vector<int64_t> tile_off(count + 1, 0);
        if (a_tiles) {
            for (int i = 0; i < count; i++) tile_off[i + 1] = tile_off[i] + static_cast<int64_t>(qgtc_adj_tiles_words(ns[i], ns[i]));
            tile_pool = torch::empty({tile_off[count]}, torch::TensorOptions().dtype(torch::kInt32).device(dev));
            keep.push_back(tile_pool);
        }
        for (int i = 0; i < count; i++) {
            h[i].A = operand(As[i], "A");
            h[i].AT = qgtc_operand{nullptr, 0};
            if (a_tiles) {
                uint32_t *tp = words_mut(tile_pool) + tile_off[i];
                const size_t tw = static_cast<size_t>(tile_off[i + 1] - tile_off[i]);
                check_rc(qgtc_adj_tiles_from_rows(h[i].A.ptr, h[i].A.words, ns[i], ns[i], tp, tw, current_stream(tile_pool)), "EpochPlan (adjacency tiles)");
                h[i].AT = qgtc_operand{tp, static_cast<uint64_t>(tw)};
            }
            h[i].X = operand(Xs[i], "X");
            h[i].XR = Xrs.empty() ? qgtc_operand{nullptr, 0} : operand(Xrs[i], "Xr");
            h[i].XC = qgtc_operand{nullptr, 0};
            if (x_chain_bits > 0) {
                TORCH_CHECK(x_cols > 0 && x_chain_bits <= 8, "x_chain_bits in 1..8 and the feature count");
                torch::Tensor xc = torch::empty({static_cast<int64_t>(qgtc_chain_words(ns[i], x_cols, x_chain_bits))}, torch::TensorOptions().dtype(torch::kInt32).device(dev));
                check_rc(qgtc_chain_from_cols(words(Xs[i]), Xs[i].numel(), ns[i], x_cols, x_chain_bits, words_mut(xc), xc.numel(), current_stream(xc)), "EpochPlan (X in the chain format)");
                h[i].XC = qgtc_operand{words(xc), static_cast<uint64_t>(xc.numel())};
                keep.push_back(xc);
            }
            h[i].n = ns[i];
            h[i].occ = nullptr;
            h[i].occ_words = 0;
            if (zero_jump) {
                h[i].occ = reinterpret_cast<const uint64_t *>(occ_pool.data_ptr<int64_t>() + occ_off[i]);
                h[i].occ_words = (S128(ns[i]) + 63) / 64;
            }
            tmp[i] = qgtc_problem{h[i].A.ptr, h[i].A.ptr, nullptr, h[i].A.words, h[i].A.words, ns[i], ns[i], 1, 128, h[i].occ_words, h[i].occ};
            max_n = std::max(max_n, ns[i]);
            nodes.push_back(ns[i]);
        }
        if (zero_jump) {
            // bitmaps of all adjacencies in one launch; whether jumping pays (a quarter of the tiles occupied at most) is read
            // back HERE, beside the packing - the plan's descriptors then carry the bitmaps or do not
            keep.push_back(occ_pool);
            auto th = torch::empty({static_cast<int64_t>(count * sizeof(qgtc_problem))}, torch::TensorOptions().dtype(torch::kUInt8));
            std::memcpy(th.data_ptr(), tmp.data(), count * sizeof(qgtc_problem));
            torch::Tensor td = th.to(dev);
            stats = torch::empty({2}, torch::TensorOptions().dtype(torch::kInt64).device(dev));
            check_rc(qgtc_tile_occupancy_batched(reinterpret_cast<qgtc_problem *>(td.data_ptr()), count, max_n, max_n, a_bits, 2.0f,
                                                 reinterpret_cast<uint64_t *>(stats.data_ptr<int64_t>()), current_stream(td)),
                     "EpochPlan (tile occupancy)");
            auto sh = stats.cpu();
            const double set = static_cast<double>(sh.data_ptr<int64_t>()[0]), all = static_cast<double>(sh.data_ptr<int64_t>()[1]);
            occupied = all > 0.0 ? set / all : 1.0;
            // `jumping`: the rule measured for the tile kernels (a quarter of the tiles occupied at most: at 43 % they lose 15 %). The
            // table keeps the bitmaps either way - the chain entries' aggregations (one wave per row block) gain from them at any
            // occupancy (ppi-sized batches, 42 % occupied: 15.9 -> 14.7 us per epoch) - and bind() decides per stage.
            jumping = occupied <= kJumpBelow;
        }
        auto host = torch::empty({static_cast<int64_t>(count * sizeof(qgtc_batch))}, torch::TensorOptions().dtype(torch::kUInt8));
        std::memcpy(host.data_ptr(), h.data(), count * sizeof(qgtc_batch));
        batches = host.to(dev);
        host_batches = h;
    }


    // The data loader's whole job in one call (qgtc_load_batches): `count` cluster batches packed from their concatenated edge
    // lists (indices local to each batch) and feature rows - adjacency rows layout + tiles + bitmaps, X in the cols layout
    // (+ rows layout, + chain format) - six launches and three uploads for the iterator instead of eight launches per batch.
    // What sampler.py:76-106 does per batch; the per-batch tensors (.As / .Xs / .Xrs) are views into the pools.
    static std::shared_ptr<EpochPlan> load(torch::Tensor src, torch::Tensor dst, std::vector<int64_t> edge_counts, torch::Tensor feats,
                                           std::vector<int> ns, int x_bits, bool with_rows, int x_chain_bits, bool a_tiles_, bool validate) {
        CHECK_INPUT(src);
        CHECK_INPUT(dst);
        CHECK_INPUT(feats);
        TORCH_CHECK(src.scalar_type() == torch::kInt64 && dst.scalar_type() == torch::kInt64 && src.dim() == 1 && src.sizes() == dst.sizes(),
                    "src and dst must be equally long 1-D int64 tensors");
        TORCH_CHECK(feats.scalar_type() == torch::kFloat32 && feats.dim() == 2 && feats.size(1) > 0, "feats must be a 2-D float32 tensor");
        TORCH_CHECK(src.device() == feats.device() && dst.device() == feats.device(), "src, dst and feats must share a device");
        const int count = static_cast<int>(ns.size());
        TORCH_CHECK(count > 0 && count <= 65535 && static_cast<int>(edge_counts.size()) == count, "1..65535 cluster batches, one edge count each");
        TORCH_CHECK(x_bits >= 1 && x_bits <= 32 && (x_chain_bits == 0 || (x_chain_bits == x_bits && x_bits <= 8)), "bad bit widths");
        const int F = static_cast<int>(feats.size(1));
        const auto dev = feats.device();
        c10::DeviceGuard guard(dev);
        auto plan = std::make_shared<EpochPlan>();
        EpochPlan &P = *plan;
        P.count = count;
        P.a_bits = 1;
        P.x_chain = x_chain_bits > 0;
        P.a_tiles = a_tiles_;
        auto r4 = [](int64_t v) { return (v + 3) & ~int64_t(3); };
        std::vector<int64_t> a_off(count + 1, 0), t_off(count + 1, 0), o_off(count + 1, 0), x_off(count + 1, 0), xr_off(count + 1, 0), xc_off(count + 1, 0),
            e_off(count + 1, 0), f_off(count + 1, 0);
        int64_t max_e = 0;
        for (int i = 0; i < count; i++) {
            const int n = ns[i];
            TORCH_CHECK(n > 0 && edge_counts[i] >= 0, "bad node / edge count");
            a_off[i + 1] = a_off[i] + static_cast<int64_t>(qgtc_rows_words(n, n, 1));
            t_off[i + 1] = t_off[i] + (a_tiles_ ? static_cast<int64_t>(qgtc_adj_tiles_words(n, n)) : 0);
            o_off[i + 1] = o_off[i] + static_cast<int64_t>(qgtc_occupancy_words(n, n));
            x_off[i + 1] = x_off[i] + static_cast<int64_t>(qgtc_cols_words(n, F, x_bits, 0));
            xr_off[i + 1] = xr_off[i] + (with_rows ? r4(static_cast<int64_t>(qgtc_rows_words(n, F, x_bits))) : 0);
            xc_off[i + 1] = xc_off[i] + (x_chain_bits ? static_cast<int64_t>(qgtc_chain_words(n, F, x_chain_bits)) : 0);
            e_off[i + 1] = e_off[i] + edge_counts[i];
            f_off[i + 1] = f_off[i] + n;
            max_e = std::max<int64_t>(max_e, edge_counts[i]);
            P.max_n = std::max(P.max_n, n);
            P.nodes.push_back(n);
        }
        TORCH_CHECK(e_off[count] == src.numel(), "the edge counts must add up to len(src)");
        TORCH_CHECK(f_off[count] == feats.size(0), "the node counts must add up to feats.size(0)");
        TORCH_CHECK(a_off[count] < (1LL << 40), "adjacency pool too large");
        const auto i32 = torch::TensorOptions().dtype(torch::kInt32).device(dev);
        // [A of every batch | stats (two uint64) | scratch of every batch]. With a work buffer (qgtc_load_work_words > 0: the bucketed route,
        // batches of at most 5120 nodes) the call writes every word of A itself: only `stats` is cleared and there is no scratch.
        const size_t work_words = qgtc_load_work_words(count, P.max_n, static_cast<uint64_t>(e_off[count]));   // (0: the bitmap route - the library decides)
        const bool bucketed = work_words > 0;
        const int64_t zwords = (bucketed ? 1 : 3) * a_off[count] + 4;
        torch::Tensor zero = torch::empty({zwords}, i32);
        torch::Tensor work = torch::empty({static_cast<int64_t>(std::max<size_t>(work_words, 4))}, i32);
        torch::Tensor tiles = torch::empty({std::max<int64_t>(t_off[count], 4)}, i32);
        torch::Tensor occ = torch::empty({std::max<int64_t>(o_off[count], 1)}, torch::TensorOptions().dtype(torch::kInt64).device(dev));
        torch::Tensor xp = torch::empty({x_off[count]}, i32), xrp = torch::empty({std::max<int64_t>(xr_off[count], 4)}, i32),
                      xcp = torch::empty({std::max<int64_t>(xc_off[count], 4)}, i32);
        torch::Tensor bad;
        if (validate) bad = torch::empty({1}, i32);
        uint32_t *zp = words_mut(zero);
        uint64_t *stats = reinterpret_cast<uint64_t *>(zp + a_off[count]);
        uint32_t *scratch0 = zp + a_off[count] + 4;
        std::vector<qgtc_loader_batch> lt(count);
        std::vector<qgtc_batch> h(count);
        for (int i = 0; i < count; i++) {
            const int n = ns[i];
            qgtc_loader_batch &b = lt[i];
            b.edge_off = static_cast<uint64_t>(e_off[i]);
            b.n_edges = static_cast<uint64_t>(edge_counts[i]);
            b.feat_row = static_cast<uint64_t>(f_off[i]);
            b.n = n;
            b.reserved = 0;
            b.A = zp + a_off[i];
            b.scratch = bucketed ? nullptr : scratch0 + 2 * a_off[i];
            b.AT = a_tiles_ ? words_mut(tiles) + t_off[i] : nullptr;
            b.occ = reinterpret_cast<uint64_t *>(occ.data_ptr<int64_t>()) + o_off[i];
            b.X = words_mut(xp) + x_off[i];
            b.XR = with_rows ? words_mut(xrp) + xr_off[i] : nullptr;
            b.XC = x_chain_bits ? words_mut(xcp) + xc_off[i] : nullptr;
            h[i].A = qgtc_operand{b.A, static_cast<uint64_t>(a_off[i + 1] - a_off[i])};
            h[i].X = qgtc_operand{b.X, static_cast<uint64_t>(x_off[i + 1] - x_off[i])};
            h[i].XR = with_rows ? qgtc_operand{b.XR, static_cast<uint64_t>(qgtc_rows_words(n, F, x_bits))} : qgtc_operand{nullptr, 0};
            h[i].XC = x_chain_bits ? qgtc_operand{b.XC, static_cast<uint64_t>(xc_off[i + 1] - xc_off[i])} : qgtc_operand{nullptr, 0};
            h[i].AT = a_tiles_ ? qgtc_operand{b.AT, static_cast<uint64_t>(t_off[i + 1] - t_off[i])} : qgtc_operand{nullptr, 0};
            h[i].occ = b.occ;
            h[i].n = n;
            h[i].occ_words = (S128(n) + 63) / 64;
        }
        // both tables in ONE upload: [qgtc_loader_batch x count | qgtc_batch x count]
        const int64_t lt_bytes = static_cast<int64_t>(count * sizeof(qgtc_loader_batch)), bt_bytes = static_cast<int64_t>(count * sizeof(qgtc_batch));
        static_assert(sizeof(qgtc_loader_batch) % 8 == 0, "the second table starts 8-byte aligned");
        auto host = torch::empty({lt_bytes + bt_bytes}, torch::TensorOptions().dtype(torch::kUInt8));
        std::memcpy(host.data_ptr(), lt.data(), lt_bytes);
        std::memcpy(static_cast<char *>(host.data_ptr()) + lt_bytes, h.data(), bt_bytes);
        torch::Tensor tables = host.to(dev);
        P.batches = tables.narrow(0, lt_bytes, bt_bytes);
        const unsigned formats = (with_rows ? QGTC_LOAD_X_ROWS : 0u) | (x_chain_bits ? QGTC_LOAD_X_CHAIN : 0u);
        check_rc(qgtc_load_batches(reinterpret_cast<const qgtc_loader_batch *>(tables.data_ptr()), count, P.max_n, static_cast<uint64_t>(max_e),
                                   src.numel() ? src.data_ptr<int64_t>() : nullptr, src.numel() ? dst.data_ptr<int64_t>() : nullptr,
                                   feats.data_ptr<float>(), F, x_bits, bucketed ? static_cast<void *>(stats) : static_cast<void *>(zp),
                                   bucketed ? 16u : static_cast<size_t>(zwords) * 4u, stats, validate ? bad.data_ptr<int>() : nullptr, formats,
                                   bucketed ? words_mut(work) : nullptr, work_words, current_stream(zero)),
                 "EpochPlan.load");
        // the occupied-tile count comes back with the (optional) index check: ONE read-back, beside the packing (outside any epoch clock)
        torch::Tensor sh = zero.narrow(0, a_off[count], 4).cpu();
        if (validate) TORCH_CHECK(bad.item<int>() == 0, "edge index out of range");
        uint64_t set = 0;
        std::memcpy(&set, sh.data_ptr(), sizeof(set));
        double all = 0.0;
        for (int i = 0; i < count; i++) all += static_cast<double>((ns[i] + 31) / 32) * S128(ns[i]);
        P.occupied = all > 0.0 ? static_cast<double>(set) / all : 1.0;
        P.jumping = P.occupied <= kJumpBelow;
        P.pool_a = zero;
        P.pool_x = xp;
        if (with_rows) P.pool_xr = xrp;
        P.off_a = a_off;
        P.off_x = x_off;
        P.off_xr = xr_off;
        P.feat_cols = F;
        P.feat_bits = x_bits;
        P.keep = {zero, tiles, occ, xp, xrp, xcp, tables, src, dst, feats};
        P.host_batches = h;
        return plan;
    }

    // stages: (left, right, K, N, bit1, bit2, ob, mode, pad128, use_occ) per operator; launches: (kind, s1, s2, extra flags)
    // expand: (weight index, K, N, nbits, order) per pre-expanded weight the launches of kinds 3 / 4 name
    void bind(std::vector<torch::Tensor> weights_, std::vector<std::array<int, 11>> stages_, std::vector<std::array<int, 5>> launches_,
              std::vector<std::array<int, 5>> expand) {
        c10::DeviceGuard guard(batches.device());
        const int ns = static_cast<int>(stages_.size()), nw = static_cast<int>(weights_.size());
        TORCH_CHECK(ns >= 1 && ns <= QGTC_MAX_STAGES && nw <= QGTC_MAX_WEIGHTS, "too many stages / weights");
        stages.clear();
        for (const auto &t : stages_)
            stages.push_back(qgtc_stage{t[0], t[1], t[2], t[3], t[4], t[5], t[6], t[7], t[8], t[9], t[10]});
        qgtc_operand wops[QGTC_MAX_WEIGHTS];
        for (int k = 0; k < nw; k++) {
            const torch::Tensor &w = weights_[k];
            CHECK_INPUT(w);
            check_bits_tensor(w, "weight");
            TORCH_CHECK(w.device() == batches.device(), "weights must live on the batches' device");
            wops[k] = qgtc_operand{words(w), static_cast<uint64_t>(w.numel())};
        }
        weights = std::move(weights_);
        launches.clear();
        const auto dev = batches.device();
        weight_codes.clear();
        if (!expand.empty()) {   // every pre-expanded weight of the plan in ONE launch
            TORCH_CHECK(expand.size() <= QGTC_MAX_WEIGHTS, "too many pre-expanded weights");
            qgtc_expand_job jobs[QGTC_MAX_WEIGHTS];
            for (size_t i = 0; i < expand.size(); i++) {
                const auto &e = expand[i];
                TORCH_CHECK(e[0] >= 0 && e[0] < nw, "bad weight index");
                const torch::Tensor &w = weights[e[0]];
                const int64_t cap = static_cast<int64_t>(qgtc_weight_codes_words(e[1], e[2], e[3], e[4]));
                TORCH_CHECK(cap > 0, "EpochPlan.bind: bad pre-expanded weight (K, N, bits, order)");
                torch::Tensor codes = torch::empty({cap}, torch::TensorOptions().dtype(torch::kInt32).device(dev));
                jobs[i] = qgtc_expand_job{words(w), words_mut(codes), static_cast<uint64_t>(w.numel()), e[1], e[2], e[3],
                                          static_cast<int32_t>(w.numel() / (static_cast<int64_t>(e[3]) * S128(e[1]) * 4)), e[4], static_cast<uint32_t>(cap)};
                weight_codes.push_back(codes);
            }
            check_rc(qgtc_expand_weights(jobs, static_cast<int>(expand.size()), current_stream(batches)), "EpochPlan.bind (weights)");
        }
        if (!jumping)   // bitmaps only for the stages a chain-entry aggregation (kind 4) runs
            for (int s = 0; s < ns; s++) {
                bool rbw = false;
                for (const auto &l : launches_) rbw = rbw || (l[0] == 4 && l[1] == s);
                if (!rbw) stages[s].use_occ = 0;
            }
        for (const auto &l : launches_) {
            TORCH_CHECK(l[0] >= 0 && l[0] <= 4 && l[1] >= 0 && l[1] < ns, "bad launch entry");
            TORCH_CHECK(l[0] == 0 || l[0] == 3 || (l[0] == 4 && l[2] < 0) || (l[2] >= 0 && l[2] < ns), "bad launch entry");
            TORCH_CHECK(l[0] < 3 || (l[0] == 4 && l[2] < 0) || (l[4] >= 0 && l[4] < static_cast<int>(weight_codes.size())), "bad weight-codes index");
            launches.push_back(Launch{l[0], l[1], l[2], static_cast<unsigned>(l[3]), l[4]});
        }
        offsets.clear();
        fill_checked = false;
        const size_t pool_words = qgtc_epoch_pool_layout(nodes.data(), count, stages.data(), ns, nullptr);
        TORCH_CHECK(pool_words > 0 && pool_words < (1ull << 40), "bad pool size");
        pool = torch::empty({static_cast<int64_t>(pool_words)}, torch::TensorOptions().dtype(torch::kInt32).device(dev));
        descs = torch::empty({static_cast<int64_t>(ns) * count * static_cast<int64_t>(sizeof(qgtc_problem))}, torch::TensorOptions().dtype(torch::kUInt8).device(dev));
        check_rc(qgtc_epoch_plan_fill(reinterpret_cast<const qgtc_batch *>(batches.data_ptr()), count, stages.data(), ns, wops, nw,
                                      pool.data_ptr(), pool_words, reinterpret_cast<qgtc_problem *>(descs.data_ptr()), current_stream(pool)),
                 "EpochPlan.bind");
    }

    const qgtc_problem *stage_descs(int s) const { return reinterpret_cast<const qgtc_problem *>(descs.data_ptr()) + static_cast<size_t>(s) * count; }
    int dimK(const qgtc_stage &st) const { return st.K == QGTC_DIM_NODES ? max_n : st.K; }

    void run_launch(const Launch &l, unsigned check) {
        const unsigned base = mm_flags() | check;
        void *st = current_stream(descs);
        if (l.kind == 0) {
            const qgtc_stage &a = stages[l.s1];
            check_rc(qgtc_bitmm_batched(stage_descs(l.s1), count, max_n, dimK(a), a.N, a.bit1, a.bit2, a.ob, a.mode,
                                        base | ((a.use_occ && jumping) ? QGTC_ZERO_JUMP : 0u) | l.extra, st), "EpochPlan.run (grouped GEMM)");
        } else if (l.kind == 1) {
            const qgtc_stage &a = stages[l.s1], &x = stages[l.s2];
            check_rc(qgtc_gcn_chain_batched(stage_descs(l.s1), stage_descs(l.s2), count, max_n, dimK(a), a.N, x.N, a.bit1, a.bit2, a.ob, x.bit2, x.ob,
                                            x.mode, base | ((a.use_occ && jumping) ? QGTC_ZERO_JUMP : 0u) | l.extra, st), "EpochPlan.run (chained pair)");
        } else if (l.kind == 3) {
            TORCH_CHECK(base & (QGTC_ENGINE_AUTO | QGTC_ENGINE_MFMA), "EpochPlan: this plan was bound for the matrix-core chain entries (T in their "
                        "private format); after set_engine('popcount') bind it again");
            const qgtc_stage &a = stages[l.s1];
            check_rc(qgtc_chain_transform(stage_descs(l.s1), count, max_n, dimK(a), a.N, a.bit1, a.ob, words(weight_codes[l.codes]), check, st),
                     "EpochPlan.run (chain transform)");
        } else if (l.kind == 4) {
            TORCH_CHECK(base & (QGTC_ENGINE_AUTO | QGTC_ENGINE_MFMA), "EpochPlan: this plan was bound for the matrix-core chain entries (T in their "
                        "private format); after set_engine('popcount') bind it again");
            const qgtc_stage &a = stages[l.s1];
            const unsigned tiles = a.left == QGTC_SRC_AT ? QGTC_CHAIN_ADJ_TILES : 0u;
            if (l.s2 < 0) {
                check_rc(qgtc_chain_aggregate(stage_descs(l.s1), nullptr, count, max_n, dimK(a), a.N, 0, a.bit2, 0, 0, 0, nullptr, check | tiles, st),
                         "EpochPlan.run (chain aggregate, float32)");
            } else {
                const qgtc_stage &x = stages[l.s2];
                check_rc(qgtc_chain_aggregate(stage_descs(l.s1), stage_descs(l.s2), count, max_n, dimK(a), a.N, x.N, a.bit2, a.ob, x.ob, x.mode,
                                              words(weight_codes[l.codes]), check | tiles, st), "EpochPlan.run (chain aggregate)");
            }
        } else {
            const qgtc_stage &a = stages[l.s1], &b = stages[l.s2];
            check_rc(qgtc_gcn_layer_batched(stage_descs(l.s1), stage_descs(l.s2), count, max_n, dimK(a), dimK(b), std::max(a.N, b.N), a.bit1, a.bit2, a.ob,
                                            b.bit1, b.ob, b.mode, base | ((b.use_occ && jumping) ? QGTC_ZERO_JUMP : 0u), st), "EpochPlan.run (layer)");
        }
    }

    void run() {
        TORCH_CHECK(descs.defined(), "EpochPlan.run before bind");
        c10::DeviceGuard guard(descs.device());
        for (const Launch &l : launches) run_launch(l, 0u);
    }

    // one epoch with QGTC_CHECK_DESCRIPTORS on every launch; raises if a descriptor breaks a grouped entry's preconditions
    void run_checked() {
        TORCH_CHECK(descs.defined(), "EpochPlan.run before bind");
        c10::DeviceGuard guard(descs.device());
        for (const Launch &l : launches) run_launch(l, QGTC_CHECK_DESCRIPTORS);
        int problem = -1, field = 0;
        const int rc = qgtc_last_batched_violation(&problem, &field, current_stream(descs));
        TORCH_CHECK(rc == QGTC_OK, "EpochPlan: descriptor ", problem, " violates a grouped launch's preconditions (field ", field, ")");
    }

    // the per-batch outputs of one stage: views into the pool, made on demand
    std::vector<torch::Tensor> outs(int s) {
        TORCH_CHECK(descs.defined() && s >= 0 && s < static_cast<int>(stages.size()), "no such stage");
        if (!fill_checked) {   // what the fill kernel recorded (an undersized pool, a batch without nodes): read ONCE per bind, here - outs() is
            fill_checked = true;   // called after the epoch clock, and it waits for the stream anyway
            c10::DeviceGuard guard(descs.device());
            int problem = -1, field = 0;
            const int rc = qgtc_last_batched_violation(&problem, &field, current_stream(descs));
            TORCH_CHECK(rc == QGTC_OK, "EpochPlan: batch ", problem, " could not be planned (field ", field, ": pool too small or no nodes)");
        }
        if (offsets.empty()) {
            offsets.resize(stages.size() * static_cast<size_t>(count));
            qgtc_epoch_pool_layout(nodes.data(), count, stages.data(), static_cast<int>(stages.size()), offsets.data());
        }
        const qgtc_stage &st = stages[s];
        TORCH_CHECK(st.fmt == 0, "stage ", s, " is kept in the chain's private format");
        std::vector<torch::Tensor> v;
        for (int b = 0; b < count; b++) {
            const int64_t off = static_cast<int64_t>(offsets[static_cast<size_t>(s) * count + b]);
            const int n = nodes[b];
            if (st.mode == 2) v.push_back(pool.narrow(0, off, static_cast<int64_t>(n) * st.N).view(torch::kFloat32).view({n, st.N}));
            else if (st.mode == 1) v.push_back(pool.narrow(0, off, static_cast<int64_t>(st.ob) * S128(n) * 4 * P128(st.N)).view({static_cast<int64_t>(st.ob) * S128(n) * 4, P128(st.N)}));
            else v.push_back(pool.narrow(0, off, static_cast<int64_t>(st.ob) * P8(n) * S128(st.N) * 4).view({static_cast<int64_t>(st.ob) * P8(n), S128(st.N) * 4}));
        }
        return v;
    }
};


// ---------------------------------------------------------------------------------------------
// The lean call path of the four operators a driver calls per cluster batch (main_qgtc.py:128-154: six calls per batch,
// 450 per epoch, each a 3 - 4 us kernel - the host side of a call IS the epoch time of an unchanged driver). The pybind11
// functions above stay the reference for semantics and error messages; these METH_FASTCALL entries take the public names
// and do the same work without pybind11's argument casters, the ATen dispatcher (at::detail::empty_cuda allocates from
// the same caching allocator) and the device guard when the operands already live on the current device. Anything that is
// not the plain good case - keyword arguments, a wrong type, a failed check, another device, an error code - is handed to
// the pybind11 function, which raises exactly what it always raised. Measured split of a call: DESIGN.md section 6.
// ---------------------------------------------------------------------------------------------
namespace lean {

PyObject *slow_val2bit = nullptr, *slow_mm2bit = nullptr, *slow_mm2bit_col = nullptr, *slow_mm2int = nullptr;

inline bool as_int(PyObject *o, int *v) {
    if (!PyLong_CheckExact(o)) return false;   // (bools, floats, numpy scalars: the pybind11 path decides)
    int overflow = 0;
    const long x = PyLong_AsLongAndOverflow(o, &overflow);
    if (overflow || x < INT32_MIN || x > INT32_MAX) return false;
    *v = static_cast<int>(x);
    return true;
}
inline bool as_bool(PyObject *o, bool *v) {
    if (o == Py_True) { *v = true; return true; }
    if (o == Py_False) { *v = false; return true; }
    return false;
}
// a contiguous device tensor of the given dtype
inline const at::Tensor *operand(PyObject *o, c10::ScalarType dtype) {
    if (!THPVariable_Check(o)) return nullptr;
    const at::Tensor &t = THPVariable_Unpack(o);
    if (!t.defined() || !t.is_cuda() || t.scalar_type() != dtype || !t.is_contiguous()) return nullptr;
    return &t;
}
// both operands on the CURRENT device (asked only once a device tensor is in hand: without a GPU the query itself throws)
inline bool on_current(const at::Tensor &a, const at::Tensor *b, c10::DeviceIndex *dev) {
    *dev = c10::hip::current_device();
    return a.get_device() == *dev && (!b || b->get_device() == *dev);
}
inline at::Tensor fresh(std::array<int64_t, 2> shape, c10::ScalarType dtype, c10::DeviceIndex dev) {
    return at::Tensor(at::detail::empty_cuda(c10::IntArrayRef(shape.data(), 2), dtype, c10::Device(c10::DeviceType::CUDA, dev), std::nullopt));
}

// kind 0 bitMM2Bit, 1 bitMM2Bit_col, 2 bitMM2Int
template <int KIND>
PyObject *mm(PyObject *, PyObject *const *args, Py_ssize_t nargs, PyObject *kwnames) {
    PyObject *slow = KIND == 0 ? slow_mm2bit : (KIND == 1 ? slow_mm2bit_col : slow_mm2int);
    const Py_ssize_t n = PyVectorcall_NARGS(nargs);
    int M, K, N, b1, b2, ob = 1;
    bool pad = false;
    if (kwnames == nullptr && (n == 8 || (KIND == 2 && n == 7)) && as_int(args[2], &M) && as_int(args[3], &K) && as_int(args[4], &N) &&
        as_int(args[5], &b1) && as_int(args[6], &b2) && (n == 7 || (KIND == 2 ? as_bool(args[7], &pad) : as_int(args[7], &ob))) &&
        M > 0 && K > 0 && N > 0 && ob >= 1 && ob <= 32) {
        const at::Tensor *x1 = operand(args[0], at::kInt), *x2 = operand(args[1], at::kInt);
        if (x1 && x2) {
            try {
                c10::DeviceIndex dev;
                if (!on_current(*x1, x2, &dev)) return PyObject_Vectorcall(slow, args, nargs, kwnames);
                at::Tensor out = KIND == 2 ? fresh({M, N}, at::kFloat, dev)
                               : KIND == 1 ? fresh({static_cast<int64_t>(ob) * S128(M) * 4, P128(N)}, at::kInt, dev)    // QGTC_device.cu:456
                                           : fresh({static_cast<int64_t>(ob) * P8(M), S128(N) * 4}, at::kInt, dev);      // QGTC_device.cu:223
                void *st = static_cast<void *>(c10::hip::getCurrentHIPStream(dev).stream());
                int rc;
                if (KIND == 2)
                    rc = qgtc_bitmm2int(words(*x1), x1->numel(), words(*x2), x2->numel(), M, K, N, b1, b2, pad, out.data_ptr<float>(),
                                        out.numel(), mm_flags(), st);
                else
                    rc = qgtc_bitmm2bit(words(*x1), x1->numel(), words(*x2), x2->numel(), M, K, N, b1, b2, ob, words_mut(out), out.numel(),
                                        mm_flags() | (KIND == 1 ? QGTC_OUT_COLS : 0u), st);
                if (rc == QGTC_OK) return THPVariable_Wrap(std::move(out));
            } catch (...) {   // (allocation failure: the pybind11 path raises it as the Python exception it is)
            }
        }
    }
    return PyObject_Vectorcall(slow, args, nargs, kwnames);
}

PyObject *val2bit_fast(PyObject *, PyObject *const *args, Py_ssize_t nargs, PyObject *kwnames) {
    const Py_ssize_t n = PyVectorcall_NARGS(nargs);
    int nbits;
    bool col = false, outl = false;
    if (kwnames == nullptr && n >= 2 && n <= 4 && as_int(args[1], &nbits) && (n < 3 || as_bool(args[2], &col)) && (n < 4 || as_bool(args[3], &outl))) {
        const at::Tensor *x = operand(args[0], at::kFloat);
        if (x && x->dim() == 2 && nbits >= 1 && nbits <= 32 && x->size(0) > 0 && x->size(1) > 0 && x->size(0) < (1 << 30) && x->size(1) < (1 << 30)) {
            const int H = static_cast<int>(x->size(0)), W = static_cast<int>(x->size(1));
            try {
                c10::DeviceIndex dev;
                if (!on_current(*x, nullptr, &dev)) return PyObject_Vectorcall(slow_val2bit, args, nargs, kwnames);
                at::Tensor out = col ? fresh({static_cast<int64_t>(nbits) * S128(H) * 4, outl ? P8(W) : P128(W)}, at::kInt, dev)   // QGTC_device.cu:83,97
                                     : fresh({static_cast<int64_t>(nbits) * P8(H), S128(W) * 4}, at::kInt, dev);                   // QGTC_device.cu:115
                const int rc = qgtc_val2bit(x->data_ptr<float>(), H, W, nbits, col, outl, words_mut(out), out.numel(),
                                            static_cast<void *>(c10::hip::getCurrentHIPStream(dev).stream()));
                if (rc == QGTC_OK) return THPVariable_Wrap(std::move(out));
            } catch (...) {
            }
        }
    }
    return PyObject_Vectorcall(slow_val2bit, args, nargs, kwnames);
}

#define QGTC_FAST(fn) reinterpret_cast<PyCFunction>(reinterpret_cast<void (*)(void)>(fn))
PyMethodDef methods[] = {
    {"val2bit", QGTC_FAST(val2bit_fast), METH_FASTCALL | METH_KEYWORDS,
     "val2bit(input, nbits, col_major=False, output_layer=False): quantize a [ float32 --> bit ] tensor"},
    {"bitMM2Bit", QGTC_FAST(mm<0>), METH_FASTCALL | METH_KEYWORDS,
     "bitMM2Bit(bit_X1, bit_X2, X1_height, X1_width, X2_width, bit1, bit2, output_bit): QGTC [ bit_A x bit_B --> bit_C ] forward"},
    {"bitMM2Bit_col", QGTC_FAST(mm<1>), METH_FASTCALL | METH_KEYWORDS,
     "bitMM2Bit_col(bit_X1, bit_X2, X1_height, X1_width, X2_width, bit1, bit2, output_bit): QGTC [ bit_A x bit_B --> bit_C (column major) ] forward"},
    {"bitMM2Int", QGTC_FAST(mm<2>), METH_FASTCALL | METH_KEYWORDS,
     "bitMM2Int(bit_X1, bit_X2, X1_height, X1_width, X2_width, bit1, bit2, pad_128=False): QGTC [ bit_A x bit_B --> float32 ] forward"},
    {nullptr, nullptr, 0, nullptr}};
#undef QGTC_FAST

}  // namespace lean


// Diagnostic (tools/call_cost.py): host microseconds of the pieces of one bitMM2Bit call, each timed over
// `reps` repetitions with the GPU drained before and after (so that a full queue never blocks the host inside the timing).
std::vector<double> host_parts(torch::Tensor bit_X1, torch::Tensor bit_X2, int M, int K, int N, int bit1, int bit2, int ob, int reps) {
    CHECK_INPUT(bit_X1);
    CHECK_INPUT(bit_X2);
    TORCH_CHECK(reps > 0, "reps must be positive");
    c10::DeviceGuard guard(bit_X1.device());
    const auto dev = bit_X1.get_device();
    const auto opts = torch::TensorOptions().dtype(torch::kInt32).device(bit_X1.device());
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto us = [&](std::chrono::steady_clock::time_point a) { return std::chrono::duration<double, std::micro>(now() - a).count() / reps; };
    std::vector<double> r;
    TORCH_CHECK(hipDeviceSynchronize() == hipSuccess, "sync failed");
    auto t = now();
    for (int i = 0; i < reps; i++) { auto o = torch::empty({static_cast<int64_t>(ob) * P8(M), S128(N) * 4}, opts); (void)o; }
    r.push_back(us(t));                                   // 0: torch::empty through the dispatcher
    t = now();
    for (int i = 0; i < reps; i++) { auto o = lean::fresh({static_cast<int64_t>(ob) * P8(M), S128(N) * 4}, at::kInt, dev); (void)o; }
    r.push_back(us(t));                                   // 1: at::detail::empty_cuda
    t = now();
    for (int i = 0; i < reps; i++) { c10::DeviceGuard g(bit_X1.device()); void *st = current_stream(bit_X1); (void)st; }
    r.push_back(us(t));                                   // 2: device guard + current stream
    auto out = torch::empty({static_cast<int64_t>(ob) * P8(M), S128(N) * 4}, opts);
    void *st = current_stream(bit_X1);
    check_rc(qgtc_bitmm2bit(words(bit_X1), bit_X1.numel(), words(bit_X2), bit_X2.numel(), M, K, N, bit1, bit2, ob, words_mut(out), out.numel(), mm_flags(), st), "host_parts");
    TORCH_CHECK(hipDeviceSynchronize() == hipSuccess, "sync failed");
    // 3: the C-ABI call (argument checks, kernel choice, hipLaunchKernel) in bursts of 16 with a drain between bursts
    double launch = 0.0;
    for (int i = 0; i < reps; i += 16) {
        auto t0 = now();
        for (int j = 0; j < 16; j++)
            (void)qgtc_bitmm2bit(words(bit_X1), bit_X1.numel(), words(bit_X2), bit_X2.numel(), M, K, N, bit1, bit2, ob, words_mut(out), out.numel(), mm_flags(), st);
        launch += std::chrono::duration<double, std::micro>(now() - t0).count();
        TORCH_CHECK(hipDeviceSynchronize() == hipSuccess, "sync failed");
    }
    r.push_back(launch / (((reps + 15) / 16) * 16));
    return r;
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    m.doc() = "QGTC bit-GEMM operators for AMD Instinct MI355X (gfx950); drop-in for the reference's QGTC extension";
    namespace py = pybind11;
    // QGTC_ENGINE=popcount|mfma|auto picks the engine unmodified callers start with (default auto; set_engine() overrides)
    if (const char *e = std::getenv("QGTC_ENGINE")) {
        const std::string name(e);
        TORCH_CHECK(name == "popcount" || name == "mfma" || name == "auto", "QGTC_ENGINE must be 'popcount', 'mfma' or 'auto'");
        g_engine = name == "mfma" ? 1 : (name == "auto" ? 2 : 0);
    }
    // The eight reference entry points (QGTC_host.cpp:259-271). Positional call forms are
    // unchanged; the trailing bools get the C++ defaults of QGTC_host.cpp:6-7,15-16,60 so that
    // unitest.py's 7-argument bitMM2Int calls (unitest.py:72,79,143,146) also work.
    m.def("val2bit", &val2bit, "quantize a [ float32 --> bit ] tensor", py::arg("input"),
          py::arg("nbits"), py::arg("col_major") = false, py::arg("output_layer") = false);
    m.def("bit2val", &bit2val, "decode a [ bit --> int32 ] tensor", py::arg("input"), py::arg("nbits"),
          py::arg("height"), py::arg("width"), py::arg("col_major") = false,
          py::arg("output_layer") = false);
    m.def("bitMM2Bit", &bitMM2Bit, "QGTC [ bit_A x bit_B --> bit_C ] forward");
    m.def("bitMM2Bit_profile", &bitMM2Bit_profile, "QGTC [ bit_A x bit_B --> bit_C ] forward, 200 timed launches");
    m.def("bitMM2Bit_base_cnt", &bitMM2Bit_base_cnt, "QGTC [ bit_A x bit_B --> bit_C ] forward + tile-step count");
    m.def("bitMM2Bit_zerojump_cnt", &bitMM2Bit_zerojump_cnt, "QGTC [ bit_A x bit_B --> bit_C ] forward + non-zero tile-step count");
    m.def("bitMM2Bit_col", &bitMM2Bit_col, "QGTC [ bit_A x bit_B --> bit_C (column major) ] forward");
    m.def("bitMM2Int", &bitMM2Int, "QGTC [ bit_A x bit_B --> float32 ] forward", py::arg("bit_X1"),
          py::arg("bit_X2"), py::arg("X1_height"), py::arg("X1_width"), py::arg("X2_width"),
          py::arg("bit1"), py::arg("bit2"), py::arg("pad_128") = false);

    // The four per-batch operators get the lean call path (namespace lean above); the pybind11 functions stay reachable as
    // the checked path every irregular call falls back to.
    lean::slow_val2bit = py::object(m.attr("val2bit")).release().ptr();
    lean::slow_mm2bit = py::object(m.attr("bitMM2Bit")).release().ptr();
    lean::slow_mm2bit_col = py::object(m.attr("bitMM2Bit_col")).release().ptr();
    lean::slow_mm2int = py::object(m.attr("bitMM2Int")).release().ptr();
    m.attr("checked_val2bit") = m.attr("val2bit");
    m.attr("checked_bitMM2Bit") = m.attr("bitMM2Bit");
    m.attr("checked_bitMM2Bit_col") = m.attr("bitMM2Bit_col");
    m.attr("checked_bitMM2Int") = m.attr("bitMM2Int");
    TORCH_CHECK(PyModule_AddFunctions(m.ptr(), lean::methods) == 0, "could not register the lean entry points");

    // Names BASELINE.json's north_star uses for the same operators (aliases; see SURVEY.md note).
    m.attr("bit_qnt") = m.attr("val2bit");
    m.attr("mm_v1") = m.attr("bitMM2Bit");
    m.attr("mm_v2") = m.attr("bitMM2Int");

    // Additive helpers (not in the reference).
    m.def("profile", [](torch::Tensor a, torch::Tensor b, int M, int K, int N, int bit1, int bit2,
                        int ob, int reps) { return profile_impl(a, b, M, K, N, bit1, bit2, ob, reps).second; },
          "time `reps` bitMM2Bit launches; returns elapsed milliseconds (blocking)");
    m.def("last_profile_ms", [] { return g_last_profile_ms.load(); });
    m.def("bitMM2Bit_enqueue_streams", &bitMM2Bit_enqueue_streams,
          "enqueue `reps` independent bitMM2Bit launches round-robin over len(outs) HIP streams (asynchronous)");
    m.def("bitMM2Bit_enqueue", &bitMM2Bit_enqueue, py::arg("out"), py::arg("bit_X1"), py::arg("bit_X2"), py::arg("M"), py::arg("K"), py::arg("N"),
          py::arg("bit1"), py::arg("bit2"), py::arg("output_bit"), py::arg("reps"), py::arg("cols") = false,
          "enqueue `reps` bitMM2Bit (cols: bitMM2Bit_col) launches into a preallocated output (asynchronous)");
    m.def("tile_counters", [](torch::Tensor x, int M, int K, int N, int bit1, int bit2) {
        CHECK_INPUT(x);
        check_bits_tensor(x, "x");
        auto c = tile_counters(x, M, K, N, bit1, bit2);
        return py::make_tuple(c[0], c[1]);
    }, "per-call (total, non-zero) tile-step counts in the reference's 8x128-bit tile units");
    m.def("tile_occupancy", [](torch::Tensor x, int M, int K, int bit1) {
        CHECK_INPUT(x);
        check_bits_tensor(x, "x");
        c10::DeviceGuard guard(x.device());
        const int64_t nw = static_cast<int64_t>(qgtc_occupancy_words(M, K));
        auto occ = torch::empty({nw}, torch::TensorOptions().dtype(torch::kInt64).device(x.device()));
        check_rc(qgtc_tile_occupancy(words(x), x.numel(), M, K, bit1,
                                     reinterpret_cast<uint64_t *>(occ.data_ptr<int64_t>()), nw, current_stream(x)),
                 "tile_occupancy");
        return occ;
    }, "occupancy bitmap (int64 words, [row tile][k-quad / 64]) of a rows-layout operand: 32-row x 128-bit tiles");
    m.def("get_counters", [] { return py::make_tuple(g_counter_global.load(), g_counter.load()); });
    m.def("reset_counters", [] { g_counter = 0; g_counter_global = 0; });
    m.def("set_zero_skip", [](bool on) { g_zero_skip = on; });
    m.def("set_engine", [](const std::string &name) {
        TORCH_CHECK(name == "popcount" || name == "mfma" || name == "auto", "engine must be 'popcount', 'mfma' or 'auto'");
        g_engine = name == "mfma" ? 1 : (name == "auto" ? 2 : 0);
    }, "engine of bitMM2Bit / bitMM2Bit_col / bitMM2Int: 'auto' (default: per call the kernel family that measured "
       "fastest on MI355X), 'popcount' (AND + v_bcnt kernels only) or 'mfma' (matrix cores wherever the plane counts "
       "allow: FP4 / int8 expansions of the bit planes). Same results.");
    m.def("get_engine", [] { return std::string(g_engine == 1 ? "mfma" : (g_engine == 2 ? "auto" : "popcount")); });
    m.def("get_zero_skip", [] { return g_zero_skip.load(); });
    m.def("abi_version", [] { return qgtc_abi_version(); });
    m.def("host_parts", &host_parts, "diagnostic: host us of [torch::empty, at::detail::empty_cuda, guard + stream, C-ABI launch] per call");

    m.def("pack_edges", &pack_edges, "rows-layout bit planes of the [height, width] adjacency of an edge list "
          "(= val2bit of the dense matrix, without materialising it)", py::arg("src"), py::arg("dst"),
          py::arg("height"), py::arg("width"), py::arg("nbits") = 1, py::arg("validate") = true);
    m.def("i8gemm", &i8gemm, "int8 MFMA GEMM (comparison path): float32 [M,N] = A[M,K] x Bt[N,K]^T, exact");
    m.def("i8gemm_profile", &i8gemm_profile, "time `reps` i8gemm launches; returns milliseconds",
          py::arg("A"), py::arg("Bt"), py::arg("reps") = 200, py::arg("print") = true);

    m.def("gcn_layer", &gcn_layer, "one quantised GNN layer on a subgraph in one call: requant(A . requant(X . W)) "
          "(QGTC_conv.py:14-22); packed activations, or float32 with output=True",
          py::arg("bit_A"), py::arg("bit_X"), py::arg("bit_W"), py::arg("n"), py::arg("f_in"), py::arg("f_out"),
          py::arg("a_bit") = 1, py::arg("act_bit") = 2, py::arg("w_bit") = 2, py::arg("output") = false);

    m.def("val2bit_many", &val2bit_many, "val2bit of up to 8 matrices in one launch (the weights an epoch packs inside its clock)",
          py::arg("inputs"), py::arg("nbits"), py::arg("col_major"), py::arg("output_layer"));
    m.def("last_batched_violation", [](torch::Tensor on) {
        c10::DeviceGuard guard(on.device());
        int problem = -1, field = 0;
        const int rc = qgtc_last_batched_violation(&problem, &field, current_stream(on));
        return py::make_tuple(rc, problem, field);
    }, "(rc, problem, field) of the first descriptor a QGTC_CHECK_DESCRIPTORS launch on this tensor's device found in violation");
    py::class_<EpochPlan, std::shared_ptr<EpochPlan>>(m, "EpochPlan")
        .def(py::init<std::vector<torch::Tensor>, std::vector<torch::Tensor>, std::vector<torch::Tensor>, std::vector<int>, int, bool, int, int, bool>(),
             py::arg("As"), py::arg("Xs"), py::arg("Xrs"), py::arg("nodes"), py::arg("a_bits") = 1, py::arg("zero_jump") = true,
             py::arg("x_chain_bits") = 0, py::arg("x_cols") = 0, py::arg("a_tiles") = false)
        .def_static("load", &EpochPlan::load, py::arg("src"), py::arg("dst"), py::arg("edge_counts"), py::arg("feats"), py::arg("nodes"), py::arg("x_bits"),
                    py::arg("with_rows") = false, py::arg("x_chain_bits") = 0, py::arg("a_tiles") = false, py::arg("validate") = false,
                    "pack `len(nodes)` cluster batches from their concatenated edge lists (indices local to each batch) and feature rows in a "
                    "handful of launches (qgtc_load_batches); .As / .Xs / .Xrs are the per-batch packed tensors (views into pools)")
        .def("format_of", &EpochPlan::format_of, py::arg("batch"), py::arg("which"),
             "one batch's operand in a loader format (SRC_A / SRC_X / SRC_XR / SRC_XC / SRC_AT, -1 = occupancy bitmap) as a flat non-owning tensor")
        .def_property_readonly("As", [](EpochPlan &p) { p.make_views(); return p.As; })
        .def_property_readonly("Xs", [](EpochPlan &p) { p.make_views(); return p.Xs; })
        .def_property_readonly("Xrs", [](EpochPlan &p) { p.make_views(); return p.Xrs; })
        .def("bind", &EpochPlan::bind, py::arg("weights"), py::arg("stages"), py::arg("launches"), py::arg("expand") = std::vector<std::array<int, 5>>(),
             "weights: packed tensors; stages: (left, right, K, N, bit1, bit2, ob, mode, pad128, use_occ, fmt); launches: (kind, s1, s2, flags, "
             "codes); expand: (weight, K, N, nbits, order) per pre-expanded weight")
        .def("run", &EpochPlan::run)
        .def("run_checked", &EpochPlan::run_checked)
        .def("run_launch", [](EpochPlan &p, int i) {
            TORCH_CHECK(p.descs.defined() && i >= 0 && i < static_cast<int>(p.launches.size()), "no such launch");
            c10::DeviceGuard guard(p.descs.device());
            p.run_launch(p.launches[i], 0u);
        }, "one launch of the bound plan (timing a stage on its own)")
        .def("outs", &EpochPlan::outs, py::arg("stage"))
        .def_readonly("count", &EpochPlan::count)
        .def_readonly("zero_jump", &EpochPlan::jumping)
        .def_readonly("x_chain", &EpochPlan::x_chain)
        .def_readonly("a_tiles", &EpochPlan::a_tiles)
        .def_readonly("occupied_fraction", &EpochPlan::occupied)
        .def_property_readonly("n_launches", [](const EpochPlan &p) { return p.launches.size(); });
    m.attr("SRC_A") = static_cast<int>(QGTC_SRC_A);
    m.attr("SRC_X") = static_cast<int>(QGTC_SRC_X);
    m.attr("SRC_XR") = static_cast<int>(QGTC_SRC_XR);
    m.attr("SRC_XC") = static_cast<int>(QGTC_SRC_XC);
    m.attr("SRC_AT") = static_cast<int>(QGTC_SRC_AT);
    m.attr("SRC_WEIGHT") = static_cast<int>(QGTC_SRC_WEIGHT);
    m.attr("SRC_STAGE") = static_cast<int>(QGTC_SRC_STAGE);
    m.attr("DIM_NODES") = static_cast<int>(QGTC_DIM_NODES);
    m.attr("CHAIN_DISCARD") = static_cast<int>(QGTC_CHAIN_DISCARD);

    py::class_<ChainedPair>(m, "ChainedPair")
        .def(py::init<std::shared_ptr<BatchedGemm>, std::shared_ptr<BatchedGemm>, bool>(), py::arg("stage_a"), py::arg("stage_xw"), py::arg("discard") = false)
        .def_readonly("discard", &ChainedPair::discard)
        .def("run", &ChainedPair::run, "A.(XW) of one layer and X.W of the next for every cluster batch (two grouped launches)")
        .def_property_readonly("outs", [](const ChainedPair &c) { return c.sx->outs; });
    py::class_<FusedLayer>(m, "FusedLayer")
        .def(py::init<std::shared_ptr<BatchedGemm>, std::shared_ptr<BatchedGemm>>(), py::arg("stage1"), py::arg("stage2"))
        .def("run", &FusedLayer::run, "one call per layer: X.W (cols-layout re-pack) then A.(XW) for every cluster batch")
        .def_property_readonly("outs", [](const FusedLayer &f) { return f.s2->outs; });

    py::class_<BatchedGemm, std::shared_ptr<BatchedGemm>>(m, "BatchedGemm")
        .def(py::init<std::vector<torch::Tensor>, std::vector<torch::Tensor>,
                      std::vector<std::tuple<int, int, int>>, int, int, int, int, bool, bool,
                      std::vector<torch::Tensor>>(),
             py::arg("Xs"), py::arg("Ws"), py::arg("dims"), py::arg("bit1"), py::arg("bit2"),
             py::arg("output_bit"), py::arg("mode") = 0, py::arg("pad_128") = false,
             py::arg("zero_jump") = false, py::arg("occs") = std::vector<torch::Tensor>())
        .def_readonly("occs", &BatchedGemm::occs)
        .def("run", &BatchedGemm::run)
        .def("run_per_problem", &BatchedGemm::run_per_problem, py::arg("n_streams") = 1)
        .def_readonly("outs", &BatchedGemm::outs)
        .def_readonly("count", &BatchedGemm::count)
        .def_property_readonly("zero_jump", &BatchedGemm::zero_jump)
        .def_property_readonly("occupied_fraction", &BatchedGemm::occupied_fraction);
}
