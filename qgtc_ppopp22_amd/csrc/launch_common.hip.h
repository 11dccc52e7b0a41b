// launch_common.hip.h — part of libqgtc_hip.so: what the translation units (qgtc_hip.hip, qgtc_mfma.hip, qgtc_fp4.hip) need on the
// host side: launch-constant helpers, the predicates that choose a kernel family, and the launchers of the
// matrix-core kernels, which live in their own translation units (all compiled in parallel).
#pragma once

#include <algorithm>
#include <atomic>

// defined in qgtc_mfma.hip
int qgtc_launch_mfma(const qgtc_problem &pr, int a, int w, int ob, int mode, hipStream_t st);
int qgtc_launch_mfma_batched(const qgtc_problem *prs, int count, int max_M, int max_K, int max_N, int a, int w,
                             int ob, int mode, hipStream_t st);
// defined in qgtc_fp4.hip
int qgtc_launch_skinny(const qgtc_problem &pr, int a, int w, int ob, int mode, bool zero_skip, hipStream_t st);
bool qgtc_skinny_is_one(const qgtc_problem &pr, int ob, int mode);
int qgtc_launch_rows_single(const qgtc_problem &pr, int a, int w, int ob, int mode, hipStream_t st);
int qgtc_launch_rows(const qgtc_problem *prs, int count, int max_M, int max_N, int a, int w, int ob, int mode, hipStream_t st);
int qgtc_launch_xw_rows(const qgtc_problem *prs, int count, int max_M, int a, int w, int ob, hipStream_t st);

// row block per wave (bitmm_fp4_rbw.hip.h), defined in qgtc_fp4.hip
int qgtc_launch_expand_weights(const qgtc_expand_job *jobs, int n_jobs, hipStream_t st);
int qgtc_launch_cols_to_chain(const uint32_t *cols, size_t words, int H, int W, int nbits, uint32_t *chain, hipStream_t st);
int qgtc_launch_cols_to_chain_batched(const qgtc_loader_batch *batches, int count, int max_n, int W, int nbits, hipStream_t st);
int qgtc_launch_rbw_xw(const qgtc_problem *prs, int count, int max_M, int K, int N, int a, int ob, const uint32_t *w_codes, hipStream_t st);
// defined in qgtc_chainx.hip (bitmm_fp4_rbx.hip.h: 5 .. 8 bits, up to 256 columns)
int qgtc_launch_rbx_xw(const qgtc_problem *prs, int count, int max_M, int K, int N, int a, int ob, const uint32_t *w_codes, hipStream_t st);
int qgtc_launch_rbx_chain(const qgtc_problem *p1, const qgtc_problem *p2, int count, int max_M, int N1, int N2, int t_bits, int act_bits,
                          int mode2, const uint32_t *w2_codes, bool a_tiles, hipStream_t st);
int qgtc_launch_rbw_chain(const qgtc_problem *p1, const qgtc_problem *p2, int count, int max_M, int N1, int N2, int t_bits, int act_bits,
                          int out_bits, int mode2, const uint32_t *w2_codes, bool a_tiles, hipStream_t st);
int qgtc_launch_rows_to_tiles(const uint32_t *rows, size_t words, int M, int K, uint32_t *tiles, hipStream_t st);

// defined in qgtc_wide.hip
int qgtc_launch_wide(const qgtc_problem &pr, int a, int w, int ob, int mode, hipStream_t st);
// defined in qgtc_stream.hip (bitmm_fp4_stream.hip.h: one-plane operands, long K, at most 256 columns)
int qgtc_launch_stream(const qgtc_problem &pr, int ob, int mode, bool zero_skip, hipStream_t st);
// defined in qgtc_epoch.hip: QGTC_CHECK_DESCRIPTORS (kind 0 one stage / 1 layer / 2 chain / 3 one stage, `out` unused / 4 the pair of
// qgtc_chain_aggregate; p2 may be NULL; exact_N*: the descriptors' N must equal it)
int qgtc_launch_check_descriptors(const qgtc_problem *p1, const qgtc_problem *p2, int count, int max_M, int max_K1, int max_N1,
                                  int max_K2, int max_N2, int kind, hipStream_t st, int exact_N1 = 0, int exact_N2 = 0, int exact_K1 = 0);

namespace {

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a PER-DEVICE property of a kernel: one flag per device
// index (not one per process), so that a second GPU used from the same process gets its opt-in too.
// Racing threads may both set the attribute; that is harmless.
constexpr int kMaxDevices = 64;
struct PerDeviceOnce {
    std::atomic<bool> done[kMaxDevices];   // static storage: zero-initialised
    template <typename F>
    int run(F &&f) {
        int dev = 0;
        HIP_TRY(hipGetDevice(&dev));
        if (dev < 0 || dev >= kMaxDevices) return f();   // beyond the table: set it every time
        if (done[dev].load(std::memory_order_acquire)) return QGTC_OK;
        const int rc = f();
        if (rc == QGTC_OK) done[dev].store(true, std::memory_order_release);
        return rc;
    }
};

// diagnostic switches (tools/ only; the product never sets them), read ONCE per process and call site: a getenv() walks
// the whole environment, and three of them per launch were 0.3 - 0.5 us of the host side of a 3 us launch
inline bool getenv_flag_now(const char *name) {
    const char *v = std::getenv(name);
    return v && v[0] && v[0] != '0';
}
#define getenv_flag(name) ([]() -> bool { static const bool v_ = getenv_flag_now(name); return v_; }())

// A launch through the MODULE API with the kernel's function handle resolved once per device (hipGetFuncBySymbol) instead of per launch
// from the host stub's address: tools/kernarg_probe.hip measured 2.18-2.23 against 2.33-2.35 us of host time a launch (2.61-2.70 against
// 2.76 on a slower host) - the eager issue loops (the reference's K launches back to back, an unchanged driver's 450 calls an epoch) are
// bound by exactly that. Arguments are converted to the kernel's own parameter types first (the module API reads them through pointers).
// QGTC_NO_MODULE_LAUNCH=1, or a runtime without the lookup: the ordinary hipLaunchKernelGGL.
inline hipError_t &launch_error_slot() {
    static thread_local hipError_t e = hipSuccess;
    return e;
}
// what the QGTC_LAUNCHes of this thread since the last call returned (the first failure), in place of a hipGetLastError() API call per launch
inline hipError_t launch_status() {
    hipError_t &e = launch_error_slot();
    const hipError_t r = e;
    e = hipSuccess;
    return r;
}
template <auto Kern>
struct KernelLaunch;
template <class... P, void (*Kern)(P...)>
struct KernelLaunch<Kern> {
    static hipFunction_t handle() {
        constexpr int MAX_DEV = 64;
        static std::atomic<hipFunction_t> fns[MAX_DEV];
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEV) return nullptr;
        hipFunction_t fn = fns[dev].load(std::memory_order_acquire);
        if (!fn) {
            if (getenv_flag("QGTC_NO_MODULE_LAUNCH") || hipGetFuncBySymbol(&fn, reinterpret_cast<const void *>(Kern)) != hipSuccess) return nullptr;
            fns[dev].store(fn, std::memory_order_release);
        }
        return fn;
    }
    static void go(dim3 grid, dim3 block, unsigned lds, hipStream_t st, P... p) {
        if (hipFunction_t fn = handle()) {
            void *params[] = {const_cast<void *>(static_cast<const void *>(&p))...};
            const hipError_t e = hipModuleLaunchKernel(fn, grid.x, grid.y, grid.z, block.x, block.y, block.z, lds, st, params, nullptr);
            if (e != hipSuccess && launch_error_slot() == hipSuccess) launch_error_slot() = e;
        } else {
            hipLaunchKernelGGL(Kern, grid, block, lds, st, p...);
            const hipError_t e = hipGetLastError();
            if (e != hipSuccess && launch_error_slot() == hipSuccess) launch_error_slot() = e;
        }
    }
};
#define QGTC_LAUNCH(kernel, grid, block, lds, st, ...) KernelLaunch<(kernel)>::go(grid, block, lds, st, __VA_ARGS__)

inline MMShape base_shape(int a, int w, int ob, int mode) {
    MMShape sh{};
    sh.a = a;
    sh.w = w;
    sh.ob = ob;
    sh.mode = mode;
    sh.ab = a;
    sh.wb = w;
    sh.maxv = std::ldexp(1.0f, ob);
    sh.maxm1 = sh.maxv - 1.0f;
    sh.nowrap = 0;
    return sh;
}

// no int32 accumulator of a product with this K can wrap (then requantisation needs no sign test). Every bound on a sum in this file counts
// the WHOLE k-quads of a line, PAD128(K) bits: the kernels AND and multiply the padding bits of the last k-quad as the reference does
// (kernel.h:301-308), and operands that do not come from val2bit may carry set bits there (tests/test_edge_domain_gpu.py: raw words, K = 1).
inline int no_wrap(int K, int a, int w) {
    if (a > 16 || w > 16) return 0;
    return static_cast<double>(pad128(K)) * ((1u << a) - 1u) * ((1u << w) - 1u) < 2147483648.0;
}

// the FP4 form of the matrix-core engine: 2-bit values at most and float32 sums that stay exact
inline bool fp4_ok(int K, int a, int w) {
    return a <= 2 && w <= 2 && static_cast<double>(pad128(K)) * ((1 << a) - 1) * ((1 << w) - 1) < 16777216.0;
}

// wide right operands; 1, 2, 4 or 8 planes on one side with 1 or 2 on the other (a stage of 4 x 4 planes and up does
// not fit the LDS): packed words
// staged by LDS-DMA, expanded in the multiplying waves' registers (bitmm_fp4_wide.hip.h)
inline bool wide_ok(const qgtc_problem &pr, int a, int w, int ob, int mode) {
    const size_t out_bytes = mode == 2 ? static_cast<size_t>(pr.M) * pr.N * 4u
                                       : static_cast<size_t>(ob) * (mode == 1 ? pad128(pr.N) : pad8(pr.M)) * step128(mode == 1 ? pr.M : pr.N) * 16u;
    const int nl = mode == 1 ? w : a, nr = mode == 1 ? a : w;   // planes of the operand that supplies the output lines / bits
    const bool planes = ((nl == 1 || nl == 2) && (nr == 1 || nr == 2 || nr == 4 || nr == 8)) || ((nl == 4 || nl == 8) && (nr == 1 || nr == 2));
    return planes && static_cast<double>(pad128(pr.K)) * ((1 << a) - 1) * ((1 << w) - 1) < 16777216.0 && (mode == 2 || (ob >= 1 && ob <= 23)) &&
           pr.x_words < (1ull << 30) && pr.w_words < (1ull << 30) && out_bytes < (1ull << 32) && !getenv_flag("QGTC_NO_WIDE");
}
// The kernel is bound by VALU issue: per MFMA and SIMD 7.8 ns + 1.55 ns per VALU operation, two waves
// per SIMD; expanding a fragment for one MFMA position costs 5 operations per one-plane operand, 12 per base-4 digit.
// Measured against that model (tools/wide_check.py): +3 % with 128-byte groups of K (1 x 1 planes), +27 % with 64-byte
// groups, plus ~3.5 us of launch, first DMA and epilogue. rf x cf = the fragments per wave (4 x 4: 128 x 256 tiles,
// 2 x 4: 64 x 256, 4 x 2: 128 x 128, twice the workgroups either way) that give the shortest launch; returns its
// estimated time in us.
inline double wide_plan(int lines, int R, int K, int a, int w, int *rf, int *cf = nullptr) {
    auto digits = [](int p) { return p == 1 ? 1 : p / 2; };
    auto expand = [&](int p) { return p == 1 ? 5.0 : 12.0 * digits(p); };
    const double nd = digits(a) * digits(w);
    const int shapes[3][2] = {{4, 4}, {2, 4}, {4, 2}};   // fragments per wave: 128 x 256, 64 x 256, 128 x 128 tiles
    double best = 0.0;
    for (const auto &sh : shapes) {
        const int f = sh[0], c = sh[1];
        if ((a >= 4 && f == 4) || (w == 8 && c == 4)) continue;   // (many left-hand planes: 2 x 4 fragments only; eight right-hand ones: 4 x 2 only)
        const double v = (f * expand(a) + c * expand(w)) / (f * c * nd);
        const double per_group = 2.0 * f * c * 8 * nd * (7.8 + 1.55 * v) * 1e-3 * (a + w == 2 ? 1.03 : 1.27);   // us per 1024 bits of K and round
        const double tiles = static_cast<double>((lines + 32 * f - 1) / (32 * f)) * ((R + 64 * c - 1) / (64 * c));
        const double t = std::ceil(tiles / 256.0) * per_group;
        if (best == 0.0 || t < best) {
            best = t;
            if (rf) *rf = f;
            if (cf) *cf = c;
        }
    }
    return 3.5 + (step128(K) / 8.0) * best;
}
// QGTC_ENGINE_AUTO
inline bool auto_prefers_wide(int M, int K, int N, int a, int w, int mode) {
    const double t_wide = wide_plan(mode == 1 ? N : M, mode == 1 ? M : N, K, mode == 1 ? w : a, mode == 1 ? a : w, nullptr);
    const double t_pop = 3.0 + 2.0 * M * static_cast<double>(K) * N * a * w / 0.95e15 * 1e6;
    return t_wide < 0.9 * t_pop;
}

// narrow right operands: the LDS-free FP4 kernel (bitmm_fp4_skinny.hip.h). Measured at 4096 x 4096 x N: ahead of
// both other kernels up to N = 256 (N = 128: 5.4 us against 7.0 popcount / 15.5 128-tile MFMA at 1 bit, 16.6 against
// 40 / 46 at 8 bits), level with the 128-tile kernel at N = 512, behind it at N = 1024
// (plane capacities 1 / 2 for X and 1 / 2 / 4 / 8 for W are instantiated; float32 sums must stay exact)
inline bool skinny_ok(int K, int N, int a, int w) {
    return N <= 256 && a <= 2 && w <= 8 &&
           static_cast<double>(pad128(K)) * ((1 << a) - 1) * ((1 << w) - 1) < 16777216.0;
}
// one-plane operands with K beyond k_bitmm_fp4_one's 4096 (5_9_adjmatrix_size.py's M = K = 8192 .. 32768): the long-K kernel that reads
// the adjacency once (bitmm_fp4_stream.hip.h) instead of k_bitmm_fp4_skinny's 32 x 32 tiles
inline bool stream_ok(const qgtc_problem &pr, int a, int w, int ob, int mode) {
    const size_t out_bytes = mode == 2 ? static_cast<size_t>(pr.M) * pr.N * 4u
                                       : static_cast<size_t>(ob) * (mode == 1 ? pad128(pr.N) : pad8(pr.M)) * step128(mode == 1 ? pr.M : pr.N) * 16u;
    // (operands below 2 GiB: a DMA piece's offset may run 128 lines past the operand before the range check drops it)
    return a == 1 && w == 1 && pr.N <= 256 && pr.K > 4096 && pad128(pr.K) < (1 << 24) && pr.M < (1 << 24) && pr.x_words < (1ull << 29) && pr.w_words < (1ull << 29) &&
           out_bytes < (1ull << 32) && !getenv_flag("QGTC_NO_STREAM");
}

// QGTC_ENGINE_AUTO: where the long-K kernel measured ahead of k_bitmm_fp4_skinny (tools/stream_route_sweep.sh, M = 1024 .. 65536, K = 8192 ..
// 32768, N = 16 .. 256: 105 shapes): its 64-row tiles need rows to fill the chip (16384 x 16384 x 64 11.2 against 20.8 us, 8192 x 8192 x 32
// 6.0 against 6.2, 8192 x 8192 x 16 6.0 against 5.4; below that a workgroup's walk over K is bound by the latency of its three stages:
// 2048 x 32768 x 64 15.8 against 10.6), its column tiles of 64 columns pay from two tiles up (4096 x 32768 x 256 18.0 against 36.1,
// 1024 x 32768 x 256 16.1 against 11.0)
inline bool auto_prefers_stream(int M, int K, int N) {
    (void)K;
    if (N > 64) return static_cast<long long>((N + 63) / 64) * M >= 8192;
    return M >= 16384 || (M >= 8192 && N > 16);
}

// QGTC_ENGINE_AUTO: measured against the popcount kernels on the reference's micro-benchmark shapes
// (1024 / 2048 / 4096 square, N = 16 / 32 / 64, 1- and 2-bit): ahead on all of them (4096 x 4096 x 64:
// 4.0 us against 4.8 at 1 bit, 4.8 against 7.1 at 2 bits)
inline bool auto_prefers_skinny(int M, int K, int N, int a, int w) {
    // ... and on every smaller single launch tried since (1213 x 128 x 128 2-bit 3.5 against 4.2 us, 300 x 300 x 64
    // 3.3 against 4.4, 256 x 4096 x 64 3.9 against 4.6): whenever skinny_ok holds
    (void)M; (void)K; (void)N; (void)a; (void)w;
    return true;
}

// grouped "A . (XW)" stages (rows-layout bits or float32 out, at most 256 columns): one workgroup per 32-row block of a
// batch, visiting only the k-quads its occupancy word names (bitmm_fp4_rows.hip.h)
inline bool rows_ok(int max_K, int max_N, int a, int w, int ob, int mode) {
    return (mode == 0 || mode == 2) && max_K <= 8192 && max_N <= 256 && a <= 8 && w <= 8 && (mode == 2 || (ob >= 1 && ob <= 23)) &&
           static_cast<double>(pad128(max_K)) * ((1 << a) - 1) * ((1 << w) - 1) < 16777216.0 && !getenv_flag("QGTC_NO_ROWS");
}
// ... and cols-layout stages (the operands not swapped, a workgroup per word of a line): every one that _xw_rows (K, N <= 128 at 2 / 4
// bits) does not take - rounds 1-2 had a column-strip kernel (K <= 128) and a one-wave-per-tile kernel (N <= 64) for them
inline bool rows_cols_ok(int max_K, int max_N, int a, int w, int ob) { return rows_ok(max_K, max_N, a, w, ob, 0) && !getenv_flag("QGTC_NO_ROWS_COLS"); }

// grouped "X . W" stages with one k-quad of K at the epochs' widths (k_bitmm_fp4_xw_rows)
inline bool xw_rows_ok(int max_K, int max_N, int a, int w, int ob) {
    return max_K <= 128 && max_N <= 128 && ((a <= 2 && w <= 2 && ob == 2) || (a <= 4 && w <= 4 && ob == 4)) && !getenv_flag("QGTC_NO_XWROWS");
}

// One width b = 1 .. 4 per chain (planes of X, W, T alike - what main_qgtc.py's --bit_width gives); 1 / 2 bits are one base-4 digit a
// nibble, 3 / 4 bits two. N, N' <= 128; float32 sums exact (4 bits: K 15 < 2^24 for the aggregation, 8192 x 15 x 15 < 2^24 for X . W with its k-quad loop).
inline bool rbw_xw_ok(int K, int N, int x_bits, int out_bits) {
    return K >= 1 && K <= 8192 && N >= 1 && N <= 128 && out_bits >= 1 && out_bits <= 4 && x_bits >= 1 && x_bits <= (out_bits > 2 ? 4 : 2);   // (K > 128: a k-quad loop; 8192 x 15 x 15 < 2^24)
}
inline bool rbw_chain_ok(int max_K, int N1, int N2, int t_bits, int act_bits, int out_bits, int mode2) {
    if (max_K < 1 || max_K > 8192 || N1 < 1 || N1 > 128 || t_bits < 1 || t_bits > 4) return false;
    if (mode2 == 0) return true;
    const bool wide = t_bits > 2;   // the format class of T, of the aggregate and of W' must agree
    return N2 >= 1 && N2 <= 128 && act_bits >= 1 && act_bits <= 4 && (act_bits > 2) == wide && (mode2 == 2 || out_bits == act_bits);
}

// The same entries beyond those widths (bitmm_fp4_rbx.hip.h): one width of 5 .. 8 bits per chain with N, N' <= 128 (four base-4 digits a
// value), or 1 .. 4 bits with up to 256 columns on either side. The float32 sums stay exact: K (2^a - 1)(2^w - 1) < 2^24 is checked for
// the X . W product (8 x 8 bits: K <= 258); the aggregation has K <= 8192 x 255 and the second product N1 <= 256 x 255 x 255 < 2^24.
inline int chain_class(int bits) { return chain_digits(bits); }
inline bool rbx_xw_ok(int K, int N, int x_bits, int out_bits) {
    if (K < 1 || K > 8192 || N < 1 || out_bits < 1 || out_bits > 8 || x_bits < 1 || x_bits > 2 * chain_class(out_bits)) return false;
    if (static_cast<double>(pad128(K)) * ((1 << x_bits) - 1) * ((1 << out_bits) - 1) >= 16777216.0) return false;
    return out_bits > 4 ? N <= 128 : (N > 128 && N <= 256);   // (1 .. 4 bits at N <= 128 are k_rbw_xw's)
}
inline bool rbx_chain_ok(int max_K, int N1, int N2, int t_bits, int act_bits, int out_bits, int mode2) {
    if (max_K < 1 || max_K > 8192 || N1 < 1 || t_bits < 1 || t_bits > 8) return false;
    const int cls = chain_class(t_bits), lim = cls == 4 ? 128 : 256;
    if (N1 > lim) return false;
    if (mode2 == 0) return cls == 4 || N1 > 128;
    if (N2 < 1 || N2 > lim || act_bits < 1 || act_bits > 8 || chain_class(act_bits) != cls || (mode2 == 1 && out_bits != act_bits)) return false;
    return cls == 4 || N1 > 128 || N2 > 128;
}

// single launches with three to eight left-hand planes (the narrow-operand kernels of bitmm_fp4_one / _skinny take two at most) and at
// most 256 columns: the row-block kernel on ONE by-value problem (k_bitmm_fp4_rows_single: 32 x 32 x 64 MFMAs, a wave per 32 columns
// of a 32-row block, no LDS; cols-layout output with the operands not swapped). The per-batch 4 x 4-bit products of the Batched-GIN
// chain (main_qgtc.py:132,134,138): 599 x 50 x 64 in 2.83 us (a wave per 32 x 32 tile on 16 x 16 x 128 MFMAs, round 4's first route:
// 7.85; popcount: 8.6). tools/rows1_sweep.py, 4 x 4 and 4 x 8 bits, M = 599 .. 16384, K = 128 .. 8192, N = 16 .. 256, all three
// outputs: ahead of the 128-tile / wide kernels the larger products took before (4096 x 4096 x 64: 16.0 against 35.2 us;
// 16384 x 8192 x 128: 63.8 against 77.9) - except the eight-wave blocks of N > 128 on big operands (16384 x 1024 x 256: 18.6
// against 16.6), which keep the cost models below
inline bool rows_single_ok(const qgtc_problem &pr, int a, int w, int ob, int mode) {
    const int M = pr.M, K = pr.K, N = pr.N;
    const size_t out_bytes = mode == 2 ? static_cast<size_t>(M) * N * 4u : static_cast<size_t>(ob) * (mode == 1 ? pad128(N) : pad8(M)) * step128(mode == 1 ? M : N) * 16u;
    const bool exact = static_cast<double>(pad128(K)) * ((1 << a) - 1) * ((1 << w) - 1) < 16777216.0;   // (8 x 8 bits: K <= 258 - the X . W products)
    // five to eight left-hand planes run on the one <8, 8> instantiation (16 MFMAs per 64 elements of K whatever w is): the b x b-bit
    // products of the drivers at --bit_width 5 .. 8 and anything small (tools/route_sweep.py: 1213 x 128 x 128 8 x 8 bits 27.9 -> 4.2 us,
    // 4096 x 4096 x 64 5 x 5 bits 45.2 -> 32.1; but 8 x 1 bits 27.7 -> 32.7 and N = 256 5 x 5 bits 47.7 -> 54.6: those stay where they were)
    const bool small = static_cast<double>(M) * K * N <= 1213.0 * 1213.0 * 128.0;
    return a > 2 && a <= 8 && w <= 8 && K <= 8192 && N <= 256 && exact && (mode == 2 || (ob >= 1 && ob <= 23)) && !getenv_flag("QGTC_NO_ROWS") &&
           (N <= 128 || M <= 8192 || K <= 512) && (a <= 4 || small || (w > 2 && N <= 128)) && M < (1 << 24) &&
           pr.x_words < (1ull << 30) && pr.w_words < (1ull << 30) && out_bytes < (1ull << 32) && !getenv_flag("QGTC_NO_ROWS1");
}

// the MFMA engine handles up to 8 planes per operand (8: offset by 128, corrected in the epilogue)
inline bool mfma_ok(int a, int w) { return a >= 1 && a <= 8 && w >= 1 && w <= 8; }

// QGTC_ENGINE_AUTO: pick the engine by a two-line cost model fitted to the round-1 measurements
// (DESIGN.md section 5.4b): popcount runs at ~0.95e15 bit-ops/s plus ~3 us of launch and tail; the
// matrix-core engine pays ~6 us fixed and ~0.46 us per k-quad and 128 x 128 tile round (a quarter
// more per extra plane to expand), rounds = tiles / 256 CUs. MFMA only when it wins by 10 %.
inline bool auto_prefers_mfma(int M, int K, int N, int a, int w) {
    if (!mfma_ok(a, w)) return false;
    const double tiles = static_cast<double>((M + 128 - 1) / 128) * ((N + 128 - 1) / 128);
    const double rounds = tiles <= 256.0 ? 1.0 : tiles / 256.0 * 0.9;
    const int maxp = a > w ? a : w;
    // per k-quad and round: 0.32 us in the FP4 form (2-bit values at most), else 0.46 us plus 15 % per extra plane
    const double per_kq = fp4_ok(K, a, w) ? 0.32 * (1.0 + 0.25 * (maxp - 1)) : 0.46 * (1.0 + 0.15 * (maxp - 1));
    const double t_mfma = 6.0 + per_kq * step128(K) * rounds;
    const double t_pop = 3.0 + 2.0 * M * static_cast<double>(K) * N * a * w / 0.95e15 * 1e6;
    return t_mfma < 0.9 * t_pop;
}

// QGTC_ENGINE_AUTO for grouped launches (cluster batches: many small products, every workgroup short-
// lived). Measured on the ogbn-arxiv- and ppi-sized epochs: the matrix-core
// engine wins when its 128-wide tile is mostly full (N = 128: X.W 13 us against 24, A.(XW) 22
// against 27) or when four or more plane pairs share one expansion at N >= 48 (ppi: 4 x 4 bits at
// N = 64 14 against 22, 1 x 4 bits at N = 50 18.6 against 20.8); narrow outputs (N = 10 classes)
// and one- or two-pair products at N <= 64 stay on the popcount kernels.
inline bool auto_prefers_mfma_batched(int max_M, int max_N, int a, int w) {
    if (!mfma_ok(a, w) || max_M < 128) return false;
    return max_N >= 96 || (a * w >= 4 && max_N >= 48);
}

}  // namespace
