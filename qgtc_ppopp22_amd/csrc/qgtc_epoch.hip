// qgtc_epoch.hip — fifth translation unit of libqgtc_hip.so: what surrounds the products of a grouped epoch (epoch_plan.hip.h):
// batched val2bit of the weights, the device-side plan fill, the descriptor checks behind QGTC_CHECK_DESCRIPTORS.
#include <hip/hip_runtime.h>

#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "qgtc.h"

#include "common.hip.h"
#include "epoch_plan.hip.h"

namespace {

// The per-device violation record of QGTC_CHECK_DESCRIPTORS: one int, INT_MAX = nothing recorded. Allocated on first use,
// never freed (four bytes per device for the life of the process).
constexpr int kMaxDev = 64;
std::mutex g_record_mutex;
int *g_record[kMaxDev] = {};

int record_for_current_device(int **out, bool create) {
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= kMaxDev) return QGTC_ENODEVICE;
    std::lock_guard<std::mutex> lock(g_record_mutex);
    if (!g_record[dev] && create) {
        int *p = nullptr;
        HIP_TRY(hipMalloc(&p, sizeof(int)));
        const int none = INT_MAX;
        const hipError_t e = hipMemcpy(p, &none, sizeof(int), hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            (void)hipFree(p);
            return hip_fail(e, "violation record init");
        }
        g_record[dev] = p;
    }
    *out = g_record[dev];
    return QGTC_OK;
}

}  // namespace

// called by the grouped entry points of qgtc_hip.hip when QGTC_CHECK_DESCRIPTORS is set (declared in launch_common.hip.h)
int qgtc_launch_check_descriptors(const qgtc_problem *p1, const qgtc_problem *p2, int count, int max_M, int max_K1, int max_N1,
                                  int max_K2, int max_N2, int kind, hipStream_t st, int exact_N1, int exact_N2, int exact_K1) {
    int *rec = nullptr;
    const int rc = record_for_current_device(&rec, true);
    if (rc != QGTC_OK) return rc;
    hipLaunchKernelGGL(k_check_descriptors, dim3((count + 255) / 256), dim3(256), 0, st, p1, p2, count, max_M, max_K1, max_N1,
                       max_K2, max_N2, kind, exact_N1, exact_N2, exact_K1, rec);
    HIP_TRY(hipGetLastError());
    return QGTC_OK;
}

extern "C" {

int qgtc_last_batched_violation(int *problem, int *field, void *stream) {
    if (problem) *problem = -1;
    if (field) *field = QGTC_VIOL_NONE;
    int *rec = nullptr;
    const int rc = record_for_current_device(&rec, false);
    if (rc != QGTC_OK) return rc;
    if (!rec) return QGTC_OK;   // no checked launch has run on this device
    HIP_TRY(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    int v = INT_MAX;
    HIP_TRY(hipMemcpy(&v, rec, sizeof(int), hipMemcpyDeviceToHost));
    if (v == INT_MAX) return QGTC_OK;
    const int none = INT_MAX;
    HIP_TRY(hipMemcpy(rec, &none, sizeof(int), hipMemcpyHostToDevice));
    if (problem) *problem = v / 8;
    if (field) *field = v % 8;
    return QGTC_EINVAL;
}

int qgtc_val2bit_batched(const qgtc_pack_job *jobs, int n_jobs, void *stream) {
    if (!jobs || n_jobs <= 0 || n_jobs > QGTC_MAX_PACK_JOBS) return QGTC_EINVAL;
    PackJobs pj{};
    pj.n = n_jobs;
    size_t most = 0;
    for (int i = 0; i < n_jobs; i++) {
        const qgtc_pack_job &j = jobs[i];
        if (!j.x || !j.out || j.H <= 0 || j.W <= 0 || !bits_ok(j.nbits)) return QGTC_EINVAL;
        const size_t need = j.col_major ? static_cast<size_t>(j.nbits) * step128(j.H) * 4u * (j.output_layer ? pad8(j.W) : pad128(j.W))
                                        : static_cast<size_t>(j.nbits) * pad8(j.H) * step128(j.W) * 4u;
        if (j.out_words < need) return QGTC_ESIZE;
        most = std::max(most, need / j.nbits);
        pj.job[i] = j;
    }
    const unsigned gx = static_cast<unsigned>(std::min<size_t>((most + 255) / 256, 4096));
    hipLaunchKernelGGL(k_val2bit_jobs, dim3(gx, n_jobs), dim3(256), 0, static_cast<hipStream_t>(stream), pj);
    HIP_TRY(hipGetLastError());
    return QGTC_OK;
}

static bool stages_ok(const qgtc_stage *stages, int n_stages, int n_weights) {
    if (!stages || n_stages <= 0 || n_stages > QGTC_MAX_STAGES) return false;
    for (int s = 0; s < n_stages; s++) {
        const qgtc_stage &st = stages[s];
        // an operand is one of the batch's own tensors, a shared weight, or the packed output of an EARLIER stage (rows or
        // cols layout: the reference's literal chains feed rows-layout results as right operands, SURVEY.md section 3.1 -
        // reads are bounds-safe and the launch is the reference's)
        auto src_ok = [&](int src) {
            if (src >= QGTC_SRC_STAGE) return src - QGTC_SRC_STAGE < s && stages[src - QGTC_SRC_STAGE].mode != 2;
            if (src >= QGTC_SRC_WEIGHT) return src - QGTC_SRC_WEIGHT < n_weights;
            return src == QGTC_SRC_A || src == QGTC_SRC_X || src == QGTC_SRC_XR || src == QGTC_SRC_XC || src == QGTC_SRC_AT;
        };
        const bool l_ok = src_ok(st.left), r_ok = src_ok(st.right);
        if (!l_ok || !r_ok) return false;
        if ((st.K <= 0 && st.K != QGTC_DIM_NODES) || st.N <= 0) return false;
        if (!bits_ok(st.bit1) || !bits_ok(st.bit2) || st.mode < 0 || st.mode > 2 || (st.mode != 2 && !bits_ok(st.ob))) return false;
        if (st.use_occ && st.left != QGTC_SRC_A && st.left != QGTC_SRC_AT) return false;
        if (st.fmt != 0 && (st.fmt != 1 || st.mode != 1)) return false;
    }
    return true;
}

size_t qgtc_epoch_pool_layout(const int32_t *nodes, int count, const qgtc_stage *stages, int n_stages, uint64_t *offsets) {
    if (!nodes || count <= 0 || !stages || n_stages <= 0 || n_stages > QGTC_MAX_STAGES) return 0;
    unsigned long long total = 0ull;
    for (int s = 0; s < n_stages; s++)
        for (int b = 0; b < count; b++) {
            if (offsets) offsets[static_cast<size_t>(s) * count + b] = total;
            total += stage_out_words(stages[s], nodes[b]);
        }
    return static_cast<size_t>(total);
}

int qgtc_epoch_plan_fill(const qgtc_batch *batches, int count, const qgtc_stage *stages, int n_stages,
                         const qgtc_operand *weights, int n_weights, void *pool, size_t pool_words,
                         qgtc_problem *descs, void *stream) {
    if (!batches || !pool || !descs || count <= 0 || count > 65535) return QGTC_EINVAL;
    if (n_weights < 0 || n_weights > QGTC_MAX_WEIGHTS || (n_weights && !weights)) return QGTC_EINVAL;
    if (!stages_ok(stages, n_stages, n_weights)) return QGTC_EINVAL;
    if (!aligned16(pool)) return QGTC_EALIGN;
    for (int k = 0; k < n_weights; k++) {
        if (!weights[k].ptr || weights[k].words >= (1ull << 30)) return QGTC_EINVAL;
        if (!aligned16(weights[k].ptr)) return QGTC_EALIGN;
    }
    int *rec = nullptr;   // (the batches' node counts live on the device: a pool smaller than qgtc_epoch_pool_layout's figure is found there)
    const int rrc = record_for_current_device(&rec, true);
    if (rrc != QGTC_OK) return rrc;
    PlanArgs pa{};
    pa.n_stages = n_stages;
    pa.n_weights = n_weights;
    pa.count = count;
    for (int s = 0; s < n_stages; s++) pa.stage[s] = stages[s];
    for (int k = 0; k < n_weights; k++) pa.weight[k] = weights[k];
    hipLaunchKernelGGL(k_epoch_plan_fill, dim3(1), dim3(PLAN_THREADS), 0, static_cast<hipStream_t>(stream), batches, pa,
                       static_cast<uint32_t *>(pool), static_cast<unsigned long long>(pool_words), descs, rec);
    HIP_TRY(hipGetLastError());
    return QGTC_OK;
}

}  // extern "C"
