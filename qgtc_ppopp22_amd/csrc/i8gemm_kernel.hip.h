// i8gemm_kernel.hip.h — part of libqgtc_hip.so (included by qgtc_hip.hip, one translation unit).
// The int8 MFMA comparison GEMM (the analogue of the reference's cuBLASGemmEX benchmark).
#pragma once

namespace {

// ------------------------------------------------------------------------------------------
// int8 MFMA GEMM, the comparison path (analogue of the reference's cuBLASGemmEX benchmark,
// cublas_main.cu:123-172): C[M,N] (float32) = A[M,K] (int8, K contiguous) x B[K,N] given as
// Bt[N,K] (int8, K contiguous), int32 accumulation on v_mfma_i32_16x16x64_i8, exact.
// A workgroup owns a 16-row x 64-column tile; its waves split K (each wave streams its slice of
// the A rows and B lines straight into MFMA fragments, 16 B per lane per load) and are summed
// through LDS. Out-of-range rows / columns read as zero through buffer range checks.
// ------------------------------------------------------------------------------------------
constexpr int I8_TM = 16, I8_TN = 64, I8_WAVES = 8;

__global__ __launch_bounds__(64 * I8_WAVES) void k_i8gemm(const int8_t *__restrict__ A,
                                                          const int8_t *__restrict__ Bt, int M, int K,
                                                          int N, float *__restrict__ C, int tiles_n) {
    __shared__ int red[I8_WAVES][4][4][64];  // [wave][column sub-tile][acc register][lane]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
    const int m0 = tm * I8_TM, n0 = tn * I8_TN;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<int8_t *>(A), 0, static_cast<int>(static_cast<uint32_t>(static_cast<size_t>(M) * K)), 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<int8_t *>(Bt), 0, static_cast<int>(static_cast<uint32_t>(static_cast<size_t>(N) * K)), 0x00020000);
    // fragment maps of mfma_i32_16x16x64_i8: lane l holds A[row l&15][k = 16*(l>>4) + 0..15] and
    // B[k = 16*(l>>4) + 0..15][col l&15]; C/D: col = l&15, row = 4*(l>>4) + reg
    const int fr = lane & 15, fk = (lane >> 4) * 16;
    const int ksteps = (K + 63) / 64;
    const int per = (ksteps + I8_WAVES - 1) / I8_WAVES;
    const int s0 = wv * per, s1 = min(s0 + per, ksteps);
    const bool row_ok = m0 + fr < M;
    uint32_t a_off = static_cast<uint32_t>(m0 + fr) * K + fk;
    uint32_t b_off[4];
    bool col_ok[4];
#pragma unroll
    for (int c = 0; c < 4; c++) {
        col_ok[c] = n0 + 16 * c + fr < N;
        b_off[c] = static_cast<uint32_t>(n0 + 16 * c + fr) * K + fk;
    }
    i32x4 acc[4];
#pragma unroll
    for (int c = 0; c < 4; c++) acc[c] = i32x4{0, 0, 0, 0};
    // Loads are unconditional: a k-step beyond this wave's slice (or beyond K) loads from offset
    // 0xffffffff, which the range check turns into zeros without touching memory, so the loop
    // has no branches around loads and hipcc can count vmcnt exactly. Four k-steps in flight.
    auto load = [&](int s, i32x4 &af, i32x4 (&bf)[4]) {
        const bool k_ok = s < s1 && s * 64 + fk < K;  // K % 16 == 0: 16-byte groups are all-in or all-out
        af = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                           ra, (row_ok && k_ok) ? a_off + s * 64 : 0xffffffffu, 0, 0));
#pragma unroll
        for (int c = 0; c < 4; c++)
            bf[c] = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                  rb, (col_ok[c] && k_ok) ? b_off[c] + s * 64 : 0xffffffffu, 0, 0));
    };
    auto mac = [&](const i32x4 &af, const i32x4 (&bf)[4]) {
#pragma unroll
        for (int c = 0; c < 4; c++) acc[c] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af, bf[c], acc[c], 0, 0, 0);
    };
    i32x4 a0, b0[4], a1, b1[4], a2, b2[4], a3, b3[4];
    load(s0, a0, b0);
    load(s0 + 1, a1, b1);
    load(s0 + 2, a2, b2);
    for (int s = s0; s < s1; s += 4) {
        load(s + 3, a3, b3);
        mac(a0, b0);
        load(s + 4, a0, b0);
        mac(a1, b1);
        load(s + 5, a1, b1);
        mac(a2, b2);
        load(s + 6, a2, b2);
        mac(a3, b3);
    }
#pragma unroll
    for (int c = 0; c < 4; c++)
#pragma unroll
        for (int r = 0; r < 4; r++) red[wv][c][r][lane] = acc[c][r];
    __syncthreads();
    // 16 x 64 outputs, 1024 (sub-tile, register, lane) slots over 512 threads
    for (int e = tid; e < 4 * 4 * 64; e += 64 * I8_WAVES) {
        const int l = e & 63, r = (e >> 6) & 3, c = e >> 8;
        int v = 0;
#pragma unroll
        for (int k = 0; k < I8_WAVES; k++) v += red[k][c][r][l];
        const int m = m0 + 4 * (l >> 4) + r, n = n0 + 16 * c + (l & 15);
        if (m < M && n < N) C[static_cast<size_t>(m) * N + n] = static_cast<float>(v);
    }
}

}  // namespace
