// bitmm_fp4_stream.hip.h — part of libqgtc_hip.so (qgtc_stream.hip).
// The 1-bit x 1-bit bit-GEMM on the matrix cores for LONG K and narrow right operands: the throughput-bound half of the
// reference's adjacency-size study (5_9_adjmatrix_size.py:15-18: M = K up to 32768, where the packed adjacency alone is
// 128 MiB and the HBM roofline is the one that binds). Same arithmetic as bitmm_fp4_one / _skinny (E2M1 codes of the bits
// on the block-scaled FP4 MFMAs, float32 sums of exact integers, K < 2^24), same words out.
#pragma once

namespace {

// ------------------------------------------------------------------------------------------
// What the launch is bound by: at 32768 x 32768 x 64 the adjacency is 128 MiB (16.8 us at 8 TB/s) and the product is 1 M MFMAs
// of 32 x 32 x 64 (32 cycles each: 15 us over 1024 SIMDs at 2.2 GHz) - both at once - and the vector ISSUE of a SIMD is as scarce
// as its matrix pipe: an MFMA holds it for 8 cycles whatever its shape, a VALU operation for 4 (MI355X guide, 'vector-instruction
// ISSUE cost'), and expanding bits into E2M1 codes is VALU work.
//   * X is read from HBM exactly ONCE: a workgroup owns 32 RF rows x up to 64 columns for the whole K;
//   * the eight waves of a workgroup have ROLES. Waves 0-3, one a SIMD, MULTIPLY: wave wk takes 64 of the 256 bytes a group of K
//     holds of every line, for all 32 RF rows x 32 CF columns (RF x CF fragments of v_mfma_scale_f32_32x32x64_f8f6f4). Waves 4-7
//     do nothing but FETCH: LDS-DMA pieces of 8 lines x 128 bytes (full cache lines; a piece costs its wave ~110 cycles of issue),
//     three stages, one barrier a group. The first forms had all eight waves do both, two multiplying waves a SIMD: those got LESS
//     out of the SIMD together than one alone (per group the older wave of a SIMD took 2200 cycles for its 32 MFMAs with the
//     younger getting 7 of its own done meanwhile, then 1000 more alone: 3650 a group where the pipe needs 2048; 33.7 us) - the
//     arbiter serves the older wave first and the two streams of MFMAs and expansions do not interleave. One multiplying wave
//     a SIMD with the whole 128 x 64 tile: RF = 4, CF = 2 needs (4 + 2) x 4 x 1.25 / 8 = 3.75 VALU operations per MFMA where
//     2 x 2 fragments needed 5: 8 + 15 = 23 issue cycles of the 32 an MFMA runs;
//   * the order of a step's instructions is written out and pinned (volatile asm anchors + sched_barrier): one MFMA, then at most
//     one expansion (four ANDs) and a few shifts for LATER MFMAs, per the tables of tools/stream_schedule.py. The MFMAs of a bit go
//     through the fragment pairs in snake order (one new operand each); expansions land in a ring of four A registers / two sets
//     of B registers, so none writes what an MFMA in flight reads; the last gaps of a step make the next step's first operands;
//   * which 128 elements of K an MFMA covers is free as long as X and W agree, so MFMA s = 0..3 of a step takes the bits
//     s, s + 4, .. of the lane's four words IN PLACE (nibble code 1 << s = 0.5, 1, 2 with the E8M0 scale 2^(1 - s); bit 3 is
//     E2M1's sign: the words are shifted right by 3, in place, between their bit-2 and bit-3 expansions);
//   * the 16-byte chunks of a line are XOR-swizzled on the SOURCE address of the DMA (chunk c of line r sits in slot
//     c ^ (r & 6)): a fragment read (16 lines x 4 chunks, one ds_read_b128) touches every bank once; the fragments of step u + 1
//     are read under the MFMAs of step u;
//   * the four partial tiles (the waves split K) meet in LDS once, at the end; then the whole workgroup re-quantises and packs:
//     all three outputs, every padding word written;
//   * all-zero (32 RF rows) x 256-bit steps of X are skipped with one ballot (a real adjacency is mostly that);
//   * more than 64 columns: column tiles of 64, the workgroups of a row tile consecutive on ONE XCD (they share X in L2).
// ------------------------------------------------------------------------------------------
constexpr int ST_WAVES = 8;     // waves per workgroup
constexpr int ST_MUL = 4;       // of them multiply (waves 0 .. 3, along K); the rest fetch
constexpr int ST_COLS = 64;     // columns of the LDS tile
constexpr int ST_PITCH = 68;    // ints between its rows: the four row groups of an MFMA's C registers land on banks 0 / 16 / 32 / 48
constexpr int ST_PIECE = 1024;  // bytes one LDS-DMA wave-instruction lands: 8 lines x 128 bytes
constexpr int ST_GB = 64 * ST_MUL;   // bytes of a line per group of K
constexpr int ST_STAGES = 3;

constexpr int st_pieces(int rf, int cf) { return (32 * rf / 8 + 32 * cf / 8) * (ST_GB / 128); }
constexpr int st_lds_bytes(int rf, int cf) {   // the stages; behind the loop the multiplying waves' partial tiles (4 rf cf quads x 1 KB each), then the int32 tile
    const int stages = ST_STAGES * st_pieces(rf, cf) * ST_PIECE, part = ST_MUL * 4 * rf * cf * 1024, tile = 32 * rf * ST_PITCH * 4;
    return stages > part ? (stages > tile ? stages : tile) : (part > tile ? part : tile);
}

// Behind MFMA n of a step: how many expansions (counted in the order the MFMAs need them, the step's own first three included; past the
// step's last: the NEXT step's first three) and how many one-dword shifts are out. tools/stream_schedule.py prints these rows.
template <int RF, int CF>
__device__ __forceinline__ constexpr int st_exp_by(int n) {
    if constexpr (RF == 2 && CF == 1) { constexpr int t[] = {5, 6, 8, 9, 11, 12, 14, 15}; return t[n]; }
    else if constexpr (RF == 2 && CF == 2) { constexpr int t[] = {4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19}; return t[n]; }
    else if constexpr (RF == 4 && CF == 1) { constexpr int t[] = {4, 5, 7, 8, 9, 10, 12, 13, 14, 15, 17, 18, 19, 20, 22, 23}; return t[n]; }
    else { constexpr int t[] = {4, 4, 5, 5, 6, 7, 8, 9, 10, 10, 11, 11, 12, 13, 14, 15, 16, 16, 17, 17, 18, 19, 20, 21, 22, 22, 23, 23, 24, 25, 26, 27}; return t[n]; }
}
template <int RF, int CF>
__device__ __forceinline__ constexpr int st_shift_by(int n) {
    if constexpr (RF == 2 && CF == 1) { constexpr int t[] = {0, 0, 0, 8, 12, 12, 12, 12}; return t[n]; }
    else if constexpr (RF == 2 && CF == 2) { constexpr int t[] = {0, 0, 0, 0, 0, 2, 4, 6, 8, 10, 12, 16, 16, 16, 16, 16}; return t[n]; }
    else if constexpr (RF == 4 && CF == 1) { constexpr int t[] = {0, 0, 0, 0, 0, 0, 0, 2, 4, 8, 12, 16, 20, 20, 20, 20}; return t[n]; }
    else { constexpr int t[] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 2, 4, 6, 8, 14, 16, 20, 22, 24, 24, 24, 24, 24, 24, 24, 24, 24, 24, 24}; return t[n]; }
}

// f(integral_constant<int, 0>) .. f(integral_constant<int, N - 1>): the step below is written out at compile time (its register arrays
// must be indexed by constants; hipcc's unroller gives up on a loop of 32 MFMAs with nested loops)
template <class F, int... I>
__device__ __forceinline__ void st_for_impl(F &&f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void st_for(F &&f) { st_for_impl(f, std::make_integer_sequence<int, N>{}); }

// one LDS-DMA instruction: lane i's 16 bytes at (voff + soff) of the buffer land at LDS byte lds_dst + 16 i.
// hipcc does not count this load: the kernel waits for it with its own s_waitcnt vmcnt.
__device__ __forceinline__ void st_dma(uint32_t lds_dst, uint32_t voff, i32x4 rsrc, uint32_t soff) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "s"(lds_dst), "v"(voff), "s"(rsrc), "s"(soff)
                 : "memory");
}

template <int MODE, int RF, int CF>
__global__ __launch_bounds__(64 * ST_WAVES) void k_bitmm_fp4_stream(
    const uint32_t *__restrict__ Xp, const uint32_t *__restrict__ Wp, void *__restrict__ outp, uint32_t x_bytes, uint32_t w_bytes,
    uint32_t out_bytes, int M, int K, int N, int w_lines,
    uint32_t cfg /* ob | zero_skip << 8 | column tiles << 16; host: every byte count < 2^31, M < 2^24 */, int n_wg) {
    static_assert((RF == 2 || RF == 4) && (CF == 1 || CF == 2), "32-line fragments of a multiplying wave: left, right");
    constexpr int TR = 32 * RF, WK = ST_MUL;
    constexpr int GB = ST_GB, BPG = GB / 128, GQ = GB / 16;              // bytes / 128-byte blocks / k-quads of a line per group
    constexpr int XP = (TR / 8) * BPG, WP = (32 * CF / 8) * BPG, TOT = XP + WP;   // pieces per stage: X's, W's
    constexpr int FW = ST_WAVES - ST_MUL;                                // fetching waves
    static_assert(XP % FW == 0 && TOT % FW == 0, "whole rounds of the fetching waves");
    constexpr int STAGE = TOT * ST_PIECE, STAGES = ST_STAGES;
    constexpr int DMAS = TOT / FW;                                       // per fetching wave and group
    static_assert(DMAS >= 1 && DMAS <= 31, "vmcnt immediate");
    extern __shared__ __attribute__((aligned(1024))) unsigned char st_lds[];
    int (*tile)[ST_PITCH] = reinterpret_cast<int (*)[ST_PITCH]>(st_lds);   // (after the last group: the stages' place)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fl = lane & 31, hf = lane >> 5;   // line of a 32-line fragment, half of the MFMA's 64 elements of K
#ifdef QGTC_STAMPS   // (tools/kbench.hip -DQGTC_STAMPS: s_memtime stamps of every wave's phases, kept in scalar registers)
    unsigned long long st_[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define ST_STAMP(i) st_[i] = __builtin_amdgcn_s_memtime()
#else
#define ST_STAMP(i) do { } while (0)
#endif
    ST_STAMP(0);
#ifdef QGTC_STAMPS
    st_[13] = __builtin_amdgcn_s_memrealtime();   // (100 MHz: slots 13 / 14 around the loop give the clock the chip held, MI355X guide DVFS item 6)
#endif
    const int ob = cfg & 255u, tiles_n = static_cast<int>(cfg >> 16);
    const bool zero_skip = ((cfg >> 8) & 1u) != 0u;
    int t = static_cast<int>(blockIdx.x);
    if (tiles_n > 1) t = xcd_consecutive(t, n_wg);   // the column tiles of a row tile run side by side on one XCD
    const int tm = t / tiles_n, tn = t - tm * tiles_n;
    const int m0 = tm * TR, n0 = tn * ST_COLS;
    const int kq = step128(K);
    const uint32_t row_bytes = static_cast<uint32_t>(kq) * 16u;
    const uint32_t lds0 = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(st_lds));
    (void)x_bytes; (void)w_bytes; (void)out_bytes; (void)w_lines;
    const int ng = (kq + GQ - 1) / GQ;
    constexpr int NQ = RF * CF * 4;   // register quads of a multiplying wave's accumulators
    f32x16 acc[RF][CF];

    if (wv >= ST_MUL) {
        // ---- a fetching wave: pieces fw, fw + 4, .. of the stage's list [X: line group, block][W: line group, block]. Lane (rr, cc) of
        // a piece fetches chunk cc ^ (rr & 6) of line rr. Lines past an operand are dropped by the range check (or read the zero padding
        // lines): the epilogue keeps nothing of them; chunks past K are zeroed where X is read from LDS.
        const int fw = wv - ST_MUL;
        const i32x4 rs_x = {static_cast<int>(reinterpret_cast<uintptr_t>(Xp)), static_cast<int>((reinterpret_cast<uintptr_t>(Xp) >> 32) & 0xffffu), static_cast<int>(x_bytes), 0x00020000};
        const i32x4 rs_w = {static_cast<int>(reinterpret_cast<uintptr_t>(Wp)), static_cast<int>((reinterpret_cast<uintptr_t>(Wp) >> 32) & 0xffffu), static_cast<int>(w_bytes), 0x00020000};
        const int rr = lane >> 3, cc = lane & 7;
        const uint32_t swz = static_cast<uint32_t>(cc ^ (rr & 6)) * 16u;
#ifdef QGTC_STREAM_TUNE   // timing-only build (tools/kbench.hip, ABL_X=1): every workgroup fetches the FIRST row tile of X - the same DMAs out of L2, no HBM
        const uint32_t voff_x = static_cast<uint32_t>(((cfg >> 9) & 1u ? 0 : m0) + rr) * row_bytes + swz, voff_w = static_cast<uint32_t>(n0 + rr) * row_bytes + swz;
        const bool no_dma = ((cfg >> 10) & 1u) != 0u;   // (ABL_NODMA=1: no piece is fetched at all)
#else
        const uint32_t voff_x = static_cast<uint32_t>(m0 + rr) * row_bytes + swz, voff_w = static_cast<uint32_t>(n0 + rr) * row_bytes + swz;
        constexpr bool no_dma = false;
#endif
        auto issue = [&](int g) {   // group g -> stage g % STAGES; groups past the last are not fetched (the last groups' waits count on that)
            if (g >= ng || no_dma) return;
            const uint32_t base = lds0 + static_cast<uint32_t>(g % STAGES) * STAGE;
            const uint32_t ko = static_cast<uint32_t>(g) * GB;
#pragma unroll
            for (int j = 0; j < DMAS; j++) {
                const int p = fw + FW * j;   // (scalar; X's pieces are whole rounds: rounds j < XP / FW are X's for every wave)
                if (FW * j < XP) {
                    st_dma(base + static_cast<uint32_t>(p) * ST_PIECE, voff_x, rs_x, ko + static_cast<uint32_t>(8 * (p / BPG)) * row_bytes + 128u * (p % BPG));
                } else {
                    const int pw = p - XP;
                    st_dma(base + static_cast<uint32_t>(p) * ST_PIECE, voff_w, rs_w, ko + static_cast<uint32_t>(8 * (pw / BPG)) * row_bytes + 128u * (pw % BPG));
                }
            }
        };
        issue(0);
        issue(1);
        for (int g = 0; g < ng; g++) {
            // group g has landed (behind its DMAs only group g + 1's are outstanding - none when g is the last group) ...
            if (g + 1 < ng) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DMAS) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef QGTC_STAMPS
            if (g == 0) ST_STAMP(1); else if (g == 5) ST_STAMP(3); else if (g == 6) ST_STAMP(7);
#endif
            __builtin_amdgcn_s_barrier();   // ... and the multiplying waves have read the last of group g - 1: its stage is free
#ifdef QGTC_STAMPS
            if (g == 0) ST_STAMP(2); else if (g == 5) ST_STAMP(4); else if (g == 6) ST_STAMP(8);
#endif
            issue(g + 2);
        }
        ST_STAMP(11);
    } else {
        // ---- a multiplying wave: its 64 bytes of a group are block wk >> 1, half wk & 1, in two steps u = 0, 1 of 32 bytes; lane
        // (fl, hf) takes chunk 4 (wk & 1) + 2 u + hf of that block of line fl of a fragment. Eight consecutive lanes read eight lines
        // of one piece: every bank once.
        const int wk = wv;
        const uint32_t frag_off = static_cast<uint32_t>(fl & 7) * 128u + static_cast<uint32_t>((4 * (wk & 1) + hf) ^ (fl & 6)) * 16u + static_cast<uint32_t>(wk >> 1) * ST_PIECE;
        const uint32_t xa0 = static_cast<uint32_t>((fl >> 3) * BPG) * ST_PIECE + frag_off;          // + 4 i line groups; step u: ^ 32 (chunk bit 1; fl & 6 keeps it)
        const uint32_t wa0 = static_cast<uint32_t>(XP + (fl >> 3) * BPG) * ST_PIECE + frag_off;     // + 4 j line groups
        const int my_q = 8 * (wk >> 1) + 4 * (wk & 1) + hf;   // the lane's k-quad of a group, step 0 (step 1: + 2)
#pragma unroll
        for (int i = 0; i < RF; i++)
#pragma unroll
            for (int j = 0; j < CF; j++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[i][j][r] = 0.0f;

        // (tail: the group may be the last one, whose chunks past K - whatever the DMA found there - must not count)
        auto fetch = [&](auto tail, int g, int u, u32x4 (&xr)[RF], u32x4 (&wr)[CF]) {
            const unsigned char *stage = st_lds + (g % STAGES) * STAGE;
#pragma unroll
            for (int i = 0; i < RF; i++) xr[i] = *reinterpret_cast<const u32x4 *>(stage + ((xa0 + static_cast<uint32_t>(4 * i * BPG) * ST_PIECE) ^ (32u * u)));
#pragma unroll
            for (int j = 0; j < CF; j++) wr[j] = *reinterpret_cast<const u32x4 *>(stage + ((wa0 + static_cast<uint32_t>(4 * j * BPG) * ST_PIECE) ^ (32u * u)));
            if constexpr (decltype(tail)::value) {
                if (g == ng - 1 && g * GQ + my_q + 2 * u >= kq) {
#pragma unroll
                    for (int i = 0; i < RF; i++) xr[i] = u32x4{0u, 0u, 0u, 0u};
                }
            }
        };
        // The operands of the MFMAs: a ring of four A registers (expansion a_i of bit s -> slot (s RF + i) & 3), two sets of B registers (s & 1)
        constexpr int MN = 4 * RF * CF, OPB = RF + CF, EN = 4 * OPB, EP = 3;   // per step: MFMAs; operands of a bit; expansions (EP of them made by the step before)
        i32x4 A[4], B[2][CF];
        // expansion e of a step, in the order its MFMAs need them: bit s = e / OPB, then a0, b0, (b1,) a1, (a2, a3)
        auto expand = [&](auto e_, u32x4 (&xr)[RF], u32x4 (&wr)[CF]) {
            constexpr int e = decltype(e_)::value, s = e / OPB, o = e % OPB;
            constexpr uint32_t mask = s < 3 ? 0x11111111u << s : 0x11111111u;   // (bit 3: the words are shifted by then)
            constexpr bool is_a = o == 0 || o > CF;
            constexpr int f = o == 0 ? 0 : (o > CF ? o - CF : o - 1);
            u32x4 v;
            if constexpr (is_a) v = xr[f]; else v = wr[f];
            const i32x4 r = {static_cast<int>(v.x & mask), static_cast<int>(v.y & mask), static_cast<int>(v.z & mask), static_cast<int>(v.w & mask)};
            // (the volatile asm anchors keep their order: an expansion stays in the gap it is written in)
            if constexpr (is_a) { A[(s * RF + f) & 3] = r; asm volatile("" : "+v"(A[(s * RF + f) & 3])); }
            else { B[s & 1][f] = r; asm volatile("" : "+v"(B[s & 1][f])); }
        };
        // shift z of a step: dword z & 3 of operand z >> 2 (the order above), in place
        auto shift = [&](auto z_, u32x4 (&xr)[RF], u32x4 (&wr)[CF]) {
            constexpr int z = decltype(z_)::value, o = z >> 2, el = z & 3;
            constexpr bool is_a = o == 0 || o > CF;
            constexpr int f = o == 0 ? 0 : (o > CF ? o - CF : o - 1);
            if constexpr (is_a) { xr[f][el] >>= 3; if constexpr (el == 3) asm volatile("" : "+v"(xr[f])); }
            else { wr[f][el] >>= 3; if constexpr (el == 3) asm volatile("" : "+v"(wr[f])); }
        };
        // wave-uniform: an all-zero 32 RF rows x 256-bit step of X is skipped. The OR of a step's fragments is made a step AHEAD, in a gap of
        // the step before (any_of; at 4 x 2 a gap that has no expansion: free); at the step itself only the ballot is left - ahead of
        // the step the eight ORs cost the one multiplying wave of a SIMD 0.7 us a launch
        auto any_of = [&](const u32x4 (&xr)[RF]) {
            uint32_t any = 0u;
#pragma unroll
            for (int i = 0; i < RF; i++) any |= (xr[i].x | xr[i].y) | (xr[i].z | xr[i].w);
            return any;
        };
        auto skipped = [&](uint32_t any) { return zero_skip && __ballot(any != 0u) == 0ull; };
        auto first = [&](u32x4 (&xr)[RF], u32x4 (&wr)[CF]) {
            st_for<EP>([&](auto e_) { expand(e_, xr, wr); });
        };
        // one step: its first EP operands are made; with next: the next step's fragments (nx, nw), whose first EP operands its last gaps make
        auto step = [&](auto with_next, u32x4 (&xr)[RF], u32x4 (&wr)[CF], u32x4 (&nx)[RF], u32x4 (&nw)[CF], uint32_t &any_next) {
            constexpr bool NEXT = decltype(with_next)::value;
            __builtin_amdgcn_sched_barrier(0);
            st_for<MN>([&](auto n_) {
                constexpr int n = decltype(n_)::value, s = n / (RF * CF), q = n % (RF * CF);
                constexpr int i = q / CF, j = (i & 1) ? CF - 1 - q % CF : q % CF;   // snake: one new operand per MFMA
                constexpr int sc = s < 3 ? 128 - s : 128;   // E8M0: code 1 << s counts as 1
                const i32x4 a4 = A[(s * RF + i) & 3], b4 = B[s & 1][j];
                const i32x8 a8 = __builtin_shufflevector(a4, a4, 0, 1, 2, 3, -1, -1, -1, -1);   // (an FP4 operand is the first 128 bits of the register tuple)
                const i32x8 b8 = __builtin_shufflevector(b4, b4, 0, 1, 2, 3, -1, -1, -1, -1);
                // cbsz = blgp = 4: E2M1 operands; lane (fl, hf) register r holds C[row 32 i + 8 (r >> 2) + 4 hf + (r & 3)][column 32 j + fl]
                acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, acc[i][j], 4, 4, 0, sc, 0, sc);
                constexpr int e0 = n == 0 ? EP : st_exp_by<RF, CF>(n == 0 ? 0 : n - 1), e1 = st_exp_by<RF, CF>(n);
                st_for<e1 - e0>([&](auto k_) {
                    constexpr int e = e0 + decltype(k_)::value;
                    if constexpr (e < EN) expand(std::integral_constant<int, e>{}, xr, wr);
                    else if constexpr (NEXT) expand(std::integral_constant<int, e - EN>{}, nx, nw);
                });
                constexpr int z0 = n == 0 ? 0 : st_shift_by<RF, CF>(n == 0 ? 0 : n - 1), z1 = st_shift_by<RF, CF>(n);
                st_for<z1 - z0>([&](auto k_) { shift(std::integral_constant<int, z0 + decltype(k_)::value>{}, xr, wr); });
                if constexpr (NEXT && n == MN - 7) {   // (the next fragments landed long ago)
                    any_next = any_of(nx);
                    asm volatile("" : "+v"(any_next));
                }
                __builtin_amdgcn_sched_barrier(0);
            });
        };
        using with_next = std::integral_constant<bool, true>;
        using last_step = std::integral_constant<bool, false>;
        uint32_t any_now = 0u;   // the OR of the fragments the next multiply takes
        auto multiply = [&](u32x4 (&xr)[RF], u32x4 (&wr)[CF], u32x4 (&nx)[RF], u32x4 (&nw)[CF]) {
            uint32_t any_next;
            if (skipped(any_now)) {
                first(nx, nw);
                any_next = any_of(nx);
            } else {
                step(with_next{}, xr, wr, nx, nw, any_next);
            }
            any_now = any_next;
        };
        // A group's barrier says: its pieces have landed, for every wave. The fragments of step u + 1 are read under the MFMAs of step u,
        // the first step of group g + 1 under the last of group g (behind barrier g + 1): nothing else covers a read's latency.
        u32x4 xp[RF], wp[CF], xq[RF], wq[CF];
        using any_group = std::integral_constant<bool, true>;
        using not_last = std::integral_constant<bool, false>;
        // one trip: the two steps of group g (its first fragments are in xp / wp, their first operands made), group g + 1's first fragments
        auto trip = [&](auto tail, int g) {
            fetch(tail, g, 1, xq, wq);
            multiply(xp, wp, xq, wq);                            // (g, 0)
#ifdef QGTC_STAMPS
            if (g == 5) ST_STAMP(6); else if (g == 6) ST_STAMP(10);
#endif
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // my reads of group g have returned: behind the barrier its stage is refilled
            __builtin_amdgcn_s_barrier();
#ifdef QGTC_STAMPS
            if (g == 4) ST_STAMP(4); else if (g == 5) ST_STAMP(8);
#endif
            fetch(tail, g + 1, 0, xp, wp);
            multiply(xq, wq, xp, wp);                            // (g, 1)
        };
        __builtin_amdgcn_s_barrier();   // group 0
        ST_STAMP(2);
        fetch(any_group{}, 0, 0, xp, wp);
        first(xp, wp);
        any_now = any_of(xp);
        int g = 0;
        for (; g + 2 < ng; g++) trip(not_last{}, g);
        for (; g + 1 < ng; g++) trip(any_group{}, g);            // (at most one trip: the one that fetches from the last group)
        fetch(any_group{}, g, 1, xq, wq);
        multiply(xp, wp, xq, wq);                                // (ng - 1, 0)
        if (!skipped(any_now)) step(last_step{}, xq, wq, xp, wp, any_now);     // (ng - 1, 1)
#ifdef QGTC_STAMPS
        asm volatile("" : "+v"(acc[0][0]));
#endif
        ST_STAMP(11);
    }
#ifdef QGTC_STAMPS
    st_[14] = __builtin_amdgcn_s_memrealtime();
#endif
    __syncthreads();   // (every multiplying wave has read its last fragments, every piece has landed: the stages' place is free)

    // ---- the WK partial tiles meet through LDS: every multiplying wave leaves its accumulators there as they sit in its registers (16
    // bytes a lane and register quad: lane-linear, no conflicts), then each of the eight waves sums ITS share of the quads over the WK
    // copies and puts the sums - exact integers below 2^24 as float32, any order - into the int32 tile the epilogue reads.
    constexpr int QW = NQ / ST_WAVES;   // quads a wave sums
    static_assert(NQ % ST_WAVES == 0, "the quads split evenly over the waves");
    {
        f32x4 *part = reinterpret_cast<f32x4 *>(st_lds);   // [multiplying wave][quad][lane]
        if (wv < ST_MUL) {
#pragma unroll
            for (int i = 0; i < RF; i++)
#pragma unroll
                for (int j = 0; j < CF; j++)
#pragma unroll
                    for (int rq = 0; rq < 4; rq++)
                        part[(wv * NQ + (i * CF + j) * 4 + rq) * 64 + lane] = f32x4{acc[i][j][4 * rq], acc[i][j][4 * rq + 1], acc[i][j][4 * rq + 2], acc[i][j][4 * rq + 3]};
        }
        __syncthreads();
        f32x4 sum[QW];
#pragma unroll
        for (int tq = 0; tq < QW; tq++) {
            sum[tq] = part[(wv * QW + tq) * 64 + lane];
#pragma unroll
            for (int w2 = 1; w2 < WK; w2++) sum[tq] += part[(w2 * NQ + wv * QW + tq) * 64 + lane];
        }
        __syncthreads();   // (the tile overlaps the partial tiles)
#pragma unroll
        for (int tq = 0; tq < QW; tq++) {
            const int q = wv * QW + tq, f = q >> 2, rq = q & 3;   // fragment (f / CF, f % CF), rows 8 rq + 4 hf .. + 3 of it, column fl
            int *dst = &tile[32 * (f / CF) + 8 * rq + 4 * hf][32 * (f % CF) + fl];
#pragma unroll
            for (int e = 0; e < 4; e++) dst[e * ST_PITCH] = static_cast<int>(sum[tq][e]);
        }
        if (CF == 1) {   // (the tile's columns 32 .. 63 are read by the epilogue: nothing was computed there)
            for (int e = tid; e < TR * 32; e += 64 * ST_WAVES) tile[e >> 5][32 + (e & 31)] = 0;
        }
    }
    __syncthreads();
#ifdef QGTC_STAMPS
    ST_STAMP(12);
    if (lane == 0 && (blockIdx.x == 0 || blockIdx.x == 100))   // slots 0 .. 7: the eight waves of workgroup 0; 8 .. 15: of workgroup 100
        for (int i = 0; i < 16; i++) g_stamps[((blockIdx.x ? 8 : 0) + wv) * 16 + i] = st_[i];
#endif
#undef ST_STAMP

    const float maxv = __builtin_ldexpf(1.0f, ob), maxm1 = maxv - 1.0f;
    const int maxi = 1 << (ob & 31);
    const bool int_rq = ob <= 23;   // float(c) > 2^ob  <=>  c > 2^ob for every int 0 <= c < 2^24
    const int tiles_m = (M + TR - 1) / TR;
    const bool last_m = tm == tiles_m - 1, last_n = tn == tiles_n - 1;
    (void)last_m;

    if (MODE == 1) {
        // cols layout [ob][PAD128(N)][STEP128(M) * 4] (intended semantics of kernel.h:651-810): word (n, m / 32). Thread (col, rq):
        // rows 4 rq .. 4 rq + 3 of every 32-row group of its column
        const int col = tid >> 3, rq = tid & 7;
        const int lines = pad128(N), line_words = step128(M) * 4;
        const size_t oplane = static_cast<size_t>(lines) * line_words;
        uint32_t *out = static_cast<uint32_t *>(outp);
        const int n1 = n0 + col, word0 = m0 >> 5;
#pragma unroll
        for (int gq = 0; gq < TR / 32; gq++) {
            uint32_t q[4];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int c = tile[32 * gq + 4 * rq + e][col];
                const int r = int_rq ? (c > maxi ? maxi - 1 : c) : requant(c, maxv, maxm1);
                q[e] = (m0 + 32 * gq + 4 * rq + e < M && n1 < N) ? static_cast<uint32_t>(r) : 0u;
            }
            uint32_t *dst = out + static_cast<size_t>(n1) * line_words + word0 + gq;
            for (int p = 0; p < ob; p++, dst += oplane) {
                const uint32_t nib = (((q[0] >> p) & 1u) << 3) | (((q[1] >> p) & 1u) << 2) | (((q[2] >> p) & 1u) << 1) | ((q[3] >> p) & 1u);
                const uint32_t wrd = or_reduce8(nib << (28u - 4u * static_cast<uint32_t>(rq)));   // row 4 rq + e at bit 31 - 4 rq - e
                if (rq == 0 && n1 < lines && word0 + gq < line_words) dst[0] = wrd;
            }
        }
        // zero what no tile computes: words past the last row tile, lines past the last column tile
        const int w_core = min(line_words, word0 + TR / 32);
        if (last_m && w_core < line_words) {
            for (int e = tid; e < ob * ST_COLS; e += 64 * ST_WAVES) {
                const int line = n0 + (e & (ST_COLS - 1)), p = e / ST_COLS;
                if (line < lines)
                    for (int wi = w_core; wi < line_words; wi++) out[p * oplane + static_cast<size_t>(line) * line_words + wi] = 0u;
            }
        }
        if (last_n && n0 + ST_COLS < lines) {
            const int nl = lines - (n0 + ST_COLS), w_end = last_m ? line_words : w_core;
            for (int e = tid; e < ob * nl; e += 64 * ST_WAVES) {
                const int line = n0 + ST_COLS + e % nl, p = e / nl;
                for (int wi = word0; wi < w_end; wi++) out[p * oplane + static_cast<size_t>(line) * line_words + wi] = 0u;
            }
        }
        return;
    }

    // thread (row, quad) of a pass over 32 rows: four consecutive columns 4 quad .. of its row
    constexpr int QPR = ST_COLS / 4;
    const int rows_pad = pad8(M), row_words = step128(N) * 4;
    const size_t oplane = static_cast<size_t>(rows_pad) * row_words;
#pragma unroll
    for (int pass = 0; pass < TR / 32; pass++) {
        const int row = 32 * pass + tid / QPR, quad = tid % QPR;
        const i32x4 c4 = *reinterpret_cast<const i32x4 *>(&tile[row][4 * quad]);
        const int m = m0 + row, n = n0 + 4 * quad;
        if (MODE == 2) {   // float32 [M,N] (reference kernel.h:915-930)
            if (m < M) {
                float *dst = static_cast<float *>(outp) + static_cast<size_t>(m) * N + n;
                if (n + 3 < N && (N & 3) == 0) {
                    *reinterpret_cast<f32x4 *>(dst) = f32x4{static_cast<float>(c4.x), static_cast<float>(c4.y), static_cast<float>(c4.z), static_cast<float>(c4.w)};
                } else {
#pragma unroll
                    for (int e = 0; e < 4; e++)
                        if (n + e < N) dst[e] = static_cast<float>(c4[e]);
                }
            }
            continue;
        }
        // rows layout [ob][PAD8(M)][STEP128(N) * 4] (reference kernel.h:357-389): word (m, n / 32)
        uint32_t q[4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const int c = c4[e];
            const int r = int_rq ? (c > maxi ? maxi - 1 : c) : requant(c, maxv, maxm1);
            q[e] = (m < M && n + e < N) ? static_cast<uint32_t>(r) : 0u;
        }
        const bool store = (quad & 7) == 0 && m < rows_pad;
        const uint32_t sh_n = 28u - 4u * static_cast<uint32_t>(quad & 7);
        const int word = (n0 >> 5) + (quad >> 3);
        // the last column tile also zeroes the row words past it (the kernels write every word of the output)
        const int extra = (last_n && quad == QPR - 8) ? row_words - word - 1 : 0;
        uint32_t *dst = static_cast<uint32_t *>(outp) + static_cast<size_t>(m) * row_words + word;
        for (int p = 0; p < ob; p++, dst += oplane) {
            const uint32_t nib = (((q[0] >> p) & 1u) << 3) | (((q[1] >> p) & 1u) << 2) | (((q[2] >> p) & 1u) << 1) | ((q[3] >> p) & 1u);
            const uint32_t wrd = or_reduce8(nib << sh_n);
            if (store) {
                if (word < row_words) dst[0] = wrd;
                for (int x = 1; x <= extra; x++) dst[x] = 0u;
            }
        }
    }
}

}  // namespace
