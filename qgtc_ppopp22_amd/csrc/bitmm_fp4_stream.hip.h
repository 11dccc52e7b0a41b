// bitmm_fp4_stream.hip.h — part of libqgtc_hip.so (qgtc_stream.hip).
// The 1-bit x 1-bit bit-GEMM on the matrix cores for LONG K and narrow right operands: the throughput-bound half of the
// reference's adjacency-size study (5_9_adjmatrix_size.py:15-18: M = K up to 32768, where the packed adjacency alone is
// 128 MiB and the HBM roofline is the one that binds). Same arithmetic as bitmm_fp4_one / _skinny (E2M1 codes of the bits
// on v_mfma_scale_f32_16x16x128_f8f6f4, float32 sums of exact integers, K < 2^24), same words out.
#pragma once

namespace {

// ------------------------------------------------------------------------------------------
// What the launch is bound by: at 32768 x 32768 x 64 the adjacency is 128 MiB (16.8 us at 8 TB/s) and the product is
// 2.1 M MFMAs of 16 x 16 x 128 (16.2 us at the 8.5 PFLOP/s the matrix pipes sustain, tools/mfma_rates2.hip) - both at
// once, and every VALU operation a wave spends on expanding operands costs its SIMD 1.55 ns on top of an MFMA's 7.8.
//   * X is read from HBM exactly ONCE: a workgroup owns 64 WM rows x up to 64 columns for the whole K (k_bitmm_fp4_skinny's
//     32 x 32 tiles read the adjacency once per 32 columns and expand both operands four times per MFMA);
//   * a WAVE multiplies 2 x CF fragments of v_mfma_scale_f32_32x32x64_f8f6f4 (64 x 64 outputs at CF = 2): 5 VALU operations per
//     MFMA with the in-place expansion below. The 32 x 32 shape, not 16 x 16 x 128: an MFMA holds its SIMD's vector issue for 8
//     cycles whatever its shape (MI355X guide, 'vector-instruction ISSUE cost'), so per 16 x 16 x 128 MFMA (16 cycles of the
//     pipe) 8 + 2.5 x 4 = 18 issue cycles were needed - the first DMA form ran 30 cycles per MFMA and SIMD by its stamps,
//     37 us - where a 32 x 32 x 64 MFMA (32 cycles) needs 8 + 5 x 4 = 28: the matrix pipe is the bound again;
//   * the operands reach the waves through LDS, fetched by LDS-DMA in pieces of 8 lines x 128 bytes = FULL cache lines.
//     The first form of this kernel loaded the MFMA operand layout straight from memory - lane (line, k-group) takes 16 bytes
//     of ITS line, so the 16 lanes of a quarter wave touch 16 different cache lines - and the launch's time followed the
//     number of such load instructions whatever they hit (47 us; 25 us with every load an L1 hit and no MFMA at all:
//     the texture path handles one line a clock). A DMA piece is two lines per quarter wave;
//   * K is walked in groups of GB = 64 WK bytes of every line, three stages (two groups in flight; two stages where three do
//     not fit); the waves of a workgroup are WM (rows) x WK (K): wave (wm, wk) multiplies the 64 bytes wk of the group for
//     its 64 rows - one step of 64 MFMAs a group - so the cross-wave sum over K happens once, at the end;
//   * the 16-byte chunks of a line are XOR-swizzled on the SOURCE address of the DMA (chunk c of line r sits in slot
//     c ^ (r & 6)): a fragment read (16 lines x 4 chunks, one ds_read_b128) touches every bank once;
//   * which 128 elements of K an MFMA covers is free as long as X and W agree, so MFMA s = 0..3 of a step takes the bits
//     s, s + 4, .. of the lane's four words IN PLACE (nibble code 1 << s = 0.5, 1, 2 with the E8M0 scale 2^(1 - s); the
//     fourth with one shift: code 8 is -0): one AND per operand dword;
//   * the waves' partial tiles meet in ONE int32 tile in LDS (ds_add_u32: exact, order-free; it takes the stages' place),
//     then the whole workgroup re-quantises and packs it: all three outputs, every padding word written;
//   * all-zero (64 rows) x 512-bit steps of X are skipped with one ballot (a real adjacency is mostly that);
//   * more than 64 columns: column tiles of 64, the workgroups of a row tile consecutive on ONE XCD (they share X in L2).
// ------------------------------------------------------------------------------------------
constexpr int ST_WAVES = 8;     // waves per workgroup: WM x WK
constexpr int ST_COLS = 64;     // columns of the LDS tile
constexpr int ST_PITCH = 68;    // ints between its rows: the four row groups of an MFMA's C registers land on banks 0 / 16 / 32 / 48
constexpr int ST_PIECE = 1024;  // bytes one LDS-DMA wave-instruction lands: 8 lines x 128 bytes
constexpr int ST_LDS_MAX = 160 * 1024;

constexpr int st_gb(int wm) { return 64 * (ST_WAVES / wm); }                                   // bytes of a line per group of K
constexpr int st_pieces(int wm, int cf) { return (64 * wm / 8 + 32 * cf / 8) * (st_gb(wm) / 128); }
constexpr int st_spare(int wm, int cf) { return st_pieces(wm, cf) % ST_WAVES ? 1 : 0; }
constexpr int st_stages(int wm, int cf) { return (3 * st_pieces(wm, cf) + st_spare(wm, cf)) * ST_PIECE <= ST_LDS_MAX ? 3 : 2; }
constexpr int st_lds_bytes(int wm, int cf) {   // the stages; behind the loop the waves' partial tiles (8 waves x 8 cf quads x 1 KB), then the int32 tile
    const int stages = (st_stages(wm, cf) * st_pieces(wm, cf) + st_spare(wm, cf)) * ST_PIECE, part = ST_WAVES * 8 * cf * 1024, tile = 64 * wm * ST_PITCH * 4;
    return stages > part ? (stages > tile ? stages : tile) : (part > tile ? part : tile);
}

// one LDS-DMA instruction: lane i's 16 bytes at (voff + soff) of the buffer land at LDS byte lds_dst + 16 i.
// hipcc does not count this load: the kernel waits for it with its own s_waitcnt vmcnt.
__device__ __forceinline__ void st_dma(uint32_t lds_dst, uint32_t voff, i32x4 rsrc, uint32_t soff) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "s"(lds_dst), "v"(voff), "s"(rsrc), "s"(soff)
                 : "memory");
}

template <int MODE, int WM, int CF>
__global__ __launch_bounds__(64 * ST_WAVES) void k_bitmm_fp4_stream(
    const uint32_t *__restrict__ Xp, const uint32_t *__restrict__ Wp, void *__restrict__ outp, uint32_t x_bytes, uint32_t w_bytes,
    uint32_t out_bytes, int M, int K, int N, int w_lines,
    uint32_t cfg /* ob | zero_skip << 8 | column tiles << 16; host: every byte count < 2^31, M < 2^24 */, int n_wg) {
    static_assert((WM == 1 || WM == 2) && (CF == 1 || CF == 2), "waves along M, right-hand 32-line fragments of a wave");
    constexpr int RF = 2, WK = ST_WAVES / WM, TR = 64 * WM;
    constexpr int GB = st_gb(WM), BPG = GB / 128, GQ = GB / 16;          // bytes / 128-byte blocks / k-quads of a line per group
    constexpr int XP = (TR / 8) * BPG, WP = (32 * CF / 8) * BPG, TOT = XP + WP;   // pieces per stage: X's, W's
    static_assert(XP % ST_WAVES == 0, "X's pieces: whole rounds of the waves");
    constexpr int STAGE = TOT * ST_PIECE, STAGES = st_stages(WM, CF);
    constexpr int DMAS = (TOT + ST_WAVES - 1) / ST_WAVES;                // per wave and group
    static_assert(DMAS >= 1 && DMAS <= 15, "vmcnt immediate");
    extern __shared__ __attribute__((aligned(1024))) unsigned char st_lds[];
    int (*tile)[ST_PITCH] = reinterpret_cast<int (*)[ST_PITCH]>(st_lds);   // (after the last group: the stages' place)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fl = lane & 31, hf = lane >> 5;   // line of a 32-line fragment, half of the MFMA's 64 elements of K
    const int wm = wv / WK, wk = wv % WK;
#ifdef QGTC_STAMPS   // (tools/kbench.hip -DQGTC_STAMPS: s_memtime stamps of wave 0's phases, kept in scalar registers)
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

    // ---- the DMA plan of this wave: pieces wv, wv + 8, .. of the stage's list [X: line group, block][W: line group, block]. Lane
    // (rr, cc) of a piece fetches chunk cc ^ (rr & 6) of line rr. Lines past an operand are dropped by the range check (or read
    // the zero padding lines): the epilogue keeps nothing of them; chunks past K are zeroed where X is read from LDS.
    const i32x4 rs_x = {static_cast<int>(reinterpret_cast<uintptr_t>(Xp)), static_cast<int>((reinterpret_cast<uintptr_t>(Xp) >> 32) & 0xffffu), static_cast<int>(x_bytes), 0x00020000};
    const i32x4 rs_w = {static_cast<int>(reinterpret_cast<uintptr_t>(Wp)), static_cast<int>((reinterpret_cast<uintptr_t>(Wp) >> 32) & 0xffffu), static_cast<int>(w_bytes), 0x00020000};
    const int rr = lane >> 3, cc = lane & 7;
    const uint32_t swz = static_cast<uint32_t>(cc ^ (rr & 6)) * 16u;
#ifdef QGTC_STREAM_TUNE   // timing-only build (tools/kbench.hip, ABL_X=1): every workgroup fetches the FIRST row tile of X - the same DMAs out of L2, no HBM
    const uint32_t voff_x = static_cast<uint32_t>(((cfg >> 9) & 1u ? 0 : m0) + rr) * row_bytes + swz, voff_w = static_cast<uint32_t>(n0 + rr) * row_bytes + swz;
#else
    const uint32_t voff_x = static_cast<uint32_t>(m0 + rr) * row_bytes + swz, voff_w = static_cast<uint32_t>(n0 + rr) * row_bytes + swz;
#endif
    const int ng = (kq + GQ - 1) / GQ;
    auto issue_one = [&](int g, int j) {   // DMA j of this wave for group g -> stage g % STAGES (the callers never ask for a group past the last)
        const uint32_t base = lds0 + static_cast<uint32_t>(g % STAGES) * STAGE;
        const uint32_t ko = static_cast<uint32_t>(g) * GB;
        const int p = wv + ST_WAVES * j;   // (scalar; X's pieces are a multiple of eight: rounds j < XP / 8 are X's for every wave)
        if (ST_WAVES * j < XP) {
            st_dma(base + static_cast<uint32_t>(p) * ST_PIECE, voff_x, rs_x, ko + static_cast<uint32_t>(8 * (p / BPG)) * row_bytes + 128u * (p % BPG));
        } else {   // (a wave without a piece in the last round: the same count of outstanding loads, dropped by the range check)
            const int pw = p - XP;
            const bool real = TOT % ST_WAVES == 0 || p < TOT;
            st_dma(real ? base + static_cast<uint32_t>(p) * ST_PIECE : lds0 + static_cast<uint32_t>(STAGES) * STAGE, voff_w, rs_w,
                   real ? ko + static_cast<uint32_t>(8 * (pw / BPG)) * row_bytes + 128u * (pw % BPG) : 0xfffffff0u);
        }
    };
#ifdef QGTC_STREAM_TUNE   // timing-only (ABL_NODMA=1): no piece is fetched at all - the loop's barriers, fragment reads, expansions and MFMAs alone
    const bool no_dma = ((cfg >> 10) & 1u) != 0u;
#else
    constexpr bool no_dma = false;
#endif
    auto issue = [&](int g) {   // (wave-uniform; the last group's wait counts on groups past the last not being fetched: publish)
        if (g >= ng || no_dma) return;
#pragma unroll
        for (int j = 0; j < DMAS; j++) issue_one(g, j);
    };
    issue(0);
    if constexpr (STAGES == 3) issue(1);   // (group 1 of a one-group K: not fetched)

    // ---- the fragment reads of this wave: its 64 bytes of the group are block wk >> 1, half wk & 1, in two steps u = 0, 1 of 32 bytes;
    // lane (fl, hf) takes chunk 4 (wk & 1) + 2 u + hf of that block of line fl of the fragment. Eight consecutive lanes read eight
    // lines of one piece: every bank once.
    const uint32_t frag_off = static_cast<uint32_t>(fl & 7) * 128u + static_cast<uint32_t>((4 * (wk & 1) + hf) ^ (fl & 6)) * 16u + static_cast<uint32_t>(wk >> 1) * ST_PIECE;
    const uint32_t xa0 = static_cast<uint32_t>((8 * wm + (fl >> 3)) * BPG) * ST_PIECE + frag_off;    // + 4 i line groups; step u: ^ 32 (chunk bit 1; fl & 6 keeps it)
    const uint32_t wa0 = static_cast<uint32_t>(XP + (fl >> 3) * BPG) * ST_PIECE + frag_off;          // + 4 j line groups
    const int my_q = 8 * (wk >> 1) + 4 * (wk & 1) + hf;   // the lane's k-quad of a group, step 0 (step 1: + 2)

    f32x16 acc[RF][CF];
#pragma unroll
    for (int i = 0; i < RF; i++)
#pragma unroll
        for (int j = 0; j < CF; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.0f;

    // group g has landed for every wave that passes the barrier; the stage of group g - 1 - read into registers an iteration ago - is
    // free again (lgkmcnt: those reads have returned)
    auto publish = [&](int g) {
        // (behind group g's DMAs only group g + 1's are outstanding - none when g is the last group)
        if (STAGES == 3 && g + 1 < ng) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(DMAS) : "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#ifdef QGTC_STAMPS
        if (g == 0) ST_STAMP(1); else if (g == 5) ST_STAMP(3); else if (g == 6) ST_STAMP(7);
#endif
#ifdef QGTC_STREAM_TUNE   // timing-only (ABL_NOBAR=1): no barrier inside the loop
        if (!(((cfg >> 12) & 1u) && g > 0))
#endif
        __builtin_amdgcn_s_barrier();
#ifdef QGTC_STAMPS
        if (g == 0) ST_STAMP(2); else if (g == 5) ST_STAMP(4); else if (g == 6) ST_STAMP(8);
#endif
    };
    auto fetch = [&](int g, u32x4 (&xr)[2][RF], u32x4 (&wr)[2][CF]) {
#ifdef QGTC_STREAM_TUNE   // timing-only (ABL_NOLDS=1): the fragments are read once, for group 0, and multiplied ng times
        if (((cfg >> 11) & 1u) && g > 1) return;
#endif
        const unsigned char *stage = st_lds + (g % STAGES) * STAGE;
#pragma unroll
        for (int u = 0; u < 2; u++) {
#pragma unroll
            for (int i = 0; i < RF; i++) xr[u][i] = *reinterpret_cast<const u32x4 *>(stage + ((xa0 + static_cast<uint32_t>(4 * i * BPG) * ST_PIECE) ^ (32u * u)));
#pragma unroll
            for (int j = 0; j < CF; j++) wr[u][j] = *reinterpret_cast<const u32x4 *>(stage + ((wa0 + static_cast<uint32_t>(4 * j * BPG) * ST_PIECE) ^ (32u * u)));
        }
        if (g == ng - 1) {   // a chunk past K: whatever the DMA found there must not count
#pragma unroll
            for (int u = 0; u < 2; u++)
                if (g * GQ + my_q + 2 * u >= kq)
#pragma unroll
                    for (int i = 0; i < RF; i++) xr[u][i] = u32x4{0u, 0u, 0u, 0u};
        }
    };
    // ... and the DMAs of group gn = g + STAGES - 1 go out BETWEEN the MFMAs of group g - 1, one per two blocks of CF MFMAs: issued in
    // a burst behind the barrier they held both waves of every SIMD at once (~100 cycles a piece, stamps: 590 of a group's 4100).
    //
    // The order of a step's instructions is written out and pinned (sched_barrier), one MFMA and then ONE operand expansion (four ANDs)
    // and one or two of the step's shifts: the vector issue of a SIMD is as scarce here as its matrix pipe (8 cycles an MFMA + 4 a VALU
    // operation = 28 of the 32 an MFMA runs), and hipcc's own order - the expansions of a bit s in bursts of twelve, MFMAs back to back,
    // the operand registers of an MFMA in flight rewritten right behind it - ran 47 cycles per MFMA and SIMD with no DMA, read or
    // barrier in the loop (tools/mfma_overlap.hip: 42.8 / 37.2 for one / two waves in isolation, 38.0 / 35.9 in this order).
    // The MFMAs of a bit s go (a0,b0) (a0,b1) (a1,b1) (a1,b0): one new operand each. The gaps behind them make a1(s), a0(s+1), b0(s+1),
    // b1(s+1) in the OTHER register set (s & 1), so no expansion writes what an MFMA in flight reads; the gaps behind step 0's last
    // MFMAs make step 1's first three operands.
    constexpr int DH = (DMAS + 1) / 2;   // DMAs that go out under step u = 0; the rest under u = 1
    constexpr int MN = 4 * RF * CF, EN = 4 * (RF + CF), EP = 3, SN = 4 * (RF + CF);   // per step: MFMAs, expansions (the first EP ahead of MFMA 0), shifts
    // dma: group gn's DMAs go out between the MFMAs (a template flag, not a branch: a branch between two MFMAs splits the block the order is pinned in)
    auto multiply = [&](auto dma, const u32x4 (&xr)[2][RF], const u32x4 (&wr)[2][CF], int gn) {
        constexpr bool DMA = decltype(dma)::value;
        i32x4 A[2][RF], B[2][CF];
        u32x4 xs[RF], ws[CF];
        // expansion e of step u, in the order the MFMAs need them: bit s = e / (RF + CF), then a0, b0, (b1,) a1
        auto expand = [&](int u, int e) {
            const int s = e / (RF + CF), o = e % (RF + CF);
            const uint32_t mask = s < 3 ? 0x11111111u << s : 0x11111111u;
            const bool is_a = o == 0 || o == RF + CF - 1;
            const int f = is_a ? (o == 0 ? 0 : 1) : o - 1;
            const u32x4 v = is_a ? (s < 3 ? xr[u][f] : xs[f]) : (s < 3 ? wr[u][f] : ws[f]);
            const i32x4 r = {static_cast<int>(v.x & mask), static_cast<int>(v.y & mask), static_cast<int>(v.z & mask), static_cast<int>(v.w & mask)};
            if (is_a) A[s & 1][f] = r; else B[s & 1][f] = r;
        };
        // shift z of step u (bit 3 of a nibble is E2M1's sign: it is multiplied as bit 0 of the word >> 3): xs0, ws0, (ws1,) xs1 - the order bit 3's expansions follow
        auto shift = [&](int u, int z) {
            const int o = z >> 2, el = z & 3;
            if (o == 0) xs[0][el] = xr[u][0][el] >> 3;
            else if (o == RF + CF - 1) xs[1][el] = xr[u][1][el] >> 3;
            else ws[o - 1][el] = wr[u][o - 1][el] >> 3;
        };
        auto skipped = [&](int u) {   // wave-uniform: an all-zero 64 rows x 256-bit step of X is skipped
            uint32_t any = 0u;
#pragma unroll
            for (int i = 0; i < RF; i++) any |= (xr[u][i].x | xr[u][i].y) | (xr[u][i].z | xr[u][i].w);
            return zero_skip && __ballot(any != 0u) == 0ull;
        };
        auto first = [&](int u) {
#pragma unroll
            for (int e = 0; e < EP; e++) expand(u, e);
        };
        auto dmas = [&](int u) {
            if (DMA) {
#pragma unroll
                for (int j = u * DH; j < (u ? DMAS : DH); j++) issue_one(gn, j);
            }
        };
        auto step = [&](int u) {   // step u's first EP operands are made
            __builtin_amdgcn_sched_barrier(0);
            int made = EP, shifted = 0;
#pragma unroll
            for (int n = 0; n < MN; n++) {
                const int s = n / (RF * CF), q = n % (RF * CF);
                const int i = q / CF, j = CF == 2 ? ((q == 1 || q == 2) ? 1 : 0) : 0;
                const int sc = s < 3 ? 128 - s : 128;   // E8M0: code 1 << s counts as 1
                const i32x8 a8 = __builtin_shufflevector(A[s & 1][i], A[s & 1][i], 0, 1, 2, 3, -1, -1, -1, -1);   // (an FP4 operand is the first 128 bits of the register tuple)
                const i32x8 b8 = __builtin_shufflevector(B[s & 1][j], B[s & 1][j], 0, 1, 2, 3, -1, -1, -1, -1);
                // cbsz = blgp = 4: E2M1 operands; lane (fl, hf) register r holds C[row 32 i + 8 (r >> 2) + 4 hf + (r & 3)][column 32 j + fl]
                acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, acc[i][j], 4, 4, 0, sc, 0, sc);
                // the gap behind MFMA n: the expansions MFMA n + 2 .. needs (CF = 2: one a gap; CF = 1: two, one, two, ..)
                const int want = CF == 2 ? EP + n + 1 : EP + (3 * (n + 1) + 1) / 2;
#pragma unroll
                for (int e = 0; e < 2; e++)
                    if (made < EN && made < want) expand(u, made++);
                const int want_z = CF == 2 ? (n >= 10 ? SN : (SN * (n + 1) + 10) / 11) : 3 * (n + 1);
#pragma unroll
                for (int z = 0; z < 3; z++)
                    if (shifted < SN && shifted < want_z) shift(u, shifted++);
                if (u == 0 && n >= MN - EP) expand(1, n - (MN - EP));   // step 1's first operands (bit 0: register set 0, which bit 3 does not use)
                const int blk = n / CF;   // block 0 .. 7 of this step; a DMA behind blocks 0, 2, 4, 6 (and 1, 3, .. if there are more)
                if (DMA && n % CF == CF - 1) {
                    if (blk % 2 == 0 && u * DH + blk / 2 < (u ? DMAS : DH)) issue_one(gn, u * DH + blk / 2);
                    if (blk % 2 == 1 && u * DH + 4 + blk / 2 < (u ? DMAS : DH)) issue_one(gn, u * DH + 4 + blk / 2);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        if (skipped(0)) {
            dmas(0);
            first(1);
        } else {
            first(0);
            step(0);
        }
        if (skipped(1)) dmas(1);
        else step(1);
    };
    using with_dma = std::integral_constant<bool, true>;
    using without_dma = std::integral_constant<bool, false>;
    // The fragment reads of group g are issued right behind its barrier and consumed an iteration later, under the MFMAs of group
    // g - 1: all eight waves of the CU pass the same barrier, so nothing else would cover a read's latency (first form: 39 us at
    // 32768 x 32768 x 64 where the DMAs alone take 24 and the MFMAs alone 24).
    u32x4 xa[2][RF], wa[2][CF], xb[2][RF], wb[2][CF];
    publish(0);
    fetch(0, xa, wa);
    issue(STAGES - 1);
    int g = 1;
    for (; g + STAGES < ng && !no_dma; g += 2) {   // both multiplies of a trip have a group to fetch
        publish(g);
        fetch(g, xb, wb);
        multiply(with_dma{}, xa, wa, g + STAGES - 1);
#ifdef QGTC_STAMPS
        asm volatile("" : "+v"(acc[0][0]));
        if (g == 5) ST_STAMP(6);
#endif
        publish(g + 1);
        fetch(g + 1, xa, wa);
        multiply(with_dma{}, xb, wb, g + STAGES);
#ifdef QGTC_STAMPS
        asm volatile("" : "+v"(acc[0][0]));
        if (g == 5) ST_STAMP(10);
#endif
    }
    for (; g + 1 < ng; g += 2) {   // the last trips: at most one group is still to be fetched (wave-uniform branches around whole multiplies)
        publish(g);
        fetch(g, xb, wb);
        if (g + STAGES - 1 < ng && !no_dma) multiply(with_dma{}, xa, wa, g + STAGES - 1);
        else multiply(without_dma{}, xa, wa, ng);
        publish(g + 1);
        fetch(g + 1, xa, wa);
        multiply(without_dma{}, xb, wb, ng);
    }
    if (g < ng) {
        publish(g);
        fetch(g, xb, wb);
        multiply(without_dma{}, xa, wa, ng);
        multiply(without_dma{}, xb, wb, ng);
    } else {
        multiply(without_dma{}, xa, wa, ng);
    }
#ifdef QGTC_STAMPS
    asm volatile("" : "+v"(acc[0][0]));
#endif
    ST_STAMP(11);
#ifdef QGTC_STAMPS
    st_[14] = __builtin_amdgcn_s_memrealtime();
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();   // (every wave has read its last fragments: the stages' place is free)

    // ---- the WK partial tiles of a row half meet through LDS: every wave leaves its accumulators there as they sit in its registers
    // (16 bytes a lane and register quad: lane-linear, no conflicts), then sums ITS share of the quads over the WK waves and puts the
    // sums - exact integers below 2^24 as float32, any order - into the int32 tile the epilogue reads. (First form: 64 ds_add_u32 a wave
    // into a zeroed tile - 5300 cycles by the stamps, 2.4 us behind every workgroup's last MFMA.)
    constexpr int NQ = RF * CF * 4, QW = NQ / WK;   // register quads of a wave's accumulators; quads a wave sums
    static_assert(NQ % WK == 0, "the quads split evenly over the waves along K");
    {
        f32x4 *part = reinterpret_cast<f32x4 *>(st_lds);   // [wave][quad][lane]
#pragma unroll
        for (int i = 0; i < RF; i++)
#pragma unroll
            for (int j = 0; j < CF; j++)
#pragma unroll
                for (int rq = 0; rq < 4; rq++)
                    part[(wv * NQ + (i * CF + j) * 4 + rq) * 64 + lane] = f32x4{acc[i][j][4 * rq], acc[i][j][4 * rq + 1], acc[i][j][4 * rq + 2], acc[i][j][4 * rq + 3]};
        __syncthreads();
        f32x4 sum[QW];
#pragma unroll
        for (int tq = 0; tq < QW; tq++) {
            sum[tq] = part[((wm * WK) * NQ + wk * QW + tq) * 64 + lane];
#pragma unroll
            for (int w2 = 1; w2 < WK; w2++) sum[tq] += part[((wm * WK + w2) * NQ + wk * QW + tq) * 64 + lane];
        }
        __syncthreads();   // (the tile overlaps the partial tiles)
#pragma unroll
        for (int tq = 0; tq < QW; tq++) {
            const int q = wk * QW + tq, f = q >> 2, rq = q & 3;   // fragment (f / CF, f % CF), rows 8 rq + 4 hf .. + 3 of it, column fl
            int *dst = &tile[64 * wm + 32 * (f / CF) + 8 * rq + 4 * hf][32 * (f % CF) + fl];
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
