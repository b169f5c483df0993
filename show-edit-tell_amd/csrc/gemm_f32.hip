// Grouped skinny fp32 GEMM for the decode step:  C[M,N] = A[M,K] * W[N,K]^T  (nn.Linear layout).
//
// Why this shape: every contraction of the step (gate products of the two LSTM cells, the
// attention projections, the context gate, the vocabulary projection) has M = the decode batch
// (<= a few hundred rows) against a weight matrix that is streamed once.  fp32 in / fp32
// accumulate is required by the 1e-4-logit / bit-exact-token target, so the math runs on
// v_mfma_f32_32x32x2_f32 (exact fp32 FMA chains, 157 TFLOP/s peak = 256 flop/clk/CU).
//
// Structure (one workgroup = 4 waves = one SIMD each):
//   * tile BM x BN x 32; global -> registers (float4, whole 128-B rows) -> LDS (unpadded 128-B rows with
//     XOR-swizzled 16-byte chunks: conflict-free ds_write_b128 and ds_read_b128), two LDS buffers, one
//     barrier per k-tile; the global loads run two k-tiles ahead of the MFMAs.
//   * K order inside an 8-wide block is permuted (lanes 0-31 take k0..3, lanes 32-63 k4..7) so one
//     ds_read_b128 per operand row feeds four MFMAs; A and W use the same permutation.
//   * several independent problems ride one launch (GemmLaunch holds up to 6 task descriptors)
//     and each may be split along K into slabs (deterministic: slab s is written by exactly one
//     workgroup; the consuming pointwise kernel adds the slabs in a fixed order).
//   * K may be a concatenation of up to three (activation, weight-column-block) segments, so
//     cat([emb, h2, h1]) x [W_ih[:, :D] | W_ih[:, 2D:3D] | W_hh] needs no copies of weights.
#include <cstdlib>
#include <cstring>
#include "set_common.h"
#include "split3.h"

namespace set {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
// loads of A / W always target global memory: say so, or the compiler emits flat_load (which also
// counts on lgkmcnt and would make every LDS wait drain the HBM loads)
typedef const f32x4 __attribute__((address_space(1)))* gptr4;

// LDS rows are 32 floats (128 B) UNPADDED; the 16-byte chunk c of row r is stored at chunk c ^ ((r >> 1) & 7).
// ds_write_b128 (8 lanes = one row) and the MFMA fragment ds_read_b128 (lane groups of 16 rows, one chunk
// each) are then both bank-conflict-free, and a 128x64x32 stage pair is 48 KB -> three workgroups per CU.
constexpr int LDS_STRIDE = 32;

#ifdef SET_EXPERIMENTAL_GEMMS
static int gemm_split_mode() { static int v = env_int("SET_GEMM_SPLIT", 0); return v; }
#else
static constexpr int gemm_split_mode() { return 0; }     // (the bf16x3 kernel is not in the shipped library)
#endif

struct GemmTask {
    const float* A[GEMM_MAX_SEG];
    const float* W[GEMM_MAX_SEG];
    long long lda[GEMM_MAX_SEG], ldw[GEMM_MAX_SEG];
    int kt_end[GEMM_MAX_SEG];     // cumulative k-tiles at the end of each segment
    float* C;
    long long ldc, slab_stride;
    const float* bias;
    int M, N, act, nseg, ktiles, ksplit, tiles_m, tiles_n, wg_begin, tm_stride;
    int vec_store;                // 16-byte epilogue stores allowed (N, ldc, slab stride multiples of 4 floats, C 16-byte aligned)
};
struct GemmLaunch {
    GemmTask t[GEMM_MAX_TASKS];
    int ntasks;
    RowGate gate;                    // set_common.h: loop-left test and the compacted row list of the decode loops
#ifdef SET_EXP_STAMPS
    unsigned long long* stamps;      // debug: 8 wall-clock stamps (10 ns units) per workgroup
#endif
};
#ifdef SET_EXP_STAMPS
unsigned long long* g_gemm_stamps = nullptr;
#define SET_STAMP(i) if (tid == 0) stamp_[i] = wall_clock64();
#else
#define SET_STAMP(i)
#endif

__device__ __forceinline__ float apply_act(float v, int act) {
    switch (act) {
        case SET_ACT_RELU: return v > 0.f ? v : 0.f;
        case SET_ACT_TANH: return tanhf(v);
        case SET_ACT_SIGMOID: return 1.f / (1.f + expf(-v));
        default: return v;
    }
}

// KG = 1: four waves, each owns a (BM/WAVES_M) x (BN/WAVES_N) piece of the tile over the whole k-tile.
// KG = 2: eight waves; waves 0-3 contract the first two 8-wide k-blocks of every k-tile, waves 4-7 the last two, and
//         the two partial tiles are added through LDS after the loop (group 0 + group 1, fixed order).  Same tile, same
//         LDS stages and slab plan as KG = 1, but six instead of three resident waves per SIMD at three workgroups per
//         CU and half as many MFMAs between two barriers, which fills more of the matrix pipe's idle slots.
// GATE (row gate of the decode loops, set_common.h): 0 = none (prologue, training, every other caller: no gate code at all),
// 1 = loop-left test.  Separate instantiations: the ungated kernel keeps the round-3 instruction stream (the gate's branch
// costs 0.4-0.5 us per launch when compiled in).
template <int BM, int BN, int WAVES_M, int WAVES_N, int KG = 1, int GATE = 0>
__global__ void __launch_bounds__(256 * KG) gemm_nt_f32(const int ntasks, const int wb1, const int wb2, const int wb3,
                                                        const int wb4, const int wb5, const int* const gate_alive,
        const int* const gate_nrows, const GemmLaunch L) {
    // ntasks / wb1..wb5 repeat L.ntasks and L.t[1..5].wg_begin as leading scalar arguments: this file is compiled with
    // -mllvm -amdgpu-kernarg-preload-count=6, so they arrive in SGPRs with the wave and the task lookup below needs no
    // memory round trip before the task's own fields can be requested (about 0.5 us of every workgroup's start-up)
    static_assert(WAVES_M * WAVES_N == 4, "4 waves per k-group");
    static_assert(KG == 1 || KG == 2, "one or two k-groups");
    constexpr int TM = BM / WAVES_M / 32, TN = BN / WAVES_N / 32;
    constexpr int RP = 32 * KG;                          // rows staged per pass of the whole workgroup
    constexpr int LA = BM / RP, LW = BN / RP;            // float4 loads per thread per k-tile
    static_assert(TM >= 1 && TN >= 1 && LA >= 1 && LW >= 1, "wave tile is a multiple of 32x32");
    __shared__ __attribute__((aligned(16))) float lds[2][(BM + BN) * LDS_STRIDE];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = (tid >> 6) & 3, kg = tid >> 8;
    const int wm = wave % WAVES_M, wn = wave / WAVES_M;
#ifdef SET_EXP_STAMPS
    unsigned long long stamp_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long cyc0_ = 0, cyc1_ = 0;
#endif
    SET_STAMP(0);

    // ---- which task / tile / k-slice is this workgroup
    static_assert(GEMM_MAX_TASKS == 6, "five leading wg_begin arguments");
    int ti = 0;
    {
        const int bid = (int)blockIdx.x;
        if (1 < ntasks && bid >= wb1) ti = 1;
        if (2 < ntasks && bid >= wb2) ti = 2;
        if (3 < ntasks && bid >= wb3) ti = 3;
        if (4 < ntasks && bid >= wb4) ti = 4;
        if (5 < ntasks && bid >= wb5) ti = 5;
    }
    const GemmTask& T = L.t[ti];
    // workgroup -> (row tile tm, column tile tn, k-slice ks).  Row tiles of the same (tn, ks) read the same
    // weight block; their block ids differ by tm_stride, a multiple of 8, so they land on the SAME XCD (blocks
    // go round-robin over the 8 XCDs) and the second reader finds the block in that XCD's L2.
    const int local = (int)blockIdx.x - T.wg_begin;
    const int tm = local / T.tm_stride;
    const int rem = local - tm * T.tm_stride;
    if (rem >= T.tiles_n * T.ksplit) return;          // padding slot
    const int ks = rem % T.ksplit;
    const int tn = rem / T.ksplit;
    const int m0 = tm * BM, n0 = tn * BN;
    const int kt0 = (int)(((long long)ks * T.ktiles) / T.ksplit);
    const int kt1 = (int)(((long long)(ks + 1) * T.ktiles) / T.ksplit);
    // ---- row gate of the decode loops (set_common.h): nothing to do once the reference has left its loop
    // (gate_alive repeats L.gate.alive_prev as a leading, SGPR-preloaded argument: its value is requested with the wave's
    // first instructions, next to the task descriptor, not behind it; gate_nrows is an unused slot of the preloaded block)
    const int epi_m = T.M;
    if constexpr (GATE >= 1) {
        if (gate_alive && *gate_alive == 0) return;
    }
    auto epi_row = [&](int row) { return row; };

    // ---- staging assignment: thread -> (row = tid/8 + RP*i, 16-byte column = tid%8)
    const int srow = tid >> 3, scol = (tid & 7) * 4;
    const int sswz = ((tid & 7) ^ ((srow >> 1) & 7)) * 4;          // swizzled chunk (rows srow+RP*i share (r>>1)&7)
    int arow[LA], wrow[LW];
#pragma unroll
    for (int i = 0; i < LA; ++i) { int r = m0 + srow + RP * i; r = r < epi_m ? r : epi_m - 1; arow[i] = r; }
#pragma unroll
    for (int i = 0; i < LW; ++i) { int r = n0 + srow + RP * i; wrow[i] = r < T.N ? r : T.N - 1; }
#ifdef SET_EXP_SAMEW
    for (int i = 0; i < LW; ++i) wrow[i] = srow + RP * i;      // diagnostic: every workgroup streams the same weight rows (L2 hits)
#endif

    f32x4 ra0[LA], rw0[LW], ra1[LA], rw1[LW];      // two register stages: loads run two k-tiles ahead
    // running per-thread row pointers inside the current K segment; re-derived only when the
    // k-tile index crosses into the next (activation, weight) segment
    const float* pa[LA];
    const float* pw[LW];
    int seg_end = 0;       // first k-tile index beyond the current segment
#define SET_SEEK(KT)                                                                                    \
    {                                                                                                   \
        const int kt_ = (KT);                                                                           \
        int s_ = 0, kbase_ = 0;                                                                         \
        _Pragma("unroll") for (int i = 0; i < GEMM_MAX_SEG - 1; ++i)                                    \
            if (i + 1 < T.nseg && kt_ >= T.kt_end[i]) { s_ = i + 1; kbase_ = T.kt_end[i]; }             \
        const float* Ab_ = T.A[0];                                                                      \
        const float* Wb_ = T.W[0];                                                                      \
        long long lda_ = T.lda[0], ldw_ = T.ldw[0];                                                     \
        seg_end = T.kt_end[0];                                                                          \
        _Pragma("unroll") for (int i = 1; i < GEMM_MAX_SEG; ++i)                                        \
            if (s_ == i) { Ab_ = T.A[i]; Wb_ = T.W[i]; lda_ = T.lda[i]; ldw_ = T.ldw[i]; seg_end = T.kt_end[i]; } \
        const long long koff_ = (long long)(kt_ - kbase_) * GEMM_BK + scol;                             \
        _Pragma("unroll") for (int i = 0; i < LA; ++i) pa[i] = Ab_ + arow[i] * lda_ + koff_;            \
        _Pragma("unroll") for (int i = 0; i < LW; ++i) pw[i] = Wb_ + wrow[i] * ldw_ + koff_;            \
    }
#define SET_GLOAD(RA, RW)                                                                               \
    {                                                                                                   \
        _Pragma("unroll") for (int i = 0; i < LA; ++i) { RA[i] = *(gptr4)(pa[i]); pa[i] += GEMM_BK; } \
        _Pragma("unroll") for (int i = 0; i < LW; ++i) { RW[i] = *(gptr4)(pw[i]); pw[i] += GEMM_BK; } \
    }
#define SET_LSTORE(BUF, RA, RW)                                                                         \
    {                                                                                                   \
        float* sA_ = lds[(BUF)];                                                                        \
        float* sW_ = lds[(BUF)] + BM * LDS_STRIDE;                                                      \
        _Pragma("unroll") for (int i = 0; i < LA; ++i)                                                  \
            *reinterpret_cast<f32x4*>(sA_ + (srow + RP * i) * LDS_STRIDE + sswz) = RA[i];              \
        _Pragma("unroll") for (int i = 0; i < LW; ++i)                                                  \
            *reinterpret_cast<f32x4*>(sW_ + (srow + RP * i) * LDS_STRIDE + sswz) = RW[i];              \
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int frow = lane & 31;
    constexpr int KB = 4 / KG;                                    // 8-wide k-blocks of a k-tile contracted by this wave
    int fo[KB];                                                   // swizzled float offset of k-chunk 2*kk + (lane>>5)
#pragma unroll
    for (int kk = 0; kk < KB; ++kk) fo[kk] = ((((kg * KB + kk) * 2 + (lane >> 5)) ^ ((frow >> 1) & 7)) * 4);
    // Software pipeline (one barrier per k-tile, two LDS buffers, one register stage):
    //   iteration kt:  MFMAs of tile kt from lds[buf]   ||  ds_write tile kt+1 -> lds[buf^1]
    //                                                    ||  global loads of tile kt+2 -> registers
    // so the LDS stores and the HBM/L2 loads are issued in the shadow of the 64-cycle MFMAs.
#define SET_FRAG_LOAD(KK, FA, FB)                                                                        \
    {                                                                                                   \
        _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                  \
            FA[i] = *reinterpret_cast<const f32x4*>(sA + i * 32 * LDS_STRIDE + fo[KK]);                 \
        _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                  \
            FB[j] = *reinterpret_cast<const f32x4*>(sW + j * 32 * LDS_STRIDE + fo[KK]);                 \
    }
#define SET_FRAG_MFMA(FA, FB)                                                                           \
    {                                                                                                   \
        _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                  \
            _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                            \
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(FA[i].x, FB[j].x, acc[i][j], 0, 0, 0); \
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(FA[i].y, FB[j].y, acc[i][j], 0, 0, 0); \
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(FA[i].z, FB[j].z, acc[i][j], 0, 0, 0); \
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(FA[i].w, FB[j].w, acc[i][j], 0, 0, 0); \
            }                                                                                           \
    }
    // stage(kt): global loads of tile kt into a register set (no-op past the end of the slice)
#ifdef SET_EXP_NOLOAD
#define SET_STAGE(KT, RA, RW) if ((KT) < kt0 + 3 && (KT) < kt1) { if ((KT) == seg_end) SET_SEEK(KT); SET_GLOAD(RA, RW); }
#else
#define SET_STAGE(KT, RA, RW)                                                                           \
    if ((KT) < kt1) {                                                                                   \
        if ((KT) == seg_end) SET_SEEK(KT);                                                              \
        SET_GLOAD(RA, RW);                                                                              \
    }
#endif
    // one k-tile: MFMAs from lds[BUF]; meanwhile registers (tile kt+1) -> lds[BUF^1], then reload them
    // with tile kt+3
#define SET_ITER(KT, BUF, RA, RW)                                                                       \
    {                                                                                                   \
        const float* sA = lds[BUF] + (wm * TM * 32 + frow) * LDS_STRIDE;                                \
        const float* sW = lds[BUF] + BM * LDS_STRIDE + (wn * TN * 32 + frow) * LDS_STRIDE;              \
        f32x4 fa0[TM], fb0[TN], fa1[TM], fb1[TN];     /* fragments of k-block kk+1 are fetched from LDS */ \
        SET_FRAG_LOAD(0, fa0, fb0);                   /* before the MFMAs of k-block kk are issued      */ \
        SET_FRAG_LOAD(1, fa1, fb1);                                                                     \
        SET_FRAG_MFMA(fa0, fb0);                                                                        \
        if ((KT) + 1 < kt1) {                                                                           \
            SET_LSTORE((BUF) ^ 1, RA, RW);                                                              \
            SET_STAGE((KT) + 3, RA, RW);                                                                \
        }                                                                                               \
        if constexpr (KG == 1) {                                                                        \
            SET_FRAG_LOAD(2, fa0, fb0);                                                                 \
            SET_FRAG_MFMA(fa1, fb1);                                                                    \
            SET_FRAG_LOAD(3, fa1, fb1);                                                                 \
            SET_FRAG_MFMA(fa0, fb0);                                                                    \
        }                                                                                               \
        SET_FRAG_MFMA(fa1, fb1);                                                                        \
        __syncthreads();                                                                                \
    }
    if (kt0 < kt1) {
        SET_SEEK(kt0);
        SET_STAMP(1);
        SET_GLOAD(ra0, rw0);                 // tile kt0
        SET_STAGE(kt0 + 1, ra1, rw1);        // tile kt0+1
        SET_LSTORE(0, ra0, rw0);
        SET_STAMP(2);
        SET_STAGE(kt0 + 2, ra0, rw0);        // tile kt0+2
        __syncthreads();
        SET_STAMP(3);
#ifdef SET_EXP_STAMPS
        if (tid == 0) cyc0_ = clock64();
#endif
    }
    // invariant at the top of an even step: lds[0] = tile kt, ra1/rw1 = tile kt+1, ra0/rw0 = tile kt+2
    for (int kt = kt0; kt < kt1; kt += 2) {
        SET_ITER(kt, 0, ra1, rw1);
#ifdef SET_EXP_STAMPS
        if (kt == kt0) { SET_STAMP(4); }
#endif
        if (kt + 1 < kt1) SET_ITER(kt + 1, 1, ra0, rw0);
    }
    SET_STAMP(5);
#ifdef SET_EXP_STAMPS
    if (tid == 0) cyc1_ = clock64();
#endif
#undef SET_ITER
#undef SET_STAGE
#undef SET_FRAG_LOAD
#undef SET_FRAG_MFMA

#include "gemm_f32_epilogue.inc"
#ifdef SET_EXP_STAMPS
    SET_STAMP(6);
    if (tid == 0 && L.stamps) {
        unsigned long long* o = L.stamps + (size_t)blockIdx.x * 8;
        for (int i = 0; i < 7; ++i) o[i] = stamp_[i];
        unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        unsigned hwid; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        o[7] = ((unsigned long long)xcc << 60) | ((cyc1_ - cyc0_) & 0xfffffffffffffffull);
        (void)hwid;
    }
#endif
}

#undef SET_GLOAD
#undef SET_SEEK
#undef SET_LSTORE

// ---------------------------------------------------------------------------------------------
// Round 4: gemm_nt_f32<64,64,2,2> with the pipelined k-loop written by hand (`gemm_nt_f32_asm`, SET_GEMM_ASM).
//
// Same tile, same LDS image (XOR-swizzled 128-byte rows), same K permutation, same MFMA order per accumulator — the
// results are BIT-IDENTICAL to the default kernel — and the same task descriptors, split-K slabs and epilogue.  What the
// hand-written loop controls and the compiler-scheduled one does not (EXPERIMENTS_r1-r4.md 4.2):
//   * the waits: a register stage is waited for with `s_waitcnt vmcnt(4)` — exactly "all but the four requests of the
//     younger stage" — so the prefetch distance is two k-tiles in EVERY round (the compiler's placement merges request
//     queues at the loop head and waits for vmcnt(0) in every second round: one k-tile);
//   * no loop-carried copies, no address arithmetic on the vector ALU beyond four 32-bit offset increments per k-tile
//     (operands are addressed as scalar base + 32-bit lane offset);
//   * requests past the end of the slice are not issued at all (scalar branches on the number of k-tiles left), so
//     nothing is in flight when the loop ends.
// The loop lives in ONE asm statement (prologue included): a compiler-visible loop around asm blocks would get loop-carried
// phi copies of registers whose loads are still in flight.  Registers v48-v95 are the loop's own (two register stages of
// four 16-byte pieces, four fragment quads); everything else is an operand.  Restriction (checked by the launcher, which
// otherwise takes the default kernel): every workgroup's K slice lies inside ONE (activation, weight) segment.
// ---------------------------------------------------------------------------------------------
#define ASM_HALF_ROUND(BUF, NBUF, SA0, SA1, SW0, SW1, LBL)                                                        \
    /* fragments of k-blocks 0 and 1 of the tile in buffer BUF */                                                 \
    "ds_read_b128 v[80:83], %[rdA0] offset:" BUF "\n"                                                             \
    "ds_read_b128 v[84:87], %[rdW0] offset:" BUF "\n"                                                             \
    "ds_read_b128 v[88:91], %[rdA1] offset:" BUF "\n"                                                             \
    "ds_read_b128 v[92:95], %[rdW1] offset:" BUF "\n"                                                             \
    "s_waitcnt lgkmcnt(2)\n"                                                                                      \
    "v_mfma_f32_32x32x2_f32 %[acc], v80, v84, %[acc]\n"                                                           \
    "v_mfma_f32_32x32x2_f32 %[acc], v81, v85, %[acc]\n"                                                           \
    "v_mfma_f32_32x32x2_f32 %[acc], v82, v86, %[acc]\n"                                                           \
    "v_mfma_f32_32x32x2_f32 %[acc], v83, v87, %[acc]\n"                                                           \
    /* tile kt+1 (if any): its register stage -> the other LDS buffer; the younger stage (tile kt+2) may still fly */ \
    "s_cmp_lt_i32 %[rem], 2\n"                                                                                    \
    "s_cbranch_scc1 " LBL "_nostore%=\n"                                                                          \
    "s_cmp_gt_i32 %[rem], 2\n"                                                                                    \
    "s_cbranch_scc1 " LBL "_w4%=\n"                                                                               \
    "s_waitcnt vmcnt(0)\n"                                                                                        \
    "s_branch " LBL "_wd%=\n"                                                                                     \
    LBL "_w4%=:\n"                                                                                                \
    "s_waitcnt vmcnt(4)\n"                                                                                        \
    LBL "_wd%=:\n"                                                                                                \
    "ds_write_b128 %[wr], " SA0 " offset:" NBUF "\n"                                                              \
    "ds_write_b128 %[wr], " SA1 " offset:" NBUF "+4096\n"                                                         \
    "ds_write_b128 %[wr], " SW0 " offset:" NBUF "+8192\n"                                                         \
    "ds_write_b128 %[wr], " SW1 " offset:" NBUF "+12288\n"                                                        \
    /* tile kt+3 (if any) into the stage that has just been stored */                                             \
    "s_cmp_lt_i32 %[rem], 4\n"                                                                                    \
    "s_cbranch_scc1 " LBL "_nostore%=\n"                                                                          \
    "global_load_dwordx4 " SA0 ", %[oA0], %[bA]\n"                                                                \
    "global_load_dwordx4 " SA1 ", %[oA1], %[bA]\n"                                                                \
    "global_load_dwordx4 " SW0 ", %[oW0], %[bW]\n"                                                                \
    "global_load_dwordx4 " SW1 ", %[oW1], %[bW]\n"                                                                \
    "v_add_u32 %[oA0], 128, %[oA0]\n"                                                                             \
    "v_add_u32 %[oA1], 128, %[oA1]\n"                                                                             \
    "v_add_u32 %[oW0], 128, %[oW0]\n"                                                                             \
    "v_add_u32 %[oW1], 128, %[oW1]\n"                                                                             \
    LBL "_nostore%=:\n"                                                                                           \
    "ds_read_b128 v[80:83], %[rdA2] offset:" BUF "\n"                                                             \
    "ds_read_b128 v[84:87], %[rdW2] offset:" BUF "\n"                                                             \
    "s_waitcnt lgkmcnt(2)\n"                         /* (LDS operations retire in order: all but the two newest) */ \
    "v_mfma_f32_32x32x2_f32 %[acc], v88, v92, %[acc]\n"                                                           \
    "v_mfma_f32_32x32x2_f32 %[acc], v89, v93, %[acc]\n"                                                           \
    "v_mfma_f32_32x32x2_f32 %[acc], v90, v94, %[acc]\n"                                                           \
    "v_mfma_f32_32x32x2_f32 %[acc], v91, v95, %[acc]\n"                                                           \
    "ds_read_b128 v[88:91], %[rdA3] offset:" BUF "\n"                                                             \
    "ds_read_b128 v[92:95], %[rdW3] offset:" BUF "\n"                                                             \
    "s_waitcnt lgkmcnt(2)\n"                                                                                      \
    "v_mfma_f32_32x32x2_f32 %[acc], v80, v84, %[acc]\n"                                                           \
    "v_mfma_f32_32x32x2_f32 %[acc], v81, v85, %[acc]\n"                                                           \
    "v_mfma_f32_32x32x2_f32 %[acc], v82, v86, %[acc]\n"                                                           \
    "v_mfma_f32_32x32x2_f32 %[acc], v83, v87, %[acc]\n"                                                           \
    "s_waitcnt lgkmcnt(0)\n"                                                                                      \
    "s_barrier\n"                                                                                                 \
    "v_mfma_f32_32x32x2_f32 %[acc], v88, v92, %[acc]\n"                                                           \
    "v_mfma_f32_32x32x2_f32 %[acc], v89, v93, %[acc]\n"                                                           \
    "v_mfma_f32_32x32x2_f32 %[acc], v90, v94, %[acc]\n"                                                           \
    "v_mfma_f32_32x32x2_f32 %[acc], v91, v95, %[acc]\n"

template <int GATE>
__global__ void __launch_bounds__(256) gemm_nt_f32_asm(const int ntasks, const int wb1, const int wb2, const int wb3,
                                                       const int wb4, const int wb5, const int* const gate_alive,
        const int* const gate_nrows, const GemmLaunch L) {
    constexpr int BM = 64, BN = 64, TM = 1, TN = 1, KG = 1;
    __shared__ __attribute__((aligned(16))) float lds[2][(BM + BN) * LDS_STRIDE];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6, kg = 0;
    const int wm = wave % 2, wn = wave / 2;
    int ti = 0;
    {
        const int bid = (int)blockIdx.x;
        if (1 < ntasks && bid >= wb1) ti = 1;
        if (2 < ntasks && bid >= wb2) ti = 2;
        if (3 < ntasks && bid >= wb3) ti = 3;
        if (4 < ntasks && bid >= wb4) ti = 4;
        if (5 < ntasks && bid >= wb5) ti = 5;
    }
    const GemmTask& T = L.t[ti];
    const int local = (int)blockIdx.x - T.wg_begin;
    const int tm = local / T.tm_stride;
    const int rem0 = local - tm * T.tm_stride;
    if (rem0 >= T.tiles_n * T.ksplit) return;          // padding slot
    if constexpr (GATE >= 1) {
        if (gate_alive && *gate_alive == 0) return;
    }
    (void)gate_nrows;
    const int ks = rem0 % T.ksplit;
    const int tn = rem0 / T.ksplit;
    const int m0 = tm * BM, n0 = tn * BN;
    const int kt0 = (int)(((long long)ks * T.ktiles) / T.ksplit);
    const int kt1 = (int)(((long long)(ks + 1) * T.ktiles) / T.ksplit);
    const int epi_m = T.M;
    auto epi_row = [](int row) { return row; };

    // the slice's segment (the launcher guarantees [kt0, kt1) lies inside one)
    int sg = 0, kbase = 0;
#pragma unroll
    for (int i = 0; i < GEMM_MAX_SEG - 1; ++i)
        if (i + 1 < T.nseg && kt0 >= T.kt_end[i]) { sg = i + 1; kbase = T.kt_end[i]; }
    const float* Ab = T.A[0];
    const float* Wb = T.W[0];
    long long lda = T.lda[0], ldw = T.ldw[0];
#pragma unroll
    for (int i = 1; i < GEMM_MAX_SEG; ++i)
        if (sg == i) { Ab = T.A[i]; Wb = T.W[i]; lda = T.lda[i]; ldw = T.ldw[i]; }
    Ab += (long long)(kt0 - kbase) * GEMM_BK;
    Wb += (long long)(kt0 - kbase) * GEMM_BK;

    // staging assignment (as gemm_nt_f32): thread -> rows tid/8 and tid/8 + 32, 16-byte column tid%8
    const int srow = tid >> 3, scol = (tid & 7) * 4;
    const int sswz = ((tid & 7) ^ ((srow >> 1) & 7)) * 4;
    unsigned oA0, oA1, oW0, oW1;                  // byte offsets of this thread's pieces from the scalar bases
    {
        int r0 = m0 + srow, r1 = m0 + srow + 32;
        r0 = r0 < T.M ? r0 : T.M - 1; r1 = r1 < T.M ? r1 : T.M - 1;
        oA0 = (unsigned)((r0 * lda + scol) * 4); oA1 = (unsigned)((r1 * lda + scol) * 4);
        int c0 = n0 + srow, c1 = n0 + srow + 32;
        c0 = c0 < T.N ? c0 : T.N - 1; c1 = c1 < T.N ? c1 : T.N - 1;
        oW0 = (unsigned)((c0 * ldw + scol) * 4); oW1 = (unsigned)((c1 * ldw + scol) * 4);
    }
    typedef __attribute__((address_space(3))) float* lds_fptr;
    const unsigned lbase = (unsigned)(unsigned long long)(lds_fptr)&lds[0][0];
    const unsigned wr = lbase + (unsigned)(srow * LDS_STRIDE + sswz) * 4u;
    const int frow = lane & 31;
    unsigned rdA[4], rdW[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        const unsigned fo = (unsigned)((((kk * 2 + (lane >> 5)) ^ ((frow >> 1) & 7)) * 4));
        rdA[kk] = lbase + ((unsigned)((wm * 32 + frow) * LDS_STRIDE) + fo) * 4u;
        rdW[kk] = lbase + ((unsigned)((BM + wn * 32 + frow) * LDS_STRIDE) + fo) * 4u;
    }
    f32x16 acc[1][1];
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[0][0][e] = 0.f;
    int rem = kt1 - kt0;                           // k-tiles left, the current one included
    if (rem > 0) {
        asm volatile(
            // ---- prologue: tile 0 -> stage 0 -> buffer 0; tile 1 -> stage 1; tile 2 -> stage 0
            "global_load_dwordx4 v[48:51], %[oA0], %[bA]\n"
            "global_load_dwordx4 v[52:55], %[oA1], %[bA]\n"
            "global_load_dwordx4 v[56:59], %[oW0], %[bW]\n"
            "global_load_dwordx4 v[60:63], %[oW1], %[bW]\n"
            "v_add_u32 %[oA0], 128, %[oA0]\n"
            "v_add_u32 %[oA1], 128, %[oA1]\n"
            "v_add_u32 %[oW0], 128, %[oW0]\n"
            "v_add_u32 %[oW1], 128, %[oW1]\n"
            "s_cmp_lt_i32 %[rem], 2\n"
            "s_cbranch_scc1 P_one%=\n"
            "global_load_dwordx4 v[64:67], %[oA0], %[bA]\n"
            "global_load_dwordx4 v[68:71], %[oA1], %[bA]\n"
            "global_load_dwordx4 v[72:75], %[oW0], %[bW]\n"
            "global_load_dwordx4 v[76:79], %[oW1], %[bW]\n"
            "v_add_u32 %[oA0], 128, %[oA0]\n"
            "v_add_u32 %[oA1], 128, %[oA1]\n"
            "v_add_u32 %[oW0], 128, %[oW0]\n"
            "v_add_u32 %[oW1], 128, %[oW1]\n"
            "s_waitcnt vmcnt(4)\n"
            "s_branch P_st%=\n"
            "P_one%=:\n"
            "s_waitcnt vmcnt(0)\n"
            "P_st%=:\n"
            "ds_write_b128 %[wr], v[48:51]\n"
            "ds_write_b128 %[wr], v[52:55] offset:4096\n"
            "ds_write_b128 %[wr], v[56:59] offset:8192\n"
            "ds_write_b128 %[wr], v[60:63] offset:12288\n"
            "s_cmp_lt_i32 %[rem], 3\n"
            "s_cbranch_scc1 P_go%=\n"
            "global_load_dwordx4 v[48:51], %[oA0], %[bA]\n"
            "global_load_dwordx4 v[52:55], %[oA1], %[bA]\n"
            "global_load_dwordx4 v[56:59], %[oW0], %[bW]\n"
            "global_load_dwordx4 v[60:63], %[oW1], %[bW]\n"
            "v_add_u32 %[oA0], 128, %[oA0]\n"
            "v_add_u32 %[oA1], 128, %[oA1]\n"
            "v_add_u32 %[oW0], 128, %[oW0]\n"
            "v_add_u32 %[oW1], 128, %[oW1]\n"
            "P_go%=:\n"
            "s_waitcnt lgkmcnt(0)\n"
            "s_barrier\n"
            // ---- the loop: an even round on buffer 0 (stores stage 1), an odd round on buffer 1 (stores stage 0)
            "L_top%=:\n"
            ASM_HALF_ROUND("0", "16384", "v[64:67]", "v[68:71]", "v[72:75]", "v[76:79]", "A")
            "s_sub_i32 %[rem], %[rem], 1\n"
            "s_cmp_eq_u32 %[rem], 0\n"
            "s_cbranch_scc1 L_end%=\n"
            ASM_HALF_ROUND("16384", "0", "v[48:51]", "v[52:55]", "v[56:59]", "v[60:63]", "B")
            "s_sub_i32 %[rem], %[rem], 1\n"
            "s_cmp_eq_u32 %[rem], 0\n"
            "s_cbranch_scc0 L_top%=\n"
            "L_end%=:\n"
            // the compiler does not know that this statement ends in MFMAs: a VALU read of the accumulator (the epilogue's
            // v_accvgpr_read) needs 18+ wait states behind a 16-pass MFMA, which nothing else would insert
            "s_nop 15\n"
            "s_nop 7\n"
            : [acc] "+a"(acc[0][0]), [oA0] "+v"(oA0), [oA1] "+v"(oA1), [oW0] "+v"(oW0), [oW1] "+v"(oW1), [rem] "+s"(rem)
            : [rdA0] "v"(rdA[0]), [rdA1] "v"(rdA[1]), [rdA2] "v"(rdA[2]), [rdA3] "v"(rdA[3]),
              [rdW0] "v"(rdW[0]), [rdW1] "v"(rdW[1]), [rdW2] "v"(rdW[2]), [rdW3] "v"(rdW[3]),
              [wr] "v"(wr), [bA] "s"(Ab), [bW] "s"(Wb)
            : "memory", "scc",
              "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63",
              "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79",
              "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95");
    }
#include "gemm_f32_epilogue.inc"
}
#undef ASM_HALF_ROUND

// ---------------------------------------------------------------------------------------------
// Variants that were built, measured and LOST (EXPERIMENTS_r1-r4.md 3.1b, 4.1): the weights-to-registers kernel, the LDS-DMA
// kernel and the bf16x3 split-precision kernel live in experimental/gemm_variants.inc and are compiled only with
// -DSET_EXPERIMENTAL_GEMMS (tools/ubench builds; SET_HIPCC_FLAGS=-DSET_EXPERIMENTAL_GEMMS python -m show_edit_tell_amd.build --force).
// The shipped library does not contain them; their environment switches are ignored there.
// ---------------------------------------------------------------------------------------------
#ifdef SET_EXPERIMENTAL_GEMMS
#include "experimental/gemm_variants.inc"
#endif

// ---------------------------------------------------------------------------------------------
// <= 16 rows (the decode batch of BASELINE.json configs[0], beam search, the tail of a ragged teacher-forced batch): the
// launch is pure weight streaming (0.5 flop per weight byte at M = 4) and the 32x128 LDS-staged tile above reaches ~3 TB/s
// of it.  Here nothing goes through LDS: a wave owns 16 weight rows (= 16 output columns) over the workgroup's K slice and
// streams them straight into registers — lane (n = l & 15, g = l >> 4) fetches W[n][k + 4g .. +3] as ONE 16-byte load that
// feeds four v_mfma_f32_16x16x4_f32 (K inside a 16-block is permuted identically for the activations), eight such loads
// in flight per lane before the first MFMA — against the <= 16 activation rows (rows >= M repeat row M - 1 and are never
// stored), which come from L1/L2.  Same task descriptors, K segments, split-K slabs and fused bias / activation as
// gemm_nt_f32; workgroup = 4 waves = 64 columns of one task x one K slice, no barrier anywhere.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) gemv_nt_f32(const int ntasks, const int wb1, const int wb2, const int wb3,
                                                   const int wb4, const int wb5, const int* const gate_alive,
        const int* const gate_nrows, const GemmLaunch L) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, g = lane >> 4;
    int ti = 0;
    {
        const int bid = (int)blockIdx.x;
        if (1 < ntasks && bid >= wb1) ti = 1;
        if (2 < ntasks && bid >= wb2) ti = 2;
        if (3 < ntasks && bid >= wb3) ti = 3;
        if (4 < ntasks && bid >= wb4) ti = 4;
        if (5 < ntasks && bid >= wb5) ti = 5;
    }
    const GemmTask& T = L.t[ti];
    const int local = (int)blockIdx.x - T.wg_begin;
    if (local >= T.tiles_n * T.ksplit) return;
    if (gate_alive && *gate_alive == 0) return;                    // the decode loop has been left (set_common.h RowGate)
    const int ks = local % T.ksplit;
    const int tn = local / T.ksplit;
    const int n0 = tn * 64 + wave * 16;
    if (n0 >= T.N) return;
    const int kt0 = (int)(((long long)ks * T.ktiles) / T.ksplit);
    const int kt1 = (int)(((long long)(ks + 1) * T.ktiles) / T.ksplit);
    const int wrow = n0 + r < T.N ? n0 + r : T.N - 1;
    const int arow = r < T.M ? r : T.M - 1;

    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#define GV_MFMA(A4, W4)                                                                  \
    {                                                                                    \
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32((A4).x, (W4).x, acc, 0, 0, 0);        \
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32((A4).y, (W4).y, acc, 0, 0, 0);        \
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32((A4).z, (W4).z, acc, 0, 0, 0);        \
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32((A4).w, (W4).w, acc, 0, 0, 0);        \
    }
    int sb = 0;
#pragma unroll
    for (int s = 0; s < GEMM_MAX_SEG; ++s) {
        if (s >= T.nseg) break;
        const int se = T.kt_end[s];
        const int a = kt0 > sb ? kt0 : sb, b = kt1 < se ? kt1 : se;
        if (a < b) {
            const float* pa = T.A[s] + (long long)arow * T.lda[s] + (long long)(a - sb) * GEMM_BK + 4 * g;
            const float* pw = T.W[s] + (long long)wrow * T.ldw[s] + (long long)(a - sb) * GEMM_BK + 4 * g;
            int kt = a;
            for (; kt + 4 <= b; kt += 4) {              // 4 k-tiles = eight 16-byte weight loads in flight per lane
                f32x4 w[8], x[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) w[i] = *(gptr4)(pw + 16 * i);
#pragma unroll
                for (int i = 0; i < 8; ++i) x[i] = *(gptr4)(pa + 16 * i);
                __builtin_amdgcn_sched_barrier(0);      // all sixteen requests out before the first MFMA waits for one
#pragma unroll
                for (int i = 0; i < 8; ++i) GV_MFMA(x[i], w[i]);
                __builtin_amdgcn_sched_barrier(0);
                pa += 4 * GEMM_BK; pw += 4 * GEMM_BK;
            }
            for (; kt < b; ++kt) {
                const f32x4 w0 = *(gptr4)(pw), w1 = *(gptr4)(pw + 16), x0 = *(gptr4)(pa), x1 = *(gptr4)(pa + 16);
                GV_MFMA(x0, w0);
                GV_MFMA(x1, w1);
                pa += GEMM_BK; pw += GEMM_BK;
            }
        }
        sb = se;
    }
#undef GV_MFMA
    // C/D map of the 16x16 MFMA: col = lane & 15, row = 4 (lane >> 4) + reg
    float* Cs = T.C + (long long)ks * T.slab_stride;
    const int col = n0 + r;
    if (col < T.N) {
        const bool fused = (T.ksplit == 1);
        const float bv = (fused && T.bias) ? T.bias[col] : 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int row = 4 * g + e;
            if (row < T.M) {
                float v = acc[e];
                if (fused) v = apply_act(v + bv, T.act);
                Cs[(long long)row * T.ldc + col] = v;
            }
        }
    }
}

int gemm_tile_m(int M) {
    // 64x64 tiles up to M = 512 (measured: +4-5 % at M = 128, +10 % at M = 256-384; big-M prologue / training products
    // stay on 128x64): at the decode batch two 64-row tiles per weight block (the second one hits the
    // same XCD's L2) halve the split-K factor -> half the slab bytes written here and re-read by the consumer,
    // and 32 KB workgroups pack three per CU.  Measured +4-5 % on the bench against the 128x64 tile.
    static const int bm64_upto = env_int("SET_GEMM_BM64_UPTO", gemm_split_mode() ? 64 : 512);   // the split kernel is 128x64 only
    static const int bm32_upto = env_int("SET_GEMM_BM32_UPTO", 32);
    // <= 16 rows (configs[0]'s batch of 4, beam search, the tail of a ragged teacher-forced batch): the launch only streams
    // weights; gemv_nt_f32 (no LDS, one 16x16 MFMA tile per wave, loads straight into registers) instead of a 32-row tile
    static const int bm16_upto = env_int("SET_GEMM_BM16_UPTO", 16);
    return M <= bm16_upto ? 16 : (M <= bm32_upto ? 32 : (M <= bm64_upto ? 64 : 128));
}
#ifdef SET_EXPERIMENTAL_GEMMS
static int gemm_dma() { static int v = env_int("SET_GEMM_DMA", 0); return v; }
static int gemm_kgroups() { static int v = env_int("SET_GEMM_KGROUPS", 1); return v; }
static int gemm_bn128() { static int v = env_int("SET_GEMM_BN128", 0); return v; }
#else
static constexpr int gemm_dma() { return 0; }
static constexpr int gemm_kgroups() { return 1; }
static constexpr int gemm_bn128() { return 0; }
#endif
int g_gemm_asm_force = -1;         // tools/ubench: switch kernels inside one process (-1: the environment decides)
static int gemm_asm() { static int v = env_int("SET_GEMM_ASM", 1); return g_gemm_asm_force >= 0 ? g_gemm_asm_force : v; }
int g_gemm_wreg_force = -1;        // tools/ubench: switch kernels inside one process (-1: the environment decides)
#ifdef SET_EXPERIMENTAL_GEMMS
static int gemm_wreg() { static int v = env_int("SET_GEMM_WREG", 0); return g_gemm_wreg_force >= 0 ? g_gemm_wreg_force : v; }
#else
static constexpr int gemm_wreg() { return 0; }
#endif
static int tile_m_of(const GemmProb& p) { return (p.bm_hint == 64 || p.bm_hint == 128) ? p.bm_hint : gemm_tile_m(p.M); }
// Row-tile class of one launch.  Up to 512 rows: from M (above).  Beyond: 128x64 tiles unless 64x64 tiles leave the CUs a
// more even load — a launch takes about ceil(workgroups / 256 CUs) rounds of one tile's k-loop, and a 64-row tile's k-loop
// is half as long: att_embed (4608 x 1024: 576 tiles = 3 rounds, as 1152 half-size tiles 5) 203 -> 170 us, features_att
// (288 tiles = 2 rounds, as 576: 3 half rounds) 75 -> 56 us; the large products of the training step (fc over all
// timesteps, the all-timestep region projection: >= 11 rounds either way) stay on 128x64.
// SET_GEMM_BN32=1 (experiment, round 3): decode batches of 65..128 rows on 128x32 tiles — ONE workgroup owns both 64-row
// halves of a weight block (same workgroup count as two 64x64 tiles, each weight byte fetched once instead of relying on
// the second row tile's L2 hit; the activation rows are re-read by twice as many workgroups, from L2)
static bool use_bn32(const GemmProb* probs, int n) {
#ifdef SET_EXPERIMENTAL_GEMMS
    static const int on = env_int("SET_GEMM_BN32", 0);
#else
    constexpr int on = 0;
#endif
    if (!on || n <= 0 || probs[0].bm_hint || gemm_split_mode()) return false;
    for (int i = 0; i < n; ++i)
        if (probs[i].M <= 64 || probs[i].M > 128 || gemm_tile_m(probs[i].M) != 64) return false;
    return true;
}
static int launch_tile_m(const GemmProb* probs, int n) {
    if (use_bn32(probs, n)) return 128;
    // one row-tile class per launch: the largest any problem asks for (the teacher-forced loop merges fc over this step's
    // rows with phase A over the next step's, and the sorted batch may shrink across a class boundary in between)
    int bm = tile_m_of(probs[0]);
    if (!probs[0].bm_hint)
        for (int i = 1; i < n; ++i) { const int c = tile_m_of(probs[i]); if (c > bm) bm = c; }
    static const int model = env_int("SET_GEMM_TILE_MODEL", 1);
    if (!model || probs[0].bm_hint || bm != 128 || gemm_split_mode() || gemm_bn128()) return bm;
    long long t128 = 0, t64 = 0;
    for (int i = 0; i < n; ++i) {
        if (gemm_tile_m(probs[i].M) != 128) return bm;
        t128 += (long long)cdiv(probs[i].M, 128) * cdiv(probs[i].N, 64);
        t64 += (long long)cdiv(probs[i].M, 64) * cdiv(probs[i].N, 64);
    }
    const double c128 = (double)((t128 + 255) / 256) * 1.08, c64 = (double)((t64 + 255) / 256) * 0.57;
    return c64 < 0.97 * c128 ? 64 : 128;
}
static int launch_tile_n(const GemmProb* probs, int n) {
    if (use_bn32(probs, n)) return 32;
    const int bm = launch_tile_m(probs, n);
    return (bm == 32 || (bm == 128 && gemm_bn128())) ? 128 : 64;      // (class 16: 4 waves x 16 columns)
}

// Split-K plan for one grouped launch: every workgroup should run about the same number of k-tiles
// (`kper`) and the whole launch should fit the chip in ONE round: 256 CUs x 2 resident workgroups =
// `cap_wgs` slots.  A second, partially filled round costs a full workgroup latency, so kper is the
// smallest value (>= 4 k-tiles, to amortise the per-workgroup prologue/epilogue) for which the
// launch has at most cap_wgs workgroups.
void plan_ksplit(GemmProb* probs, int n, int cap_wgs) {
    // the cap is given for 128-row tiles (2 workgroups per CU); 64x64 workgroups are half as large: 3 per CU
    static const int pct64 = env_int("SET_GEMM_WGS64_PCT", 150);
    static const int pct32 = env_int("SET_GEMM_WGS32_PCT", 50);
    const int bm_l = n > 0 ? launch_tile_m(probs, n) : 128, bn_l = n > 0 ? launch_tile_n(probs, n) : 64;
    if (n > 0 && (bm_l == 64 || bn_l == 32)) cap_wgs = cap_wgs * pct64 / 100;
    // <= 32 rows: the launch only streams weights; fewer, longer workgroups halve the slab traffic (measured +5 %)
    if (n > 0 && bm_l == 32) cap_wgs = cap_wgs * pct32 / 100;
    // <= 16 rows: 64-column workgroups without LDS, many fit a CU; slabs are a few KB, so split generously for bytes in flight
    static const int pct16 = env_int("SET_GEMM_WGS16_PCT", 200);
    if (n > 0 && bm_l == 16) cap_wgs = cap_wgs * pct16 / 100;
    int tiles[GEMM_MAX_TASKS], kts[GEMM_MAX_TASKS], max_kt = 1;
    for (int i = 0; i < n; ++i) {
        const int bm = bm_l, bn = bn_l;
        tiles[i] = cdiv(probs[i].M, bm) * cdiv(probs[i].N, bn);
        kts[i] = probs[i].ktiles();
        if (kts[i] > max_kt) max_kt = kts[i];
    }
    auto split_of = [&](int i, int kper) {
        int ks = cdiv(kts[i], kper);
        if (ks > probs[i].max_ksplit) ks = probs[i].max_ksplit;
        if (ks > kts[i]) ks = kts[i];
        return ks < 1 ? 1 : ks;
    };
    static const int min_kper = env_int("SET_GEMM_MIN_KPER", 4);
    int kper = min_kper;
    for (; kper < max_kt; ++kper) {
        long long wgs = 0;
        for (int i = 0; i < n; ++i) wgs += (long long)tiles[i] * split_of(i, kper);
        if (wgs <= cap_wgs) break;
    }
    long long total = 0;
    for (int i = 0; i < n; ++i) { probs[i].ksplit = split_of(i, kper); total += (long long)tiles[i] * probs[i].ksplit; }
    // A common kper rarely lands on the slot count (e.g. the phase-B group: 224 tiles x 3 slices = 672 of 768 slots,
    // so a third of the CUs run 2 workgroups and the rest 3).  Fill the spare slots: give one more K-slice to the
    // problems that fit, smallest first (their workgroups get shorter, nobody's gets longer).
    static const int fill = env_int("SET_GEMM_FILL_SLOTS", 1);
    if (fill && n > 1) {
        bool grew = true;
        while (grew) {
            grew = false;
            int best = -1;
            for (int i = 0; i < n; ++i) {
                const int ks = probs[i].ksplit;
                if (ks >= probs[i].max_ksplit || ks >= kts[i] || kts[i] / (ks + 1) < min_kper) continue;
                if (total + tiles[i] > cap_wgs) continue;
                if (best < 0 || tiles[i] < tiles[best]) best = i;
            }
            if (best >= 0) { probs[best].ksplit += 1; total += tiles[best]; grew = true; }
        }
    }
}

// every K slice of every task lies inside one (activation, weight) segment (what gemm_nt_f32_asm needs)
static bool slices_in_one_segment(const GemmLaunch& L) {
    for (int i = 0; i < L.ntasks; ++i) {
        const GemmTask& t = L.t[i];
        // (the hand-written loop addresses its operands as scalar base + 32-bit byte offset of the lane's row)
        for (int sgi = 0; sgi < t.nseg; ++sgi)
            if ((unsigned long long)t.M * (unsigned long long)t.lda[sgi] * 4ull >= (1ull << 32) ||
                (unsigned long long)t.N * (unsigned long long)t.ldw[sgi] * 4ull >= (1ull << 32)) return false;
        for (int ks = 0; ks < t.ksplit; ++ks) {
            const int a = (int)(((long long)ks * t.ktiles) / t.ksplit), b = (int)(((long long)(ks + 1) * t.ktiles) / t.ksplit);
            for (int sgi = 0; sgi + 1 < t.nseg; ++sgi)
                if (a < t.kt_end[sgi] && b > t.kt_end[sgi]) return false;
        }
    }
    return true;
}

int gemm_group(const GemmProb* probs, int n, hipStream_t stream, const char* tag) {
    if (n <= 0) return SET_OK;
    if (n > GEMM_MAX_TASKS) return SET_ERR_ARG;
    GemmLaunch L;
    L.ntasks = n;
    L.gate = g_row_gate;
#ifdef SET_EXP_STAMPS
    L.stamps = g_gemm_stamps;
#endif
    int wg = 0;
    const int bm = launch_tile_m(probs, n), bn = launch_tile_n(probs, n);
    for (int i = 0; i < n; ++i) {
        const GemmProb& p = probs[i];
        GemmTask& t = L.t[i];
        if (p.M <= 0 || p.N <= 0 || p.nseg <= 0 || p.nseg > GEMM_MAX_SEG || !p.C) return SET_ERR_ARG;
        if (bm == 16 && p.M > 16) return SET_ERR_ARG;
        int kt = 0;
        for (int s = 0; s < GEMM_MAX_SEG; ++s) {
            if (s < p.nseg) {
                const GemmSeg& g = p.seg[s];
                if (!g.A || !g.W || g.K <= 0) return SET_ERR_ARG;
                if (g.K % GEMM_BK) return SET_ERR_UNSUPPORTED;
                if (!aligned16(g.A) || !aligned16(g.W) || (g.lda & 3) || (g.ldw & 3)) return SET_ERR_ARG;
                kt += g.K / GEMM_BK;
                t.A[s] = g.A; t.W[s] = g.W; t.lda[s] = g.lda; t.ldw[s] = g.ldw;
            } else {
                t.A[s] = nullptr; t.W[s] = nullptr; t.lda[s] = 0; t.ldw[s] = 0;
            }
            t.kt_end[s] = kt;
        }
        t.C = p.C; t.ldc = p.ldc; t.slab_stride = p.slab_stride; t.bias = p.bias;
        t.M = p.M; t.N = p.N; t.act = p.act; t.nseg = p.nseg; t.ktiles = kt;
        t.ksplit = p.ksplit < 1 ? 1 : p.ksplit;
        if (t.ksplit > kt) t.ksplit = kt;
        if (t.ksplit > 1 && (p.act != SET_ACT_NONE)) return SET_ERR_ARG;
        t.tiles_m = cdiv(p.M, bm); t.tiles_n = cdiv(p.N, bn);
        static const int vec_epi = env_int("SET_GEMM_VEC_EPILOGUE", 1);
        t.vec_store = vec_epi && !(p.N & 3) && !(p.ldc & 3) && !(p.slab_stride & 3) && aligned16(p.C);
        t.wg_begin = wg;
        static const int xcd_align = env_int("SET_GEMM_XCD_ALIGN", 1);
        t.tm_stride = (t.tiles_m > 1 && xcd_align) ? (int)round_up((size_t)t.tiles_n * t.ksplit, 8) : t.tiles_n * t.ksplit;
        wg += t.tiles_m * t.tm_stride;
    }
    for (int i = n; i < GEMM_MAX_TASKS; ++i) { L.t[i] = L.t[0]; L.t[i].wg_begin = 0x7fffffff; }
    double flops = 0.0, bytes = 0.0;
    for (int i = 0; i < n; ++i) {
        const GemmTask& t = L.t[i];
        const double K = (double)t.ktiles * GEMM_BK;
        flops += 2.0 * t.M * t.N * K;
        bytes += 4.0 * ((double)t.M * K + (double)t.N * K + (double)t.M * t.N * t.ksplit);
    }
    const char* kname = (bm == 64 && bn == 64 && gemm_asm() && slices_in_one_segment(L)) ? "gemm_nt_f32_asm<64,64>" :
                        (bm == 128 && bn == 32) ? "gemm_nt_f32<128,32>" : bm == 128 ? "gemm_nt_f32<128,64>" : (bm == 64 ? "gemm_nt_f32<64,64>" : (bm == 32 ? "gemm_nt_f32<32,128>" : "gemv_nt_f32<16,64>"));
    ProfScope ps(kname, stream, flops, bytes);
    static const bool sites = env_int("SET_PROFILE_SITES", 0) != 0;   // per-call-site breakdown (nested events)
    ProfScope ps2(sites ? (tag ? tag : "gemm:other") : nullptr, stream, flops, bytes);
    dim3 grid(wg), block(256);
    static const char* split_tag = getenv("SET_GEMM_SPLIT_TAG");     // debug: restrict the split kernel to one call site
    const bool split_here = gemm_split_mode() && (!split_tag || !*split_tag || strstr(tag ? tag : "untagged", split_tag));
#ifdef SET_EXPERIMENTAL_GEMMS
    if (bm == 128 && bn == 64 && split_here) {
        constexpr int kLds = 2 * 3 * (128 + 64) * 64;
        static bool attr_set = false;
        if (!attr_set) {
            SET_HIP_TRY(hipFuncSetAttribute((const void*)gemm_nt_split_bf16, hipFuncAttributeMaxDynamicSharedMemorySize,
                                            kLds));
            attr_set = true;
        }
        hipLaunchKernelGGL(gemm_nt_split_bf16, grid, block, kLds, stream, L);
    } else
#endif
    {
        (void)split_here;
        const int nt = L.ntasks, w1 = L.t[1].wg_begin, w2 = L.t[2].wg_begin, w3 = L.t[3].wg_begin, w4 = L.t[4].wg_begin,
                  w5 = L.t[5].wg_begin;
        const int* ga = L.gate.alive_prev;
        const int* gn = nullptr;                 // (unused slot of the preloaded argument block)
        const int gate_mode = ga ? 1 : 0;
        if (bm == 16)
            hipLaunchKernelGGL(gemv_nt_f32, grid, block, 0, stream, nt, w1, w2, w3, w4, w5, ga, gn, L);
#ifdef SET_EXPERIMENTAL_GEMMS
        else if (bm == 128 && bn == 32)
            hipLaunchKernelGGL((gemm_nt_f32<128, 32, 4, 1>), grid, block, 0, stream, nt, w1, w2, w3, w4, w5, ga, gn, L);
        else if (bm == 128 && bn == 128)
            hipLaunchKernelGGL((gemm_nt_f32<128, 128, 2, 2>), grid, block, 0, stream, nt, w1, w2, w3, w4, w5, ga, gn, L);
        else if (bm == 128 && gemm_dma())
            hipLaunchKernelGGL((gemm_nt_f32_dma<128, 64, 2, 2>), grid, block, 0, stream, nt, w1, w2, w3, w4, w5, ga, gn, L);
#endif
        else if (bm == 128 && gate_mode)
            hipLaunchKernelGGL((gemm_nt_f32<128, 64, 2, 2, 1, 1>), grid, block, 0, stream, nt, w1, w2, w3, w4, w5, ga, gn, L);
        else if (bm == 128)
            hipLaunchKernelGGL((gemm_nt_f32<128, 64, 2, 2>), grid, block, 0, stream, nt, w1, w2, w3, w4, w5, ga, gn, L);
#ifdef SET_EXPERIMENTAL_GEMMS
        else if (bm == 64 && gemm_wreg())
            hipLaunchKernelGGL(gemm_nt_f32_wreg, grid, block, 0, stream, nt, w1, w2, w3, w4, w5, ga, gn, L);
        else if (bm == 64 && gemm_kgroups() == 2)
            hipLaunchKernelGGL((gemm_nt_f32<64, 64, 2, 2, 2>), grid, dim3(512), 0, stream, nt, w1, w2, w3, w4, w5, ga, gn, L);
        else if (bm == 64 && gemm_dma())
            hipLaunchKernelGGL((gemm_nt_f32_dma<64, 64, 2, 2>), grid, block, 0, stream, nt, w1, w2, w3, w4, w5, ga, gn, L);
#endif
        else if (bm == 64 && bn == 64 && gemm_asm() && slices_in_one_segment(L))
            if (gate_mode) hipLaunchKernelGGL((gemm_nt_f32_asm<1>), grid, block, 0, stream, nt, w1, w2, w3, w4, w5, ga, gn, L);
            else hipLaunchKernelGGL((gemm_nt_f32_asm<0>), grid, block, 0, stream, nt, w1, w2, w3, w4, w5, ga, gn, L);
        else if (bm == 64 && gate_mode == 1)
            hipLaunchKernelGGL((gemm_nt_f32<64, 64, 2, 2, 1, 1>), grid, block, 0, stream, nt, w1, w2, w3, w4, w5, ga, gn, L);
        else if (bm == 64)
            hipLaunchKernelGGL((gemm_nt_f32<64, 64, 2, 2>), grid, block, 0, stream, nt, w1, w2, w3, w4, w5, ga, gn, L);
        else if (gate_mode)
            hipLaunchKernelGGL((gemm_nt_f32<32, 128, 1, 4, 1, 1>), grid, block, 0, stream, nt, w1, w2, w3, w4, w5, ga, gn, L);
        else
            hipLaunchKernelGGL((gemm_nt_f32<32, 128, 1, 4>), grid, block, 0, stream, nt, w1, w2, w3, w4, w5, ga, gn, L);
    }
    SET_LAUNCH_CHECK();
    return SET_OK;
}

}  // namespace set
