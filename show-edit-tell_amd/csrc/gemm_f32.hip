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

static int gemm_split_mode() { static int v = env_int("SET_GEMM_SPLIT", 0); return v; }

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
// hand-written loop controls and the compiler-scheduled one does not (EXPERIMENTS.md 4.2):
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
// Round 4: the 64-row class with the WEIGHT operand taken off LDS (`gemm_nt_f32_wreg`, SET_GEMM_WREG).
//
// gemm_nt_f32<64,64,2,2> moves 3 LDS "operand words" per MFMA (each wave reads one A and one W value per lane and MFMA
// from LDS, and the workgroup writes 16 KB per 64 MFMAs into it) and is bound by what the chip sustains for that mix of
// MFMA + LDS + L2 traffic (clock 2.0 GHz under this load against 2.4 for bare MFMAs).  Here
//   * the four waves are 2 (column halves of the 64x64 tile) x 2 (K groups): a wave owns a 64 x 32 piece of the output —
//     TWO 32x32 accumulators that share every weight fragment — over every other k-tile of the workgroup's K slice;
//   * the weight fragments never see LDS: lane (j = l & 31, h = l >> 5) loads W[n0 + j][8 kk + 4 h .. +3] of its wave's 32
//     columns straight from global memory as 16-byte pieces — already the B operand of four v_mfma_f32_32x32x2_f32 (K inside
//     an 8-block permuted exactly as the LDS fragment reads of the activations permute it) — two k-tiles ahead, each piece
//     re-requested as soon as its MFMAs have issued;
//   * only the 64 x 32 activation tile of each K group goes through LDS (shared by the group's two waves; same XOR-swizzled
//     image, same conflict-free ds_write_b128 / ds_read_b128 as above), two register stages + two LDS buffers;
//   * per MFMA: 1 LDS operand word read (was 2), 0.5 written (was 1), one barrier per 32 MFMAs of a wave (was 16);
//   * after the loop the two K groups exchange one 32x32 accumulator each through LDS (group 0 keeps rows 0-31, group 1
//     rows 32-63; sum order group 0 + group 1, fixed), so all four waves store, through the same epilogue.
// Same task descriptors, K segments, split-K slabs, bias / activation epilogue as gemm_nt_f32.
// ---------------------------------------------------------------------------------------------
#ifdef SET_WREG_A2
#define WR_WAVES_PER_SIMD 2
#else
#define WR_WAVES_PER_SIMD 3
#endif
__global__ void __launch_bounds__(256, WR_WAVES_PER_SIMD) gemm_nt_f32_wreg(const int ntasks, const int wb1, const int wb2, const int wb3,
                                                        const int wb4, const int wb5, const int* const gate_alive,
        const int* const gate_nrows, const GemmLaunch L) {
    constexpr int BM = 64, BN = 64;
    constexpr int GROUP_FLOATS = BM * LDS_STRIDE;                  // one K group's activation tile (8 KB)
    __shared__ __attribute__((aligned(16))) float lds[2][2 * GROUP_FLOATS];

    const int tid = threadIdx.x;
    // the wave index decides control flow here (K group, tile counts): make it a scalar for the compiler, or every
    // `j < nt` becomes an exec-mask branch and the segment selects turn into per-lane loads of the kernel arguments
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave & 1, wk = wave >> 1;                       // column half, K group
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
    const int rem = local - tm * T.tm_stride;
    if (rem >= T.tiles_n * T.ksplit) return;          // padding slot
    const int ks = rem % T.ksplit;
    const int tn = rem / T.ksplit;
    const int m0 = tm * BM, n0 = tn * BN;
    const int kt0 = (int)(((long long)ks * T.ktiles) / T.ksplit);
    const int kt1 = (int)(((long long)(ks + 1) * T.ktiles) / T.ksplit);
    if (gate_alive && *gate_alive == 0) return;                    // (the compacted row list is not supported here: gemm_group)
    const int epi_m = T.M;
    auto epi_row = [](int row) { return row; };
    // this K group's k-tiles: kt0 + wk, kt0 + wk + 2, ...  (local index j <-> global k-tile kt0 + wk + 2 j)
    const int nt_wg = kt1 - kt0;
    const int nt = (nt_wg - wk + 1) >> 1;                          // tiles of this group
    const int nt_max = (nt_wg + 1) >> 1;                           // tiles of group 0 = barrier rounds of the workgroup

    // ---- activation staging: thread g (0..127 of the group) -> rows g/8 + 16 i, 16-byte column g%8
    const int tg = tid & 127;
    const int srow = tg >> 3, scol = (tg & 7) * 4;
    const int sswz = ((tg & 7) ^ ((srow >> 1) & 7)) * 4;           // rows srow + 16 i share (r >> 1) & 7
    int arow[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { const int r = m0 + srow + 16 * i; arow[i] = r < T.M ? r : T.M - 1; }
    // ---- weight fragments: lane -> row n0 + 32 wn + (lane & 31), floats 4 (lane >> 5) + 8 kk of a k-tile
    const int frow = lane & 31, fh = lane >> 5;
    int wrow = n0 + 32 * wn + frow;
    wrow = wrow < T.N ? wrow : T.N - 1;

    // Operand addresses are derived per k-tile from the (wave-uniform) global k-tile index with branch-free selects over the
    // <= 3 K segments: a dozen SALU + a few 64-bit VALU adds per 32 MFMAs, no running pointers, no seek branches (the seek
    // code of gemm_nt_f32, instantiated for two streams x two rounds, cost 540 spilled SGPRs here).
    const long long arow_l[4] = {arow[0], arow[1], arow[2], arow[3]};
    const int nseg_ = T.nseg, ke0_ = T.kt_end[0], ke1_ = T.kt_end[1];
    const float *A0_ = T.A[0], *A1_ = T.A[1], *A2_ = T.A[2], *W0_ = T.W[0], *W1_ = T.W[1], *W2_ = T.W[2];
    const long long la0_ = T.lda[0], la1_ = T.lda[1], la2_ = T.lda[2], lw0_ = T.ldw[0], lw1_ = T.ldw[1], lw2_ = T.ldw[2];
#define WR_SEG(G)                                                                                       \
        const int g_ = (G);                                                                             \
        const bool s1_ = nseg_ > 1 && g_ >= ke0_, s2_ = nseg_ > 2 && g_ >= ke1_;                        \
        const int kb_ = s2_ ? ke1_ : (s1_ ? ke0_ : 0);                                                  \
        const long long ko_ = (long long)(g_ - kb_) * GEMM_BK;
    // Loads are UNCONDITIONAL (a request past the group's last tile repeats that tile: a cache hit whose result is never
    // used): with a branch around a load the compiler's s_waitcnt placement must assume the path with fewer requests
    // outstanding and ends up draining the whole queue (vmcnt(0)) in the middle of every round.
    const int jlast = nt > 0 ? nt - 1 : 0;
    const int glast = kt1 - 1;
#define WR_TILE(J) ({ int j_ = (J) < jlast ? (J) : jlast; int g_ = kt0 + wk + 2 * j_; g_ < glast ? g_ : glast; })
#define WR_STAGE_A(J, RA)                                                                               \
    {                                                                                                   \
        WR_SEG(WR_TILE(J))                                                                              \
        const float* Ab_ = (s2_ ? A2_ : (s1_ ? A1_ : A0_)) + ko_ + scol;                                \
        const long long lda_ = s2_ ? la2_ : (s1_ ? la1_ : la0_);                                        \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) RA[i] = *(gptr4)(Ab_ + arow_l[i] * lda_);         \
    }
#define WR_LSTORE(BUF, RA)                                                                              \
    {                                                                                                   \
        float* sA_ = lds[(BUF)] + wk * GROUP_FLOATS;                                                    \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                   \
            *reinterpret_cast<f32x4*>(sA_ + (srow + 16 * i) * LDS_STRIDE + sswz) = RA[i];               \
    }
    // weight stream: one row pointer per tile (WR_W_BEGIN), its four 16-byte pieces are requested one by one
    const float* pw = nullptr;
#define WR_W_BEGIN(J)                                                                                   \
    {                                                                                                   \
        WR_SEG(WR_TILE(J))                                                                              \
        const float* Wb_ = (s2_ ? W2_ : (s1_ ? W1_ : W0_)) + ko_ + 4 * fh;                              \
        const long long ldw_ = s2_ ? lw2_ : (s1_ ? lw1_ : lw0_);                                        \
        pw = Wb_ + (long long)wrow * ldw_;                                                              \
    }
#define WR_W_PIECE(KK, WR) WR[KK] = *(gptr4)(pw + 8 * (KK));

    f32x16 acc2[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc2[i][e] = 0.f;
    int fo[4];                                                     // swizzled float offset of k-chunk 2 kk + (lane >> 5)
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) fo[kk] = (((kk * 2 + fh) ^ ((frow >> 1) & 7)) * 4);

    f32x4 we[4], wo[4];                   // weight fragments of the even / odd local tiles
#define WR_FRAG(KK, FA)                                                                                  \
    {                                                                                                   \
        FA[0] = *reinterpret_cast<const f32x4*>(sA + fo[KK]);                                           \
        FA[1] = *reinterpret_cast<const f32x4*>(sA + 32 * LDS_STRIDE + fo[KK]);                         \
    }
#define WR_MFMA(FA, W4)                                                                                 \
    {                                                                                                   \
        acc2[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(FA[0].x, (W4).x, acc2[0], 0, 0, 0);              \
        acc2[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(FA[1].x, (W4).x, acc2[1], 0, 0, 0);              \
        acc2[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(FA[0].y, (W4).y, acc2[0], 0, 0, 0);              \
        acc2[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(FA[1].y, (W4).y, acc2[1], 0, 0, 0);              \
        acc2[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(FA[0].z, (W4).z, acc2[0], 0, 0, 0);              \
        acc2[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(FA[1].z, (W4).z, acc2[1], 0, 0, 0);              \
        acc2[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(FA[0].w, (W4).w, acc2[0], 0, 0, 0);              \
        acc2[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(FA[1].w, (W4).w, acc2[1], 0, 0, 0);              \
    }
    // one round: MFMAs of local tile J (activations from lds[BUF], weights from WR); meanwhile the register stage RA (tile
    // J + 1) goes to lds[BUF ^ 1] and is re-requested with tile J + 3, and every weight piece is re-requested with tile
    // J + 2's as soon as its MFMAs have issued.  Straight-line code: both K groups run the same `nfull` rounds; the odd
    // k-tile of a slice, if any, is group 0's and is contracted after the loop (WR_TAIL).
    // The hand-written order must survive the compiler: MFMAs are pure values to LLVM and get sunk past the weight
    // re-requests (which then go through temporaries and come back as v_mov copies behind s_waitcnt vmcnt(0)); an IR-level
    // sched_barrier does not stop that.  An empty volatile asm that "rewrites" both accumulators and clobbers memory does:
    // every MFMA before it must have issued, every load / LDS access after it stays after it.
#define WR_SB() asm volatile("" : "+a"(acc2[0]), "+a"(acc2[1]) : : "memory")
#define WR_ROUND(J, BUF, RA, WR)                                                                        \
    {                                                                                                   \
        const float* sA = lds[BUF] + wk * GROUP_FLOATS + frow * LDS_STRIDE;                             \
        f32x4 fa0[2], fa1[2];                                                                           \
        WR_FRAG(0, fa0);                                                                                \
        WR_FRAG(1, fa1);                                                                                \
        WR_SB();                                                                                        \
        WR_MFMA(fa0, WR[0]);                                                                            \
        WR_SB();                                                                                        \
        WR_W_BEGIN((J) + 2);                                                                            \
        WR_W_PIECE(0, WR);                                                                              \
        WR_LSTORE((BUF) ^ 1, RA);                                                                       \
        WR_STAGE_A((J) + WR_ADIST, RA);                                                                 \
        WR_FRAG(2, fa0);                                                                                \
        WR_SB();                                                                                        \
        WR_MFMA(fa1, WR[1]);                                                                            \
        WR_SB();                                                                                        \
        WR_W_PIECE(1, WR);                                                                              \
        WR_FRAG(3, fa1);                                                                                \
        WR_SB();                                                                                        \
        WR_MFMA(fa0, WR[2]);                                                                            \
        WR_SB();                                                                                        \
        WR_W_PIECE(2, WR);                                                                              \
        WR_SB();                                                                                        \
        WR_MFMA(fa1, WR[3]);                                                                            \
        WR_SB();                                                                                        \
        WR_W_PIECE(3, WR);                                                                              \
        __syncthreads();                                                                                \
    }
#define WR_TAIL(BUF, WR)                                                                                \
    {                                                                                                   \
        const float* sA = lds[BUF] + frow * LDS_STRIDE;                                                 \
        f32x4 fa0[2], fa1[2];                                                                           \
        WR_FRAG(0, fa0);                                                                                \
        WR_FRAG(1, fa1);                                                                                \
        WR_MFMA(fa0, WR[0]);                                                                            \
        WR_FRAG(2, fa0);                                                                                \
        WR_MFMA(fa1, WR[1]);                                                                            \
        WR_FRAG(3, fa1);                                                                                \
        WR_MFMA(fa0, WR[2]);                                                                            \
        WR_MFMA(fa1, WR[3]);                                                                            \
    }
#ifdef SET_WREG_A2
    // two activation register stages (tile j + 1 landed, tile j + 2 in flight at the top of round j): 16 more registers,
    // two waves per SIMD
#define WR_ADIST 3
    f32x4 ra0[4], ra1[4];
    {
        WR_STAGE_A(0, ra0);                  // tile 0
        WR_W_BEGIN(0);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) WR_W_PIECE(kk, we);
        WR_STAGE_A(1, ra1);                  // tile 1
        WR_W_BEGIN(1);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) WR_W_PIECE(kk, wo);
        WR_LSTORE(0, ra0);
        WR_STAGE_A(2, ra0);                  // tile 2
    }
    __syncthreads();
#define WR_RA_EVEN ra1
#define WR_RA_ODD ra0
#else
    // one activation register stage: tile j + 1 is requested in round j - 1 (right after the stage went to LDS) and stored
    // in round j — the activations are a few MB that every workgroup of the launch re-reads, i.e. L2 hits
#define WR_ADIST 2
    f32x4 ra0[4];
    {
        // requests in the order a round issues them (weight piece 0, the activation tile, pieces 1-3): the queue the first
        // round finds is then the queue every round finds, and the compiler's vmcnt values at the loop head stay exact
        WR_W_BEGIN(0);
        WR_W_PIECE(0, we);
        WR_STAGE_A(0, ra0);                  // tile 0
        WR_W_PIECE(1, we);
        WR_W_PIECE(2, we);
        WR_W_PIECE(3, we);
        WR_W_BEGIN(1);
        WR_W_PIECE(0, wo);
        WR_LSTORE(0, ra0);
        WR_STAGE_A(1, ra0);                  // tile 1
        WR_W_PIECE(1, wo);
        WR_W_PIECE(2, wo);
        WR_W_PIECE(3, wo);
    }
    __syncthreads();
#define WR_RA_EVEN ra0
#define WR_RA_ODD ra0
#endif
    // invariant at the top of an even round j: lds[0] = tile j, the register stage(s) hold tile j + 1 (and j + 2), we = W(j),
    // wo = W(j + 1)
    const int nfull = nt_wg >> 1;
    const bool odd_tail = (nt_wg & 1) && wk == 0;        // the slice's odd k-tile is group 0's
    int j = 0;
    for (; j + 2 <= nfull; j += 2) {
        WR_ROUND(j, 0, WR_RA_EVEN, we);
        WR_ROUND(j + 1, 1, WR_RA_ODD, wo);
    }
    if (j < nfull) {                                     // odd number of full rounds: one more on buffer 0
        WR_ROUND(j, 0, WR_RA_EVEN, we);
        if (odd_tail) WR_TAIL(1, wo);
    } else if (odd_tail) {
        WR_TAIL(0, we);
    }
    __syncthreads();                                     // (all fragment reads of the stages are done: the epilogue reuses them)
#undef WR_ROUND
#undef WR_ADIST
#undef WR_RA_EVEN
#undef WR_RA_ODD
#undef WR_SB
#undef WR_TAIL
#undef WR_MFMA
#undef WR_FRAG
#undef WR_W_PIECE
#undef WR_W_BEGIN
#undef WR_LSTORE
#undef WR_STAGE_A
#undef WR_TILE
#undef WR_SEG
    // ---- the two K groups exchange one 32x32 accumulator each (lane-major in lds[1]: conflict-free): group 0 keeps the
    // tile's rows 0-31, group 1 rows 32-63; every sum is (group 0's partial) + (group 1's partial)
    {
        // (static register indices only: `acc2[wk]` with a run-time wk turns into thousands of v_cndmask; the result lands in
        // acc2[0] for both groups so that no third accumulator is live)
        float* sX = &lds[1][0] + wave * 1024 + lane;
        const float* sY = &lds[1][0] + (wave ^ 2) * 1024 + lane;
        if (wk == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sX[r * 64] = acc2[1][r];
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) sX[r * 64] = acc2[0][r];
        }
        __syncthreads();
        if (wk == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[0][r] = acc2[0][r] + sY[r * 64];
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[0][r] = sY[r * 64] + acc2[1][r];
        }
    }
    f32x16 (&acc)[1][1] = *reinterpret_cast<f32x16 (*)[1][1]>(&acc2[0]);
    constexpr int TM = 1, TN = 1, KG = 1;
    const int kg = 0, wm = wk;
    // (the shared epilogue transposes through lds[0], which nobody reads any more: the last round's barrier is behind us)
#include "gemm_f32_epilogue.inc"
}

// ---------------------------------------------------------------------------------------------
// The same grouped GEMM with the operand tiles staged by LDS-DMA (`global_load_lds_dwordx4`): no VGPR round trip and no
// ds_write pass.  A wave-instruction lands 64 x 16 B = 1 KB = eight 128-byte rows of the stage CONTIGUOUSLY (the LDS
// destination is wave-uniform base + lane * 16), so the XOR swizzle of the 16-byte chunks is applied on the SOURCE side:
// lane l of the piece that fills rows r0..r0+7 writes chunk position c' = l & 7 of row r = r0 + (l >> 3) and therefore
// fetches logical chunk c = c' ^ ((r >> 1) & 7) of that row — still one full 128-byte line per 8 lanes.  The MFMA fragment
// reads are those of gemm_nt_f32 (same image).  Three stages: while tile kt is contracted, tile kt+1 has landed or is
// landing and tile kt+2 is being requested; one raw s_barrier per k-tile, counted vmcnt so that a DMA stays in flight
// across the barrier (__syncthreads() would drain it: an LDS-DMA is a pending LDS write on the VM counter).
// ---------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void* lds_vptr;
typedef const __attribute__((address_space(1))) void* glb_vptr;

template <int BM, int BN, int WAVES_M, int WAVES_N>
__global__ void __launch_bounds__(256) gemm_nt_f32_dma(const int ntasks, const int wb1, const int wb2, const int wb3,
                                                       const int wb4, const int wb5, const int* const gate_alive,
        const int* const gate_nrows, const GemmLaunch L) {
    static_assert(WAVES_M * WAVES_N == 4, "4 waves");
    constexpr int KG = 1;
    constexpr int TM = BM / WAVES_M / 32, TN = BN / WAVES_N / 32;
    constexpr int ROWS = BM + BN;                        // rows of one stage: A tile rows, then W tile rows
    constexpr int PIECES = ROWS / 32;                    // 8-row DMA pieces per wave per k-tile
    static_assert(ROWS % 32 == 0 && TM >= 1 && TN >= 1, "tile");
    constexpr int NSTAGE = 3;
    __shared__ __attribute__((aligned(16))) float lds[NSTAGE][ROWS * LDS_STRIDE];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6, kg = 0;
    const int wm = wave % WAVES_M, wn = wave / WAVES_M;
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
    const int rem = local - tm * T.tm_stride;
    if (rem >= T.tiles_n * T.ksplit) return;          // padding slot
    if (gate_alive && *gate_alive == 0) return;                    // (the compacted row list is not supported here: gemm_group)
    const int ks = rem % T.ksplit;
    const int tn = rem / T.ksplit;
    const int m0 = tm * BM, n0 = tn * BN;
    const int kt0 = (int)(((long long)ks * T.ktiles) / T.ksplit);
    const int kt1 = (int)(((long long)(ks + 1) * T.ktiles) / T.ksplit);
    const int epi_m = T.M;
    auto epi_row = [](int row) { return row; };

    // ---- DMA assignment: piece i of this wave fills stage rows [8 * (wave * PIECES + i), +8); lane -> (row, chunk position)
    int prow[PIECES];            // global row (clamped) of this lane's stage row, in A (stage row < BM) or W
    int pcol[PIECES];            // float offset of the logical chunk this lane fetches
    bool pisw[PIECES];
#pragma unroll
    for (int i = 0; i < PIECES; ++i) {
        const int sr = 8 * (wave * PIECES + i) + (lane >> 3);        // stage row
        pisw[i] = sr >= BM;
        int r = pisw[i] ? n0 + (sr - BM) : m0 + sr;
        const int lim = pisw[i] ? T.N : T.M;
        prow[i] = r < lim ? r : lim - 1;
        pcol[i] = ((lane & 7) ^ ((sr >> 1) & 7)) * 4;
    }
    const float* pp[PIECES];     // running per-lane source pointers inside the current K segment
    int seg_end = 0;
#define DMA_SEEK(KT)                                                                                    \
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
        const long long koff_ = (long long)(kt_ - kbase_) * GEMM_BK;                                    \
        _Pragma("unroll") for (int i = 0; i < PIECES; ++i)                                              \
            pp[i] = (pisw[i] ? Wb_ + prow[i] * ldw_ : Ab_ + prow[i] * lda_) + koff_ + pcol[i];          \
    }
    // request tile KT into stage SLOT (no-op past the end of the slice)
#define DMA_ISSUE(KT, SLOT)                                                                             \
    if ((KT) < kt1) {                                                                                   \
        if ((KT) == seg_end) DMA_SEEK(KT);                                                              \
        _Pragma("unroll") for (int i = 0; i < PIECES; ++i) {                                            \
            __builtin_amdgcn_global_load_lds((glb_vptr)pp[i],                                           \
                (lds_vptr)(&lds[SLOT][(8 * (wave * PIECES + i)) * LDS_STRIDE]), 16, 0, 0);              \
            pp[i] += GEMM_BK;                                                                           \
        }                                                                                               \
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int frow = lane & 31;
    int fo[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) fo[kk] = (((kk * 2 + (lane >> 5)) ^ ((frow >> 1) & 7)) * 4);

    if (kt0 < kt1) {
        DMA_SEEK(kt0);
        DMA_ISSUE(kt0, 0);
        DMA_ISSUE(kt0 + 1, 1);
    }
    int slot = 0;
    for (int kt = kt0; kt < kt1; ++kt) {
        // tile kt has landed once this wave's older pieces are done (the PIECES of tile kt+1 may still fly) and every wave
        // has said so at the barrier; the barrier also tells that everybody is done READING stage (kt-1) % 3 = (kt+2) % 3
        if (kt + 1 < kt1) { if constexpr (PIECES == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                            else if constexpr (PIECES == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                            else asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const int nslot = slot == 0 ? 2 : slot - 1;             // (slot + 2) % 3
        DMA_ISSUE(kt + 2, nslot);
        const float* sA = &lds[slot][0] + (wm * TM * 32 + frow) * LDS_STRIDE;
        const float* sW = &lds[slot][0] + BM * LDS_STRIDE + (wn * TN * 32 + frow) * LDS_STRIDE;
        f32x4 fa0[TM], fb0[TN], fa1[TM], fb1[TN];
#define DMA_FRAG(KK, FA, FB)                                                                            \
        {                                                                                               \
            _Pragma("unroll") for (int i = 0; i < TM; ++i)                                              \
                FA[i] = *reinterpret_cast<const f32x4*>(sA + i * 32 * LDS_STRIDE + fo[KK]);             \
            _Pragma("unroll") for (int j = 0; j < TN; ++j)                                              \
                FB[j] = *reinterpret_cast<const f32x4*>(sW + j * 32 * LDS_STRIDE + fo[KK]);             \
        }
#define DMA_MFMA(FA, FB)                                                                                \
        {                                                                                               \
            _Pragma("unroll") for (int i = 0; i < TM; ++i)                                              \
                _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                        \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(FA[i].x, FB[j].x, acc[i][j], 0, 0, 0); \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(FA[i].y, FB[j].y, acc[i][j], 0, 0, 0); \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(FA[i].z, FB[j].z, acc[i][j], 0, 0, 0); \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(FA[i].w, FB[j].w, acc[i][j], 0, 0, 0); \
                }                                                                                       \
        }
        DMA_FRAG(0, fa0, fb0);
        DMA_FRAG(1, fa1, fb1);
        DMA_MFMA(fa0, fb0);
        DMA_FRAG(2, fa0, fb0);
        DMA_MFMA(fa1, fb1);
        DMA_FRAG(3, fa1, fb1);
        DMA_MFMA(fa0, fb0);
        DMA_MFMA(fa1, fb1);
        slot = slot == 2 ? 0 : slot + 1;
    }
#undef DMA_FRAG
#undef DMA_MFMA
#undef DMA_ISSUE
#undef DMA_SEEK
    __syncthreads();                                     // all fragment reads done: the stages are free for the epilogue
#include "gemm_f32_epilogue.inc"
}

// ---------------------------------------------------------------------------------------------
// EXPERIMENTAL, opt-in (SET_GEMM_SPLIT=1; never the default, never the headline number): the same grouped
// GEMM on the bf16 matrix pipe with every fp32 operand split EXACTLY into three bf16 values
//      x = hi + mid + lo        (8 + 8 + 8 significand bits, by truncation and exact fp32 subtraction)
// and six of the nine partial products accumulated in fp32 (hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid; the
// dropped ones are <= 2^-24 relative, i.e. at the level of one fp32 rounding).  v_mfma_f32_32x32x16_bf16 has 16x
// the rate of v_mfma_f32_32x32x2_f32, so six of them cost 2.67x less matrix-pipe time than the fp32 chain.
// Splitting happens on the fly when a k-tile is staged into LDS (weights stay fp32 in HBM, nothing is
// repacked); LDS holds three bf16 planes per operand, 64-B rows, 16-B chunks XOR-swizzled by (row>>2)&3 so
// that the ds_write_b64 of the staging and the ds_read_b128 of the fragments are bank-conflict-free.
// Same task descriptors, K segments, split-K slabs and epilogue as gemm_nt_f32<128,64>.
// ---------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) gemm_nt_split_bf16(const GemmLaunch L) {
    constexpr int BM = 128, BN = 64, TM = 2, LA = 4, LW = 2;
    constexpr int ROWB = 64;                              // bytes per LDS row (32 bf16)
    constexpr int PLANE_A = BM * ROWB, PLANE_W = BN * ROWB;
    constexpr int BUF = 3 * (PLANE_A + PLANE_W);          // 36 KB per stage
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;

    int ti = 0;
#pragma unroll
    for (int i = 1; i < GEMM_MAX_TASKS; ++i)
        if (i < L.ntasks && (int)blockIdx.x >= L.t[i].wg_begin) ti = i;
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

    const int srow = tid >> 3, kc = tid & 7, scol = kc * 4;
    // byte offset of this thread's 8-byte store inside a plane row
    const int soff = srow * ROWB + (((kc >> 1) ^ ((srow >> 2) & 3)) << 4) + ((kc & 1) << 3);
    int arow[LA], wrow[LW];
#pragma unroll
    for (int i = 0; i < LA; ++i) { int r = m0 + srow + 32 * i; arow[i] = r < T.M ? r : T.M - 1; }
#pragma unroll
    for (int i = 0; i < LW; ++i) { int r = n0 + srow + 32 * i; wrow[i] = r < T.N ? r : T.N - 1; }

    f32x4 ra0[LA], rw0[LW], ra1[LA], rw1[LW];
    const float* pa[LA];
    const float* pw[LW];
    int seg_end = 0;
#define SPL_SEEK(KT)                                                                                    \
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
#define SPL_GLOAD(RA, RW)                                                                           \
    {                                                                                                   \
        _Pragma("unroll") for (int i = 0; i < LA; ++i) { RA[i] = *(gptr4)(pa[i]); pa[i] += GEMM_BK; } \
        _Pragma("unroll") for (int i = 0; i < LW; ++i) { RW[i] = *(gptr4)(pw[i]); pw[i] += GEMM_BK; } \
    }
#define SPL_STAGE(KT, RA, RW)                                                                       \
    if ((KT) < kt1) {                                                                                   \
        if ((KT) == seg_end) SPL_SEEK(KT);                                                              \
        SPL_GLOAD(RA, RW);                                                                          \
    }
    // registers (fp32) -> three bf16 planes in LDS
#define SPL_LSTORE(B, RA, RW)                                                                       \
    {                                                                                                   \
        char* sA_ = smem + (B) * BUF + soff;                                                            \
        char* sW_ = smem + (B) * BUF + 3 * PLANE_A + soff;                                              \
        _Pragma("unroll") for (int i = 0; i < LA; ++i) {                                                \
            u32x2 h_, m_, l_;                                                                           \
            split3(RA[i], h_, m_, l_);                                                                  \
            *reinterpret_cast<u32x2*>(sA_ + i * 32 * ROWB) = h_;                                        \
            *reinterpret_cast<u32x2*>(sA_ + i * 32 * ROWB + PLANE_A) = m_;                              \
            *reinterpret_cast<u32x2*>(sA_ + i * 32 * ROWB + 2 * PLANE_A) = l_;                          \
        }                                                                                               \
        _Pragma("unroll") for (int i = 0; i < LW; ++i) {                                                \
            u32x2 h_, m_, l_;                                                                           \
            split3(RW[i], h_, m_, l_);                                                                  \
            *reinterpret_cast<u32x2*>(sW_ + i * 32 * ROWB) = h_;                                        \
            *reinterpret_cast<u32x2*>(sW_ + i * 32 * ROWB + PLANE_W) = m_;                              \
            *reinterpret_cast<u32x2*>(sW_ + i * 32 * ROWB + 2 * PLANE_W) = l_;                          \
        }                                                                                               \
    }

    f32x16 acc[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

    const int frow = lane & 31;
    const int fsw = (frow >> 2) & 3;
    // byte offsets of this lane's fragment rows (A: 2 sub-tiles, W: 1) and of its chunk in k16-step 0 / 1
    const int fa0 = (wm * 64 + frow) * ROWB, fw0 = (wn * 32 + frow) * ROWB;
    const int fc[2] = {(((lane >> 5)) ^ fsw) << 4, ((2 + (lane >> 5)) ^ fsw) << 4};
#define SPL_FRAG(S, FA, FW)                                                                             \
    {                                                                                                   \
        _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                  \
            _Pragma("unroll") for (int p = 0; p < 3; ++p)                                               \
                FA[i][p] = *reinterpret_cast<const u32x4*>(sA + p * PLANE_A + fa0 + i * 32 * ROWB + fc[S]); \
        _Pragma("unroll") for (int p = 0; p < 3; ++p)                                                   \
            FW[p] = *reinterpret_cast<const u32x4*>(sW + p * PLANE_W + fw0 + fc[S]);                    \
    }
#define SPL_MM(I, PA, PW) acc[I] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(                             \
        __builtin_bit_cast(bf16x8, fa_[I][PA]), __builtin_bit_cast(bf16x8, fw_[PW]), acc[I], 0, 0, 0)
    // six products per sub-tile, smallest first; the two sub-tiles alternate so consecutive MFMAs are independent
#define SPL_MFMA(FA, FW)                                                                                \
    {                                                                                                   \
        auto& fa_ = FA; auto& fw_ = FW;                                                                 \
        SPL_MM(0, 0, 2); SPL_MM(1, 0, 2);                                                               \
        SPL_MM(0, 2, 0); SPL_MM(1, 2, 0);                                                               \
        SPL_MM(0, 1, 1); SPL_MM(1, 1, 1);                                                               \
        SPL_MM(0, 0, 1); SPL_MM(1, 0, 1);                                                               \
        SPL_MM(0, 1, 0); SPL_MM(1, 1, 0);                                                               \
        SPL_MM(0, 0, 0); SPL_MM(1, 0, 0);                                                               \
    }
    // One k-tile.  The split of tile kt+1 (VALU) and its LDS stores are issued IN BETWEEN the 24 MFMAs of tile kt
    // (sched_group_barrier pins the interleave: the matrix pipe runs 32 cycles per MFMA, enough for ~7 VALU
    // issues), so the conversion work hides under the matrix pipe instead of alternating with it.  The store
    // is unconditional: on the last tile it writes stale registers into the buffer nobody reads again.
#define SPL_ITER(KT, B, RA, RW)                                                                     \
    {                                                                                                   \
        const char* sA = smem + (B) * BUF;                                                              \
        const char* sW = smem + (B) * BUF + 3 * PLANE_A;                                                \
        u32x4 fA0[TM][3], fW0[3], fA1[TM][3], fW1[3];                                                   \
        SPL_FRAG(0, fA0, fW0);                                                                          \
        SPL_FRAG(1, fA1, fW1);                                                                          \
        SPL_LSTORE((B) ^ 1, RA, RW);                                                                \
        SPL_MFMA(fA0, fW0);                                                                             \
        SPL_MFMA(fA1, fW1);                                                                             \
        _Pragma("unroll") for (int q_ = 0; q_ < 24; ++q_) {                                             \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                          \
            __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);                                          \
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);                                          \
        }                                                                                               \
        SPL_STAGE((KT) + 3, RA, RW);                                                                \
        __syncthreads();                                                                                \
    }
    if (kt0 < kt1) {
        SPL_SEEK(kt0);
        SPL_GLOAD(ra0, rw0);
        SPL_STAGE(kt0 + 1, ra1, rw1);
        SPL_LSTORE(0, ra0, rw0);
        SPL_STAGE(kt0 + 2, ra0, rw0);
        __syncthreads();
    }
    for (int kt = kt0; kt < kt1; kt += 2) {
        SPL_ITER(kt, 0, ra1, rw1);
        if (kt + 1 < kt1) SPL_ITER(kt + 1, 1, ra0, rw0);
    }
#undef SPL_ITER
#undef SPL_MFMA
#undef SPL_MM
#undef SPL_FRAG
#undef SPL_LSTORE
#undef SPL_STAGE
#undef SPL_GLOAD
#undef SPL_SEEK

    float* Cs = T.C + (long long)ks * T.slab_stride;
    const bool fused = (T.ksplit == 1);
    const int crow0 = m0 + wm * 64 + 4 * (lane >> 5);
    const int col = n0 + wn * 32 + (lane & 31);
    if (col < T.N) {
        const float bv = (fused && T.bias) ? T.bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = crow0 + i * 32 + (r & 3) + 8 * (r >> 2);
                if (row < T.M) {
                    float v = acc[i][r];
                    if (fused) v = apply_act(v + bv, T.act);
                    Cs[(long long)row * T.ldc + col] = v;
                }
            }
        }
    }
}




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
static int gemm_dma() { static int v = env_int("SET_GEMM_DMA", 0); return v; }
int g_gemm_asm_force = -1;         // tools/ubench: switch kernels inside one process (-1: the environment decides)
static int gemm_asm() { static int v = env_int("SET_GEMM_ASM", 1); return g_gemm_asm_force >= 0 ? g_gemm_asm_force : v; }
int g_gemm_wreg_force = -1;        // tools/ubench: switch kernels inside one process (-1: the environment decides)
static int gemm_wreg() { static int v = env_int("SET_GEMM_WREG", 0); return g_gemm_wreg_force >= 0 ? g_gemm_wreg_force : v; }
static int gemm_kgroups() { static int v = env_int("SET_GEMM_KGROUPS", 1); return v; }
static int gemm_bn128() { static int v = env_int("SET_GEMM_BN128", 0); return v; }
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
    static const int on = env_int("SET_GEMM_BN32", 0);
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
    if (bm == 128 && bn == 64 && split_here) {
        constexpr int kLds = 2 * 3 * (128 + 64) * 64;
        static bool attr_set = false;
        if (!attr_set) {
            SET_HIP_TRY(hipFuncSetAttribute((const void*)gemm_nt_split_bf16, hipFuncAttributeMaxDynamicSharedMemorySize,
                                            kLds));
            attr_set = true;
        }
        hipLaunchKernelGGL(gemm_nt_split_bf16, grid, block, kLds, stream, L);
    } else {
        const int nt = L.ntasks, w1 = L.t[1].wg_begin, w2 = L.t[2].wg_begin, w3 = L.t[3].wg_begin, w4 = L.t[4].wg_begin,
                  w5 = L.t[5].wg_begin;
        const int* ga = L.gate.alive_prev;
        const int* gn = nullptr;                 // (unused slot of the preloaded argument block)
        const int gate_mode = ga ? 1 : 0;
        if (bm == 16)
            hipLaunchKernelGGL(gemv_nt_f32, grid, block, 0, stream, nt, w1, w2, w3, w4, w5, ga, gn, L);
        else if (bm == 128 && bn == 32)
            hipLaunchKernelGGL((gemm_nt_f32<128, 32, 4, 1>), grid, block, 0, stream, nt, w1, w2, w3, w4, w5, ga, gn, L);
        else if (bm == 128 && bn == 128)
            hipLaunchKernelGGL((gemm_nt_f32<128, 128, 2, 2>), grid, block, 0, stream, nt, w1, w2, w3, w4, w5, ga, gn, L);
        else if (bm == 128 && gemm_dma())
            hipLaunchKernelGGL((gemm_nt_f32_dma<128, 64, 2, 2>), grid, block, 0, stream, nt, w1, w2, w3, w4, w5, ga, gn, L);
        else if (bm == 128 && gate_mode)
            hipLaunchKernelGGL((gemm_nt_f32<128, 64, 2, 2, 1, 1>), grid, block, 0, stream, nt, w1, w2, w3, w4, w5, ga, gn, L);
        else if (bm == 128)
            hipLaunchKernelGGL((gemm_nt_f32<128, 64, 2, 2>), grid, block, 0, stream, nt, w1, w2, w3, w4, w5, ga, gn, L);
        else if (bm == 64 && gemm_wreg())
            hipLaunchKernelGGL(gemm_nt_f32_wreg, grid, block, 0, stream, nt, w1, w2, w3, w4, w5, ga, gn, L);
        else if (bm == 64 && gemm_kgroups() == 2)
            hipLaunchKernelGGL((gemm_nt_f32<64, 64, 2, 2, 2>), grid, dim3(512), 0, stream, nt, w1, w2, w3, w4, w5, ga, gn, L);
        else if (bm == 64 && gemm_dma())
            hipLaunchKernelGGL((gemm_nt_f32_dma<64, 64, 2, 2>), grid, block, 0, stream, nt, w1, w2, w3, w4, w5, ga, gn, L);
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
