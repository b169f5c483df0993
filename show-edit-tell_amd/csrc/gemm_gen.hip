// General-layout fp32 MFMA GEMM for the training path (gfx950):
//
//      C[m, n] (+)= sum_k a(m, k) * b(n, k)
//      a(m, k) = A[m*lda + k]  ("k-major", rows of A are contraction vectors)   or  A[k*lda + m]  ("k-minor")
//      b(n, k) = B[n*ldb + k]                                                   or  B[k*ldb + n]
//
// The decode path only ever needs (k-major, k-major) -- activations x nn.Linear weights -- and has its own
// grouped kernel (gemm_f32.hip).  The backward of a Linear needs the other two:
//      dX = dY . W          a = dY k-major,  b = W  k-minor       (contraction over the output features)
//      dW += dY^T . X       a = dY k-minor,  b = X  k-minor       (contraction over the rows of the batch)
// Same tiling as the forward kernel (128x64x32 or 64x64x32 tile, 2x2 waves, v_mfma_f32_32x32x2_f32, two LDS
// buffers + two register stages, one barrier per k-tile).  A k-minor operand is staged [k][BX] (rows of the
// source are copied as they lie, coalesced) and its MFMA fragment is four ds_read_b32 of consecutive lanes
// instead of one ds_read_b128; no transpose is ever materialised in HBM.
// Split-K writes slabs into caller scratch; a small kernel sums them in slab order (deterministic) and
// optionally accumulates into C, which is how `.grad` buffers are updated in place.
#include "set_common.h"

namespace set {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef const f32x4 __attribute__((address_space(1)))* gptr4;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int GEN_MAX_TASKS = 6;
struct GenTask {
    const float* A;
    const float* B;
    float* C;            // C itself (ksplit == 1) or the slab base
    long long lda, ldb, ldc, slab_stride;
    int M, N, K, ktiles, ksplit, tiles_n, accumulate, wg_begin;
    int vec_store;       // 16-byte epilogue stores allowed (N, ldc, slab stride multiples of 4 floats, C 16-byte aligned)
    unsigned ext_a, ext_b;   // HWB kernels: readable bytes of A / B (the buffer resource's range: loads beyond it return zeros)
    // reduction pass (ksplit > 1): out (+)= sum of slabs
    float* out;
    long long ldo;
    int red_begin;       // first 256-thread block of this task in the grouped reduction launch
    // in-launch combine (SET_GEN_COMBINE=1): one arrival counter per output tile, zero on entry and on exit; the last of a
    // tile's ksplit workgroups adds the slabs (in slab order, like slab_reduce_k) and writes `out`
    unsigned* counters;
};
struct GenLaunch {
    GenTask t[GEN_MAX_TASKS];
    int ntasks;
};

// COMBINE: the in-launch split-K combine (experiment, SET_GEN_COMBINE=1) is compiled into its own instantiations — its
// read-back holds BM/16 x 4 slab pieces per thread, which would otherwise set the register budget (and the occupancy) of
// the default kernels too
// HWB (round 5, the default): operands are read with raw buffer loads whose range check does the kernel's bounds work in
// hardware — a k-minor operand's rows k >= K lie beyond the resource's range and read as zeros, a load that must not happen
// (k >= K inside a k-major row, a k-tile beyond this workgroup's slice) is given an offset beyond the range — so the
// pipelined k-loop has NO branch besides its back edge: the compiler keeps two register stages in flight with partial
// vmcnt waits instead of draining every outstanding request at the join of a predicated load (six exec-mask branches per
// k-tile before).  HWB = false: global loads + predicates, for operands whose extent does not fit a 31-bit byte offset.
template <int BM, int BN, bool A_KMAJ, bool B_KMAJ, bool COMBINE = false, bool HWB = false>
__global__ void __launch_bounds__(256) gemm_gen_f32(const GenLaunch L) {
    constexpr int TM = BM / 64, TN = BN / 64;            // 2x2 waves
    constexpr int LA = BM / 32, LB = BN / 32;            // float4 loads per thread per k-tile
    __shared__ __attribute__((aligned(16))) float lds[2][(BM + BN) * 32];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    int ti = 0;
#pragma unroll
    for (int i = 1; i < GEN_MAX_TASKS; ++i)
        if (i < L.ntasks && (int)blockIdx.x >= L.t[i].wg_begin) ti = i;
    const GenTask& T = L.t[ti];
    const int local = (int)blockIdx.x - T.wg_begin;
    const int ks = local % T.ksplit;
    const int tile = local / T.ksplit;
    const int tn = tile % T.tiles_n, tm = tile / T.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int kt0 = (int)(((long long)ks * T.ktiles) / T.ksplit);
    const int kt1 = (int)(((long long)(ks + 1) * T.ktiles) / T.ksplit);

    // ---- staging maps
    // k-major: thread -> (row = tid/8 + 32 i, k chunk = tid%8), LDS [row][32] with the chunk XOR-swizzled
    // k-minor: float4 index f = tid + 256 i -> (k row = f / (BX/4), column chunk = f % (BX/4)), LDS [k][BX]
    const int srow = tid >> 3, scol = (tid & 7) * 4;
    const int sswz = ((tid & 7) ^ ((srow >> 1) & 7)) * 4;
    const float* pa[LA];
    const float* pb[LB];
    unsigned va[LA], vb[LB];         // HWB: byte offsets of the loads inside A / B
    int ka[LA], kb[LB];              // the k index this thread's load i covers inside a k-tile
    int sa[LA], sb[LB];              // LDS float offset of the store
    long long stepa, stepb;          // pointer advance per k-tile
#pragma unroll
    for (int i = 0; i < LA; ++i) {
        if constexpr (A_KMAJ) {
            int r = m0 + srow + 32 * i; r = r < T.M ? r : T.M - 1;
            pa[i] = T.A + (long long)r * T.lda + scol;
            va[i] = (unsigned)(((long long)r * T.lda + scol) * 4);
            ka[i] = scol; sa[i] = (srow + 32 * i) * 32 + sswz;
        } else {
            const int f = tid + 256 * i, kr = f / (BM / 4), c4 = (f % (BM / 4)) * 4;
            const int Mr = (T.M + 3) & ~3;       // a ragged M is accepted when the rows are readable up to Mr (host check)
            int c = m0 + c4; c = c + 4 <= Mr ? c : Mr - 4;
            pa[i] = T.A + (long long)kr * T.lda + c;
            va[i] = (unsigned)(((long long)kr * T.lda + c) * 4);
            ka[i] = kr; sa[i] = kr * BM + c4;
        }
    }
#pragma unroll
    for (int i = 0; i < LB; ++i) {
        if constexpr (B_KMAJ) {
            int r = n0 + srow + 32 * i; r = r < T.N ? r : T.N - 1;
            pb[i] = T.B + (long long)r * T.ldb + scol;
            vb[i] = (unsigned)(((long long)r * T.ldb + scol) * 4);
            kb[i] = scol; sb[i] = (srow + 32 * i) * 32 + sswz;
        } else {
            const int f = tid + 256 * i, kr = f / (BN / 4), c4 = (f % (BN / 4)) * 4;
            int c = n0 + c4; c = c + 4 <= T.N ? c : T.N - 4;
            pb[i] = T.B + (long long)kr * T.ldb + c;
            vb[i] = (unsigned)(((long long)kr * T.ldb + c) * 4);
            kb[i] = kr; sb[i] = kr * BN + c4;
        }
    }
    stepa = A_KMAJ ? 32 : 32 * T.lda;
    stepb = B_KMAJ ? 32 : 32 * T.ldb;
#pragma unroll
    for (int i = 0; i < LA; ++i) { pa[i] += (long long)kt0 * stepa; va[i] += (unsigned)(kt0 * stepa * 4); }
#pragma unroll
    for (int i = 0; i < LB; ++i) { pb[i] += (long long)kt0 * stepb; vb[i] += (unsigned)(kt0 * stepb * 4); }
    const unsigned vstepa = (unsigned)(stepa * 4), vstepb = (unsigned)(stepb * 4);
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)T.A, 0, HWB ? T.ext_a : 0u, 0x00027000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)T.B, 0, HWB ? T.ext_b : 0u, 0x00027000);
    constexpr unsigned GEN_OOB = 0x80000000u;    // beyond every resource's range (host: extents < 2^31 bytes): reads as zeros

    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    // loads of k-tile KT; contraction indices >= K contribute zeros (K need not be a multiple of 32)
#define GEN_GLOAD_PTR(KT, RA, RB)                                                                       \
    {                                                                                                   \
        const int kbase_ = (KT) * 32;                                                                   \
        _Pragma("unroll") for (int i = 0; i < LA; ++i) {                                                \
            RA[i] = (kbase_ + ka[i] < T.K) ? *(gptr4)(pa[i]) : zero4;                                   \
            pa[i] += stepa;                                                                             \
        }                                                                                               \
        _Pragma("unroll") for (int i = 0; i < LB; ++i) {                                                \
            RB[i] = (kbase_ + kb[i] < T.K) ? *(gptr4)(pb[i]) : zero4;                                   \
            pb[i] += stepb;                                                                             \
        }                                                                                               \
    }
    // HWB: every load is issued; what must not be read gets an out-of-range offset (a select, not a branch).  k-minor rows
    // k >= K are out of range by themselves (the resource ends with row K - 1)
#define GEN_GLOAD_BUF(KT, RA, RB)                                                                       \
    {                                                                                                   \
        const int kbase_ = (KT) * 32;                                                                   \
        const bool live_ = (KT) < kt1;                                                                  \
        _Pragma("unroll") for (int i = 0; i < LA; ++i) {                                                \
            const bool ok_ = A_KMAJ ? (live_ && kbase_ + ka[i] < T.K) : live_;                          \
            RA[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, ok_ ? va[i] : GEN_OOB, 0, 0)); \
            va[i] += vstepa;                                                                            \
        }                                                                                               \
        _Pragma("unroll") for (int i = 0; i < LB; ++i) {                                                \
            const bool ok_ = B_KMAJ ? (live_ && kbase_ + kb[i] < T.K) : live_;                          \
            RB[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB, ok_ ? vb[i] : GEN_OOB, 0, 0)); \
            vb[i] += vstepb;                                                                            \
        }                                                                                               \
    }
#define GEN_GLOAD(KT, RA, RB) { if constexpr (HWB) GEN_GLOAD_BUF(KT, RA, RB) else GEN_GLOAD_PTR(KT, RA, RB) }
#define GEN_STAGE(KT, RA, RB) { if constexpr (HWB) GEN_GLOAD_BUF(KT, RA, RB) else if ((KT) < kt1) GEN_GLOAD_PTR(KT, RA, RB) }
#define GEN_LSTORE(BUF, RA, RB)                                                                         \
    {                                                                                                   \
        float* sA_ = lds[(BUF)];                                                                        \
        float* sB_ = lds[(BUF)] + BM * 32;                                                              \
        _Pragma("unroll") for (int i = 0; i < LA; ++i) *reinterpret_cast<f32x4*>(sA_ + sa[i]) = RA[i];  \
        _Pragma("unroll") for (int i = 0; i < LB; ++i) *reinterpret_cast<f32x4*>(sB_ + sb[i]) = RB[i];  \
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int frow = lane & 31, kh = lane >> 5;
    int fo[4];                       // k-major: swizzled float offset of k-chunk 2*kk + kh in row frow
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) fo[kk] = (((kk * 2 + kh) ^ ((frow >> 1) & 7)) * 4);
    // fragment of sub-tile `sub` of one operand for k-block KK (k = 8 KK + 4 kh + e, e = 0..3)
#define GEN_FRAG(KMAJ, BX, S, WB, SUB, KK, OUT)                                                          \
    if constexpr (KMAJ) {                                                                               \
        OUT = *reinterpret_cast<const f32x4*>((S) + ((WB) + (SUB) * 32 + frow) * 32 + fo[KK]);           \
    } else {                                                                                            \
        const float* q_ = (S) + ((KK) * 8 + kh * 4) * (BX) + (WB) + (SUB) * 32 + frow;                   \
        OUT.x = q_[0]; OUT.y = q_[BX]; OUT.z = q_[2 * (BX)]; OUT.w = q_[3 * (BX)];                       \
    }
#define GEN_FRAG_LOAD(KK, FA, FB)                                                                       \
    {                                                                                                   \
        _Pragma("unroll") for (int i = 0; i < TM; ++i) GEN_FRAG(A_KMAJ, BM, sA, wm * TM * 32, i, KK, FA[i]) \
        _Pragma("unroll") for (int j = 0; j < TN; ++j) GEN_FRAG(B_KMAJ, BN, sB, wn * TN * 32, j, KK, FB[j]) \
    }
#define GEN_FRAG_MFMA(FA, FB)                                                                           \
    {                                                                                                   \
        _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                  \
            _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                            \
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(FA[i].x, FB[j].x, acc[i][j], 0, 0, 0); \
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(FA[i].y, FB[j].y, acc[i][j], 0, 0, 0); \
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(FA[i].z, FB[j].z, acc[i][j], 0, 0, 0); \
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(FA[i].w, FB[j].w, acc[i][j], 0, 0, 0); \
            }                                                                                           \
    }
#define GEN_ITER(KT, BUF, RA, RB)                                                                       \
    {                                                                                                   \
        const float* sA = lds[BUF];                                                                     \
        const float* sB = lds[BUF] + BM * 32;                                                           \
        f32x4 fa0[TM], fb0[TN], fa1[TM], fb1[TN];                                                       \
        GEN_FRAG_LOAD(0, fa0, fb0);                                                                     \
        GEN_FRAG_LOAD(1, fa1, fb1);                                                                     \
        GEN_FRAG_MFMA(fa0, fb0);                                                                        \
        if (HWB || (KT) + 1 < kt1) {      /* HWB: unconditional (behind the slice: zeros into the idle buffer) */ \
            GEN_LSTORE((BUF) ^ 1, RA, RB);                                                              \
            GEN_STAGE((KT) + 3, RA, RB);                                                                \
        }                                                                                               \
        GEN_FRAG_LOAD(2, fa0, fb0);                                                                     \
        GEN_FRAG_MFMA(fa1, fb1);                                                                        \
        GEN_FRAG_LOAD(3, fa1, fb1);                                                                     \
        GEN_FRAG_MFMA(fa0, fb0);                                                                        \
        GEN_FRAG_MFMA(fa1, fb1);                                                                        \
        __syncthreads();                                                                                \
    }
    f32x4 ra0[LA], rb0[LB], ra1[LA], rb1[LB];
    if (kt0 < kt1) {
        GEN_GLOAD(kt0, ra0, rb0);
        GEN_STAGE(kt0 + 1, ra1, rb1);
        GEN_LSTORE(0, ra0, rb0);
        GEN_STAGE(kt0 + 2, ra0, rb0);
        __syncthreads();
    }
    // invariant at the top of an even step: lds[0] = tile kt, ra1/rb1 = tile kt+1, ra0/rb0 = tile kt+2
    // (measured and not kept, round 5: four k-tiles per trip so that only one LDS store in four drains both register stages
    // — the compiler's wait-count pass forgets the order of the requests in flight at the loop header — made the 128x64
    // TN products 19 % SLOWER, 677 -> 804 us for the x2h weight gradient)
    for (int kt = kt0; kt < kt1; kt += 2) {
        GEN_ITER(kt, 0, ra1, rb1);
        if (kt + 1 < kt1) GEN_ITER(kt + 1, 1, ra0, rb0);
    }
#undef GEN_ITER
#undef GEN_FRAG_MFMA
#undef GEN_FRAG_LOAD
#undef GEN_FRAG
#undef GEN_LSTORE
#undef GEN_STAGE
#undef GEN_GLOAD
#undef GEN_GLOAD_BUF
#undef GEN_GLOAD_PTR

    // ---- epilogue: C/D map of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    float* Cs = T.C + (long long)ks * T.slab_stride;
    const bool acc_c = T.accumulate && T.ksplit == 1;
    const int crow0 = m0 + wm * TM * 32 + 4 * kh;
    const int ccol0 = n0 + wn * TN * 32 + frow;
    if constexpr (COMBINE) if (T.ksplit > 1 && T.counters) {
        // ---- in-launch combine.  The partial tile goes out write-through (sc1), the workgroup drains its stores and takes a
        // ticket; the last arrival of the tile reads all slabs back with sc1 loads (both sides at the coherence point: no
        // L2 write-back, no invalidate — MI355X_MICROARCH.md, Guideline 16 / splitk-seam) and does slab_reduce_k's job.
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)T.C, 0, 0x7ffffff0, 0x00027000);
        const int sbase = ks * (int)T.slab_stride;                 // floats; slabs of one task stay below 2^29 floats (host)
        float* sT = &lds[0][0] + wave * 1024;
        const int trow = lane >> 3, tcol = (lane & 7) * 4;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + wn * TN * 32 + j * 32 + tcol;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                if (i + j) __syncthreads();
#pragma unroll
                for (int r = 0; r < 16; ++r) sT[((r & 3) + 8 * (r >> 2) + 4 * kh) * 32 + frow] = acc[i][j][r];
                __syncthreads();
                const int rbase = m0 + wm * TM * 32 + i * 32 + trow;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 v4 = *reinterpret_cast<const f32x4*>(sT + q * 256 + lane * 4);
                    const int row = rbase + 8 * q;
                    if (row < T.M && col < T.N)
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v4), rs,
                                                               (sbase + row * (int)T.ldc + col) * 4, 0, 16);
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        unsigned* flag = reinterpret_cast<unsigned*>(&lds[1][0]);
        if (tid == 0) {
            unsigned* cnt = T.counters + tile;
            const unsigned old = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const bool last = old + 1u == (unsigned)T.ksplit;
            if (last) __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *flag = last ? 1u : 0u;
        }
        __syncthreads();
        if (*flag == 0u) return;
        // BM x 64 tile = BM*16 float4, 256 threads
        // read-back in groups of 4 output pieces x 2 slabs (8 loads of 16 bytes in flight per thread: the accumulator and
        // staging registers are dead here, so this fits the k-loop's register budget); slabs in index order, 0 + s0 + s1 + ...
        // = slab_reduce_k's sum bit for bit
        constexpr int PER = BM * BN / 4 / 256, GRP = 4;
        static_assert(PER % GRP == 0, "whole groups");
        const int sstep = (int)T.slab_stride * 4;
#pragma unroll 1
        for (int u0 = 0; u0 < PER; u0 += GRP) {
            int off[GRP];
            long long oo[GRP];
            bool ok[GRP];
            f32x4 v[GRP];
#pragma unroll
            for (int u = 0; u < GRP; ++u) {
                const int f = tid + 256 * (u0 + u), row = m0 + f / (BN / 4), col = n0 + (f % (BN / 4)) * 4;
                ok[u] = row < T.M && col < T.N;
                off[u] = ok[u] ? (row * (int)T.ldc + col) * 4 : 0;
                oo[u] = (long long)row * T.ldo + col;
                v[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
            int sidx = 0;
            for (; sidx + 2 <= T.ksplit; sidx += 2) {
                f32x4 w[2][GRP];
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int u = 0; u < GRP; ++u)
                        w[q][u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off[u] + (sidx + q) * sstep, 0, 16));
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int u = 0; u < GRP; ++u) v[u] += w[q][u];
            }
            if (sidx < T.ksplit) {
                f32x4 w[GRP];
#pragma unroll
                for (int u = 0; u < GRP; ++u)
                    w[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off[u] + sidx * sstep, 0, 16));
#pragma unroll
                for (int u = 0; u < GRP; ++u) v[u] += w[u];
            }
#pragma unroll
            for (int u = 0; u < GRP; ++u) {
                if (!ok[u]) continue;
                float* o = T.out + oo[u];
                if (T.accumulate) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] += v[u][e];
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = v[u][e];
                }
            }
        }
        return;
    }
#include "gemm_gen_epilogue.inc"
}

// (A hand-written k-loop of the 128x64 class was built in round 5 — bit-identical to gemm_gen_f32 and not faster: x2h weight
// gradient 122.7 vs 122.2 TFLOP/s; once the predicated loads were gone — HWB above — the compiler-scheduled loop runs at the
// clock-limited rate inside a round of tiles, what is left is tile quantisation.  EXPERIMENTS.md 5.2; removed in round 6.)

// out[m, n] (+)= sum over slabs, slab 0 first; one launch serves every split task of a group
__global__ void __launch_bounds__(256) slab_reduce_k(const GenLaunch L) {
    int ti = -1;
#pragma unroll
    for (int i = 0; i < GEN_MAX_TASKS; ++i)
        if (i < L.ntasks && L.t[i].ksplit > 1 && (int)blockIdx.x >= L.t[i].red_begin) ti = i;
    if (ti < 0) return;
    const GenTask& T = L.t[ti];
    const int per_row = T.N >> 2;
    const long long idx = (long long)((int)blockIdx.x - T.red_begin) * blockDim.x + threadIdx.x;
    if (idx >= (long long)T.M * per_row) return;
    const long long m = idx / per_row;
    const int j = (int)(idx - m * per_row) << 2;
    const float* p = T.C + m * T.N + j;
    f32x4 v = *reinterpret_cast<const f32x4*>(p);
    for (int s = 1; s < T.ksplit; ++s) v += *reinterpret_cast<const f32x4*>(p + (long long)s * T.slab_stride);
    float* o = T.out + m * T.ldo + j;
    if (T.accumulate) {
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] += v[e];
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = v[e];
    }
}

template <int BM, int BN, bool COMBINE, bool HWB>
static void launch_gen(const GenLaunch& L, int a_kmaj, int b_kmaj, unsigned wgs, hipStream_t s) {
    dim3 grid(wgs), block(256);
    if (a_kmaj && b_kmaj) hipLaunchKernelGGL((gemm_gen_f32<BM, BN, true, true, COMBINE, HWB>), grid, block, 0, s, L);
    else if (a_kmaj) hipLaunchKernelGGL((gemm_gen_f32<BM, BN, true, false, COMBINE, HWB>), grid, block, 0, s, L);
    else if (b_kmaj) hipLaunchKernelGGL((gemm_gen_f32<BM, BN, false, true, COMBINE, HWB>), grid, block, 0, s, L);
    else hipLaunchKernelGGL((gemm_gen_f32<BM, BN, false, false, COMBINE, HWB>), grid, block, 0, s, L);
}

// n independent problems of the same operand layout in ONE launch (+ one reduction launch when any is split):
// the Linear backward of a module computes several dX = dY.W products per timestep that are each too small to
// fill the chip.  All problems must fall into the same row-tile class.
int gemm_gen_group(const SetGemmDesc* d, int n, int a_kminor, int b_kminor, void* ws, size_t ws_bytes, hipStream_t s,
                   SetSlabSrc* slabs_out) {
    if (n <= 0) return SET_OK;
    if (n > GEN_MAX_TASKS || !d) return SET_ERR_ARG;
    static const int bm64_upto = env_int("SET_GEMM_GEN_BM64_UPTO", 512);    // same finding as the forward kernel
    GenLaunch L;
    L.ntasks = n;
    static const int hwb_on = env_int("SET_GEMM_GEN_HWB", 1);
    bool hwb = hwb_on != 0;
    int bm = 0;
    long long tiles[GEN_MAX_TASKS], tiles_total = 0;
    int max_kt = 1;
    for (int i = 0; i < n; ++i) {
        const SetGemmDesc& p = d[i];
        if (p.M <= 0 || p.N <= 0 || p.K <= 0 || !p.A || !p.B || !p.C) return SET_ERR_ARG;
        if (!aligned16(p.A) || !aligned16(p.B) || (p.lda & 3) || (p.ldb & 3)) return SET_ERR_ARG;
        // k-major operands are read in float4 along k; k-minor ones in float4 along their own dimension.  A ragged
        // contraction length (K % 4 != 0, e.g. a 9490-word vocabulary) is accepted when every k-major operand's rows
        // are readable AND ZERO up to the next multiple of 4 (ld >= round_up(K,4), zero padding supplied by the caller):
        // rows k >= K of a k-minor operand are never read.
        if (p.K & 3) {
            const long long K4 = ((long long)p.K + 3) & ~3LL;
            if ((!a_kminor && p.lda < K4) || (!b_kminor && p.ldb < K4)) return SET_ERR_UNSUPPORTED;
        }
        // a k-minor A is read in float4 along M: a ragged M (e.g. the 9490 rows of d fc.weight = dY^T X) is accepted when
        // A's rows are readable up to the next multiple of 4 (lda >= round_up(M, 4)); output rows >= M are never stored
        if (a_kminor && (p.M < 4 || ((p.M & 3) && p.lda < (((long long)p.M + 3) & ~3LL)))) return SET_ERR_UNSUPPORTED;
        if (b_kminor && ((p.N & 3) || p.N < 4)) return SET_ERR_UNSUPPORTED;
        const int b = p.M <= bm64_upto ? 64 : 128;
        if (bm && b != bm) return SET_ERR_ARG;
        bm = b;
        GenTask& T = L.t[i];
        T.A = p.A; T.B = p.B; T.lda = p.lda; T.ldb = p.ldb;
        T.M = p.M; T.N = p.N; T.K = p.K; T.ktiles = cdiv(p.K, GEMM_BK); T.tiles_n = cdiv(p.N, 64);
        T.accumulate = p.accumulate; T.out = p.C; T.ldo = p.ldc;
        if (T.ktiles > max_kt) max_kt = T.ktiles;
        // readable extent of each operand = the range of its buffer resource (HWB kernels).  k-major: M (N) rows of K4 floats;
        // k-minor: K rows, the last one read up to the (4-padded) row / column count.  The furthest offset a load may carry
        // (32 rows past the last k-tile) must stay below 2^31 as well
        const long long K4 = ((long long)p.K + 3) & ~3LL, M4 = ((long long)p.M + 3) & ~3LL, Kt = (long long)T.ktiles * 32 + 64;
        const long long ea = 4 * (a_kminor ? ((long long)(p.K - 1) * p.lda + M4) : ((long long)(p.M - 1) * p.lda + K4));
        const long long eb = 4 * (b_kminor ? ((long long)(p.K - 1) * p.ldb + p.N) : ((long long)(p.N - 1) * p.ldb + K4));
        const long long fa = 4 * (a_kminor ? Kt * p.lda + M4 : (long long)(p.M - 1) * p.lda + Kt);
        const long long fb = 4 * (b_kminor ? Kt * p.ldb + p.N : (long long)(p.N - 1) * p.ldb + Kt);
        if (ea >= 0x7fffffffLL || eb >= 0x7fffffffLL || fa >= 0x7fffffffLL || fb >= 0x7fffffffLL) hwb = false;
        T.ext_a = (unsigned)(ea < 0x7fffffffLL ? ea : 0); T.ext_b = (unsigned)(eb < 0x7fffffffLL ? eb : 0);
    }
    static const int plan_model = env_int("SET_GEMM_GEN_PLAN", 1);
    // Split depth (and, above 512 rows, the row-tile class).  Cost model, measured with tools/bench_gen_split.py:
    // workgroups that share a CU share its MFMA pipe, so a launch of W workgroups of `kper` k-tiles each takes about
    // ceil(W / 256) x (kper + c0) k-tile times — W just above a multiple of the CU count wastes a whole round (304 tiles
    // of the all-timestep fc dX: 2 rounds of 297 k-tiles unsplit, 6 rounds of 60 split five ways: 607 -> 410 us; 576
    // 128-row tiles of the region dX: 3 rounds, as 1152 64-row tiles 5 half-sized ones: 61 -> 53 us) — plus, when
    // anything is split, the reduction launch and one slab write + read per split.  c0 = prologue + epilogue.
    auto plan = [&](int bm_c, int* kper_out) -> double {
        const double unit_us = bm_c == 64 ? 0.57 : 1.08, c0 = 3.0, red_fixed_us = 4.0;
        long long tl[GEN_MAX_TASKS];
        for (int i = 0; i < n; ++i) tl[i] = (long long)cdiv(L.t[i].M, bm_c) * L.t[i].tiles_n;
        double best = 1e30;
        for (int kp = max_kt; kp >= 4; --kp) {
            long long wgs = 0;
            double slab_bytes = 0.0;
            int deepest = 1;
            for (int i = 0; i < n; ++i) {
                const int kt = L.t[i].ktiles;
                int ks = kt < 8 ? 1 : cdiv(kt, kp);
                if (ks > 64) ks = 64;
                wgs += tl[i] * ks;
                if (cdiv(kt, ks) > deepest) deepest = cdiv(kt, ks);
                if (ks > 1) slab_bytes += 4.0 * ks * L.t[i].M * L.t[i].N;
            }
            const long long rounds = (wgs + 255) / 256;
            // one workgroup per CU cannot hide its own load latency (measured ~1.3x per k-tile)
            double us = rounds * (deepest + c0) * unit_us * (rounds == 1 ? 1.3 : 1.0);
            // slabs are written once and read once: from the last-level cache while they fit, else HBM
            if (slab_bytes > 0.0) us += red_fixed_us + 2.0 * slab_bytes / (slab_bytes < 48e6 ? 8.0e6 : 3.0e6);
            if (us < best * 0.98) { best = us; *kper_out = kp; }     // deeper splits must pay for themselves
        }
        return best;
    };
    int kper = max_kt;                                    // = no split
    if (plan_model && !getenv("SET_EXP_GEN_KPER")) {
        int kp64 = max_kt, kp128 = max_kt;
        if (bm == 128) {
            const double us128 = plan(128, &kp128), us64 = plan(64, &kp64);
            if (us64 < us128 * 0.97) bm = 64;
        } else plan(64, &kp64);
        kper = bm == 64 ? kp64 : kp128;
    }
    for (int i = 0; i < n; ++i) {
        tiles[i] = (long long)cdiv(L.t[i].M, bm) * L.t[i].tiles_n;
        tiles_total += tiles[i];
    }
    const int slots = bm == 64 ? 768 : 512;
    if (const char* e = getenv("SET_EXP_GEN_KPER")) {      // experiment knob (tools/bench_gen_split.py)
        kper = atoi(e) > 0 ? atoi(e) : max_kt;
    } else if (plan_model) {
    } else if (tiles_total < slots * 3 / 4) {
        for (kper = 4; kper < max_kt; ++kper) {
            long long wgs = 0;
            for (int i = 0; i < n; ++i) wgs += tiles[i] * cdiv(L.t[i].ktiles, kper);
            if (wgs <= slots) break;
        }
    }
    // SET_GEN_COMBINE=1 (experiment): split products are combined inside the launch by each tile's last workgroup instead of
    // by slab_reduce_k.  The arrival counters live in the last GEN_COUNTER_BYTES of the scratch.
    static const int combine = env_int("SET_GEN_COMBINE", 0);
    constexpr size_t GEN_COUNTER_BYTES = 64 << 10;
    unsigned* counters = nullptr;
    long long counters_used = 0;
    if (combine && ws && ws_bytes > 4 * GEN_COUNTER_BYTES) {
        ws_bytes -= GEN_COUNTER_BYTES;
        counters = reinterpret_cast<unsigned*>((char*)ws + ws_bytes);
        // the arrival counters are cleared on the launch stream before EVERY launch (one 64 KB fill): the scratch is the
        // caller's — it may hold anything, and a launch that was aborted would leave counts behind
        SET_HIP_TRY(hipMemsetAsync(counters, 0, GEN_COUNTER_BYTES, s));
    }
    size_t ws_off = 0;
    int wg = 0, red_blocks = 0;
    double flops = 0.0, bytes = 0.0, red_bytes = 0.0;
    for (int i = 0; i < n; ++i) {
        GenTask& T = L.t[i];
        int ksplit = cdiv(T.ktiles, kper);
        if (ksplit > 64) ksplit = 64;
        const bool vec_ok = !(T.N & 3) && !(T.ldo & 3) && aligned16(T.out) && ws && aligned16(ws);
        const size_t slab = round_up((size_t)T.M * T.N * sizeof(float), 256);
        if (!vec_ok || T.ktiles < 8) ksplit = 1;
        else if (ws_off + (size_t)ksplit * slab > ws_bytes) ksplit = (int)((ws_bytes - ws_off) / slab);
        if (ksplit < 2) ksplit = 1;
        T.ksplit = ksplit;
        T.red_begin = 0x7fffffff;
        T.counters = nullptr;
        if (ksplit == 1) { T.C = T.out; T.ldc = T.ldo; T.slab_stride = 0; }
        else {
            T.C = (float*)((char*)ws + ws_off); T.ldc = T.N; T.slab_stride = (long long)(slab / sizeof(float));
            ws_off += (size_t)ksplit * slab;
            if (slabs_out) {
                // (the consumer reduces: set_gemm_group_slabs_f32)
            } else if (counters && (counters_used + tiles[i]) * sizeof(unsigned) <= GEN_COUNTER_BYTES &&
                (size_t)ksplit * slab < ((size_t)1 << 31)) {
                T.counters = counters + counters_used;
                counters_used += tiles[i];
            } else {
                T.red_begin = red_blocks;
                red_blocks += (int)(((long long)T.M * (T.N >> 2) + 255) / 256);
                red_bytes += 4.0 * T.M * T.N * (ksplit + 1.0);
            }
        }
        if (slabs_out) {
            SetSlabSrc& o = slabs_out[i];
            o.p = ksplit > 1 ? T.C : nullptr; o.slab_stride = T.slab_stride; o.ld = T.ldc; o.nslab = ksplit > 1 ? ksplit : 0;
            o.rows = T.M;
        }
        static const int vec_epi = env_int("SET_GEMM_VEC_EPILOGUE", 1);
        T.vec_store = vec_epi && !(T.N & 3) && !(T.ldc & 3) && !(T.slab_stride & 3) && aligned16(T.C);
        T.wg_begin = wg;
        wg += (int)(tiles[i] * ksplit);
        flops += 2.0 * T.M * T.N * (double)T.K;
        bytes += 4.0 * ((double)T.M * T.K + (double)T.N * T.K + (double)T.M * T.N * ksplit);
    }
    for (int i = n; i < GEN_MAX_TASKS; ++i) { L.t[i] = L.t[0]; L.t[i].wg_begin = 0x7fffffff; L.t[i].ksplit = 1; }
    {
        const char* name = a_kminor ? (b_kminor ? "gemm_gen_f32<tn>" : "gemm_gen_f32<tt>")
                                    : (b_kminor ? "gemm_gen_f32<nn>" : "gemm_gen_f32<nt>");
        ProfScope ps(name, s, flops, bytes);
        if (counters_used > 0) {
            if (bm == 128) launch_gen<128, 64, true, false>(L, !a_kminor, !b_kminor, (unsigned)wg, s);
            else launch_gen<64, 64, true, false>(L, !a_kminor, !b_kminor, (unsigned)wg, s);
        } else if (hwb) {
            if (bm == 128) launch_gen<128, 64, false, true>(L, !a_kminor, !b_kminor, (unsigned)wg, s);
            else launch_gen<64, 64, false, true>(L, !a_kminor, !b_kminor, (unsigned)wg, s);
        } else {
            if (bm == 128) launch_gen<128, 64, false, false>(L, !a_kminor, !b_kminor, (unsigned)wg, s);
            else launch_gen<64, 64, false, false>(L, !a_kminor, !b_kminor, (unsigned)wg, s);
        }
        SET_LAUNCH_CHECK();
    }
    if (red_blocks > 0) {
        ProfScope ps("slab_reduce", s, 0.0, red_bytes);
        hipLaunchKernelGGL(slab_reduce_k, dim3((unsigned)red_blocks), dim3(256), 0, s, L);
        SET_LAUNCH_CHECK();
    }
    return SET_OK;
}

int gemm_gen(const float* A, long long lda, int a_kminor, const float* B, long long ldb, int b_kminor, float* C,
             long long ldc, int M, int N, int K, int accumulate, void* ws, size_t ws_bytes, hipStream_t s) {
    if (M <= 0 || N <= 0) return SET_OK;
    SetGemmDesc d{A, lda, B, ldb, C, ldc, M, N, K, accumulate};
    return gemm_gen_group(&d, 1, a_kminor, b_kminor, ws, ws_bytes, s);
}

}  // namespace set
