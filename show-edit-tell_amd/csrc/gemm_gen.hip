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

struct GenTask {
    const float* A;
    const float* B;
    float* C;            // C itself (ksplit == 1) or the slab base
    long long lda, ldb, ldc, slab_stride;
    int M, N, K, ktiles, ksplit, tiles_n, accumulate;
};

template <int BM, int BN, bool A_KMAJ, bool B_KMAJ>
__global__ void __launch_bounds__(256) gemm_gen_f32(const GenTask T) {
    constexpr int TM = BM / 64, TN = BN / 64;            // 2x2 waves
    constexpr int LA = BM / 32, LB = BN / 32;            // float4 loads per thread per k-tile
    __shared__ __attribute__((aligned(16))) float lds[2][(BM + BN) * 32];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int ks = (int)blockIdx.x % T.ksplit;
    const int tile = (int)blockIdx.x / T.ksplit;
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
    int ka[LA], kb[LB];              // the k index this thread's load i covers inside a k-tile
    int sa[LA], sb[LB];              // LDS float offset of the store
    long long stepa, stepb;          // pointer advance per k-tile
#pragma unroll
    for (int i = 0; i < LA; ++i) {
        if constexpr (A_KMAJ) {
            int r = m0 + srow + 32 * i; r = r < T.M ? r : T.M - 1;
            pa[i] = T.A + (long long)r * T.lda + scol;
            ka[i] = scol; sa[i] = (srow + 32 * i) * 32 + sswz;
        } else {
            const int f = tid + 256 * i, kr = f / (BM / 4), c4 = (f % (BM / 4)) * 4;
            int c = m0 + c4; c = c + 4 <= T.M ? c : T.M - 4;
            pa[i] = T.A + (long long)kr * T.lda + c;
            ka[i] = kr; sa[i] = kr * BM + c4;
        }
    }
#pragma unroll
    for (int i = 0; i < LB; ++i) {
        if constexpr (B_KMAJ) {
            int r = n0 + srow + 32 * i; r = r < T.N ? r : T.N - 1;
            pb[i] = T.B + (long long)r * T.ldb + scol;
            kb[i] = scol; sb[i] = (srow + 32 * i) * 32 + sswz;
        } else {
            const int f = tid + 256 * i, kr = f / (BN / 4), c4 = (f % (BN / 4)) * 4;
            int c = n0 + c4; c = c + 4 <= T.N ? c : T.N - 4;
            pb[i] = T.B + (long long)kr * T.ldb + c;
            kb[i] = kr; sb[i] = kr * BN + c4;
        }
    }
    stepa = A_KMAJ ? 32 : 32 * T.lda;
    stepb = B_KMAJ ? 32 : 32 * T.ldb;
#pragma unroll
    for (int i = 0; i < LA; ++i) pa[i] += (long long)kt0 * stepa;
#pragma unroll
    for (int i = 0; i < LB; ++i) pb[i] += (long long)kt0 * stepb;

    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    // loads of k-tile KT; contraction indices >= K contribute zeros (K need not be a multiple of 32)
#define GEN_GLOAD(KT, RA, RB)                                                                           \
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
#define GEN_STAGE(KT, RA, RB) if ((KT) < kt1) GEN_GLOAD(KT, RA, RB)
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
        if ((KT) + 1 < kt1) {                                                                           \
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

    // ---- epilogue: C/D map of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    float* Cs = T.C + (long long)ks * T.slab_stride;
    const bool acc_c = T.accumulate && T.ksplit == 1;
    const int crow0 = m0 + wm * TM * 32 + 4 * kh;
    const int ccol0 = n0 + wn * TN * 32 + frow;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = ccol0 + j * 32;
        if (col >= T.N) continue;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = crow0 + i * 32 + (r & 3) + 8 * (r >> 2);
                if (row < T.M) {
                    float* p = Cs + (long long)row * T.ldc + col;
                    *p = acc_c ? *p + acc[i][j][r] : acc[i][j][r];
                }
            }
        }
    }
}

// out[m, n] (+)= sum over slabs, slab 0 first
__global__ void __launch_bounds__(256) slab_reduce_k(const float* slabs, long long slab_stride, int nslab, float* out,
                                                     long long ldo, int M, int N, int accumulate) {
    const int per_row = N >> 2;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)M * per_row) return;
    const long long m = idx / per_row;
    const int j = (int)(idx - m * per_row) << 2;
    const float* p = slabs + m * N + j;
    f32x4 v = *reinterpret_cast<const f32x4*>(p);
    for (int s = 1; s < nslab; ++s) v += *reinterpret_cast<const f32x4*>(p + (long long)s * slab_stride);
    float* o = out + m * ldo + j;
    if (accumulate) {
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] += v[e];
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = v[e];
    }
}

template <int BM, int BN>
static void launch_gen(const GenTask& T, int a_kmaj, int b_kmaj, unsigned wgs, hipStream_t s) {
    dim3 grid(wgs), block(256);
    if (a_kmaj && b_kmaj) hipLaunchKernelGGL((gemm_gen_f32<BM, BN, true, true>), grid, block, 0, s, T);
    else if (a_kmaj) hipLaunchKernelGGL((gemm_gen_f32<BM, BN, true, false>), grid, block, 0, s, T);
    else if (b_kmaj) hipLaunchKernelGGL((gemm_gen_f32<BM, BN, false, true>), grid, block, 0, s, T);
    else hipLaunchKernelGGL((gemm_gen_f32<BM, BN, false, false>), grid, block, 0, s, T);
}

int gemm_gen(const float* A, long long lda, int a_kminor, const float* B, long long ldb, int b_kminor, float* C,
             long long ldc, int M, int N, int K, int accumulate, void* ws, size_t ws_bytes, hipStream_t s) {
    if (M <= 0 || N <= 0) return SET_OK;
    if (K <= 0 || !A || !B || !C) return SET_ERR_ARG;
    if (!aligned16(A) || !aligned16(B) || (lda & 3) || (ldb & 3)) return SET_ERR_ARG;
    // k-major operands are read in float4 along k; k-minor ones in float4 along their own dimension
    if ((!a_kminor || !b_kminor) && (K & 3)) return SET_ERR_UNSUPPORTED;
    if (a_kminor && ((M & 3) || M < 4)) return SET_ERR_UNSUPPORTED;
    if (b_kminor && ((N & 3) || N < 4)) return SET_ERR_UNSUPPORTED;
    static const int bm64_upto = env_int("SET_GEMM_GEN_BM64_UPTO", 128);    // same finding as the forward kernel
    const int bm = M <= bm64_upto ? 64 : 128, bn = 64;
    const int tiles_m = cdiv(M, bm), tiles_n = cdiv(N, bn);
    const long long tiles = (long long)tiles_m * tiles_n;
    const int ktiles = cdiv(K, GEMM_BK);
    // split the contraction only when the output alone cannot fill the chip (512 workgroup slots)
    int ksplit = 1;
    const int slots = bm == 64 ? 768 : 512;              // workgroups that are resident at once
    if (tiles < slots * 3 / 4 && ktiles >= 8) {
        ksplit = (int)(slots / tiles);
        if (ksplit > ktiles / 4) ksplit = ktiles / 4;
        if (ksplit > 64) ksplit = 64;
        const bool vec_ok = !(N & 3) && !(ldc & 3) && aligned16(C) && ws && aligned16(ws);
        const size_t slab = (size_t)M * N * sizeof(float);
        if (!vec_ok) ksplit = 1;
        else if ((size_t)ksplit * slab > ws_bytes) ksplit = (int)(ws_bytes / slab);
        if (ksplit < 2) ksplit = 1;
    }
    GenTask T;
    T.A = A; T.B = B; T.lda = lda; T.ldb = ldb;
    T.M = M; T.N = N; T.K = K; T.ktiles = ktiles; T.ksplit = ksplit; T.tiles_n = tiles_n;
    T.accumulate = accumulate;
    if (ksplit == 1) { T.C = C; T.ldc = ldc; T.slab_stride = 0; }
    else { T.C = (float*)ws; T.ldc = N; T.slab_stride = (long long)M * N; }
    const double flops = 2.0 * M * N * (double)K;
    const double bytes = 4.0 * ((double)M * K + (double)N * K + (double)M * N * ksplit);
    {
        const char* name = a_kminor ? (b_kminor ? "gemm_gen_f32<tn>" : "gemm_gen_f32<tt>")
                                    : (b_kminor ? "gemm_gen_f32<nn>" : "gemm_gen_f32<nt>");
        ProfScope ps(name, s, flops, bytes);
        const unsigned wgs = (unsigned)(tiles * ksplit);
        if (bm == 128) launch_gen<128, 64>(T, !a_kminor, !b_kminor, wgs, s);
        else launch_gen<64, 64>(T, !a_kminor, !b_kminor, wgs, s);
        SET_LAUNCH_CHECK();
    }
    if (ksplit > 1) {
        ProfScope ps("slab_reduce", s, 0.0, 4.0 * M * N * (ksplit + 1.0));
        const long long n = (long long)M * (N >> 2);
        hipLaunchKernelGGL(slab_reduce_k, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const float*)ws,
                           (long long)M * N, ksplit, C, ldc, M, N, accumulate);
        SET_LAUNCH_CHECK();
    }
    return SET_OK;
}

}  // namespace set
