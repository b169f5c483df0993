// Small-tile fp32-MFMA GEMM with a fused pointwise epilogue, for the contractions of the step that
// are too small to amortise a split-K launch + a separate reduction kernel (K = 1024, N = 1024):
//   * context gating   zt = sig(W_g[word,h1,ctx]), out = zt*tanh(W_sc ctx) + (1-zt)*tanh(W_tc[word,h1])
//                      (editnet.py:378-380; the [word,h1] columns arrive as slabs from phase B)
//   * copy gate        copy = sig(W_n c_new + W_m sel), c2, h2            (editnet.py:281-283)
//   * encoder LSTM     gates = h W_hh^T + (hoisted x W_xh^T + b), cell update, H/M stores
//                      (editnet.py:333-338; nn.LSTM directions of dcnet.py:233)
// One workgroup owns a 32(rows) x 32(columns) output tile of up to two accumulators over the FULL K,
// so no slabs are written and the epilogue runs in the same launch.  The 4 waves split every
// BK-wide k-tile between them (intra-workgroup split-K: wave w takes 8-wide k-blocks w, w+4, ...),
// operands are staged coalesced through LDS (row stride BK+4: conflict-free b128 access), the
// partial accumulators are combined through LDS in fixed wave order (deterministic).
#include "set_common.h"

namespace set {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef const f32x4 __attribute__((address_space(1)))* gptr4;

enum { EPI_CTXGATE = 0, EPI_COPYGATE = 1, EPI_ENCLSTM = 2, EPI_COPYGATE1 = 3 };

struct FusedArgs {
    // accumulator a = A[a] (M,K) x W[a] (N rows, K)^T ; both accumulators share K
    const float* A[2];
    const float* W[2];
    long long lda[2], ldw[2];
    int K, M, N;          // N = output columns (hidden units for ENCLSTM)
    int gate_stride;      // ENCLSTM: W row of (gate q, unit u) = q*gate_stride + u ; tile = 8 units x 4 gates
    // epilogue operands
    Slabs s0, s1;         // CTXGATE: s0 = context_gate[word,h1] slabs, s1 = tc_affine slabs
    const float *b0, *b1, *b2;          // CTXGATE: b_gate, b_sc, b_tc ; COPYGATE: b_cnew, b_cmem ; ENCLSTM: b_extra
    const float *e0, *e1, *e2;          // COPYGATE: c_new, sel, ogate ; ENCLSTM: xg, h_in, (unused)
    const float* e3;                    // COPYGATE1: the hoisted gate_cmem(sel) product (M,N), its GEMM is not run here
    float *o0, *o1, *o2, *o3;           // CTXGATE: out ; COPYGATE: c2, h2 ; ENCLSTM: h_out, c (in place), H, Mem
    const int64_t* lens;                // ENCLSTM
    long long ld_xg_row, ld_xg_t, ld_out_b, ld_out_t;
    int t, reverse, out_col0;
    RowGather g0, g1;                   // CTXGATE: token-table addends of z and of the tc_affine term
    const int64_t* seq; int seq_T, seq_V; // ENCLSTM with a token table: xg row = e0 + seq[m*seq_T + pos] * ld_xg_row (V rows)
    // ENCLSTM, optional: rows visited in order of decreasing length (perm[sorted position] = row) so that whole 32-row
    // tiles whose rows have all finished (sorted position >= nactive[t]) skip the contraction (the reference shrinks
    // its length-sorted batch prefix the same way, editnet.py:333-335)
    const int* perm; const int* nactive;
    // ENCLSTM in a grad-enabled forward (gates != nullptr): the cell state is read from c_in and written to o1 (the
    // backward needs both), the post-activation gates (i, f, g, o) go to gates[b, q*N + unit], the incoming h to hprev
    // (same indexing as H), and finished rows store zeros to H / Mem / gates and carry h and c
    const float* c_in; float* gates; float* hprev;
    // decode loops (set_common.h RowGate): COPYGATE*: `alive_prev` = the loop-left test
    const int* alive_prev;
};

__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + expf(-x)); }

// NW = waves of the workgroup (4, or 8: the same tile with every k-tile split eight ways — the workgroup's serial k-loop is
// what bounds these launches (1.15 us per 128-wide k-tile at one or two workgroups per CU), twice the waves halve it; the
// epilogue is run by the first 256 threads)
// ST = register stages of the global -> LDS pipeline (2, or 4: tiles kt + 1 .. kt + 4 in registers while tile kt is
// contracted from LDS — a k-tile takes about half a memory latency with two, the loads of a workgroup are what it waits for)
template <int NACC, bool SHARED_A, int BK, int EPI, int NW = 4, int ST = 2>
__global__ void __launch_bounds__(64 * NW) gemm_fused_k(const FusedArgs P) {
    constexpr int NT = 64 * NW;
    static_assert(ST == 2 || ST == 4, "two or four register stages");
    constexpr int STRIDE = BK + 4;
    constexpr int NA = SHARED_A ? 1 : NACC;
    constexpr int TILE = 32 * STRIDE;                       // floats per staged operand tile
    constexpr int STAGE = (NA + NACC) * TILE;
    constexpr int LPT = 8 * BK / NT;                        // float4 loads per thread per operand tile
    constexpr int RED = NW * NACC * 32 * 33;                // cross-wave reduction scratch
    static_assert(LPT >= 1 && BK % (8 * NW) == 0, "tile / wave count mismatch");
    extern __shared__ __attribute__((aligned(16))) float lds[];   // fused_lds_bytes<...>() bytes

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // ENCLSTM: accumulator x holds the 4 gates of the 8 units n0 + 8 x .. (NACC = 2: 16 units per workgroup, the activation
    // tile is staged once for both)
    constexpr int EU = 8 * NACC;
    const int tiles_n = (EPI == EPI_ENCLSTM) ? (P.N + EU - 1) / EU : (P.N + 31) / 32;
    const int tn = blockIdx.x % tiles_n, tm = blockIdx.x / tiles_n;
    const int m0 = tm * 32;
    const int n0 = (EPI == EPI_ENCLSTM) ? tn * EU : tn * 32;
    constexpr bool ROWLIST = (EPI == EPI_ENCLSTM);          // rows visited through P.perm
    const int Meff = P.M;

    // staging: thread -> row (tid / (BK/4)) .. covers 32 rows x BK floats with LPT float4 per thread
    constexpr int TPR = BK / 4;                             // threads per row
    constexpr int RPP = NT / TPR;                           // rows per pass
    const int srow = tid / TPR, scol = (tid % TPR) * 4;
    gptr4 pa[NA][LPT];
    gptr4 pw[NACC][LPT];
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int i = 0; i < LPT; ++i) {
            int r = m0 + srow + RPP * i;
            r = r < Meff ? r : Meff - 1;
            if (ROWLIST && P.perm) r = P.perm[r];
            pa[a][i] = (gptr4)(P.A[a] + (long long)r * P.lda[a] + scol);
        }
#pragma unroll
    for (int a = 0; a < NACC; ++a)
#pragma unroll
        for (int i = 0; i < LPT; ++i) {
            const int j = srow + RPP * i;                   // tile column 0..31
            long long wr;
            if (EPI == EPI_ENCLSTM) {
                int u = n0 + 8 * a + (j & 7);
                u = u < P.N ? u : P.N - 1;
                wr = (long long)(j >> 3) * P.gate_stride + u;
                pw[a][i] = (gptr4)(P.W[0] + wr * P.ldw[0] + scol);     // (one weight matrix for every accumulator)
                continue;
            } else {
                int c = n0 + j;
                wr = c < P.N ? c : P.N - 1;
            }
            pw[a][i] = (gptr4)(P.W[a] + wr * P.ldw[a] + scol);
        }

    f32x4 ra0[NA][LPT], rw0[NACC][LPT], ra1[NA][LPT], rw1[NACC][LPT];   // two register stages
    f32x4 ra2[ST == 4 ? NA : 1][LPT], rw2[ST == 4 ? NACC : 1][LPT], ra3[ST == 4 ? NA : 1][LPT], rw3[ST == 4 ? NACC : 1][LPT];
#define FS_GLOAD(RA, RW)                                                                                \
    {                                                                                                   \
        _Pragma("unroll") for (int a = 0; a < NA; ++a)                                                  \
            _Pragma("unroll") for (int i = 0; i < LPT; ++i) { RA[a][i] = *pa[a][i]; pa[a][i] += BK / 4; } \
        _Pragma("unroll") for (int a = 0; a < NACC; ++a)                                                \
            _Pragma("unroll") for (int i = 0; i < LPT; ++i) { RW[a][i] = *pw[a][i]; pw[a][i] += BK / 4; } \
    }
#define FS_LSTORE(BUF, RA, RW)                                                                          \
    {                                                                                                   \
        float* base_ = lds + (BUF) * STAGE;                                                             \
        _Pragma("unroll") for (int a = 0; a < NA; ++a)                                                  \
            _Pragma("unroll") for (int i = 0; i < LPT; ++i)                                             \
                *reinterpret_cast<f32x4*>(base_ + a * TILE + (srow + RPP * i) * STRIDE + scol) = RA[a][i]; \
        _Pragma("unroll") for (int a = 0; a < NACC; ++a)                                                \
            _Pragma("unroll") for (int i = 0; i < LPT; ++i)                                             \
                *reinterpret_cast<f32x4*>(base_ + (NA + a) * TILE + (srow + RPP * i) * STRIDE + scol) = RW[a][i]; \
    }

    f32x16 acc[NACC];
#pragma unroll
    for (int a = 0; a < NACC; ++a)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;

#ifdef SET_EXP_ENC_NOK
    const int nkt = (EPI == EPI_ENCLSTM) ? SET_EXP_ENC_NOK : P.K / BK;     // diagnostic: truncated contraction
#else
    const int nkt = P.K / BK;
#endif
    const int frow = lane & 31, fk = (lane >> 5) * 4;
    FS_GLOAD(ra0, rw0);                                  // tile 0
    if (nkt > 1) FS_GLOAD(ra1, rw1);                     // tile 1
    if constexpr (ST == 4) {
        if (nkt > 2) FS_GLOAD(ra2, rw2);                 // tile 2
        if (nkt > 3) FS_GLOAD(ra3, rw3);                 // tile 3
    }

    // ---- epilogue operands do not depend on the contraction: fetch them now, under the k-loop
    const int erow = (tid & 255) >> 3, ec4 = (tid & 7) * 4, eu = tid & 7;
    const bool erow_ok = tid < 256 && m0 + erow < Meff;
    const long long em = (ROWLIST && P.perm) ? (long long)P.perm[erow_ok ? m0 + erow : Meff - 1] : (long long)(m0 + erow);
    if (EPI == EPI_ENCLSTM && P.nactive && m0 >= P.nactive[P.t]) {
        // every row of this tile has finished (rows are visited longest first): carry the state, skip the contraction
#pragma unroll
        for (int x = 0; x < NACC; ++x) {
            const int unit = n0 + 8 * x + eu;
            if (erow_ok && unit < P.N) P.o0[em * P.N + unit] = P.e1[em * P.N + unit];
        }
        return;
    }
    f32x4 pre0 = {0.f, 0.f, 0.f, 0.f}, pre1 = {0.f, 0.f, 0.f, 0.f}, pre2 = {0.f, 0.f, 0.f, 0.f}, pre3 = {0.f, 0.f, 0.f, 0.f};
    float eg[NACC][4], ecp[NACC], ehin[NACC];
#pragma unroll
    for (int x = 0; x < NACC; ++x) { eg[x][0] = eg[x][1] = eg[x][2] = eg[x][3] = 0.f; ecp[x] = 0.f; ehin[x] = 0.f; }
    int elen = 0, epos = 0;
    if (EPI == EPI_CTXGATE) {
        if (erow_ok && n0 + ec4 < P.N) {                 // N % 4 == 0 (host check)
            if (P.s0.n > 0) pre0 = slab_sum4_at(P.s0, em * P.s0.ld + n0 + ec4);
            if (P.s1.n > 0) pre1 = slab_sum4_at(P.s1, em * P.s1.ld + n0 + ec4);
            if (P.g0.tab) pre0 += *(gptr4)(P.g0.row(em) + n0 + ec4);
            if (P.g1.tab) pre1 += *(gptr4)(P.g1.row(em) + n0 + ec4);
        }
    } else if (EPI == EPI_COPYGATE || EPI == EPI_COPYGATE1) {
        if (erow_ok && n0 + ec4 < P.N) {
            pre0 = *(gptr4)(P.e0 + em * P.N + n0 + ec4);
            pre1 = *(gptr4)(P.e1 + em * P.N + n0 + ec4);
            pre2 = *(gptr4)(P.e2 + em * P.N + n0 + ec4);
            if (EPI == EPI_COPYGATE1) pre3 = *(gptr4)(P.e3 + em * P.N + n0 + ec4);
        }
    } else {
        // The chain row -> length -> word -> table row is three dependent round trips: only its head is requested here, the
        // word after the first k-tile and the table row after the second (ENC_STAGE1 / ENC_STAGE2 below) — requested ahead of
        // the contraction, a wave waits for each link before it stages its first tile (4 of the step's 11 fixed microseconds)
        if (erow_ok) {
            elen = (int)P.lens[em];
#pragma unroll
            for (int x = 0; x < NACC; ++x) {
                const int unit = n0 + 8 * x + eu;
                if (unit >= P.N) continue;
                ehin[x] = P.e1[em * P.N + unit];
                ecp[x] = (P.gates ? P.c_in : P.o1)[em * P.N + unit];
            }
        }
    }
    long long etok = 0;
    float eb[NACC][4];
#define ENC_STAGE1()                                                                                    \
    if (EPI == EPI_ENCLSTM && erow_ok) {                                                                \
        if (P.t < elen) epos = P.reverse ? (elen - 1 - P.t) : P.t;                                      \
        etok = (P.seq && P.t < elen) ? P.seq[em * P.seq_T + epos] : 0;                                  \
        _Pragma("unroll") for (int x = 0; x < NACC; ++x)                                                \
            _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                             \
                const int unit = n0 + 8 * x + eu;                                                       \
                eb[x][q] = (P.b0 && unit < P.N) ? P.b0[q * P.N + unit] : 0.f;                           \
            }                                                                                           \
    }
#define ENC_STAGE2()                                                                                    \
    if (EPI == EPI_ENCLSTM && erow_ok && P.t < elen) {                                                  \
        long long tok = etok < 0 ? 0 : (etok >= P.seq_V ? P.seq_V - 1 : etok);    /* same clamp as embed_relu_k */ \
        const float* xr = P.seq ? P.e0 + tok * P.ld_xg_row : P.e0 + em * P.ld_xg_row + (long long)epos * P.ld_xg_t; \
        _Pragma("unroll") for (int x = 0; x < NACC; ++x) {                                              \
            const int unit = n0 + 8 * x + eu;                                                           \
            if (unit < P.N) {                                                                           \
                _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                         \
                    eg[x][q] = xr[q * P.N + unit];                                                      \
                    if (P.b0) eg[x][q] += eb[x][q];                                                     \
                }                                                                                       \
            }                                                                                           \
        }                                                                                               \
    }

    FS_LSTORE(0, ra0, rw0);
    if constexpr (ST == 4) { if (nkt > 4) FS_GLOAD(ra0, rw0); }          // tile 4
    else { if (nkt > 2) FS_GLOAD(ra0, rw0); }                            // tile 2
    __syncthreads();
    // invariant at an even step kt: lds[0] = tile kt, ra1 = tile kt+1, ra0 = tile kt+2
#define FS_ITER(KT, BUF, RA, RW)                                                                        \
    {                                                                                                   \
        const float* sb = lds + (BUF) * STAGE;                                                          \
        _Pragma("unroll") for (int kb = 0; kb < BK / (8 * NW); ++kb) {                                  \
            const int koff = (wave + NW * kb) * 8 + fk;                                                 \
            f32x4 a[NA], b[NACC];                                                                       \
            _Pragma("unroll") for (int x = 0; x < NA; ++x)                                              \
                a[x] = *reinterpret_cast<const f32x4*>(sb + x * TILE + frow * STRIDE + koff);           \
            _Pragma("unroll") for (int x = 0; x < NACC; ++x)                                            \
                b[x] = *reinterpret_cast<const f32x4*>(sb + (NA + x) * TILE + frow * STRIDE + koff);    \
            _Pragma("unroll") for (int x = 0; x < NACC; ++x) {                                          \
                const f32x4 av = a[SHARED_A ? 0 : x];                                                   \
                acc[x] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, b[x].x, acc[x], 0, 0, 0);           \
                acc[x] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, b[x].y, acc[x], 0, 0, 0);           \
                acc[x] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, b[x].z, acc[x], 0, 0, 0);           \
                acc[x] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, b[x].w, acc[x], 0, 0, 0);           \
            }                                                                                           \
            if (kb == 0 && (KT) + 1 < nkt) {                                                            \
                FS_LSTORE((BUF) ^ 1, RA, RW);                                                           \
                if ((KT) + ST + 1 < nkt) FS_GLOAD(RA, RW);                                              \
            }                                                                                           \
        }                                                                                               \
        __syncthreads();                                                                                \
    }
    if constexpr (ST == 4) {
        // invariant at kt % 4 == 0: lds[0] = tile kt, ra1 .. ra3 = tiles kt + 1 .. kt + 3, ra0 = tile kt + 4
        for (int kt = 0; kt < nkt; kt += 4) {
            FS_ITER(kt, 0, ra1, rw1);
            if (kt == 0) { ENC_STAGE1(); if (nkt < 2) { ENC_STAGE2(); } }
            if (kt + 1 < nkt) FS_ITER(kt + 1, 1, ra2, rw2);
            if (kt == 0 && nkt >= 2) { ENC_STAGE2(); }
            if (kt + 2 < nkt) FS_ITER(kt + 2, 0, ra3, rw3);
            if (kt + 3 < nkt) FS_ITER(kt + 3, 1, ra0, rw0);
        }
    } else {
        for (int kt = 0; kt < nkt; kt += 2) {
            FS_ITER(kt, 0, ra1, rw1);
            if (kt == 0) { ENC_STAGE1(); if (nkt < 2) { ENC_STAGE2(); } }
            if (kt + 1 < nkt) FS_ITER(kt + 1, 1, ra0, rw0);
            if (kt == 0 && nkt >= 2) { ENC_STAGE2(); }
        }
    }
#undef ENC_STAGE1
#undef ENC_STAGE2
#undef FS_ITER
#undef FS_GLOAD
#undef FS_LSTORE

    // ---- cross-wave reduction through LDS: red[wave][acc][row][33]
    float* red = lds;
#pragma unroll
    for (int x = 0; x < NACC; ++x)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), col = lane & 31;
            red[((wave * NACC + x) * 32 + row) * 33 + col] = acc[x][r];
        }
    __syncthreads();
    auto rsum = [&](int x, int row, int col) {
        float v = red[((0 * NACC + x) * 32 + row) * 33 + col];
#pragma unroll
        for (int wv = 1; wv < NW; ++wv) v += red[((wv * NACC + x) * 32 + row) * 33 + col];
        return v;
    };

    if (EPI == EPI_ENCLSTM) {
        // thread -> (row = tid/8, unit = tid%8 of every accumulator); tile columns q*8+u hold gate q of unit u
#pragma unroll
        for (int x = 0; x < NACC; ++x) {
            const int unit = n0 + 8 * x + eu;
            if (!erow_ok || unit >= P.N) continue;
            const int D = P.N;
            if (P.t < elen) {
                float g[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) g[q] = rsum(x, erow, q * 8 + eu) + eg[x][q];
                const float ai = sigm(g[0]), af = sigm(g[1]), ag = tanhf(g[2]), ao = sigm(g[3]);
                const float cn = af * ecp[x] + ai * ag;
                const float hn = ao * tanhf(cn);
                P.o1[em * D + unit] = cn;
                P.o0[em * D + unit] = hn;
                P.o2[em * P.ld_out_b + (long long)epos * P.ld_out_t + P.out_col0 + unit] = hn;
                if (P.o3) P.o3[em * P.ld_out_b + (long long)epos * P.ld_out_t + P.out_col0 + unit] = cn;
                if (P.gates) {
                    float* gr = P.gates + em * 4 * D + unit;
                    gr[0] = ai; gr[D] = af; gr[2 * D] = ag; gr[3 * D] = ao;
                }
            } else {
                P.o0[em * D + unit] = ehin[x];               // finished rows carry their state
                if (P.gates) {                               // grad-enabled forward: position t of a finished row
                    const long long o = em * P.ld_out_b + (long long)P.t * P.ld_out_t + P.out_col0 + unit;
                    P.o1[em * D + unit] = ecp[x];
                    P.o2[o] = 0.f;
                    if (P.o3) P.o3[o] = 0.f;
                    float* gr = P.gates + em * 4 * D + unit;
                    gr[0] = 0.f; gr[D] = 0.f; gr[2 * D] = 0.f; gr[3 * D] = 0.f;
                }
            }
            if (P.hprev) P.hprev[em * P.ld_out_b + (long long)P.t * P.ld_out_t + P.out_col0 + unit] = ehin[x];
        }
        return;
    }

    // CTXGATE / COPYGATE: thread -> (row = tid/8, 4 consecutive columns)
    if (!erow_ok || n0 + ec4 >= P.N) return;
    f32x4 out0, out1;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int n = n0 + ec4 + e;
        if (EPI == EPI_CTXGATE) {
            // reference: one Linear over cat([word, h1, ctx]); [word,h1] slabs first, then the ctx part
            const float z = (pre0[e] + rsum(0, erow, ec4 + e)) + P.b0[n];
            const float sv = rsum(1, erow, ec4 + e) + P.b1[n];
            const float tt = pre1[e] + P.b2[n];
            const float zt = sigm(z);
            out0[e] = zt * tanhf(sv) + (1.f - zt) * tanhf(tt);
        } else {
            // reference order (editnet.py:281): (gate_cnew(c_new) + b) + (gate_cmem(c_memory) + b)
            const float a = rsum(0, erow, ec4 + e) + P.b0[n];
            const float b = (EPI == EPI_COPYGATE1 ? pre3[e] : rsum(NACC - 1, erow, ec4 + e)) + P.b1[n];
            const float cg = sigm(a + b);
            const float co = cg * pre1[e] + (1.f - cg) * pre0[e];
            out0[e] = co;
            out1[e] = pre2[e] * tanhf(co);
        }
    }
    *reinterpret_cast<f32x4*>(P.o0 + em * P.N + n0 + ec4) = out0;
    if (EPI == EPI_COPYGATE || EPI == EPI_COPYGATE1) *reinterpret_cast<f32x4*>(P.o1 + em * P.N + n0 + ec4) = out1;
}

template <int NACC, bool SHARED_A, int BK, int NW = 4>
constexpr int fused_lds_bytes() {
    constexpr int stage = ((SHARED_A ? 1 : NACC) + NACC) * 32 * (BK + 4);
    constexpr int red = NW * NACC * 32 * 33;
    return 4 * ((2 * stage > red) ? 2 * stage : red);
}

template <int NACC, bool SHARED_A, int BK, int EPI, int NW = 4, int ST = 2>
static int launch_fused(const FusedArgs& P, int grid, hipStream_t s) {
    constexpr int bytes = fused_lds_bytes<NACC, SHARED_A, BK, NW>();
    static bool configured = false;          // raise the dynamic-LDS cap once (idempotent)
    if (!configured) {
        SET_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_fused_k<NACC, SHARED_A, BK, EPI, NW, ST>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
        configured = true;
    }
    hipLaunchKernelGGL((gemm_fused_k<NACC, SHARED_A, BK, EPI, NW, ST>), dim3(grid), dim3(64 * NW), bytes, s, P);
    SET_LAUNCH_CHECK();
    return SET_OK;
}

// context gating fused with its ctx-side contractions (phase C of the step)
int fused_context_gate(const float* ctx, const float* w_gate_ctx, long long ld_gate, const float* w_sc, Slabs cg_ab,
                       Slabs tc, const float* b_gate, const float* b_sc, const float* b_tc, float* out, int M, int D,
                       hipStream_t s, RowGather gz, RowGather gtc) {
    if (D % 64) return SET_ERR_UNSUPPORTED;
    FusedArgs P{};
    P.A[0] = ctx; P.A[1] = ctx; P.lda[0] = P.lda[1] = D;
    P.W[0] = w_gate_ctx; P.ldw[0] = ld_gate; P.W[1] = w_sc; P.ldw[1] = D;
    P.K = D; P.M = M; P.N = D;
    P.s0 = cg_ab; P.s1 = tc; P.b0 = b_gate; P.b1 = b_sc; P.b2 = b_tc; P.o0 = out;
    P.g0 = gz; P.g1 = gtc;
    const int grid = cdiv(M, 32) * cdiv(D, 32);
    ProfScope ps("fused_context_gate", s, 4.0 * M * D * D, 4.0 * (2.0 * D * D + 2.0 * M * D));
    return launch_fused<2, true, 64, EPI_CTXGATE>(P, grid, s);
}

// copy gate fused with gate_cnew(c_new) and gate_cmem(sel) (phase E of the step)
int fused_copy_gate(const float* c_new, const float* sel, const float* ogate, const float* w_cnew, const float* w_cmem,
                    const float* b_cnew, const float* b_cmem, float* c_out, float* h_out, int M, int D, hipStream_t s) {
    if (D % 64) return SET_ERR_UNSUPPORTED;
    FusedArgs P{};
    P.A[0] = c_new; P.A[1] = sel; P.lda[0] = P.lda[1] = D;
    P.W[0] = w_cnew; P.W[1] = w_cmem; P.ldw[0] = P.ldw[1] = D;
    P.K = D; P.M = M; P.N = D;
    P.b0 = b_cnew; P.b1 = b_cmem; P.e0 = c_new; P.e1 = sel; P.e2 = ogate; P.o0 = c_out; P.o1 = h_out;
    P.alive_prev = g_row_gate.alive_prev;
    const int grid = cdiv(M, 32) * cdiv(D, 32);
    ProfScope ps("fused_copy_gate", s, 4.0 * M * D * D, 4.0 * (2.0 * D * D + 6.0 * M * D));
    return launch_fused<2, false, 64, EPI_COPYGATE>(P, grid, s);
}

// the same with gate_cmem(sel) supplied (hoisted to the prologue: row gather in the attention kernel): one accumulator
int fused_copy_gate_pre(const float* c_new, const float* sel, const float* cmem_pre, const float* ogate,
                        const float* w_cnew, const float* b_cnew, const float* b_cmem, float* c_out, float* h_out, int M,
                        int D, hipStream_t s) {
    if (D % 64) return SET_ERR_UNSUPPORTED;
    FusedArgs P{};
    P.A[0] = c_new; P.lda[0] = D; P.W[0] = w_cnew; P.ldw[0] = D;
    P.K = D; P.M = M; P.N = D;
    P.b0 = b_cnew; P.b1 = b_cmem; P.e0 = c_new; P.e1 = sel; P.e2 = ogate; P.e3 = cmem_pre; P.o0 = c_out; P.o1 = h_out;
    P.alive_prev = g_row_gate.alive_prev;
    const int grid = cdiv(M, 32) * cdiv(D, 32);
    ProfScope ps("fused_copy_gate", s, 2.0 * M * D * D, 4.0 * (1.0 * D * D + 7.0 * M * D));
    static const int bk128 = env_int("SET_COPYGATE_BK128", 1);
    // SET_FUSED_NW8 (bit 0: copy gate, bit 1: encoder step): 8-wave workgroups — 13.5 -> 12.7 us / 19.2 -> 18.1 us at B = 128.
    // (Four register stages, ST = 4, on top of that: no gain — 13.0-13.5 / 18.5-18.7 us; the instantiation is not built.)
    static const int nw8 = env_int("SET_FUSED_NW8", 3);
    if (nw8 && bk128 && D % 128 == 0 && grid <= 256) return launch_fused<1, true, 128, EPI_COPYGATE1, 8>(P, grid, s);
    if (bk128 && D % 128 == 0) return launch_fused<1, true, 128, EPI_COPYGATE1>(P, grid, s);
    return launch_fused<1, true, 64, EPI_COPYGATE1>(P, grid, s);
}

// perm[p] = row with the p-th longest sequence (stable: ties keep row order); nactive[t] = #rows with len > t
__global__ void __launch_bounds__(1024) encoder_order_k(const int64_t* lens, int B, int T, int* perm, int* nactive) {
    __shared__ int sl[4096];                              // B <= 4096 (host check): the lengths once, then LDS only
    for (int b = threadIdx.x; b < B; b += blockDim.x) sl[b] = (int)lens[b];
    __syncthreads();
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        const int lb = sl[b];
        int rank = 0;
        for (int j = 0; j < B; ++j) {
            const int lj = sl[j];
            rank += (lj > lb) || (lj == lb && j < b);
        }
        perm[rank] = b;
    }
    for (int t = threadIdx.x; t < T; t += blockDim.x) {
        int n = 0;
        for (int j = 0; j < B; ++j) n += sl[j] > t;
        nactive[t] = n;
    }
}

int encoder_order(const int64_t* lens, int B, int T, int* perm, int* nactive, hipStream_t s) {
    hipLaunchKernelGGL(encoder_order_k, dim3(1), dim3(1024), 0, s, lens, B, T, perm, nactive);
    SET_LAUNCH_CHECK();
    return SET_OK;
}

// one encoder timestep: gates = h_in W_hh^T + xg[b,pos] + b_extra -> (h_out, c, H[b,pos], Mem[b,pos])
int fused_encoder_step(const float* h_in, float* h_out, float* c, const float* w_hh, const float* xg,
                       long long ld_xg_row, long long ld_xg_t, const float* b_extra, const int64_t* lens, int t,
                       int reverse, float* H, float* Mem, long long ld_out_b, long long ld_out_t, int out_col0, int B,
                       int D, hipStream_t s, const int64_t* seq, int seq_T, int seq_V, const int* perm, const int* nactive) {
    if (D % 128) return SET_ERR_UNSUPPORTED;
    FusedArgs P{};
    P.A[0] = h_in; P.lda[0] = D; P.W[0] = w_hh; P.ldw[0] = D;
    P.K = D; P.M = B; P.N = D; P.gate_stride = D;
    P.b0 = b_extra; P.e0 = xg; P.e1 = h_in; P.o0 = h_out; P.o1 = c; P.o2 = H; P.o3 = Mem; P.lens = lens;
    P.ld_xg_row = ld_xg_row; P.ld_xg_t = ld_xg_t; P.ld_out_b = ld_out_b; P.ld_out_t = ld_out_t;
    P.t = t; P.reverse = reverse; P.out_col0 = out_col0; P.seq = seq; P.seq_T = seq_T; P.seq_V = seq_V > 0 ? seq_V : 1;
    P.perm = perm; P.nactive = nactive;
    // SET_ENC_UNITS16 (experiment, round 5): 16 units per workgroup — two accumulators on one staged activation tile, 98 MB
    // through L2 per step at B = 128 instead of 134 — is SLOWER (22.9 / 23.7 us with BK 64 / 128 against 20.5): the step is not
    // bound by L2 bytes but by each workgroup's serial k-loop (1.15 us per 128-wide k-tile) on top of 11 us of fixed cost
    static const int u16 = env_int("SET_ENC_UNITS16", 0);
    static const int u16_min = env_int("SET_ENC_UNITS16_MINB", 64);
    ProfScope ps("fused_encoder_step", s, 8.0 * B * D * D, 4.0 * (4.0 * D * D + 8.0 * B * D));
    if (u16 && B >= u16_min && D % 16 == 0) {
        const int grid16 = cdiv(B, 32) * cdiv(D, 16);
        return u16 == 2 ? launch_fused<2, true, 128, EPI_ENCLSTM>(P, grid16, s) : launch_fused<2, true, 64, EPI_ENCLSTM>(P, grid16, s);
    }
    const int grid = cdiv(B, 32) * cdiv(D, 8);
    static const int nw8 = env_int("SET_FUSED_NW8", 3);
    if (nw8 & 2) return launch_fused<1, true, 128, EPI_ENCLSTM, 8>(P, grid, s);
    return launch_fused<1, true, 128, EPI_ENCLSTM>(P, grid, s);
}

// the same step inside a grad-enabled forward (position = t for every row, no token table, no row ordering): c is not
// updated in place, gates / previous h are kept for the backward, finished rows are written as zeros
int fused_encoder_step_train(const float* h_in, const float* c_in, float* h_out, float* c_out, const float* w_hh,
                             const float* xg, long long ld_xg_row, long long ld_xg_t, const float* b_hh, const int64_t* lens,
                             int t, float* H, float* Mem, float* Hprev, long long ld_out_b, long long ld_out_t, int out_col0,
                             float* gates, int B, int D, hipStream_t s) {
    if (D % 128) return SET_ERR_UNSUPPORTED;
    FusedArgs P{};
    P.A[0] = h_in; P.lda[0] = D; P.W[0] = w_hh; P.ldw[0] = D;
    P.K = D; P.M = B; P.N = D; P.gate_stride = D;
    P.b0 = b_hh; P.e0 = xg; P.e1 = h_in; P.o0 = h_out; P.o1 = c_out; P.o2 = H; P.o3 = Mem; P.lens = lens;
    P.ld_xg_row = ld_xg_row; P.ld_xg_t = ld_xg_t; P.ld_out_b = ld_out_b; P.ld_out_t = ld_out_t;
    P.t = t; P.reverse = 0; P.out_col0 = out_col0; P.seq = nullptr; P.seq_T = 0; P.seq_V = 1;
    P.c_in = c_in; P.gates = gates; P.hprev = Hprev;
    const int grid = cdiv(B, 32) * cdiv(D, 8);
    ProfScope ps("fused_encoder_step(train)", s, 8.0 * B * D * D, 4.0 * (4.0 * D * D + 16.0 * B * D));
    return launch_fused<1, true, 128, EPI_ENCLSTM>(P, grid, s);
}

}  // namespace set
