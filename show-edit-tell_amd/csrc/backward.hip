// Backward (gradient) kernels of the decode-step operators + their C ABI.
// The reference has no backward code: it is PyTorch autograd over a1-a12 (editnet.py:579).  Here the
// pointwise / attention parts of every operator's backward are hand-written HIP kernels; the plain
// dX = dY W and dW = dY^T X contractions of the backward stay library GEMMs on the PyTorch side
// (autograd_ops.py), as BASELINE.json's north star keeps autograd host-side.
// Conventions: `gates` holds POST-activation (i, f, g, o) as saved by the train-mode forward;
// every kernel handles one float4 of hidden units per thread; all outputs are fully overwritten.
#include "set_common.h"
#include "philox.h"

namespace set {

typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 ldb4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void stb4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ float wsum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// ---------------------------------------------------------------------------------------------
// LSTM cell backward (nn.LSTMCell / LSTMCellC, editnet.py:235-242):
//   c' = f c + i g ; h' = o tanh(c')
//   given dh' and dc' (external, may be NULL):  dct = dc' + dh' o (1 - tanh^2 c')
//   d pre-activations: di = dct g i(1-i), df = dct c f(1-f), dg = dct i (1-g^2), do = dh' tanh(c') o(1-o)
//   dc_prev = dct f
// ---------------------------------------------------------------------------------------------
template <bool SRC>
__global__ void __launch_bounds__(256) lstm_cell_bwd_k(const float* dh, const float* dc_in, const float* gates,
                                                       const float* c_prev, const float* c_new, float* dgates,
                                                       float* dc_prev, int M, int D, const SrcList S) {
    const int per_row = D >> 2;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)M * per_row) return;
    const long long m = idx / per_row;
    const int j = (int)(idx - m * per_row) << 2;
    const float* gr = gates + m * 4 * D + j;
    const f32x4 gi = ldb4(gr), gf = ldb4(gr + D), gg = ldb4(gr + 2 * D), go = ldb4(gr + 3 * D);
    const f32x4 cp = ldb4(c_prev + m * D + j), cn = ldb4(c_new + m * D + j);
    f32x4 dhv = {0.f, 0.f, 0.f, 0.f}, dcv = {0.f, 0.f, 0.f, 0.f};
    if constexpr (SRC) dhv = S.load4(dh, m * D + j, m, j);   // dh (may be NULL) + the addends still in split-K partials
    else if (dh) dhv = ldb4(dh + m * D + j);
    if (dc_in) dcv = ldb4(dc_in + m * D + j);
    f32x4 di, df, dg, dou, dcp;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float tc = tanhf(cn[e]);
        const float dct = dcv[e] + dhv[e] * go[e] * (1.f - tc * tc);
        di[e] = dct * gg[e] * gi[e] * (1.f - gi[e]);
        df[e] = dct * cp[e] * gf[e] * (1.f - gf[e]);
        dg[e] = dct * gi[e] * (1.f - gg[e] * gg[e]);
        dou[e] = dhv[e] * tc * go[e] * (1.f - go[e]);
        dcp[e] = dct * gf[e];
    }
    float* dr = dgates + m * 4 * D + j;
    stb4(dr, di); stb4(dr + D, df); stb4(dr + 2 * D, dg); stb4(dr + 3 * D, dou);
    stb4(dc_prev + m * D + j, dcp);
}

// ---------------------------------------------------------------------------------------------
// CopyLSTMCellC backward, stage 1 (editnet.py:281-283):
//   cg = sig(u) ; adp = cg cm + (1-cg) cn ; h = o tanh(adp)
//   given dh and dadp (external, may be NULL):
//     dadp_t = dadp + dh o (1 - tanh^2 adp) ; do_pre = dh tanh(adp) o(1-o)
//     du = dadp_t (cm - cn) cg (1-cg) ; dcm_direct = dadp_t cg ; dcn_direct = dadp_t (1-cg)
// stage 2 (after dcn = dcn_direct + du W_n on the host side) is lstm_gates_bwd_k.
// ---------------------------------------------------------------------------------------------
template <bool SRC>
__global__ void __launch_bounds__(256) copy_gate_bwd_k(const float* dh, const float* dadp_in, const float* ogate,
                                                       const float* adp, const float* cg, const float* cmem,
                                                       const float* c_new, float* du, float* dcm_direct,
                                                       float* dcn_direct, float* do_pre, int M, int D, long long ld_og,
                                                       const SrcList S, const float* dh_drop, long long ld_drop, float p_drop,
                                                       float sc_drop, unsigned long long seed, unsigned long long offset) {
    const int per_row = D >> 2;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)M * per_row) return;
    const long long o = idx * 4;
    const long long m = idx / per_row;
    f32x4 dhv = {0.f, 0.f, 0.f, 0.f}, dav = {0.f, 0.f, 0.f, 0.f};
    const int j = (int)(o - m * D);
    if constexpr (SRC) dhv = S.load4(dh, o, m, j);
    else if (dh) dhv = ldb4(dh + o);
    if (SRC && dh_drop) {                                           // + the output dropout's backward (dropout_bwd_philox_k, fused)
        const f32x4 g = ldb4(dh_drop + m * ld_drop + j);
        if (p_drop > 0.f) {
            uint32_t k[4] = {(uint32_t)m, (uint32_t)(j >> 2), (uint32_t)offset, (uint32_t)(offset >> 32)};
            philox4x32_10(k, (uint32_t)seed, (uint32_t)(seed >> 32));
#pragma unroll
            for (int e = 0; e < 4; ++e) dhv[e] += ((float)(k[e] >> 8) * (1.0f / 16777216.0f) >= p_drop) ? g[e] * sc_drop : 0.f;
        } else dhv += g;
    }
    if (dadp_in) dav = ldb4(dadp_in + o);
    const f32x4 og = ldb4(ogate + m * ld_og + (o - m * D)), ad = ldb4(adp + o), g = ldb4(cg + o), cm = ldb4(cmem + o), cn = ldb4(c_new + o);
    f32x4 duv, dcm, dcn, dop;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float ta = tanhf(ad[e]);
        const float dat = dav[e] + dhv[e] * og[e] * (1.f - ta * ta);
        dop[e] = dhv[e] * ta * og[e] * (1.f - og[e]);
        duv[e] = dat * (cm[e] - cn[e]) * g[e] * (1.f - g[e]);
        dcm[e] = dat * g[e];
        dcn[e] = dat * (1.f - g[e]);
    }
    stb4(du + o, duv); stb4(dcm_direct + o, dcm); stb4(dcn_direct + o, dcn); stb4(do_pre + o, dop);
}

// LSTM gate backward given the gradient of the new cell state and the o-gate pre-activation gradient
template <bool SRC>
__device__ __forceinline__ void lstm_gates_bwd_body(const int blk, const float* dcn, const float* do_pre, const float* gates,
                                                    const float* c_prev, float* dgates, float* dc_prev, int M, int D,
                                                    const SrcList& S) {
    const int per_row = D >> 2;
    const long long idx = (long long)blk * blockDim.x + threadIdx.x;
    if (idx >= (long long)M * per_row) return;
    const long long m = idx / per_row;
    const int j = (int)(idx - m * per_row) << 2;
    const float* gr = gates + m * 4 * D + j;
    const f32x4 gi = ldb4(gr), gf = ldb4(gr + D), gg = ldb4(gr + 2 * D);
    const f32x4 cp = ldb4(c_prev + m * D + j), dct = SRC ? S.load4(dcn, m * D + j, m, j) : ldb4(dcn + m * D + j), dop = ldb4(do_pre + m * D + j);
    f32x4 di, df, dg, dcp;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        di[e] = dct[e] * gg[e] * gi[e] * (1.f - gi[e]);
        df[e] = dct[e] * cp[e] * gf[e] * (1.f - gf[e]);
        dg[e] = dct[e] * gi[e] * (1.f - gg[e] * gg[e]);
        dcp[e] = dct[e] * gf[e];
    }
    float* dr = dgates + m * 4 * D + j;
    stb4(dr, di); stb4(dr + D, df); stb4(dr + 2 * D, dg); stb4(dr + 3 * D, dop);
    stb4(dc_prev + m * D + j, dcp);
}
template <bool SRC>
__global__ void __launch_bounds__(256) lstm_gates_bwd_k(const float* dcn, const float* do_pre, const float* gates,
                                                        const float* c_prev, float* dgates, float* dc_prev, int M,
                                                        int D, const SrcList S) {
    lstm_gates_bwd_body<SRC>(blockIdx.x, dcn, do_pre, gates, c_prev, dgates, dc_prev, M, D, S);
}

// ---------------------------------------------------------------------------------------------
// context gating backward (editnet.py:378-380): out = zt s + (1-zt) t, zt = sig(z), s = tanh(.), t = tanh(.)
//   dz_pre = dout (s - t) zt (1-zt) ; ds_pre = dout zt (1-s^2) ; dt_pre = dout (1-zt)(1-t^2)
// ---------------------------------------------------------------------------------------------
template <bool SRC>
__device__ __forceinline__ void context_gate_bwd_body(const int blk, const float* dout, const float* zt, const float* s,
                                                      const float* t, float* dz, float* ds, float* dt, long long n4, int D4,
                                                      long long ld_out, const SrcList& S) {
    const long long idx = (long long)blk * blockDim.x + threadIdx.x;
    if (idx >= n4) return;
    const long long o = idx * 4;
    const long long m = idx / D4;
    const long long oo = m * ld_out + (o - m * 4 * D4);       // the three outputs share a row stride (>= D)
    const f32x4 d = SRC ? S.load4(dout, o, m, (int)(o - m * 4 * D4)) : ldb4(dout + o), z = ldb4(zt + o), sv = ldb4(s + o), tv = ldb4(t + o);
    f32x4 a, b, c;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        a[e] = d[e] * (sv[e] - tv[e]) * z[e] * (1.f - z[e]);
        b[e] = d[e] * z[e] * (1.f - sv[e] * sv[e]);
        c[e] = d[e] * (1.f - z[e]) * (1.f - tv[e] * tv[e]);
    }
    stb4(dz + oo, a); stb4(ds + oo, b); stb4(dt + oo, c);
}
template <bool SRC>
__global__ void __launch_bounds__(256) context_gate_bwd_k(const float* dout, const float* zt, const float* s,
                                                          const float* t, float* dz, float* ds, float* dt, long long n4,
                                                          int D4, long long ld_out, const SrcList S) {
    context_gate_bwd_body<SRC>(blockIdx.x, dout, zt, s, t, dz, ds, dt, n4, D4, ld_out, S);
}

// ---------------------------------------------------------------------------------------------
// additive attention backward (caption: tanh, editnet.py:370-376 ; visual: relu, :443-446).
// One workgroup per sample.  Given dctx (M,Dv) and an optional external dalpha (M,L):
//   dalpha_l = <dctx, V_l> + dalpha_ext_l ; de = alpha (dalpha - sum alpha dalpha)
//   dpre[l,a] = de_l w_full[a] act'(att1[l,a] + att2[a])        -> datt1 (M,L,A), datt2 = sum_l dpre (M,A)
//   dwfull_part[a] = sum_l de_l act(att1[l,a] + att2[a])         (M,A), reduced over M by the caller
//   dV_l = alpha_l dctx   (only when dV != NULL: the caption features get gradient, the image features do not)
// att2 already contains the decoder-projection bias.
// ---------------------------------------------------------------------------------------------
constexpr int ATTB_MAX = 256;
template <bool TANH>
__global__ void __launch_bounds__(256) attention_bwd_k(const float* dctx, const float* dalpha_ext, const float* alpha,
                                                       const float* Vals, const float* att1, const float* att2,
                                                       const float* w_full, float* datt1, float* datt2,
                                                       float* dwfull_part, float* dV, float* de_out, int L, int Dv,
                                                       int A, int acc_datt1, int acc_dv, long long ld_datt2) {
    __shared__ float s_da[ATTB_MAX];
    __shared__ float s_de[ATTB_MAX];
    __shared__ float s_dot;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* dc = dctx + (long long)b * Dv;
    // dalpha_l = <dctx, V_l>: one wave per row; the wave's slice of dctx stays in registers and the loads of two
    // rows (up to 16 float4 per lane) are in flight together
    constexpr int NQ = 8;                                   // Dv <= 2048 on this path (host check)
    f32x4 dcr[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int d = lane * 4 + 256 * q;
        dcr[q] = d < Dv ? ldb4(dc + d) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    for (int l0 = wave; l0 < L; l0 += 8) {
        const int l1 = l0 + 4;
        const float* v0 = Vals + ((long long)b * L + l0) * Dv;
        const float* v1 = Vals + ((long long)b * L + (l1 < L ? l1 : l0)) * Dv;
        f32x4 x0[NQ], x1[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int d = lane * 4 + 256 * q;
            x0[q] = d < Dv ? ldb4(v0 + d) : (f32x4){0.f, 0.f, 0.f, 0.f};
            x1[q] = d < Dv ? ldb4(v1 + d) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {                      // same accumulation order as a d-ascending loop
            s0 += x0[q][0] * dcr[q][0] + x0[q][1] * dcr[q][1] + x0[q][2] * dcr[q][2] + x0[q][3] * dcr[q][3];
            s1 += x1[q][0] * dcr[q][0] + x1[q][1] * dcr[q][1] + x1[q][2] * dcr[q][2] + x1[q][3] * dcr[q][3];
        }
        s0 = wsum(s0);
        s1 = wsum(s1);
        if (lane == 0) {
            s_da[l0] = s0 + (dalpha_ext ? dalpha_ext[(long long)b * L + l0] : 0.f);
            if (l1 < L) s_da[l1] = s1 + (dalpha_ext ? dalpha_ext[(long long)b * L + l1] : 0.f);
        }
    }
    __syncthreads();
    if (tid < 64) {
        float s = 0.f;
        for (int l = tid; l < L; l += 64) s += alpha[(long long)b * L + l] * s_da[l];
        s = wsum(s);
        if (tid == 0) s_dot = s;
    }
    __syncthreads();
    for (int l = tid; l < L; l += 256) {
        const float a = alpha[(long long)b * L + l];
        const float de = a * (s_da[l] - s_dot);
        s_de[l] = de;
        if (de_out) de_out[(long long)b * L + l] = de;
    }
    __syncthreads();
    // datt1 / datt2 / dwfull: thread -> attention columns a = tid*4 .. (A <= 1024)
    for (int a = tid * 4; a < A; a += 1024) {
        const f32x4 w = ldb4(w_full + a), a2 = ldb4(att2 + (long long)b * A + a);
        f32x4 acc2 = {0.f, 0.f, 0.f, 0.f}, accw = {0.f, 0.f, 0.f, 0.f};
        constexpr int LB = 6;                               // rows of att1 in flight per pass
        for (int lb = 0; lb < L; lb += LB) {
            f32x4 pr[LB];
#pragma unroll
            for (int u = 0; u < LB; ++u)
                pr[u] = (lb + u < L) ? ldb4(att1 + ((long long)b * L + lb + u) * A + a) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < LB; ++u) {
                const int l = lb + u;
                if (l >= L) break;
                const f32x4 p = pr[u] + a2;
                const float de = s_de[l];
                f32x4 dp;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float act, dact;
                    if (TANH) { act = tanhf(p[e]); dact = 1.f - act * act; }
                    else { act = p[e] > 0.f ? p[e] : 0.f; dact = p[e] > 0.f ? 1.f : 0.f; }
                    dp[e] = de * w[e] * dact;
                    accw[e] += de * act;
                }
                acc2 += dp;
                float* o = datt1 + ((long long)b * L + l) * A + a;
                stb4(o, acc_datt1 ? ldb4(o) + dp : dp);
            }
        }
        stb4(datt2 + (long long)b * ld_datt2 + a, acc2);
        stb4(dwfull_part + (long long)b * A + a, accw);
    }
    if (dV) {
        for (int d = tid * 4; d < Dv; d += 1024) {
            const f32x4 y = ldb4(dc + d);
            for (int l = 0; l < L; ++l) {
                float* o = dV + ((long long)b * L + l) * Dv + d;
                const f32x4 v = y * alpha[(long long)b * L + l];
                stb4(o, acc_dv ? ldb4(o) + v : v);
            }
        }
    }
}

// Wide form of attention_bwd_k (round 4): 512 threads per row.  The 256-thread form walks five dependent rounds of
// value-row loads and six of att1 loads with one workgroup per sample (128 workgroups on 256 CUs): 22-25 us at B = 128
// whatever the bytes.  Here wave w takes the value rows l = w, w + 8, ... three at a time (24 float4 per lane in flight),
// and the (L x A) projection block is covered by (A / 4) column groups x 4 row slices, each thread holding ten of its rows
// before it touches any; the per-column sums over rows are combined across the 4 slices in slice order through LDS.  Same
// results as attention_bwd_k up to the order of those two sums.
constexpr int ATTW_RB = 3;       // value rows of a wave in flight
constexpr int ATTW_SB = 10;      // projection rows of a thread in flight
struct AttBwdArgs {
    const float* dctx; const float* dalpha_ext; const float* alpha; const float* Vals; const float* att1; const float* att2;
    const float* w_full; float* datt1; float* datt2; float* dwfull_part; float* de_out; int L, Dv, A, acc_datt1;
    long long ld_datt2; float* dctx_out;
};
template <bool TANH, bool SRC>
__device__ __forceinline__ void attention_bwd_wide_body(const int b, const AttBwdArgs& P, const SrcList& S) {
    __shared__ float s_da[ATTB_MAX];
    __shared__ float s_de[ATTB_MAX];
    __shared__ float s_dot;
    __shared__ __attribute__((aligned(16))) float s_acc[4 * 128 * 4 * 2];     // [slice][column group][acc2 | accw] float4
    const float *dctx = P.dctx, *dalpha_ext = P.dalpha_ext, *alpha = P.alpha, *Vals = P.Vals, *att1 = P.att1, *att2 = P.att2,
                *w_full = P.w_full;
    float *datt1 = P.datt1, *datt2 = P.datt2, *dwfull_part = P.dwfull_part, *de_out = P.de_out, *dctx_out = P.dctx_out;
    const int L = P.L, Dv = P.Dv, A = P.A, acc_datt1 = P.acc_datt1;
    const long long ld_datt2 = P.ld_datt2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int NQ = 8;                                   // Dv <= 2048
    f32x4 dcr[NQ];
    if constexpr (SRC) {
        // the gradient row is still split-K partials: summed ONCE by the block (one float4 per thread, every partial requested
        // before the first add), parked in LDS — s_acc is free until the projection pass — and handed to the eight waves
        f32x4* s_dc = reinterpret_cast<f32x4*>(s_acc);
        if (tid * 4 < Dv) {
            const f32x4 v = S.load4(dctx, (long long)b * Dv + tid * 4, b, tid * 4);
            s_dc[tid] = v;
            if (dctx_out) stb4(dctx_out + (long long)b * Dv + tid * 4, v);
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int d = lane * 4 + 256 * q;
            dcr[q] = d < Dv ? s_dc[lane + 64 * q] : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        __syncthreads();                                    // (s_acc is reused below)
    } else {
        const float* dc = dctx + (long long)b * Dv;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int d = lane * 4 + 256 * q;
            dcr[q] = d < Dv ? ldb4(dc + d) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    }
    for (int l0 = wave; l0 < L; l0 += 8 * ATTW_RB) {
        f32x4 xr[ATTW_RB][NQ];
#pragma unroll
        for (int i = 0; i < ATTW_RB; ++i) {
            const int l = l0 + 8 * i;
            const float* v = Vals + ((long long)b * L + (l < L ? l : l0)) * Dv;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int d = lane * 4 + 256 * q;
                xr[i][q] = d < Dv ? ldb4(v + d) : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
#pragma unroll
        for (int i = 0; i < ATTW_RB; ++i) {
            const int l = l0 + 8 * i;
            float s0 = 0.f;
#pragma unroll
            for (int q = 0; q < NQ; ++q)              // same accumulation order as attention_bwd_k
                s0 += xr[i][q][0] * dcr[q][0] + xr[i][q][1] * dcr[q][1] + xr[i][q][2] * dcr[q][2] + xr[i][q][3] * dcr[q][3];
            s0 = wsum(s0);
            if (lane == 0 && l < L) s_da[l] = s0 + (dalpha_ext ? dalpha_ext[(long long)b * L + l] : 0.f);
        }
    }
    __syncthreads();
    if (tid < 64) {
        float s = 0.f;
        for (int l = tid; l < L; l += 64) s += alpha[(long long)b * L + l] * s_da[l];
        s = wsum(s);
        if (tid == 0) s_dot = s;
    }
    __syncthreads();
    for (int l = tid; l < L; l += 512) {
        const float a = alpha[(long long)b * L + l];
        const float de = a * (s_da[l] - s_dot);
        s_de[l] = de;
        if (de_out) de_out[(long long)b * L + l] = de;
    }
    __syncthreads();
    const int ncg = A >> 2, t_cg = tid & 127, t_sl = tid >> 7;       // 128 column groups x 4 row slices per pass
    for (int c0 = 0; c0 < ncg; c0 += 128) {                  // A <= 512: one pass; A <= 1024: two
        const int c = c0 + t_cg, a = c * 4;
        const bool on = c < ncg;
        f32x4 w = {0.f, 0.f, 0.f, 0.f}, a2 = {0.f, 0.f, 0.f, 0.f};
        if (on) { w = ldb4(w_full + a); a2 = ldb4(att2 + (long long)b * A + a); }
        f32x4 acc2 = {0.f, 0.f, 0.f, 0.f}, accw = {0.f, 0.f, 0.f, 0.f};
        for (int lb = t_sl; lb < L; lb += 4 * ATTW_SB) {
            f32x4 pr[ATTW_SB];
#pragma unroll
            for (int u = 0; u < ATTW_SB; ++u) {
                const int l = lb + 4 * u;
                pr[u] = (on && l < L) ? ldb4(att1 + ((long long)b * L + l) * A + a) : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int u = 0; u < ATTW_SB; ++u) {
                const int l = lb + 4 * u;
                if (!on || l >= L) continue;
                const f32x4 p = pr[u] + a2;
                const float de = s_de[l];
                f32x4 dp;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float act, dact;
                    if (TANH) { act = tanhf(p[e]); dact = 1.f - act * act; }
                    else { act = p[e] > 0.f ? p[e] : 0.f; dact = p[e] > 0.f ? 1.f : 0.f; }
                    dp[e] = de * w[e] * dact;
                    accw[e] += de * act;
                }
                acc2 += dp;
                float* o = datt1 + ((long long)b * L + l) * A + a;
                stb4(o, acc_datt1 ? ldb4(o) + dp : dp);
            }
        }
        if (c0) __syncthreads();
        *reinterpret_cast<f32x4*>(s_acc + ((t_sl * 128 + t_cg) * 2 + 0) * 4) = acc2;
        *reinterpret_cast<f32x4*>(s_acc + ((t_sl * 128 + t_cg) * 2 + 1) * 4) = accw;
        __syncthreads();
        if (t_sl == 0 && on) {
            f32x4 r2 = {0.f, 0.f, 0.f, 0.f}, rw = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                r2 += *reinterpret_cast<const f32x4*>(s_acc + ((s4 * 128 + t_cg) * 2 + 0) * 4);
                rw += *reinterpret_cast<const f32x4*>(s_acc + ((s4 * 128 + t_cg) * 2 + 1) * 4);
            }
            stb4(datt2 + (long long)b * ld_datt2 + a, r2);
            stb4(dwfull_part + (long long)b * A + a, rw);
        }
    }
}
template <bool TANH, bool SRC>
__global__ void __launch_bounds__(512) attention_bwd_wide_k(const AttBwdArgs P, const SrcList S) {
    attention_bwd_wide_body<TANH, SRC>(blockIdx.x, P, S);
}
// visual attention backward + context gating backward of one timestep in one launch (both wait for the same grouped dX
// product and feed the next one): blocks [0, M) = attention rows, the rest = context-gate elements, 512 threads each
struct CtxGateArgs { const float* dout; const float* zt; const float* s; const float* t; float* dz; float* ds; float* dt;
                     long long n4; int D4; long long ld_out; };
template <bool SRC>
__global__ void __launch_bounds__(512) attention_ctxgate_bwd_k(const AttBwdArgs P, const SrcList S, const CtxGateArgs C,
                                                               const SrcList SC, int M) {
    if ((int)blockIdx.x < M) attention_bwd_wide_body<false, SRC>(blockIdx.x, P, S);
    else context_gate_bwd_body<SRC>(blockIdx.x - M, C.dout, C.zt, C.s, C.t, C.dz, C.ds, C.dt, C.n4, C.D4, C.ld_out, SC);
}

// ---------------------------------------------------------------------------------------------
// SelectC backward (editnet.py:409-420): sel = w M[j*], w = a + (1 - a_detached)
//   dM[b, j*] = w dsel ; dalpha[b, j*] = <dsel, M[b, j*]> ; zero elsewhere
// ---------------------------------------------------------------------------------------------
template <bool SRC>
__device__ __forceinline__ void select_bwd_body(const int b, const float* dsel, const float* Mem, const float* alpha, float* dM,
                                                float* dalpha, int T, int D, int acc_dm, const SrcList& S) {
    __shared__ int s_arg;
    __shared__ float s_val, s_red[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < 64) {
        float best = -INFINITY; int bi = 0x7fffffff;
        for (int t = tid; t < T; t += 64) { const float a = alpha[(long long)b * T + t]; if (a > best) { best = a; bi = t; } }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ob = __shfl_xor(best, o);
            const int oi = __shfl_xor(bi, o);
            if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
        if (tid == 0) { s_arg = bi == 0x7fffffff ? 0 : bi; s_val = best; }     // (all-NaN alpha: stay in range)
    }
    __syncthreads();
    const int js = s_arg;
    const float w = s_val * 1.f + (1.f - s_val);
    float dot = 0.f;
    if (acc_dm) {
        // accumulate: only the selected row of dM changes — no walk over the T rows
        for (int d = tid * 4; d < D; d += 1024) {
            const f32x4 g = SRC ? S.load4(dsel, (long long)b * D + d, b, d) : ldb4(dsel + (long long)b * D + d);
            const f32x4 m = ldb4(Mem + ((long long)b * T + js) * D + d);
            dot += g[0] * m[0] + g[1] * m[1] + g[2] * m[2] + g[3] * m[3];
            float* o = dM + ((long long)b * T + js) * D + d;
            stb4(o, ldb4(o) + g * w);
        }
    } else {
        for (int t = 0; t < T; ++t)
            for (int d = tid * 4; d < D; d += 1024) {
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (t == js) {
                    const f32x4 g = SRC ? S.load4(dsel, (long long)b * D + d, b, d) : ldb4(dsel + (long long)b * D + d);
                    const f32x4 m = ldb4(Mem + ((long long)b * T + t) * D + d);
                    v = g * w;
                    dot += g[0] * m[0] + g[1] * m[1] + g[2] * m[2] + g[3] * m[3];
                }
                stb4(dM + ((long long)b * T + t) * D + d, v);
            }
    }
    dot = wsum(dot);
    if (lane == 0) s_red[wave] = dot;
    __syncthreads();
    for (int t = tid; t < T; t += 256)
        dalpha[(long long)b * T + t] = (t == js) ? ((s_red[0] + s_red[1]) + (s_red[2] + s_red[3])) : 0.f;
}
// both attention backwards of one timestep in one launch: blocks [0, M) = visual rows, [M, 2M) = caption rows.  The visual
// one only needs the copy cell's input gradient, the caption one additionally the context-gate product that follows it — run
// together after that product, the 128-workgroup visual kernel (half the chip idle for 19 us) leaves the dependent chain
template <bool SRC>
__global__ void __launch_bounds__(512) attention_pair_bwd_k(const AttBwdArgs PV, const SrcList SV, const AttBwdArgs PC,
                                                            const SrcList SCp, int M) {
    if ((int)blockIdx.x < M) attention_bwd_wide_body<false, SRC>(blockIdx.x, PV, SV);
    else attention_bwd_wide_body<true, SRC>(blockIdx.x - M, PC, SCp);
}
// soft selection (editnet.py:419-420 with soft = True): the weighted sum over the T memory rows, terms added in index order
__global__ void __launch_bounds__(256) select_soft_k(const float* Mem, const float* alpha, float* sel, int T, int D) {
    const int b = blockIdx.x, tid = threadIdx.x;
    for (int d = tid * 4; d < D; d += 1024) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int t = 0; t < T; ++t) acc += ldb4(Mem + ((long long)b * T + t) * D + d) * alpha[(long long)b * T + t];
        stb4(sel + (long long)b * D + d, acc);
    }
}
__global__ void __launch_bounds__(256) select_soft_bwd_k(const float* dsel, const float* Mem, const float* alpha, float* dM,
                                                         float* dalpha, int T, int D) {
    __shared__ float s_red[4];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int t = 0; t < T; ++t) {
        const float a = alpha[(long long)b * T + t];
        float dot = 0.f;
        for (int d = tid * 4; d < D; d += 1024) {
            const f32x4 g = ldb4(dsel + (long long)b * D + d);
            const f32x4 m = ldb4(Mem + ((long long)b * T + t) * D + d);
            dot += g[0] * m[0] + g[1] * m[1] + g[2] * m[2] + g[3] * m[3];
            stb4(dM + ((long long)b * T + t) * D + d, g * a);
        }
        dot = wsum(dot);
        __syncthreads();                                   // (s_red of the previous row has been read)
        if (lane == 0) s_red[wave] = dot;
        __syncthreads();
        if (tid == 0) dalpha[(long long)b * T + t] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
    }
}
template <bool SRC>
__global__ void __launch_bounds__(256) select_bwd_k(const float* dsel, const float* Mem, const float* alpha, float* dM,
                                                    float* dalpha, int T, int D, int acc_dm, const SrcList S) {
    select_bwd_body<SRC>(blockIdx.x, dsel, Mem, alpha, dM, dalpha, T, D, acc_dm, S);
}
// LSTM gate backward + SelectC backward of one timestep in one launch (both wait for the same grouped dX product):
// blocks [0, nG) = gate elements, the next M = selection rows
template <bool SRC>
__global__ void __launch_bounds__(256) lstm_gates_select_bwd_k(const float* dcn, const float* do_pre, const float* gates,
                                                               const float* c_prev, float* dgates, float* dc_prev, int M, int D,
                                                               const SrcList S, int nG, const float* dsel, const float* Mem,
                                                               const float* alpha, float* dM, float* dalpha, int T, int acc_dm,
                                                               const SrcList SS) {
    if ((int)blockIdx.x < nG) lstm_gates_bwd_body<SRC>(blockIdx.x, dcn, do_pre, gates, c_prev, dgates, dc_prev, M, D, S);
    else select_bwd_body<SRC>(blockIdx.x - nG, dsel, Mem, alpha, dM, dalpha, T, D, acc_dm, SS);
}

// ---------------------------------------------------------------------------------------------
// Caption-encoder recurrence for the grad-enabled path (CaptionEncoderC editnet.py:333-338; the packed BiLSTM of
// dcnet.py:233): one step for ALL rows with the reference's length handling inside the kernel — rows with
// t < len[b] advance, the others carry their state and emit zeros — instead of ~12 masked elementwise torch kernels
// per step (and as many again in the backward).  xg holds the hoisted input projection x W_x^T + b_x for every (b, t).
//   forward : gates = hh slabs + xg[b,t] + b_hh ; (i,f,g,o) ; c' = f c + i g ; h' = o tanh(c')
//             valid row : h_out = h', c_out = c', H[b,t] = h', Mem[b,t] = c', gates_out = post-activations
//             else      : h_out = h,  c_out = c,  H[b,t] = 0,  Mem[b,t] = 0,  gates_out = 0
//   backward: valid row : dh' = dh_out + dH[b,t], dc' = dc_out + dM[b,t] -> LSTM cell backward (dgates, dc_prev), dh_pass = 0
//             else      : dgates = 0, dc_prev = dc_out, dh_pass = dh_out
//             (the caller adds dgates W_hh to dh_pass: that is dh of the previous step)
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) enc_cell_train_k(Slabs hh, const float* xg, long long ld_xg_row, long long ld_xg_t,
                                                        const float* b_hh, const int64_t* lens, int t, const float* h_in,
                                                        const float* c_in, float* h_out, float* c_out, float* H, float* Mem,
                                                        float* Hprev, long long ld_out_b, long long ld_out_t, int out_col0,
                                                        float* gates_out, int B, int D) {
    const int per_row = D >> 2;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)B * per_row) return;
    const long long b = idx / per_row;
    const int j = (int)(idx - b * per_row) << 2;
    const bool valid = t < (int)lens[b];
    const f32x4 hp = ldb4(h_in + b * D + j), cp = ldb4(c_in + b * D + j);
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    float* gr = gates_out + b * 4 * D + j;
    const long long o = b * ld_out_b + (long long)t * ld_out_t + out_col0 + j;
    if (Hprev) stb4(Hprev + b * (long long)(ld_out_b) + (long long)t * ld_out_t + out_col0 + j, hp);
    if (!valid) {
        stb4(h_out + b * D + j, hp); stb4(c_out + b * D + j, cp);
        stb4(H + o, zero);
        if (Mem) stb4(Mem + o, zero);
        stb4(gr, zero); stb4(gr + D, zero); stb4(gr + 2 * D, zero); stb4(gr + 3 * D, zero);
        return;
    }
    const float* xr = xg + b * ld_xg_row + (long long)t * ld_xg_t;
    f32x4 g[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        f32x4 v = zero;
        for (int i = 0; i < hh.n; ++i) v += ldb4(hh.p + (long long)i * hh.stride + b * hh.ld + q * D + j);
        v += ldb4(xr + q * D + j);
        if (b_hh) v += ldb4(b_hh + q * D + j);
        g[q] = v;
    }
    f32x4 ai, af, ag, ao, cn, hn;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        ai[e] = 1.f / (1.f + expf(-g[0][e])); af[e] = 1.f / (1.f + expf(-g[1][e]));
        ag[e] = tanhf(g[2][e]); ao[e] = 1.f / (1.f + expf(-g[3][e]));
        cn[e] = af[e] * cp[e] + ai[e] * ag[e];
        hn[e] = ao[e] * tanhf(cn[e]);
    }
    stb4(h_out + b * D + j, hn); stb4(c_out + b * D + j, cn);
    stb4(H + o, hn);
    if (Mem) stb4(Mem + o, cn);
    stb4(gr, ai); stb4(gr + D, af); stb4(gr + 2 * D, ag); stb4(gr + 3 * D, ao);
}

__global__ void __launch_bounds__(256) enc_cell_bwd_k(const float* dh_out, const float* dc_out, const float* dH, const float* dM,
                                                      long long ld_d_b, long long ld_d_t, int d_col0, const int64_t* lens,
                                                      int t, const float* gates, const float* c_prev, const float* c_new,
                                                      float* dgates, long long ld_dg, float* dc_prev, float* dh_pass, int B,
                                                      int D) {
    const int per_row = D >> 2;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)B * per_row) return;
    const long long b = idx / per_row;
    const int j = (int)(idx - b * per_row) << 2;
    const bool valid = t < (int)lens[b];
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    f32x4 dhv = dh_out ? ldb4(dh_out + b * D + j) : zero;
    f32x4 dcv = dc_out ? ldb4(dc_out + b * D + j) : zero;
    float* dr = dgates + b * ld_dg + j;
    if (!valid) {
        stb4(dr, zero); stb4(dr + D, zero); stb4(dr + 2 * D, zero); stb4(dr + 3 * D, zero);
        stb4(dc_prev + b * D + j, dcv);
        stb4(dh_pass + b * D + j, dhv);
        return;
    }
    const long long o = b * ld_d_b + (long long)t * ld_d_t + d_col0 + j;
    if (dH) dhv += ldb4(dH + o);
    if (dM) dcv += ldb4(dM + o);
    const float* gr = gates + b * 4 * D + j;
    const f32x4 gi = ldb4(gr), gf = ldb4(gr + D), gg = ldb4(gr + 2 * D), go = ldb4(gr + 3 * D);
    const f32x4 cp = ldb4(c_prev + b * D + j), cn = ldb4(c_new + b * D + j);
    f32x4 di, df, dg, dou, dcp;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float tc = tanhf(cn[e]);
        const float dct = dcv[e] + dhv[e] * go[e] * (1.f - tc * tc);
        di[e] = dct * gg[e] * gi[e] * (1.f - gi[e]);
        df[e] = dct * cp[e] * gf[e] * (1.f - gf[e]);
        dg[e] = dct * gi[e] * (1.f - gg[e] * gg[e]);
        dou[e] = dhv[e] * tc * go[e] * (1.f - go[e]);
        dcp[e] = dct * gf[e];
    }
    stb4(dr, di); stb4(dr + D, df); stb4(dr + 2 * D, dg); stb4(dr + 3 * D, dou);
    stb4(dc_prev + b * D + j, dcp);
    stb4(dh_pass + b * D + j, zero);
}

// validate a caller's addend list and copy it into the by-value kernel argument
int make_src_list(const SetSlabSrc* src, int n, SrcList* out) {
    out->n = 0;
    if (n < 0 || n > SET_MAX_SRC || (n > 0 && !src)) return SET_ERR_ARG;
    for (int i = 0; i < n; ++i) {
        const SetSlabSrc& e = src[i];
        if (e.nslab < 0 || e.rows < 0) return SET_ERR_ARG;
        if (e.nslab > 0 && (!e.p || !aligned16(e.p) || (e.ld & 3) || (e.slab_stride & 3))) return SET_ERR_UNSUPPORTED;
        out->s[i] = e;
    }
    out->n = n;
    return SET_OK;
}


// ---- merged launches of the training timestep loop (train_loop.hip): two kernels that wait for the same grouped product and
// whose outputs feed the same next product run as ONE launch (the step is bound by its ~12 dependent launches, not by bytes)
int lstm_gates_select_bwd_src(const float* dcn_base, const SetSlabSrc* src, const float* do_pre, const float* gates,
                              const float* c_prev, float* dgates, float* dc_prev, const float* dsel_base, const SetSlabSrc* ssrc,
                              const float* Mem, const float* alpha, float* dM, float* dalpha, int M, int T, int D, int acc_dM,
                              hipStream_t st) {
    if (!do_pre || !gates || !c_prev || !dgates || !dc_prev || !Mem || !alpha || !dM || !dalpha || M <= 0 || T <= 0 || D <= 0)
        return SET_ERR_ARG;
    if (D & 3) return SET_ERR_UNSUPPORTED;
    SrcList S, SS;
    SET_TRY(make_src_list(src, 1, &S));
    SET_TRY(make_src_list(ssrc, 1, &SS));
    const int nG = (int)(((long long)M * (D >> 2) + 255) / 256);
    hipLaunchKernelGGL(lstm_gates_select_bwd_k<true>, dim3(nG + M), dim3(256), 0, st, dcn_base, do_pre, gates, c_prev, dgates, dc_prev,
                       M, D, S, nG, dsel_base, Mem, alpha, dM, dalpha, T, acc_dM, SS);
    SET_LAUNCH_CHECK();
    return SET_OK;
}

static int att_bwd_check(const AttBwdArgs& P) {
    if (!P.alpha || !P.Vals || !P.att1 || !P.att2 || !P.w_full || !P.datt1 || !P.datt2 || !P.dwfull_part) return SET_ERR_ARG;
    if (P.L > ATTB_MAX || (P.A & 3) || (P.Dv & 3) || P.A > 1024 || P.Dv > 2048 || (P.ld_datt2 & 3) || P.ld_datt2 < P.A)
        return SET_ERR_UNSUPPORTED;
    return SET_OK;
}

int attention_pair_bwd_src(const SetSlabSrc* vsrc, const float* alpha_v, const float* X, const float* att1_v, const float* att2_v,
                           const float* wfull_v, float* datt1_v, float* datt2_v, float* dwf_v, float* de_v, int R, int F, int acc_v,
                           const SetSlabSrc* csrc, float* dctx_out, const float* dalpha_ext, const float* alpha_c, const float* H,
                           const float* att1_c, const float* att2_c, const float* wfull_c, float* datt1_c, float* datt2_c,
                           float* dwf_c, float* de_c, int Tc, int D, int acc_c, int M, int A, long long ld_datt2, hipStream_t st) {
    if (M <= 0) return SET_ERR_ARG;
    const AttBwdArgs PV{nullptr, nullptr, alpha_v, X, att1_v, att2_v, wfull_v, datt1_v, datt2_v, dwf_v, de_v, R, F, A, acc_v, ld_datt2,
                        nullptr};
    const AttBwdArgs PC{nullptr, dalpha_ext, alpha_c, H, att1_c, att2_c, wfull_c, datt1_c, datt2_c, dwf_c, de_c, Tc, D, A, acc_c,
                        ld_datt2, dctx_out};
    SET_TRY(att_bwd_check(PV));
    SET_TRY(att_bwd_check(PC));
    SrcList SV, SCp;
    SET_TRY(make_src_list(vsrc, 1, &SV));
    SET_TRY(make_src_list(csrc, 1, &SCp));
    hipLaunchKernelGGL(attention_pair_bwd_k<true>, dim3(2 * M), dim3(512), 0, st, PV, SV, PC, SCp, M);
    SET_LAUNCH_CHECK();
    return SET_OK;
}

int attention_ctxgate_bwd_src(const SetSlabSrc* src, const float* alpha, const float* values, const float* att1, const float* att2,
                              const float* w_full, float* datt1, float* datt2, float* dwfull_part, float* de, int M, int L, int Dv,
                              int A, int acc_datt1, long long ld_datt2, const SetSlabSrc* csrc, const float* zt, const float* s,
                              const float* t, float* dz, float* ds, float* dt, long long ld_out, int D, hipStream_t st) {
    if (!alpha || !values || !att1 || !att2 || !w_full || !datt1 || !datt2 || !dwfull_part || !zt || !s || !t || !dz || !ds || !dt ||
        M <= 0 || D <= 0)
        return SET_ERR_ARG;
    if (L > ATTB_MAX || (A & 3) || (Dv & 3) || A > 1024 || Dv > 2048 || (ld_datt2 & 3) || ld_datt2 < A || (D & 3) || (ld_out & 3) ||
        ld_out < D)
        return SET_ERR_UNSUPPORTED;
    SrcList S, SC;
    SET_TRY(make_src_list(src, 1, &S));
    SET_TRY(make_src_list(csrc, 1, &SC));
    const long long n4 = (long long)M * (D >> 2);
    const AttBwdArgs P{nullptr, nullptr, alpha, values, att1, att2, w_full, datt1, datt2, dwfull_part, de, L, Dv, A, acc_datt1, ld_datt2,
                       nullptr};
    const CtxGateArgs Cg{nullptr, zt, s, t, dz, ds, dt, n4, D >> 2, ld_out};
    hipLaunchKernelGGL(attention_ctxgate_bwd_k<true>, dim3(M + (unsigned)((n4 + 511) / 512)), dim3(512), 0, st, P, S, Cg, SC, M);
    SET_LAUNCH_CHECK();
    return SET_OK;
}

}  // namespace set

using namespace set;

extern "C" {

int set_gemm_group_slabs_f32(const SetGemmDesc* descs, int n, int a_kminor, int b_kminor, void* ws, size_t ws_bytes,
                             SetSlabSrc* out, void* stream) {
    if (!out) return SET_ERR_ARG;
    return gemm_gen_group(descs, n, a_kminor, b_kminor, ws, ws_bytes, (hipStream_t)stream, out);
}

int set_lstm_cell_bwd_src_f32(const SetSlabSrc* dh_src, int n_dh, const float* dc, const float* gates, const float* c_prev,
                              const float* c_new, float* dgates, float* dc_prev, int M, int D, void* stream) {
    if (!gates || !c_prev || !c_new || !dgates || !dc_prev || M <= 0 || D <= 0) return SET_ERR_ARG;
    if (D & 3) return SET_ERR_UNSUPPORTED;
    SrcList S;
    SET_TRY(make_src_list(dh_src, n_dh, &S));
    const long long n = (long long)M * (D >> 2);
    ProfScope ps("lstm_cell_bwd", (hipStream_t)stream, 0.0, 4.0 * M * D * 13.0);
    hipLaunchKernelGGL(lstm_cell_bwd_k<true>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const float*)nullptr,
                       dc, gates, c_prev, c_new, dgates, dc_prev, M, D, S);
    SET_LAUNCH_CHECK();
    return SET_OK;
}

int set_copy_gate_bwd_src_f32(const SetSlabSrc* dh_src, int n_dh, const float* dh_drop, int64_t ld_drop, float p, uint64_t seed,
                              uint64_t offset, const float* dadp, const float* ogate, int64_t ld_ogate, const float* adp,
                              const float* cg, const float* cmem, const float* c_new, float* du, float* dcm_direct,
                              float* dcn_direct, float* do_pre, int M, int D, void* stream) {
    if (!ogate || !adp || !cg || !cmem || !c_new || !du || !dcm_direct || !dcn_direct || !do_pre || M <= 0 || D <= 0)
        return SET_ERR_ARG;
    if (!(p >= 0.f) || !(p < 1.f)) return SET_ERR_ARG;
    if ((D & 3) || (ld_ogate & 3) || ld_ogate < D || (dh_drop && ((ld_drop & 3) || ld_drop < D || !aligned16(dh_drop))))
        return SET_ERR_UNSUPPORTED;
    SrcList S;
    SET_TRY(make_src_list(dh_src, n_dh, &S));
    const long long n = (long long)M * (D >> 2);
    hipLaunchKernelGGL(copy_gate_bwd_k<true>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const float*)nullptr,
                       dadp, ogate, adp, cg, cmem, c_new, du, dcm_direct, dcn_direct, do_pre, M, D, (long long)ld_ogate, S, dh_drop,
                       (long long)ld_drop, p, 1.0f / (1.0f - p), (unsigned long long)seed, (unsigned long long)offset);
    SET_LAUNCH_CHECK();
    return SET_OK;
}

int set_lstm_gates_bwd_src_f32(const float* dcn_base, const SetSlabSrc* src, int n_src, const float* do_pre, const float* gates,
                               const float* c_prev, float* dgates, float* dc_prev, int M, int D, void* stream) {
    if (!do_pre || !gates || !c_prev || !dgates || !dc_prev || M <= 0 || D <= 0) return SET_ERR_ARG;
    if (D & 3) return SET_ERR_UNSUPPORTED;
    SrcList S;
    SET_TRY(make_src_list(src, n_src, &S));
    const long long n = (long long)M * (D >> 2);
    hipLaunchKernelGGL(lstm_gates_bwd_k<true>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dcn_base, do_pre,
                       gates, c_prev, dgates, dc_prev, M, D, S);
    SET_LAUNCH_CHECK();
    return SET_OK;
}

int set_select_bwd_src_f32(const float* dsel_base, const SetSlabSrc* src, int n_src, const float* Mem, const float* alpha,
                           float* dM, float* dalpha, int M, int T, int D, int acc_dM, void* stream) {
    if (!Mem || !alpha || !dM || !dalpha || M <= 0 || T <= 0 || D <= 0) return SET_ERR_ARG;
    if (D & 3) return SET_ERR_UNSUPPORTED;
    SrcList S;
    SET_TRY(make_src_list(src, n_src, &S));
    hipLaunchKernelGGL(select_bwd_k<true>, dim3(M), dim3(256), 0, (hipStream_t)stream, dsel_base, Mem, alpha, dM, dalpha, T, D, acc_dM, S);
    SET_LAUNCH_CHECK();
    return SET_OK;
}

int set_context_gate_bwd_src_f32(const float* dout_base, const SetSlabSrc* src, int n_src, const float* zt, const float* s,
                                 const float* t, float* dz, float* ds, float* dt, int64_t ld_out, int M, int D, void* stream) {
    if (!zt || !s || !t || !dz || !ds || !dt || M <= 0 || D <= 0) return SET_ERR_ARG;
    if ((D & 3) || (ld_out & 3) || ld_out < D) return SET_ERR_UNSUPPORTED;
    SrcList S;
    SET_TRY(make_src_list(src, n_src, &S));
    const long long n = (long long)M * (D >> 2);
    hipLaunchKernelGGL(context_gate_bwd_k<true>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dout_base, zt, s,
                       t, dz, ds, dt, n, D >> 2, (long long)ld_out, S);
    SET_LAUNCH_CHECK();
    return SET_OK;
}

int set_attention_bwd_src_f32(const float* dctx_base, const SetSlabSrc* src, int n_src, float* dctx_out, const float* dalpha_ext,
                              const float* alpha, const float* values, const float* att1, const float* att2,
                              const float* w_full, float* datt1, float* datt2, float* dwfull_part, float* de, int M, int L,
                              int Dv, int A, int use_tanh, int acc_datt1, int64_t ld_datt2, void* stream) {
    if (!alpha || !values || !att1 || !att2 || !w_full || !datt1 || !datt2 || !dwfull_part || M <= 0) return SET_ERR_ARG;
    if (ld_datt2 <= 0) ld_datt2 = A;
    if (L > ATTB_MAX || (A & 3) || (Dv & 3) || A > 1024 || Dv > 2048 || (ld_datt2 & 3) || ld_datt2 < A) return SET_ERR_UNSUPPORTED;
    SrcList S;
    SET_TRY(make_src_list(src, n_src, &S));
    hipStream_t st = (hipStream_t)stream;
    if (use_tanh)
        hipLaunchKernelGGL((attention_bwd_wide_k<true, true>), dim3(M), dim3(512), 0, st, AttBwdArgs{dctx_base, dalpha_ext, alpha, values, att1, att2, w_full, datt1, datt2, dwfull_part, de, L, Dv, A, acc_datt1, (long long)ld_datt2, dctx_out}, S);
    else
        hipLaunchKernelGGL((attention_bwd_wide_k<false, true>), dim3(M), dim3(512), 0, st, AttBwdArgs{dctx_base, dalpha_ext, alpha, values, att1, att2, w_full, datt1, datt2, dwfull_part, de, L, Dv, A, acc_datt1, (long long)ld_datt2, dctx_out}, S);
    SET_LAUNCH_CHECK();
    return SET_OK;
}

int set_lstm_cell_bwd_f32(const float* dh, const float* dc, const float* gates, const float* c_prev, const float* c_new,
                          float* dgates, float* dc_prev, int M, int D, void* stream) {
    if (!gates || !c_prev || !c_new || !dgates || !dc_prev || M <= 0 || D <= 0) return SET_ERR_ARG;
    if (D & 3) return SET_ERR_UNSUPPORTED;
    const long long n = (long long)M * (D >> 2);
    ProfScope ps("lstm_cell_bwd", (hipStream_t)stream, 0.0, 4.0 * M * D * 13.0);
    hipLaunchKernelGGL(lstm_cell_bwd_k<false>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dh, dc, gates,
                       c_prev, c_new, dgates, dc_prev, M, D, SrcList());
    SET_LAUNCH_CHECK();
    return SET_OK;
}

int set_copy_gate_bwd_ld_f32(const float* dh, const float* dadp, const float* ogate, int64_t ld_ogate, const float* adp,
                             const float* cg, const float* cmem, const float* c_new, float* du, float* dcm_direct,
                             float* dcn_direct, float* do_pre, int M, int D, void* stream) {
    if (!ogate || !adp || !cg || !cmem || !c_new || !du || !dcm_direct || !dcn_direct || !do_pre || M <= 0 || D <= 0)
        return SET_ERR_ARG;
    if ((D & 3) || (ld_ogate & 3) || ld_ogate < D) return SET_ERR_UNSUPPORTED;
    const long long n = (long long)M * (D >> 2);
    hipLaunchKernelGGL(copy_gate_bwd_k<false>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dh, dadp, ogate,
                       adp, cg, cmem, c_new, du, dcm_direct, dcn_direct, do_pre, M, D, (long long)ld_ogate, SrcList(),
                       (const float*)nullptr, 0LL, 0.f, 1.f, 0ULL, 0ULL);
    SET_LAUNCH_CHECK();
    return SET_OK;
}

int set_copy_gate_bwd_f32(const float* dh, const float* dadp, const float* ogate, const float* adp, const float* cg,
                          const float* cmem, const float* c_new, float* du, float* dcm_direct, float* dcn_direct,
                          float* do_pre, int M, int D, void* stream) {
    return set_copy_gate_bwd_ld_f32(dh, dadp, ogate, D, adp, cg, cmem, c_new, du, dcm_direct, dcn_direct, do_pre, M, D, stream);
}

int set_lstm_gates_bwd_f32(const float* dcn, const float* do_pre, const float* gates, const float* c_prev, float* dgates,
                           float* dc_prev, int M, int D, void* stream) {
    if (!dcn || !do_pre || !gates || !c_prev || !dgates || !dc_prev || M <= 0 || D <= 0) return SET_ERR_ARG;
    if (D & 3) return SET_ERR_UNSUPPORTED;
    const long long n = (long long)M * (D >> 2);
    hipLaunchKernelGGL(lstm_gates_bwd_k<false>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dcn, do_pre,
                       gates, c_prev, dgates, dc_prev, M, D, SrcList());
    SET_LAUNCH_CHECK();
    return SET_OK;
}

int set_context_gate_bwd_ld_f32(const float* dout, const float* zt, const float* s, const float* t, float* dz, float* ds,
                                float* dt, int64_t ld_out, int M, int D, void* stream) {
    if (!dout || !zt || !s || !t || !dz || !ds || !dt || M <= 0 || D <= 0) return SET_ERR_ARG;
    if ((D & 3) || (ld_out & 3) || ld_out < D) return SET_ERR_UNSUPPORTED;
    const long long n = (long long)M * (D >> 2);
    hipLaunchKernelGGL(context_gate_bwd_k<false>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dout, zt, s,
                       t, dz, ds, dt, n, D >> 2, (long long)ld_out, SrcList());
    SET_LAUNCH_CHECK();
    return SET_OK;
}

int set_context_gate_bwd_f32(const float* dout, const float* zt, const float* s, const float* t, float* dz, float* ds,
                             float* dt, int M, int D, void* stream) {
    return set_context_gate_bwd_ld_f32(dout, zt, s, t, dz, ds, dt, D, M, D, stream);
}

// dvalues[b, l, :] (+)= sum_t alpha[t, b, l] * dctx[t, b, :]  — the attended rows' gradient of ALL timesteps of a sequence at
// once (alpha (T,B,L), dctx (T,B,Dv) are per-sequence logs); replaces T read-modify-write passes over dvalues.
__global__ void __launch_bounds__(256) attention_dvalues_k(const float* alpha, const float* dctx, float* dV, int T, int B, int L,
                                                           int Dv, int accumulate) {
    __shared__ float s_a[32 * 64];                          // alpha[:, b, :] (T <= 64 steps x L <= 32 rows per pass)
    const int b = blockIdx.x, tid = threadIdx.x;
    const int d = (blockIdx.y * 256 + tid) * 4;
    for (int l0 = 0; l0 < L; l0 += 32) {
        const int ln = L - l0 < 32 ? L - l0 : 32;
        __syncthreads();
        for (int i = tid; i < T * ln; i += 256) {
            const int t = i / ln, l = i - t * ln;
            s_a[t * 32 + l] = alpha[((long long)t * B + b) * L + l0 + l];
        }
        __syncthreads();
        if (d < Dv) {
            for (int l = 0; l < ln; ++l) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                for (int t = 0; t < T; ++t) acc += ldb4(dctx + ((long long)t * B + b) * Dv + d) * s_a[t * 32 + l];
                float* o = dV + ((long long)b * L + l0 + l) * Dv + d;
                stb4(o, accumulate ? ldb4(o) + acc : acc);
            }
        }
    }
}

extern "C" int set_attention_dvalues_f32(const float* alpha, const float* dctx, float* dvalues, int T, int B, int L, int Dv,
                                         int accumulate, void* stream) {
    if (!alpha || !dctx || !dvalues || T <= 0 || B <= 0 || L <= 0 || Dv <= 0) return SET_ERR_ARG;
    if (T > 64 || (Dv & 3)) return SET_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(attention_dvalues_k, dim3(B, cdiv(Dv, 1024)), dim3(256), 0, (hipStream_t)stream, alpha, dctx, dvalues, T,
                       B, L, Dv, accumulate);
    SET_LAUNCH_CHECK();
    return SET_OK;
}

int set_attention_bwd_acc_f32(const float* dctx, const float* dalpha_ext, const float* alpha, const float* values,
                              const float* att1, const float* att2, const float* w_full, float* datt1, float* datt2,
                              float* dwfull_part, float* dvalues, float* de, int M, int L, int Dv, int A, int use_tanh,
                              int acc_datt1, int acc_dvalues, int64_t ld_datt2, void* stream) {
    if (!dctx || !alpha || !values || !att1 || !att2 || !w_full || !datt1 || !datt2 || !dwfull_part || M <= 0)
        return SET_ERR_ARG;
    if (ld_datt2 <= 0) ld_datt2 = A;
    if (L > ATTB_MAX || (A & 3) || (Dv & 3) || A > 1024 || Dv > 2048 || (ld_datt2 & 3) || ld_datt2 < A) return SET_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    static const int wide = env_int("SET_ATT_BWD_WIDE", 1);
    if (wide && !dvalues) {                                  // 512 threads per row, several rows of every operand in flight
        const AttBwdArgs P{dctx, dalpha_ext, alpha, values, att1, att2, w_full, datt1, datt2, dwfull_part, de, L, Dv, A, acc_datt1,
                           (long long)ld_datt2, nullptr};
        if (use_tanh) hipLaunchKernelGGL((attention_bwd_wide_k<true, false>), dim3(M), dim3(512), 0, st, P, SrcList());
        else hipLaunchKernelGGL((attention_bwd_wide_k<false, false>), dim3(M), dim3(512), 0, st, P, SrcList());
        SET_LAUNCH_CHECK();
        return SET_OK;
    }
    if (use_tanh)
        hipLaunchKernelGGL(attention_bwd_k<true>, dim3(M), dim3(256), 0, st, dctx, dalpha_ext, alpha, values, att1, att2,
                           w_full, datt1, datt2, dwfull_part, dvalues, de, L, Dv, A, acc_datt1, acc_dvalues, (long long)ld_datt2);
    else
        hipLaunchKernelGGL(attention_bwd_k<false>, dim3(M), dim3(256), 0, st, dctx, dalpha_ext, alpha, values, att1, att2,
                           w_full, datt1, datt2, dwfull_part, dvalues, de, L, Dv, A, acc_datt1, acc_dvalues, (long long)ld_datt2);
    SET_LAUNCH_CHECK();
    return SET_OK;
}

int set_attention_bwd_f32(const float* dctx, const float* dalpha_ext, const float* alpha, const float* values,
                          const float* att1, const float* att2, const float* w_full, float* datt1, float* datt2,
                          float* dwfull_part, float* dvalues, float* de, int M, int L, int Dv, int A, int use_tanh,
                          void* stream) {
    return set_attention_bwd_acc_f32(dctx, dalpha_ext, alpha, values, att1, att2, w_full, datt1, datt2, dwfull_part,
                                     dvalues, de, M, L, Dv, A, use_tanh, 0, 0, A, stream);
}

int set_select_bwd_acc_f32(const float* dsel, const float* Mem, const float* alpha, float* dM, float* dalpha, int M,
                           int T, int D, int acc_dM, void* stream) {
    if (!dsel || !Mem || !alpha || !dM || !dalpha || M <= 0 || T <= 0 || D <= 0) return SET_ERR_ARG;
    if (D & 3) return SET_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(select_bwd_k<false>, dim3(M), dim3(256), 0, (hipStream_t)stream, dsel, Mem, alpha, dM, dalpha, T, D, acc_dM, SrcList());
    SET_LAUNCH_CHECK();
    return SET_OK;
}

int set_select_bwd_f32(const float* dsel, const float* Mem, const float* alpha, float* dM, float* dalpha, int M, int T,
                       int D, void* stream) {
    return set_select_bwd_acc_f32(dsel, Mem, alpha, dM, dalpha, M, T, D, 0, stream);
}

// SelectC.forward with soft = True (editnet.py:419-420): the attention weights themselves — sel = sum_t alpha_t M_t — and its
// backward: dM[b, t] = alpha[b, t] dsel[b], dalpha[b, t] = <dsel[b], M[b, t]>.  One workgroup per row.
int set_select_soft_f32(const float* Mem, const float* alpha, float* sel, int M, int T, int D, void* stream) {
    if (!Mem || !alpha || !sel || M <= 0 || T <= 0 || D <= 0) return SET_ERR_ARG;
    if (D & 3) return SET_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(select_soft_k, dim3(M), dim3(256), 0, (hipStream_t)stream, Mem, alpha, sel, T, D);
    SET_LAUNCH_CHECK();
    return SET_OK;
}

int set_select_soft_bwd_f32(const float* dsel, const float* Mem, const float* alpha, float* dM, float* dalpha, int M, int T,
                            int D, void* stream) {
    if (!dsel || !Mem || !alpha || !dM || !dalpha || M <= 0 || T <= 0 || D <= 0) return SET_ERR_ARG;
    if (D & 3) return SET_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(select_soft_bwd_k, dim3(M), dim3(256), 0, (hipStream_t)stream, dsel, Mem, alpha, dM, dalpha, T, D);
    SET_LAUNCH_CHECK();
    return SET_OK;
}

int set_gemm_f32(const float* A, long long lda, int a_kminor, const float* B, long long ldb, int b_kminor, float* C,
                 long long ldc, int M, int N, int K, int accumulate, void* ws, size_t ws_bytes, void* stream) {
    return gemm_gen(A, lda, a_kminor, B, ldb, b_kminor, C, ldc, M, N, K, accumulate, ws, ws_bytes,
                    (hipStream_t)stream);
}

int set_gemm_group_f32(const SetGemmDesc* descs, int n, int a_kminor, int b_kminor, void* ws, size_t ws_bytes,
                       void* stream) {
    return gemm_gen_group(descs, n, a_kminor, b_kminor, ws, ws_bytes, (hipStream_t)stream);
}

size_t set_encoder_cell_workspace_bytes(int B, int D) {
    if (B <= 0 || D <= 0) return 0;
    return round_up((size_t)GEMM_MAX_KSPLIT * B * 4 * D * sizeof(float), 256) + 256;
}

int set_encoder_cell_train_f32(const float* xg, int64_t ld_xg_row, int64_t ld_xg_t, const float* h, const float* c,
                               const float* w_hh, const float* b_hh, const int64_t* lens, int t, float* h_out, float* c_out,
                               float* H, float* Mem, float* Hprev, int64_t ld_out_b, int64_t ld_out_t, int out_col0,
                               float* gates, int B, int D, void* ws, size_t ws_bytes, void* stream) {
    if (!xg || !h || !c || !w_hh || !lens || !h_out || !c_out || !H || !gates || B <= 0 || D <= 0 || t < 0) return SET_ERR_ARG;
    if ((D % GEMM_BK) || (ld_xg_row & 3) || (ld_xg_t & 3) || (ld_out_b & 3) || (ld_out_t & 3) || (out_col0 & 3))
        return SET_ERR_UNSUPPORTED;
    if (!ws || !aligned16(ws) || ws_bytes < set_encoder_cell_workspace_bytes(B, D) - 256) return SET_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    // one launch: the contraction with the cell update as its epilogue (the decode path's encoder kernel, keeping what
    // the backward needs), when the hidden size allows its 128-wide k-tiles; else split-K slabs + the pointwise kernel
    static const int fused = env_int("SET_ENC_TRAIN_FUSED", 1);
    if (fused && D % 128 == 0 && h_out != h && c_out != c)
        return fused_encoder_step_train(h, c, h_out, c_out, w_hh, xg, ld_xg_row, ld_xg_t, b_hh, lens, t, H, Mem, Hprev, ld_out_b,
                                        ld_out_t, out_col0, gates, B, D, st);
    GemmProb p = slab_prob((float*)ws, B, 4 * D, B);
    p.add(h, D, w_hh, D, D);
    plan_ksplit(&p, 1, gemm_target_wgs());
    SET_TRY(gemm_group(&p, 1, st, "gemm:enc h2h (train)"));
    const long long n = (long long)B * (D >> 2);
    ProfScope ps("enc_cell_train", st, 0.0, 4.0 * B * D * (4.0 * p.ksplit + 16.0));
    hipLaunchKernelGGL(enc_cell_train_k, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, slabs_of(p), xg, ld_xg_row,
                       ld_xg_t, b_hh, lens, t, h, c, h_out, c_out, H, Mem, Hprev, ld_out_b, ld_out_t, out_col0, gates, B, D);
    SET_LAUNCH_CHECK();
    return SET_OK;
}

int set_encoder_cell_bwd_f32(const float* dh, const float* dc, const float* dH, const float* dM, int64_t ld_d_b,
                             int64_t ld_d_t, int d_col0, const int64_t* lens, int t, const float* gates,
                             const float* c_prev, const float* c_new, float* dgates, int64_t ld_dg, float* dc_prev,
                             float* dh_pass, int B, int D, void* stream) {
    if (!lens || !gates || !c_prev || !c_new || !dgates || !dc_prev || !dh_pass || B <= 0 || D <= 0 || t < 0) return SET_ERR_ARG;
    if ((D & 3) || (ld_d_b & 3) || (ld_d_t & 3) || (d_col0 & 3) || (ld_dg & 3)) return SET_ERR_UNSUPPORTED;
    const long long n = (long long)B * (D >> 2);
    ProfScope ps("enc_cell_bwd", (hipStream_t)stream, 0.0, 4.0 * B * D * 16.0);
    hipLaunchKernelGGL(enc_cell_bwd_k, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dh, dc, dH, dM,
                       ld_d_b, ld_d_t, d_col0, lens, t, gates, c_prev, c_new, dgates, ld_dg, dc_prev, dh_pass, B, D);
    SET_LAUNCH_CHECK();
    return SET_OK;
}

}  // extern "C"
