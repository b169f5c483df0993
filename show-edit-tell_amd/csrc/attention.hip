// Additive-attention kernels of the decode step (one workgroup of 4 waves per sample):
//   caption attention  (CaptionAttentionC editnet.py:370-376 + SelectC :409-420 ; DCNet dcnet.py:261-268)
//   visual attention   (VisualAttentionC  editnet.py:443-446 ; adaptive editnet_adaptive.py:449-456)
// Both are HBM/L2 streaming kernels: the loop-invariant projection att1 (hoisted to the prologue
// in eval mode) is read once per step, scored against the per-step decoder projection att2,
// soft-maxed inside one wavefront with 64-lane shuffles, and the context is accumulated over
// raw rows with one float4 column per thread (every feature element is read exactly once, so
// nothing is staged through LDS: LDS only holds the <=R attention weights).
#include "set_common.h"

namespace set {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 ld4a(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
// streamed operands (region features X, hoisted caption projections P): read once per timestep by one workgroup and, with
// several batches in flight, far larger than L2 + Infinity Cache -> non-temporal loads, so that they do not evict the
// weights the GEMM launches re-read every timestep (same-box A/B: GEMM launches -1.2 %, decode rate +0.3 ... +0.7 %;
// -DSET_ATT_PLAIN_LOADS restores ordinary loads)
__device__ __forceinline__ f32x4 ld4s(const float* p) {
#ifdef SET_ATT_PLAIN_LOADS
    return *reinterpret_cast<const f32x4*>(p);
#else
    return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
#endif
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

constexpr int ATT_MAX_ROWS = 256;   // T (caption) / R (regions) upper bound held in LDS

// softmax over sc[0..n) in place (wave 0 of the block), also returns the first arg-max.
// masked entries were filled with -1e10 by the caller (masked_fill, editnet.py:374).
__device__ __forceinline__ void block_softmax(float* sc, int n, int tid, int* argmax_out) {
    if (tid < 64) {
        float m = -INFINITY;
        for (int i = tid; i < n; i += 64) m = fmaxf(m, sc[i]);
        m = wave_max(m);
        float s = 0.f;
        for (int i = tid; i < n; i += 64) { const float e = expf(sc[i] - m); sc[i] = e; s += e; }
        s = wave_sum(s);
        float best = -1.f; int bi = 0x7fffffff;
        for (int i = tid; i < n; i += 64) {
            const float a = sc[i] / s;
            sc[i] = a;
            if (a > best) { best = a; bi = i; }        // ascending i per lane: keeps the first max
        }
        if (argmax_out) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const float ob = __shfl_xor(best, o);
                const int oi = __shfl_xor(bi, o);
                if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
            }
            if (tid == 0) *argmax_out = bi;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// caption attention + hard selection.  grid = M samples, 256 threads.
//   e_t   = w_full . tanh(att1_c[b,t,:] + att2_c[b,:] + dec_bias) + b_full ; masked -> -1e10
//   alpha = softmax_t(e) ; ctx = sum_t alpha_t H[b,t,:]
//   sel   = Mem[b,j*,:] * (alpha_j* + (1 - alpha_j*)),  j* = first argmax_t alpha     (Mem may be NULL)
// ---------------------------------------------------------------------------------------------
struct CapAttArgs {
    const float* att1_c; Slabs att2_c; const float* dec_bias; const float* w_full; const float* b_full;
    const float* mask; const float* H; const float* Mem; float* ctx; float* sel; float* alpha_out;
    int T, Dh, A;
    // Hoisted-projection mode (P != NULL; EditNet eval step): the contractions of the context (editnet.py:378-379)
    // are linear in ctx = sum_t alpha_t H_t, so P[b,t] = [context_gate.W[:, 2D:3D] H_t | sc_affine.W H_t] (B,T,2Dh) is
    // computed once per sequence and this kernel accumulates sum_t alpha_t P[b,t] instead of ctx, then applies the
    // context gate itself:  out = zt*tanh(s) + (1-zt)*tanh(tc),  zt = sig((cg_ab + zc) + b_gate),  s = sc + b_sc.
    // Q[b,t] = gate_cmem.W Mem_t (B,T,Dh) likewise turns gate_cmem(sel) (editnet.py:281) into a row gather.
    const float* P; const float* Q; float* cmem_out; float* gated_out;
    Slabs cg_ab, tc; RowGather gz, gtc; const float *b_gate, *b_sc, *b_tc;
    float* att2_out;      // (M,A) or NULL: decoder-side projection incl. its bias, as used for the scores (kept for the backward)
};
struct VisAttArgs {
    const float* att1; Slabs att2; const float* dec_bias; const float* w_full; const float* b_full;
    const float* X; const float* rmask; float* ctx; float* alpha_out;
    int R, F, A, fcols, fsn;
    int prefetch;
    float* att2_out;      // (M,A) or NULL, see CapAttArgs
};

__device__ __forceinline__ void caption_attention_body(const CapAttArgs& P, int b, float* sc, int* s_arg_p) {
    const float* att1_c = P.att1_c; const Slabs att2_c = P.att2_c; const float* dec_bias = P.dec_bias;
    const float* w_full = P.w_full; const float* b_full = P.b_full; const float* mask = P.mask;
    const float* H = P.H; const float* Mem = P.Mem; float* ctx = P.ctx; float* sel = P.sel;
    float* alpha_out = P.alpha_out; const int T = P.T, Dh = P.Dh, A = P.A;
    int& s_arg = *s_arg_p;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // per-lane slice of att2_c + bias and w_full: a = lane*4 + 256*q
    const int nq = (A + 255) / 256;
    f32x4 a2[2], wf[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        a2[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
        wf[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int a = lane * 4 + 256 * q;
        if (q < nq && a < A) {
            f32x4 v = ld4a(att2_c.p + (long long)b * att2_c.ld + a);
            for (int i = 1; i < att2_c.n; ++i) v += ld4a(att2_c.p + (long long)i * att2_c.stride + (long long)b * att2_c.ld + a);
            a2[q] = v + ld4a(dec_bias + a);
            wf[q] = ld4a(w_full + a);
            if (P.att2_out && wave == 0) *reinterpret_cast<f32x4*>(P.att2_out + (long long)b * A + a) = a2[q];
        }
    }
    const float bf = b_full[0];
    // each wave scores rows wave, wave+4, ...; RB rows are loaded before any is reduced so that their
    // HBM/L2 round trips overlap (the reductions are 6-step cross-lane chains)
    constexpr int RB = 5;
    for (int t0 = wave; t0 < T; t0 += 4 * RB) {
        f32x4 v[RB][2];
#pragma unroll
        for (int u = 0; u < RB; ++u) {
            const int t = t0 + 4 * u;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int a = lane * 4 + 256 * q;
                v[u][q] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (t < T && q < nq && a < A) v[u][q] = ld4a(att1_c + ((long long)b * T + t) * A + a);
            }
        }
#pragma unroll
        for (int u = 0; u < RB; ++u) {
            const int t = t0 + 4 * u;
            if (t >= T) break;
            float s = 0.f;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int a = lane * 4 + 256 * q;
                if (q < nq && a < A) {
                    const f32x4 x = v[u][q] + a2[q];
                    s += wf[q][0] * tanhf(x[0]) + wf[q][1] * tanhf(x[1]) + wf[q][2] * tanhf(x[2]) + wf[q][3] * tanhf(x[3]);
                }
            }
            s = wave_sum(s);
            if (lane == 0) sc[t] = (mask[(long long)b * T + t] == 0.f) ? -1e10f : (s + bf);
        }
    }
    __syncthreads();
    block_softmax(sc, T, tid, &s_arg);
    __syncthreads();
    if (alpha_out)
        for (int t = tid; t < T; t += 256) alpha_out[(long long)b * T + t] = sc[t];
    const int js = s_arg;
    const float aj = sc[js];
    const float wj = aj * 1.f + (1.f - aj);            // the reference's fp32 expression (editnet.py:417-418)
    if (P.P) {
        // ---- hoisted projections + fused context gate
        for (int d = tid * 4; d < Dh; d += 1024) {
            // operands that do not depend on the attention weights first: their latency overlaps the P stream
            const long long m = b;
            f32x4 pre0 = {0.f, 0.f, 0.f, 0.f}, pre1 = {0.f, 0.f, 0.f, 0.f};
            for (int i = 0; i < P.cg_ab.n; ++i) pre0 += ld4a(P.cg_ab.p + (long long)i * P.cg_ab.stride + m * P.cg_ab.ld + d);
            for (int i = 0; i < P.tc.n; ++i) pre1 += ld4a(P.tc.p + (long long)i * P.tc.stride + m * P.tc.ld + d);
            if (P.gz.tab) pre0 += ld4a(P.gz.row(m) + d);
            if (P.gtc.tab) pre1 += ld4a(P.gtc.row(m) + d);
            const f32x4 bg = ld4a(P.b_gate + d), bs = ld4a(P.b_sc + d), bt = ld4a(P.b_tc + d);
            const f32x4 mrow = Mem ? ld4a(Mem + ((long long)b * T + js) * Dh + d) : pre0;
            const f32x4 qrow = P.Q ? ld4a(P.Q + ((long long)b * T + js) * Dh + d) : pre0;
            const float* pp = P.P + (long long)b * T * 2 * Dh + d;
            f32x4 zc = {0.f, 0.f, 0.f, 0.f}, sv = {0.f, 0.f, 0.f, 0.f};
            int t = 0;
            for (; t + 4 <= T; t += 4) {                      // 8 loads in flight; accumulation stays in t order
                f32x4 vz[4], vs[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    vz[u] = ld4s(pp + (long long)(t + u) * 2 * Dh);
                    vs[u] = ld4s(pp + (long long)(t + u) * 2 * Dh + Dh);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) { zc += vz[u] * sc[t + u]; sv += vs[u] * sc[t + u]; }
            }
            for (; t < T; ++t) {
                zc += ld4a(pp + (long long)t * 2 * Dh) * sc[t];
                sv += ld4a(pp + (long long)t * 2 * Dh + Dh) * sc[t];
            }
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                // reference: ONE Linear over cat([word, h1, ctx]): the [word,h1] part first, then the ctx part, then bias
                const float z = (pre0[e] + zc[e]) + bg[e];
                const float zt = 1.f / (1.f + expf(-z));
                o[e] = zt * tanhf(sv[e] + bs[e]) + (1.f - zt) * tanhf(pre1[e] + bt[e]);
            }
            *reinterpret_cast<f32x4*>(P.gated_out + (long long)b * Dh + d) = o;
            if (Mem) *reinterpret_cast<f32x4*>(sel + (long long)b * Dh + d) = mrow * wj;
            if (P.Q) *reinterpret_cast<f32x4*>(P.cmem_out + (long long)b * Dh + d) = qrow * wj;
        }
        return;
    }
    for (int d = tid * 4; d < Dh; d += 1024) {
        const float* hp = H + (long long)b * T * Dh + d;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        int t = 0;
        for (; t + 8 <= T; t += 8) {                      // 8 rows in flight; accumulation stays in t order
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = ld4a(hp + (long long)(t + u) * Dh);
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += v[u] * sc[t + u];
        }
        for (; t < T; ++t) acc += ld4a(hp + (long long)t * Dh) * sc[t];
        *reinterpret_cast<f32x4*>(ctx + (long long)b * Dh + d) = acc;
        if (Mem) {
            const f32x4 m = ld4a(Mem + ((long long)b * T + js) * Dh + d);
            *reinterpret_cast<f32x4*>(sel + (long long)b * Dh + d) = m * wj;
        }
    }
}

__global__ void __launch_bounds__(256) caption_attention_k(const CapAttArgs P) {
    __shared__ float sc[ATT_MAX_ROWS];
    __shared__ int s_arg;
    caption_attention_body(P, blockIdx.x, sc, &s_arg);
}

int caption_attention(const float* att1_c, Slabs att2_c, const float* dec_bias, const float* w_full,
                      const float* b_full, const float* mask, const float* H, const float* Mem, float* ctx,
                      float* sel, float* alpha_out, int M, int T, int Dh, int A, hipStream_t s, float* att2_out) {
    if (T > ATT_MAX_ROWS || A > 512 || (A & 3) || (Dh & 3)) return SET_ERR_UNSUPPORTED;
    if (M <= 0) return SET_OK;
    ProfScope ps("caption_attention", s, 0.0, 4.0 * M * ((double)T * A + (double)T * Dh + 3.0 * Dh + att2_c.n * A));
    CapAttArgs P{};
    P.att1_c = att1_c; P.att2_c = att2_c; P.dec_bias = dec_bias; P.w_full = w_full; P.b_full = b_full; P.mask = mask;
    P.H = H; P.Mem = Mem; P.ctx = ctx; P.sel = sel; P.alpha_out = alpha_out; P.T = T; P.Dh = Dh; P.A = A;
    P.att2_out = att2_out;
    hipLaunchKernelGGL(caption_attention_k, dim3(M), dim3(256), 0, s, P);
    SET_LAUNCH_CHECK();
    return SET_OK;
}

// ---------------------------------------------------------------------------------------------
// visual attention.  grid = (M samples, FS feature slices), 256 threads; every slice recomputes
// the R scores (R*A MACs, att1 comes from L2 after the first slice) and owns F/FS output columns.
//   e_r   = w_full . relu(att1[b,r,:] + att2[b,:] + dec_bias) + b_full     (ReLU, not tanh)
//   rmask (adaptive only): e_r = -1e10 where rmask[b,r] == 0  (editnet_adaptive.py:453)
//   alpha = softmax_r(e) ; ctx = sum_r alpha_r X[b,r,:]   (context over the RAW features)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void visual_attention_body(const VisAttArgs& P, int b, int fs, float* sc) {
    const float* att1 = P.att1; const Slabs att2 = P.att2; const float* dec_bias = P.dec_bias;
    const float* w_full = P.w_full; const float* b_full = P.b_full; const float* X = P.X; const float* rmask = P.rmask;
    float* ctx = P.ctx; float* alpha_out = P.alpha_out; const int R = P.R, F = P.F, A = P.A, fcols = P.fcols;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nq = (A + 255) / 256;
    f32x4 a2[2], wf[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        a2[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
        wf[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int a = lane * 4 + 256 * q;
        if (q < nq && a < A) {
            f32x4 v = ld4a(att2.p + (long long)b * att2.ld + a);
            for (int i = 1; i < att2.n; ++i) v += ld4a(att2.p + (long long)i * att2.stride + (long long)b * att2.ld + a);
            a2[q] = v + ld4a(dec_bias + a);
            wf[q] = ld4a(w_full + a);
            if (P.att2_out && wave == 0 && fs == 0) *reinterpret_cast<f32x4*>(P.att2_out + (long long)b * A + a) = a2[q];
        }
    }
    const float bf = b_full[0];
    // the first context batch (12 regions of this thread's feature columns) does not depend on the scores:
    // request it now so its HBM latency overlaps the scoring phase
    constexpr int CB = 12;
    const int f_first = fs * fcols + tid * 4;
    const bool pre_ok = P.prefetch && f_first < fs * fcols + fcols && f_first < F && R >= CB;
    f32x4 xpre[CB];
    if (pre_ok) {
        const float* xp0 = X + (long long)b * R * F + f_first;
#pragma unroll
        for (int u = 0; u < CB; ++u) xpre[u] = ld4s(xp0 + (long long)u * F);
    }
    constexpr int RB = 9;                                 // R = 36 regions -> one batch per wave
    for (int r0 = wave; r0 < R; r0 += 4 * RB) {
        f32x4 v[RB][2];
#pragma unroll
        for (int u = 0; u < RB; ++u) {
            const int r = r0 + 4 * u;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int a = lane * 4 + 256 * q;
                v[u][q] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (r < R && q < nq && a < A) v[u][q] = ld4a(att1 + ((long long)b * R + r) * A + a);
            }
        }
#pragma unroll
        for (int u = 0; u < RB; ++u) {
            const int r = r0 + 4 * u;
            if (r >= R) break;
            float s = 0.f;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int a = lane * 4 + 256 * q;
                if (q < nq && a < A) {
                    const f32x4 x = v[u][q] + a2[q];
                    s += wf[q][0] * fmaxf(x[0], 0.f) + wf[q][1] * fmaxf(x[1], 0.f) + wf[q][2] * fmaxf(x[2], 0.f) +
                         wf[q][3] * fmaxf(x[3], 0.f);
                }
            }
            s = wave_sum(s);
            if (lane == 0) sc[r] = (rmask && rmask[(long long)b * R + r] == 0.f) ? -1e10f : (s + bf);
        }
    }
    __syncthreads();
    block_softmax(sc, R, tid, nullptr);
    __syncthreads();
    if (alpha_out && fs == 0)
        for (int r = tid; r < R; r += 256) alpha_out[(long long)b * R + r] = sc[r];
    const int f0 = fs * fcols;
    for (int f = f0 + tid * 4; f < f0 + fcols && f < F; f += 1024) {
        const float* xp = X + (long long)b * R * F + f;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        int r = 0;
        if (pre_ok && f == f_first) {                     // the batch requested before the scoring phase
#pragma unroll
            for (int u = 0; u < CB; ++u) acc += xpre[u] * sc[u];
            r = CB;
        }
        for (; r + 12 <= R; r += 12) {                    // 12 regions in flight; accumulation stays in r order
            f32x4 v[12];
#pragma unroll
            for (int u = 0; u < 12; ++u) v[u] = ld4s(xp + (long long)(r + u) * F);
#pragma unroll
            for (int u = 0; u < 12; ++u) acc += v[u] * sc[r + u];
        }
        for (; r < R; ++r) acc += ld4s(xp + (long long)r * F) * sc[r];
        *reinterpret_cast<f32x4*>(ctx + (long long)b * F + f) = acc;
    }
}

__global__ void __launch_bounds__(256) visual_attention_k(const VisAttArgs P) {
    __shared__ float sc[ATT_MAX_ROWS];
    visual_attention_body(P, blockIdx.x / P.fsn, blockIdx.x % P.fsn, sc);
}

// Both attentions of one timestep in ONE launch (they depend only on the phase-B projections and
// are independent of each other): workgroups [0, Mv*fsn) stream the image regions, the rest score
// the previous caption.  Saves one kernel boundary + one ~4.5 us launch floor per timestep and lets
// the MFMA-free caption attention overlap the HBM-bound region streaming.
__global__ void __launch_bounds__(256) step_attention_k(const VisAttArgs V, const CapAttArgs C, int nvis) {
    __shared__ float sc[ATT_MAX_ROWS];
    __shared__ int s_arg;
    if ((int)blockIdx.x < nvis)
        visual_attention_body(V, blockIdx.x / V.fsn, blockIdx.x % V.fsn, sc);
    else
        caption_attention_body(C, blockIdx.x - nvis, sc, &s_arg);
}

static int vis_fsn(int M, int F) {
    // enough workgroups to cover the chip: split the feature axis while slices stay >= 1024 columns
    // measured at B=128: one slice per sample (128 + 128 workgroups) beats 2 or 4 slices (each slice recomputes the scores)
    static const int min_cols = env_int("SET_ATT_MIN_COLS", 2048);
    int fsn = 1;
    while (M * fsn < 512 && F / (fsn * 2) >= min_cols && (F % (fsn * 2 * 4)) == 0) fsn *= 2;
    return fsn;
}

int step_attention(const float* att1, Slabs att2, const float* v_dec_bias, const float* v_w_full, const float* v_b_full,
                   const float* X, const float* rmask, float* v_ctx, float* v_alpha, int R, int F,
                   const float* att1_c, Slabs att2_c, const float* c_dec_bias, const float* c_w_full,
                   const float* c_b_full, const float* mask, const float* H, const float* Mem, float* c_ctx, float* sel,
                   float* c_alpha, int T, int Dh, int A, int M, hipStream_t s, const CapHoist* hoist, float* v_att2_out,
                   float* c_att2_out) {
    if (R > ATT_MAX_ROWS || T > ATT_MAX_ROWS || A > 512 || (A & 3) || (F & 3) || (Dh & 3)) return SET_ERR_UNSUPPORTED;
    if (M <= 0) return SET_OK;
    const int fsn = vis_fsn(M, F);
    static const int prefetch = env_int("SET_ATT_PREFETCH", 1);
    VisAttArgs V{att1, att2, v_dec_bias, v_w_full, v_b_full, X, rmask, v_ctx, v_alpha, R, F, A, F / fsn, fsn, prefetch, v_att2_out};
    CapAttArgs C{};
    C.att1_c = att1_c; C.att2_c = att2_c; C.dec_bias = c_dec_bias; C.w_full = c_w_full; C.b_full = c_b_full; C.mask = mask;
    C.H = H; C.Mem = Mem; C.ctx = c_ctx; C.sel = sel; C.alpha_out = c_alpha; C.T = T; C.Dh = Dh; C.A = A;
    C.att2_out = c_att2_out;
    if (hoist && hoist->P) {
        C.P = hoist->P; C.Q = hoist->Q; C.cmem_out = hoist->cmem_out; C.gated_out = hoist->gated_out;
        C.cg_ab = hoist->cg_ab; C.tc = hoist->tc; C.gz = hoist->gz; C.gtc = hoist->gtc;
        C.b_gate = hoist->b_gate; C.b_sc = hoist->b_sc; C.b_tc = hoist->b_tc;
    }
    const double cap_rows = (hoist && hoist->P) ? 2.0 * T * Dh + (double)(hoist->cg_ab.n + hoist->tc.n + 6) * Dh
                                                : (double)T * Dh + 3.0 * Dh;
    ProfScope ps("step_attention", s, 0.0,
                 4.0 * M * ((double)R * A * fsn + (double)R * F + F + (double)T * A + cap_rows));
    hipLaunchKernelGGL(step_attention_k, dim3(M * fsn + M), dim3(256), 0, s, V, C, M * fsn);
    SET_LAUNCH_CHECK();
    return SET_OK;
}

int visual_attention(const float* att1, Slabs att2, const float* dec_bias, const float* w_full,
                     const float* b_full, const float* X, const float* rmask, float* ctx, float* alpha_out, int M,
                     int R, int F, int A, hipStream_t s) {
    if (R > ATT_MAX_ROWS || A > 512 || (A & 3) || (F & 3)) return SET_ERR_UNSUPPORTED;
    if (M <= 0) return SET_OK;
    const int fsn = vis_fsn(M, F);
    VisAttArgs P{att1, att2, dec_bias, w_full, b_full, X, rmask, ctx, alpha_out, R, F, A, F / fsn, fsn, 1, nullptr};
    ProfScope ps("visual_attention", s, 0.0, 4.0 * M * ((double)R * A + (double)R * F + F + att2.n * A));
    hipLaunchKernelGGL(visual_attention_k, dim3(M * fsn), dim3(256), 0, s, P);
    SET_LAUNCH_CHECK();
    return SET_OK;
}

// ---------------------------------------------------------------------------------------------
// adaptive features: rmask[b,r] = 1 iff r < n_b and relu-embedded row fe[b,r,:] sums to non-zero,
// where n_b = #rows of X[b] with non-zero sum (editnet_adaptive.py:440-449; pack_padded_sequence
// keeps the FIRST n_b rows).  One wave per (b,r) row; n_b via a block-wide count.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) region_masks_k(const float* X, const float* fe, float* rmask, int R, int F,
                                                      int D) {
    __shared__ int nvalid;
    __shared__ float fsum[ATT_MAX_ROWS];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) nvalid = 0;
    __syncthreads();
    for (int r = wave; r < R; r += 4) {
        const float* xp = X + ((long long)b * R + r) * F;
        float s = 0.f;
        for (int f = lane * 4; f < F; f += 256) { const f32x4 v = ld4a(xp + f); s += (v[0] + v[1]) + (v[2] + v[3]); }
        s = wave_sum(s);
        const float* fp = fe + ((long long)b * R + r) * D;
        float t = 0.f;
        for (int d = lane * 4; d < D; d += 256) { const f32x4 v = ld4a(fp + d); t += (v[0] + v[1]) + (v[2] + v[3]); }
        t = wave_sum(t);
        if (lane == 0) {
            if (s != 0.f) atomicAdd(&nvalid, 1);
            fsum[r] = t;
        }
    }
    __syncthreads();
    for (int r = tid; r < R; r += 256) rmask[(long long)b * R + r] = (r < nvalid && fsum[r] != 0.f) ? 1.f : 0.f;
}

int region_masks(const float* X, const float* fe, float* rmask, int B, int R, int F, int D, hipStream_t s) {
    if (R > ATT_MAX_ROWS || (F & 3) || (D & 3)) return SET_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(region_masks_k, dim3(B), dim3(256), 0, s, X, fe, rmask, R, F, D);
    SET_LAUNCH_CHECK();
    return SET_OK;
}

// ---------------------------------------------------------------------------------------------
// SelectC.forward, hard mode (editnet.py:409-420) as a stand-alone operator: one workgroup per row
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) select_rows_k(const float* Mem, const float* alpha, float* sel, int T, int D) {
    __shared__ int s_arg;
    __shared__ float s_val;
    const int b = blockIdx.x, tid = threadIdx.x;
    if (tid < 64) {
        float best = -INFINITY; int bi = 0x7fffffff;
        for (int t = tid; t < T; t += 64) { const float a = alpha[(long long)b * T + t]; if (a > best) { best = a; bi = t; } }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ob = __shfl_xor(best, o);
            const int oi = __shfl_xor(bi, o);
            if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
        if (tid == 0) { s_arg = bi; s_val = best; }
    }
    __syncthreads();
    const float aj = s_val;
    const float wj = aj * 1.f + (1.f - aj);
    for (int d = tid * 4; d < D; d += 1024) {
        const f32x4 m = ld4a(Mem + ((long long)b * T + s_arg) * D + d);
        *reinterpret_cast<f32x4*>(sel + (long long)b * D + d) = m * wj;
    }
}

int select_rows(const float* Mem, const float* alpha, float* sel, int M, int T, int D, hipStream_t s) {
    if (D & 3) return SET_ERR_UNSUPPORTED;
    if (M <= 0) return SET_OK;
    hipLaunchKernelGGL(select_rows_k, dim3(M), dim3(256), 0, s, Mem, alpha, sel, T, D);
    SET_LAUNCH_CHECK();
    return SET_OK;
}

}  // namespace set
