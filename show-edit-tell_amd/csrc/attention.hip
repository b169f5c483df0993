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
            if (tid == 0) *argmax_out = bi == 0x7fffffff ? 0 : bi;   // all-NaN weights: a valid row index (the outputs are NaN anyway), never an out-of-range gather
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
    // small batches: the Dh output columns of a row are split over dsn workgroups (each recomputes the T scores and owns
    // dcols columns) so that a handful of rows still covers the chip; dsn == 0 / 1: one workgroup per row
    int dcols, dsn;
};
struct VisAttArgs {
    const float* att1; Slabs att2; const float* dec_bias; const float* w_full; const float* b_full;
    const float* X; const float* rmask; float* ctx; float* alpha_out;
    int R, F, A, fcols, fsn;
    int prefetch;
    float* att2_out;      // (M,A) or NULL, see CapAttArgs
};

__device__ __forceinline__ void caption_attention_body(const CapAttArgs& P, int b, float* sc, int* s_arg_p, int ds = 0) {
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
            const f32x4 v = slab_sum4_at(att2_c, (long long)b * att2_c.ld + a);
            a2[q] = v + ld4a(dec_bias + a);
            wf[q] = ld4a(w_full + a);
            if (P.att2_out && wave == 0 && ds == 0) *reinterpret_cast<f32x4*>(P.att2_out + (long long)b * A + a) = a2[q];
        }
    }
    const float bf = b_full[0];
    const int d_lo = P.dsn > 1 ? ds * P.dcols : 0, d_hi = P.dsn > 1 ? d_lo + P.dcols : Dh;      // this workgroup's columns
    // hoisted-projection mode: the gate epilogue's operands and the first four rows of the P stream do not depend on the
    // scores — requested here, their round trips run under the scoring phase instead of after the softmax
    constexpr int PB = 4;
    const int d_first = d_lo + tid * 4;
    const bool pre_ok = P.P && d_first < d_hi && T >= PB;
    f32x4 h_pre0 = {0.f, 0.f, 0.f, 0.f}, h_pre1 = h_pre0, h_bg = h_pre0, h_bs = h_pre0, h_bt = h_pre0, h_vz[PB], h_vs[PB];
    if (pre_ok) {
        const long long m = b;
        h_pre0 = slab_sum4_at(P.cg_ab, m * P.cg_ab.ld + d_first);
        h_pre1 = slab_sum4_at(P.tc, m * P.tc.ld + d_first);
        if (P.gz.tab) h_pre0 += ld4a(P.gz.row(m) + d_first);
        if (P.gtc.tab) h_pre1 += ld4a(P.gtc.row(m) + d_first);
        h_bg = ld4a(P.b_gate + d_first); h_bs = ld4a(P.b_sc + d_first); h_bt = ld4a(P.b_tc + d_first);
        const float* pp = P.P + (long long)b * T * 2 * Dh + d_first;
#pragma unroll
        for (int u = 0; u < PB; ++u) {
            h_vz[u] = ld4s(pp + (long long)u * 2 * Dh);
            h_vs[u] = ld4s(pp + (long long)u * 2 * Dh + Dh);
        }
    }
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
    if (alpha_out && ds == 0)
        for (int t = tid; t < T; t += 256) alpha_out[(long long)b * T + t] = sc[t];
    const int js = s_arg;
    const float aj = sc[js];
    const float wj = aj * 1.f + (1.f - aj);            // the reference's fp32 expression (editnet.py:417-418)
    if (P.P) {
        // ---- hoisted projections + fused context gate
        for (int d = d_lo + tid * 4; d < d_hi; d += 1024) {
            // operands that do not depend on the attention weights first: their latency overlaps the P stream
            const long long m = b;
            const bool first = pre_ok && d == d_first;
            f32x4 pre0 = h_pre0, pre1 = h_pre1, bg = h_bg, bs = h_bs, bt = h_bt;
            if (!first) {
                pre0 = slab_sum4_at(P.cg_ab, m * P.cg_ab.ld + d); pre1 = slab_sum4_at(P.tc, m * P.tc.ld + d);
                if (P.gz.tab) pre0 += ld4a(P.gz.row(m) + d);
                if (P.gtc.tab) pre1 += ld4a(P.gtc.row(m) + d);
                bg = ld4a(P.b_gate + d); bs = ld4a(P.b_sc + d); bt = ld4a(P.b_tc + d);
            }
            const f32x4 mrow = Mem ? ld4a(Mem + ((long long)b * T + js) * Dh + d) : pre0;
            const f32x4 qrow = P.Q ? ld4a(P.Q + ((long long)b * T + js) * Dh + d) : pre0;
            const float* pp = P.P + (long long)b * T * 2 * Dh + d;
            f32x4 zc = {0.f, 0.f, 0.f, 0.f}, sv = {0.f, 0.f, 0.f, 0.f};
            int t = 0;
            if (first) {                                      // the rows requested before the scoring phase
#pragma unroll
                for (int u = 0; u < PB; ++u) { zc += h_vz[u] * sc[u]; sv += h_vs[u] * sc[u]; }
                t = PB;
            }
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
    for (int d = d_lo + tid * 4; d < d_hi; d += 1024) {
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

__global__ void __launch_bounds__(256) caption_attention_k(const CapAttArgs P, const RowGate G) {
    __shared__ float sc[ATT_MAX_ROWS];
    __shared__ int s_arg;
    const int dsn = P.dsn > 1 ? P.dsn : 1;
    if (G.loop_left()) return;      // decode loops (set_common.h RowGate)
    caption_attention_body(P, blockIdx.x / dsn, sc, &s_arg, blockIdx.x % dsn);
}

// column slices of the caption role for small batches: enough workgroups for ~half the chip, slices of >= 128 columns
static int cap_dsn(int M, int Dh) {
    static const int on = env_int("SET_ATT_SMALL_SLICES", 1);
    int dsn = 1;
    if (!on || M >= 64) return 1;
    while (M * dsn < 128 && Dh / (dsn * 2) >= 128 && (Dh % (dsn * 2 * 4)) == 0) dsn *= 2;
    return dsn;
}

__global__ void caption_attention_v2_k(const CapAttArgs C, const RowGate G);

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
    static const int v2 = env_int("SET_ATT_V2", 1);
    static const int v2_maxm = env_int("SET_ATT_V2_MAXM", 64);
    if (v2 && M <= v2_maxm && Dh <= 1024) {  // 512 threads per row, all H rows requested before the scoring phase (see v2 below)
        hipLaunchKernelGGL(caption_attention_v2_k, dim3(M), dim3(512), 0, s, P, g_row_gate);
        SET_LAUNCH_CHECK();
        return SET_OK;
    }
    P.dsn = cap_dsn(M, Dh); P.dcols = Dh / P.dsn;
    hipLaunchKernelGGL(caption_attention_k, dim3(M * P.dsn), dim3(256), 0, s, P, g_row_gate);
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
            const f32x4 v = slab_sum4_at(att2, (long long)b * att2.ld + a);
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
__global__ void SET_VGPR_CAP_ATT __launch_bounds__(256) step_attention_k(const VisAttArgs V, const CapAttArgs C, int nvis, const RowGate G) {
    __shared__ float sc[ATT_MAX_ROWS];
    __shared__ int s_arg;
    if (G.loop_left()) return;                               // decode loops: the reference has left its loop (set_common.h)
    if ((int)blockIdx.x < nvis) {
        visual_attention_body(V, blockIdx.x / V.fsn, blockIdx.x % V.fsn, sc);
    } else {
        const int i = blockIdx.x - nvis, dsn = C.dsn > 1 ? C.dsn : 1;
        caption_attention_body(C, i / dsn, sc, &s_arg, i % dsn);
    }
}

// ---------------------------------------------------------------------------------------------
// v2 of the merged attention launch (round 3): 512 threads per row, and EVERY streamed operand of a row is requested
// before the scoring phase.  The 256-thread bodies above walk a chain of dependent round trips per row — decoder
// projection slabs -> att1 rows -> (softmax) -> 12 regions -> 12 regions -> 12 regions — which costs ~24 us whether the
// launch has 4 rows or 128; bytes per CU are not what bounds it.  Here the region rows X[b, r, :] (visual role) and the
// hoisted projection rows P[b, t, :] / encoder rows H[b, t, :] (caption role) do not depend on the scores, so each of
// the two thread halves requests its half of the rows (18 regions x 2 column groups, or 10 caption rows x 2 operands: 36 /
// 20 loads of 16 bytes in flight per thread, the whole 295 KB row at once) first; scoring runs on 8 waves underneath;
// after the softmax only FMAs on registers, one LDS exchange between the halves and the stores remain.
// Accumulation order: regions / positions ascending inside a half, then half 0 + half 1 (deterministic).
// ---------------------------------------------------------------------------------------------
constexpr int V2_PB = 18;      // prefetched regions per half (R <= 36 is fully covered)
constexpr int V2_PT = 10;      // prefetched caption positions per half (T <= 20 is fully covered)

__device__ __forceinline__ void visual_attention_v2(const VisAttArgs& P, int b, float* sc, f32x4* xch) {
    const float* att1 = P.att1; const Slabs att2 = P.att2; const float* X = P.X; const float* rmask = P.rmask;
    const int R = P.R, F = P.F, A = P.A;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = tid >> 8, cg = tid & 255;
    // fixed blocks of V2_PB regions alternate between the two halves (block k -> half k & 1): which half sums a region does
    // not depend on R, so zero-weight padding regions / positions never change a bit of the result
    const int r_lo = half * V2_PB, r_hi = (r_lo + V2_PB < R) ? r_lo + V2_PB : R;
    const float* xrow = X + (long long)b * R * F;
    const int f0 = cg * 4, f1 = (cg + 256) * 4;
    // ---- all of this thread's first V2_PB regions, both column groups: requested now
    f32x4 xp0[V2_PB], xp1[V2_PB];
#pragma unroll
    for (int u = 0; u < V2_PB; ++u) {
        const int r = r_lo + u;
        xp0[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
        xp1[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (r < r_hi) {
            if (f0 < F) xp0[u] = ld4s(xrow + (long long)r * F + f0);
            if (f1 < F) xp1[u] = ld4s(xrow + (long long)r * F + f1);
        }
    }
    // ---- decoder projection (+ bias) and the scoring vector: per lane a = lane*4 + 256 q
    const int nq = (A + 255) / 256;
    f32x4 a2[2], wf[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        a2[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
        wf[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int a = lane * 4 + 256 * q;
        if (q < nq && a < A) {
            const f32x4 v = slab_sum4_at(att2, (long long)b * att2.ld + a);
            a2[q] = v + ld4a(P.dec_bias + a);
            wf[q] = ld4a(P.w_full + a);
            if (P.att2_out && wave == 0) *reinterpret_cast<f32x4*>(P.att2_out + (long long)b * A + a) = a2[q];
        }
    }
    const float bf = P.b_full[0];
    constexpr int RB = 5;                                 // 8 waves x 5 rows: R <= 40 in one batch
    for (int r0 = wave; r0 < R; r0 += 8 * RB) {
        f32x4 v[RB][2];
#pragma unroll
        for (int u = 0; u < RB; ++u) {
            const int r = r0 + 8 * u;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int a = lane * 4 + 256 * q;
                v[u][q] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (r < R && q < nq && a < A) v[u][q] = ld4a(att1 + ((long long)b * R + r) * A + a);
            }
        }
#pragma unroll
        for (int u = 0; u < RB; ++u) {
            const int r = r0 + 8 * u;
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
    if (P.alpha_out)
        for (int r = tid; r < R; r += 512) P.alpha_out[(long long)b * R + r] = sc[r];
    // ---- context from the registers (+ the rows beyond the prefetched ones when R > 36)
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < V2_PB; ++u)
        if (r_lo + u < r_hi) { acc0 += xp0[u] * sc[r_lo + u]; acc1 += xp1[u] * sc[r_lo + u]; }
    for (int rb = r_lo + 2 * V2_PB; rb < R; rb += 2 * V2_PB) {      // further blocks of this half (R > 36: adaptive features)
        const int re = (rb + V2_PB < R) ? rb + V2_PB : R;
        for (int r = rb; r < re; r += 6) {                  // 12 loads in flight; accumulation stays in r order
            f32x4 v0[6], v1[6];
#pragma unroll
            for (int u = 0; u < 6; ++u) {
                v0[u] = (f32x4){0.f, 0.f, 0.f, 0.f}; v1[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (r + u < re) {
                    if (f0 < F) v0[u] = ld4s(xrow + (long long)(r + u) * F + f0);
                    if (f1 < F) v1[u] = ld4s(xrow + (long long)(r + u) * F + f1);
                }
            }
#pragma unroll
            for (int u = 0; u < 6; ++u)
                if (r + u < re) { acc0 += v0[u] * sc[r + u]; acc1 += v1[u] * sc[r + u]; }
        }
    }
    // halves: half 1 hands its partial sums over, half 0 adds (half 0 + half 1) and stores
    if (half == 1) { xch[cg] = acc0; xch[256 + cg] = acc1; }
    __syncthreads();
    if (half == 0) {
        if (f0 < F) *reinterpret_cast<f32x4*>(P.ctx + (long long)b * F + f0) = acc0 + xch[cg];
        if (f1 < F) *reinterpret_cast<f32x4*>(P.ctx + (long long)b * F + f1) = acc1 + xch[256 + cg];
    }
    // feature columns beyond 2048 (not a reference shape): the plain loop, all regions, by the first 256 threads
    for (int f = f0 + 2048; f < F && half == 0; f += 1024) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int r = 0; r < R; ++r) acc += ld4s(xrow + (long long)r * F + f) * sc[r];
        *reinterpret_cast<f32x4*>(P.ctx + (long long)b * F + f) = acc;
    }
}

__device__ __forceinline__ void caption_attention_v2(const CapAttArgs& P, int b, float* sc, int* s_arg_p, f32x4* xch) {
    const Slabs att2_c = P.att2_c; const float* Mem = P.Mem; const int T = P.T, Dh = P.Dh, A = P.A;
    int& s_arg = *s_arg_p;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = tid >> 8, cg = tid & 255;
    const int t_lo = half * V2_PT, t_hi = (t_lo + V2_PT < T) ? t_lo + V2_PT : T;      // fixed blocks, as in the visual role
    const int d0 = cg * 4;
    const bool col_ok = d0 < Dh;
    const bool hoisted = P.P != nullptr;
    // ---- this thread's rows of the streamed operand(s): requested now
    //      hoisted mode: P[b, t, d0] (context-gate part) and P[b, t, Dh + d0] (sc_affine part); plain mode: H[b, t, d0]
    f32x4 pz[V2_PT], ps[V2_PT];
    const float* prow = hoisted ? P.P + (long long)b * T * 2 * Dh + d0 : P.H + (long long)b * T * Dh + d0;
    const long long pstride = hoisted ? 2LL * Dh : (long long)Dh;
#pragma unroll
    for (int u = 0; u < V2_PT; ++u) {
        pz[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
        ps[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (col_ok && t_lo + u < t_hi) {
            pz[u] = hoisted ? ld4s(prow + (t_lo + u) * pstride) : ld4a(prow + (t_lo + u) * pstride);
            if (hoisted) ps[u] = ld4s(prow + (t_lo + u) * pstride + Dh);
        }
    }
    // operands of the gate epilogue that do not depend on the attention weights (half 0 runs the epilogue)
    f32x4 pre0 = {0.f, 0.f, 0.f, 0.f}, pre1 = {0.f, 0.f, 0.f, 0.f}, bg = pre0, bs = pre0, bt = pre0;
    if (hoisted && half == 0 && col_ok) {
        const long long m = b;
        pre0 = slab_sum4_at(P.cg_ab, m * P.cg_ab.ld + d0);
        pre1 = slab_sum4_at(P.tc, m * P.tc.ld + d0);
        if (P.gz.tab) pre0 += ld4a(P.gz.row(m) + d0);
        if (P.gtc.tab) pre1 += ld4a(P.gtc.row(m) + d0);
        bg = ld4a(P.b_gate + d0); bs = ld4a(P.b_sc + d0); bt = ld4a(P.b_tc + d0);
    }
    // ---- scores
    const int nq = (A + 255) / 256;
    f32x4 a2[2], wf[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        a2[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
        wf[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int a = lane * 4 + 256 * q;
        if (q < nq && a < A) {
            const f32x4 v = slab_sum4_at(att2_c, (long long)b * att2_c.ld + a);
            a2[q] = v + ld4a(P.dec_bias + a);
            wf[q] = ld4a(P.w_full + a);
            if (P.att2_out && wave == 0) *reinterpret_cast<f32x4*>(P.att2_out + (long long)b * A + a) = a2[q];
        }
    }
    const float bf = P.b_full[0];
    constexpr int RB = 3;                                 // 8 waves x 3 rows: T <= 24 in one batch
    for (int t0 = wave; t0 < T; t0 += 8 * RB) {
        f32x4 v[RB][2];
#pragma unroll
        for (int u = 0; u < RB; ++u) {
            const int t = t0 + 8 * u;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int a = lane * 4 + 256 * q;
                v[u][q] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (t < T && q < nq && a < A) v[u][q] = ld4a(P.att1_c + ((long long)b * T + t) * A + a);
            }
        }
#pragma unroll
        for (int u = 0; u < RB; ++u) {
            const int t = t0 + 8 * u;
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
            if (lane == 0) sc[t] = (P.mask[(long long)b * T + t] == 0.f) ? -1e10f : (s + bf);
        }
    }
    __syncthreads();
    block_softmax(sc, T, tid, &s_arg);
    __syncthreads();
    if (P.alpha_out)
        for (int t = tid; t < T; t += 512) P.alpha_out[(long long)b * T + t] = sc[t];
    const int js = s_arg;
    const float aj = sc[js];
    const float wj = aj * 1.f + (1.f - aj);            // the reference's fp32 expression (editnet.py:417-418)
    // the selected rows depend on the arg-max: half 1 fetches them while half 0 finishes the context
    if (half == 1 && col_ok) {
        if (Mem) *reinterpret_cast<f32x4*>(P.sel + (long long)b * Dh + d0) = ld4a(Mem + ((long long)b * T + js) * Dh + d0) * wj;
        if (hoisted && P.Q)
            *reinterpret_cast<f32x4*>(P.cmem_out + (long long)b * Dh + d0) = ld4a(P.Q + ((long long)b * T + js) * Dh + d0) * wj;
    }
    f32x4 zc = {0.f, 0.f, 0.f, 0.f}, sv = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < V2_PT; ++u)
        if (t_lo + u < t_hi) { zc += pz[u] * sc[t_lo + u]; sv += ps[u] * sc[t_lo + u]; }
    if (col_ok)
        for (int tb = t_lo + 2 * V2_PT; tb < T; tb += 2 * V2_PT) {      // further blocks of this half (T > 20)
            const int te = (tb + V2_PT < T) ? tb + V2_PT : T;
            for (int t = tb; t < te; ++t) {
                zc += ld4a(prow + t * pstride) * sc[t];
                if (hoisted) sv += ld4a(prow + t * pstride + Dh) * sc[t];
            }
        }
    if (half == 1) { xch[cg] = zc; xch[256 + cg] = sv; }
    __syncthreads();
    if (half == 0 && col_ok) {
        zc += xch[cg];
        sv += xch[256 + cg];
        if (hoisted) {
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                // reference: ONE Linear over cat([word, h1, ctx]): the [word,h1] part first, then the ctx part, then bias
                const float z = (pre0[e] + zc[e]) + bg[e];
                const float zt = 1.f / (1.f + expf(-z));
                o[e] = zt * tanhf(sv[e] + bs[e]) + (1.f - zt) * tanhf(pre1[e] + bt[e]);
            }
            *reinterpret_cast<f32x4*>(P.gated_out + (long long)b * Dh + d0) = o;
        } else {
            *reinterpret_cast<f32x4*>(P.ctx + (long long)b * Dh + d0) = zc;
        }
    }
}

__global__ void __launch_bounds__(512) step_attention_v2_k(const VisAttArgs V, const CapAttArgs C, int nvis, const RowGate G) {
    __shared__ float sc[ATT_MAX_ROWS];
    __shared__ int s_arg;
    __shared__ f32x4 xch[512];
    if (G.loop_left()) return;
    if ((int)blockIdx.x < nvis) visual_attention_v2(V, blockIdx.x, sc, xch);
    else caption_attention_v2(C, blockIdx.x - nvis, sc, &s_arg, xch);
}

__global__ void __launch_bounds__(512) caption_attention_v2_k(const CapAttArgs C, const RowGate G) {
    __shared__ float sc[ATT_MAX_ROWS];
    __shared__ int s_arg;
    __shared__ f32x4 xch[512];
    if (G.loop_left()) return;
    caption_attention_v2(C, blockIdx.x, sc, &s_arg, xch);
}

static int vis_fsn(int M, int F, int R = 36) {
    // enough workgroups to cover the chip: split the feature axis while slices stay >= 1024 columns
    // measured at B=128: one slice per sample (128 + 128 workgroups) beats 2 or 4 slices (each slice recomputes the scores)
    // small batches (round 3): a handful of rows at one workgroup each leaves the kernel at one CU's streaming rate per
    // row (B = 4: 23 us for 1.2 MB); slices down to 128 columns put ~half the chip on it
    static const int min_cols = env_int("SET_ATT_MIN_COLS", 2048);
    static const int small = env_int("SET_ATT_SMALL_SLICES", 1);
    // adaptive features (up to 100 regions: 0.8 MB of X per row): two slices per sample — 64 rows are 64 workgroups walking
    // 18 dependent rounds of region loads each; B = 64, R = 100: 39.8 -> 31.8 us (four slices: 35.0)
    const int mc = (small && M < 64) ? 128 : ((R > 48 && min_cols > 1024) ? 1024 : min_cols), want = (small && M < 64) ? 128 : 512;
    int fsn = 1;
    while (M * fsn < want && F / (fsn * 2) >= mc && (F % (fsn * 2 * 4)) == 0) fsn *= 2;
    return fsn;
}

int step_attention(const float* att1, Slabs att2, const float* v_dec_bias, const float* v_w_full, const float* v_b_full,
                   const float* X, const float* rmask, float* v_ctx, float* v_alpha, int R, int F,
                   const float* att1_c, Slabs att2_c, const float* c_dec_bias, const float* c_w_full,
                   const float* c_b_full, const float* mask, const float* H, const float* Mem, float* c_ctx, float* sel,
                   float* c_alpha, int T, int Dh, int A, int M, hipStream_t s, const CapHoist* hoist, float* v_att2_out,
                   float* c_att2_out) {
#if defined(SET_EXP_SKIP_POINTWISE) || defined(SET_EXP_SKIP_ATT)      // diagnostic build (EXPERIMENTS 5.7): the launch is dropped, results are garbage
    return SET_OK;
#endif
    if (R > ATT_MAX_ROWS || T > ATT_MAX_ROWS || A > 512 || (A & 3) || (F & 3) || (Dh & 3)) return SET_ERR_UNSUPPORTED;
    if (M <= 0) return SET_OK;
    const int fsn = vis_fsn(M, F, R);
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
    // v2 (512 threads per row, every streamed operand requested before the scoring phase) when the columns fit its layout
    // Default: up to SET_ATT_V2_MAXM rows (64; 32 until round 6).  Measured at B = 128 (round 3): 26.4 -> 24.3 us per launch single
    // stream, but a 512-thread / 236-register workgroup leaves no room for another batch's kernels on its CU: 6.75 k -> 6.41 k
    // with 7 batches in flight; at B = 4 26.3 -> 21.5 us and nothing else is there to displace.  Round 6, 33..64 rows (one
    // decode at a time is the case there): 23.1 -> 21.2 us per launch, B = 48 / 64 greedy decode 2.72 / 2.79 -> 2.66 / 2.75 ms.
    static const int v2 = env_int("SET_ATT_V2", 1);
    static const int v2_maxm = env_int("SET_ATT_V2_MAXM", 64);
    if (v2 && M <= v2_maxm && F <= 2048 && Dh <= 1024) {
        V.fcols = F; V.fsn = 1;
        hipLaunchKernelGGL(step_attention_v2_k, dim3(2 * M), dim3(512), 0, s, V, C, M, g_row_gate);
        SET_LAUNCH_CHECK();
        return SET_OK;
    }
    C.dsn = cap_dsn(M, Dh); C.dcols = Dh / C.dsn;
    hipLaunchKernelGGL(step_attention_k, dim3(M * fsn + M * C.dsn), dim3(256), 0, s, V, C, M * fsn, g_row_gate);
    SET_LAUNCH_CHECK();
    return SET_OK;
}

int visual_attention(const float* att1, Slabs att2, const float* dec_bias, const float* w_full,
                     const float* b_full, const float* X, const float* rmask, float* ctx, float* alpha_out, int M,
                     int R, int F, int A, hipStream_t s) {
    if (R > ATT_MAX_ROWS || A > 512 || (A & 3) || (F & 3)) return SET_ERR_UNSUPPORTED;
    if (M <= 0) return SET_OK;
    const int fsn = vis_fsn(M, F, R);
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
        if (tid == 0) { s_arg = bi == 0x7fffffff ? 0 : bi; s_val = best; }     // (all-NaN alpha: stay in range)
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
