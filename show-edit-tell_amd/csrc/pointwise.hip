// Pointwise / reduction kernels of the decode step.  All are HBM/L2-bound streaming kernels:
// one float4 of hidden units per thread, coalesced rows, slab partial sums of the split-K GEMMs
// added in a fixed order (slab 0, 1, 2, ...) so results are run-to-run deterministic.
#include "set_common.h"

namespace set {

typedef float f32x4 __attribute__((ext_vector_type(4)));

thread_local int g_last_hip_error = 0;

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

// sum of the K-slabs of one GEMM output at (m, n..n+3): slabs are added in index order (fixed,
// deterministic); loads are issued four slabs at a time so they are in flight together
__device__ __forceinline__ f32x4 slab_sum4(const Slabs& s, long long m, int n) {
    if (s.n <= 0) return (f32x4){0.f, 0.f, 0.f, 0.f};
    return slab_sum4_at(s, m * s.ld + n);
}

// the same for Q column blocks of one row at once (the four gates of an LSTM row): Q x 2 loads in flight
template <int Q>
__device__ __forceinline__ void slab_accum(f32x4 (&acc)[Q], const Slabs& s, long long m, int col0, int colstep) {
    if (s.n <= 0) return;
    const float* p = s.p + m * s.ld + col0;
    int i = 0;
    for (; i + 4 <= s.n; i += 4) {                  // Q x 4 independent loads in flight
        f32x4 v[4][Q];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int q = 0; q < Q; ++q) v[u][q] = ld4(p + (long long)(i + u) * s.stride + q * colstep);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int q = 0; q < Q; ++q) acc[q] += v[u][q];
    }
    for (; i + 2 <= s.n; i += 2) {
        f32x4 v0[Q], v1[Q];
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            v0[q] = ld4(p + (long long)i * s.stride + q * colstep);
            v1[q] = ld4(p + (long long)(i + 1) * s.stride + q * colstep);
        }
#pragma unroll
        for (int q = 0; q < Q; ++q) { acc[q] += v0[q]; acc[q] += v1[q]; }
    }
    if (i < s.n) {
        f32x4 v0[Q];
#pragma unroll
        for (int q = 0; q < Q; ++q) v0[q] = ld4(p + (long long)i * s.stride + q * colstep);
#pragma unroll
        for (int q = 0; q < Q; ++q) acc[q] += v0[q];
    }
}

// Up to three slab sets added one after the other (set 0's partials in index order, then set 1's, then set 2's — the order
// three slab_accum calls give) as ONE software-pipelined walk over the flattened partial list: four partials are in flight
// while the previous four are added.  Three separate calls cost a round trip per batch per set (6 for the copy cell's
// 6 + 3 + 3 partials: most of the kernel's 10.8 us at B = 128).
template <int Q>
__device__ __forceinline__ void slab_accum3(f32x4 (&acc)[Q], const Slabs& s0, const Slabs& s1, const Slabs& s2, long long m,
                                            int col0, int colstep) {
    const int n0 = s0.n > 0 ? s0.n : 0, n1 = s1.n > 0 ? s1.n : 0, n2 = s2.n > 0 ? s2.n : 0, n = n0 + n1 + n2;
    if (n == 0) return;
    auto at = [&](int i) -> const float* {          // (uniform: scalar selects)
        if (i < n0) return s0.p + (long long)i * s0.stride + m * s0.ld + col0;
        if (i < n0 + n1) return s1.p + (long long)(i - n0) * s1.stride + m * s1.ld + col0;
        return s2.p + (long long)(i - n0 - n1) * s2.stride + m * s2.ld + col0;
    };
    f32x4 a[4][Q], b[4][Q];
#define SA3_LOAD(BUF, I0)                                                                               \
    _Pragma("unroll") for (int u = 0; u < 4; ++u)                                                       \
        if ((I0) + u < n) {                                                                             \
            const float* p_ = at((I0) + u);                                                             \
            _Pragma("unroll") for (int q = 0; q < Q; ++q) BUF[u][q] = ld4(p_ + q * colstep);            \
        }
#define SA3_ADD(BUF, I0)                                                                                \
    _Pragma("unroll") for (int u = 0; u < 4; ++u)                                                       \
        if ((I0) + u < n) {                                                                             \
            _Pragma("unroll") for (int q = 0; q < Q; ++q) acc[q] += BUF[u][q];                          \
        }
    int i = 0;
    SA3_LOAD(a, 0);
    for (;;) {
        if (i + 4 < n) { SA3_LOAD(b, i + 4); }
        SA3_ADD(a, i);
        i += 4;
        if (i >= n) break;
        if (i + 4 < n) { SA3_LOAD(a, i + 4); }
        SA3_ADD(b, i);
        i += 4;
        if (i >= n) break;
    }
#undef SA3_LOAD
#undef SA3_ADD
}

// ---------------------------------------------------------------------------------------------
// LSTM cell pointwise (nn.LSTMCell editnet.py:468,532 / LSTMCellC :235-242 / CopyLSTMCellC :274-280)
//   gates[m, g*D + j] = g0 + g1 + g2 (slab sums) + pre + b0 + b1 ; order i, f, g, o
//   c' = sig(f) c + sig(i) tanh(g);  h' = sig(o) tanh(c')     (h_out / ogate_out optional)
// ---------------------------------------------------------------------------------------------
__global__ void SET_VGPR_CAP __launch_bounds__(256) lstm_pointwise_k(Slabs g0, Slabs g1, Slabs g2, const float* pre,
                                                        long long ldpre, const float* b0, const float* b1,
                                                        const float* c_in, float* c_out, float* h_out,
                                                        float* ogate_out, int M, int D, RowGather gt,
                                                        float* gates_out, const RowGate G) {
    const int per_row = D >> 2;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)M * per_row) return;
    const long long m = idx / per_row;
    const int j = (int)(idx - m * per_row) << 2;
    // operands that do not depend on the slabs are requested first (the table row needs its token id): their
    // latency overlaps the slab reads; the additions keep the order slabs, pre, table row, b0, b1
    const float* trow = gt.tab ? gt.row(m) : nullptr;
    const f32x4 c = ld4(c_in + m * D + j);
    f32x4 xpre[4], xtab[4], xb0[4], xb1[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int n = q * D + j;
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        xpre[q] = pre ? ld4(pre + m * ldpre + n) : z;
        xtab[q] = trow ? ld4(trow + n) : z;
        xb0[q] = b0 ? ld4(b0 + n) : z;
        xb1[q] = b1 ? ld4(b1 + n) : z;
    }
    f32x4 g[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) g[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
    slab_accum3<4>(g, g0, g1, g2, m, j, D);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (pre) g[q] += xpre[q];
        if (trow) g[q] += xtab[q];
        if (b0) g[q] += xb0[q];
        if (b1) g[q] += xb1[q];
    }
    f32x4 cn, hn, og;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float ig = sigmoidf_(g[0][e]), fg = sigmoidf_(g[1][e]), gg = tanhf(g[2][e]);
        og[e] = sigmoidf_(g[3][e]);
        cn[e] = fg * c[e] + ig * gg;
        hn[e] = og[e] * tanhf(cn[e]);
    }
    st4(c_out + m * D + j, cn);
    if (h_out) st4(h_out + m * D + j, hn);
    if (ogate_out) st4(ogate_out + m * D + j, og);
    if (gates_out) {                      // POST-activation i, f, g, o for the backward kernels
        f32x4 ai, af, ag;
#pragma unroll
        for (int e = 0; e < 4; ++e) { ai[e] = sigmoidf_(g[0][e]); af[e] = sigmoidf_(g[1][e]); ag[e] = tanhf(g[2][e]); }
        float* gr = gates_out + m * 4 * D + j;
        st4(gr, ai); st4(gr + D, af); st4(gr + 2 * D, ag); st4(gr + 3 * D, og);
    }
}

// Threads per workgroup of the row-major pointwise kernels.  A handful of rows (BASELINE.json configs[0]: 4) is one or two
// thousand threads: as 256-thread workgroups that is 4-8 workgroups, each pulling its slabs through ONE CU (~25-40 GB/s)
// — 10 us for 1.5 MB of slab reads.  64-thread workgroups spread the same threads over 4x as many CUs.
static int pointwise_block(long long n_threads) {
    static const int small = env_int("SET_POINTWISE_SMALL_BLOCKS", 1);
    static const int small_max = env_int("SET_POINTWISE_SMALL_MAX", 16384);
    return (small && n_threads <= small_max) ? 64 : 256;
}

int lstm_pointwise(Slabs g0, Slabs g1, Slabs g2, const float* pre, long long ldpre, const float* b0,
                   const float* b1, const float* c_in, float* c_out, float* h_out, float* ogate_out, int M,
                   int D, hipStream_t s, RowGather gt, float* gates_out) {
#if defined(SET_EXP_SKIP_POINTWISE) || defined(SET_EXP_SKIP_LSTM)      // diagnostic build (EXPERIMENTS 5.7): the launch is dropped, results are garbage
    return SET_OK;
#endif
    if (D & 3) return SET_ERR_UNSUPPORTED;
    const long long n = (long long)M * (D >> 2);
    ProfScope ps("lstm_pointwise", s, 0.0, 4.0 * M * D * (4.0 * (g0.n + g1.n + g2.n + (pre ? 1 : 0)) + 3.0));
    const int blk = pointwise_block(n);
    hipLaunchKernelGGL(lstm_pointwise_k, dim3((unsigned)((n + blk - 1) / blk)), dim3(blk), 0, s, g0, g1, g2, pre,
                       ldpre, b0, b1, c_in, c_out, h_out, ogate_out, M, D, gt, gates_out, g_row_gate);
    SET_LAUNCH_CHECK();
    return SET_OK;
}

// ---------------------------------------------------------------------------------------------
// context gating (editnet.py:378-380):
//   zt = sig(cg_a + cg_b + b_cg); out = zt*tanh(sc + b_sc) + (1-zt)*tanh(tc + b_tc)
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) context_gate_k(Slabs cg_a, Slabs cg_b, const float* cg_bias, Slabs sc,
                                                      const float* sc_bias, Slabs tc, const float* tc_bias,
                                                      float* out, int M, int D, float* zt_out, float* s_out,
                                                      float* t_out, RowGather gz, RowGather gt) {
    const int per_row = D >> 2;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)M * per_row) return;
    const long long m = idx / per_row;
    const int j = (int)(idx - m * per_row) << 2;
    f32x4 z = slab_sum4(cg_a, m, j);
    if (gz.tab) z += ld4(gz.row(m) + j);     // token-table part of [word,h1]
    z += slab_sum4(cg_b, m, j);
    z += ld4(cg_bias + j);
    f32x4 s = slab_sum4(sc, m, j) + ld4(sc_bias + j);
    f32x4 t = slab_sum4(tc, m, j);
    if (gt.tab) t += ld4(gt.row(m) + j);
    t += ld4(tc_bias + j);
    f32x4 o, zs, ss, ts;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float zt = sigmoidf_(z[e]);
        zs[e] = zt; ss[e] = tanhf(s[e]); ts[e] = tanhf(t[e]);
        o[e] = zt * ss[e] + (1.f - zt) * ts[e];
    }
    st4(out + m * D + j, o);
    if (zt_out) { st4(zt_out + m * D + j, zs); st4(s_out + m * D + j, ss); st4(t_out + m * D + j, ts); }
}

int context_gate_pointwise(Slabs cg_a, Slabs cg_b, const float* cg_bias, Slabs sc, const float* sc_bias,
                           Slabs tc, const float* tc_bias, float* out, int M, int D, hipStream_t s, float* zt_out,
                           float* s_out, float* t_out, RowGather gz, RowGather gt) {
    if (D & 3) return SET_ERR_UNSUPPORTED;
    const long long n = (long long)M * (D >> 2);
    ProfScope ps("context_gate", s, 0.0, 4.0 * M * D * (cg_a.n + cg_b.n + sc.n + tc.n + 1.0));
    hipLaunchKernelGGL(context_gate_k, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, cg_a, cg_b, cg_bias,
                       sc, sc_bias, tc, tc_bias, out, M, D, zt_out, s_out, t_out, gz, gt);
    SET_LAUNCH_CHECK();
    return SET_OK;
}

// ---------------------------------------------------------------------------------------------
// copy gate (editnet.py:281-283):
//   copy = sig(Wn c_new + bn + Wm sel + bm); c2' = copy*sel + (1-copy)*c_new; h2' = ogate*tanh(c2')
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) copy_gate_k(Slabs gn, const float* bn, Slabs gm, const float* bm,
                                                   const float* c_new, const float* sel, const float* ogate,
                                                   float* c_out, float* h_out, int M, int D, float* cg_out,
                                                   const RowGate G) {
    const int per_row = D >> 2;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)M * per_row) return;
    const long long m = idx / per_row;
    const int j = (int)(idx - m * per_row) << 2;
    // reference order: (gate_cnew(c_new) + b) + (gate_cmem(c_memory) + b)
    f32x4 a = slab_sum4(gn, m, j) + ld4(bn + j);
    f32x4 b = slab_sum4(gm, m, j) + ld4(bm + j);
    const f32x4 cn = ld4(c_new + m * D + j), sm = ld4(sel + m * D + j), og = ld4(ogate + m * D + j);
    f32x4 co, ho, cgs;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float cg = sigmoidf_(a[e] + b[e]);
        cgs[e] = cg;
        co[e] = cg * sm[e] + (1.f - cg) * cn[e];
        ho[e] = og[e] * tanhf(co[e]);
    }
    st4(c_out + m * D + j, co);
    st4(h_out + m * D + j, ho);
    if (cg_out) st4(cg_out + m * D + j, cgs);
}

int copy_gate_pointwise(Slabs gn, const float* bn, Slabs gm, const float* bm, const float* c_new,
                        const float* sel, const float* ogate, float* c_out, float* h_out, int M, int D,
                        hipStream_t s, float* cg_out) {
#if defined(SET_EXP_SKIP_POINTWISE) || defined(SET_EXP_SKIP_CG)      // diagnostic build (EXPERIMENTS 5.7): the launch is dropped, results are garbage
    return SET_OK;
#endif
    if (D & 3) return SET_ERR_UNSUPPORTED;
    const long long n = (long long)M * (D >> 2);
    ProfScope ps("copy_gate", s, 0.0, 4.0 * M * D * (gn.n + gm.n + 5.0));
    const int blk = pointwise_block(n);
    hipLaunchKernelGGL(copy_gate_k, dim3((unsigned)((n + blk - 1) / blk)), dim3(blk), 0, s, gn, bn, gm, bm, c_new,
                       sel, ogate, c_out, h_out, M, D, cg_out, g_row_gate);
    SET_LAUNCH_CHECK();
    return SET_OK;
}

// ---------------------------------------------------------------------------------------------
// out[m, n] = act(sum_slabs + b0 + b1)        (N multiple of 4)
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) reduce_bias_act_k(Slabs in, const float* b0, const float* b1, float* out,
                                                         long long ldo, int M, int N, int act) {
    const int per_row = N >> 2;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)M * per_row) return;
    const long long m = idx / per_row;
    const int j = (int)(idx - m * per_row) << 2;
    f32x4 v = slab_sum4(in, m, j);
    if (b0) v += ld4(b0 + j);
    if (b1) v += ld4(b1 + j);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float x = v[e];
        if (act == SET_ACT_RELU) x = x > 0.f ? x : 0.f;
        else if (act == SET_ACT_TANH) x = tanhf(x);
        else if (act == SET_ACT_SIGMOID) x = sigmoidf_(x);
        v[e] = x;
    }
    st4(out + m * ldo + j, v);
}

// scalar variant for rows that are not 16-byte aligned (e.g. V = 9490 predictions)
__global__ void __launch_bounds__(256) reduce_bias_act_scalar_k(Slabs in, const float* b0, const float* b1,
                                                                float* out, long long ldo, int M, int N, int act) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)M * N) return;
    const long long m = idx / N;
    const int j = (int)(idx - m * N);
    const float* p = in.p + m * in.ld + j;
    float x = p[0];
    for (int i = 1; i < in.n; ++i) x += p[(long long)i * in.stride];
    if (b0) x += b0[j];
    if (b1) x += b1[j];
    if (act == SET_ACT_RELU) x = x > 0.f ? x : 0.f;
    else if (act == SET_ACT_TANH) x = tanhf(x);
    else if (act == SET_ACT_SIGMOID) x = sigmoidf_(x);
    out[m * ldo + j] = x;
}

int reduce_bias_act(Slabs in, const float* b0, const float* b1, float* out, long long ldo, int M, int N, int act,
                    hipStream_t s) {
    if (M <= 0 || N <= 0) return SET_OK;
    const bool vec = !(N & 3) && !(ldo & 3) && !(in.ld & 3) && !(in.stride & 3) && aligned16(out) && aligned16(in.p) &&
                     (!b0 || aligned16(b0)) && (!b1 || aligned16(b1));
    ProfScope ps("reduce_bias_act", s, 0.0, 4.0 * M * N * (in.n + 1.0));
    if (vec) {
        const long long n = (long long)M * (N >> 2);
        hipLaunchKernelGGL(reduce_bias_act_k, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, b0, b1, out, ldo,
                           M, N, act);
    } else {
        const long long n = (long long)M * N;
        hipLaunchKernelGGL(reduce_bias_act_scalar_k, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, b0, b1,
                           out, ldo, M, N, act);
    }
    SET_LAUNCH_CHECK();
    return SET_OK;
}

// ---------------------------------------------------------------------------------------------
// EmbeddingC.forward, eval (editnet.py:300-304): out[i] = relu(table[ids[i*stride]])
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) embed_relu_k(const float* table, const int64_t* ids, long long ids_stride,
                                                    float* out, long long ldo, int n, int D, int V) {
    const int per_row = D >> 2;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)n * per_row) return;
    const long long i = idx / per_row;
    const int j = (int)(idx - i * per_row) << 2;
    long long id = ids[i * ids_stride];
    id = id < 0 ? 0 : (id >= V ? V - 1 : id);       // ids are validated by the caller; clamp = no OOB read
    f32x4 v = ld4(table + id * D + j);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
    st4(out + i * ldo + j, v);
}

int embed_relu(const float* table, const int64_t* ids, long long ids_stride, float* out, long long ldo, int n,
               int D, int V, hipStream_t s) {
    if ((D & 3) || (ldo & 3)) return SET_ERR_UNSUPPORTED;
    if (n <= 0) return SET_OK;
    const long long t = (long long)n * (D >> 2);
    ProfScope ps("embed_relu", s, 0.0, 8.0 * n * D);
    hipLaunchKernelGGL(embed_relu_k, dim3((unsigned)((t + 255) / 256)), dim3(256), 0, s, table, ids, ids_stride, out,
                       ldo, n, D, V);
    SET_LAUNCH_CHECK();
    return SET_OK;
}

// ---------------------------------------------------------------------------------------------
// image_mean = X.mean(1) (editnet.py:503): sequential sum over regions, then / R
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) mean_regions_k(const float* X, float* out, int B, int R, int F) {
    const int per_row = F >> 2;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)B * per_row) return;
    const long long b = idx / per_row;
    const int j = (int)(idx - b * per_row) << 2;
    const float* p = X + b * (long long)R * F + j;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < R; ++r) acc += ld4(p + (long long)r * F);
    const float rr = (float)R;
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] = acc[e] / rr;
    st4(out + b * F + j, acc);
}

int mean_regions(const float* X, float* out, int B, int R, int F, hipStream_t s) {
    if (F & 3) return SET_ERR_UNSUPPORTED;
    const long long t = (long long)B * (F >> 2);
    ProfScope ps("mean_regions", s, 0.0, 4.0 * B * F * (R + 1.0));
    hipLaunchKernelGGL(mean_regions_k, dim3((unsigned)((t + 255) / 256)), dim3(256), 0, s, X, out, B, R, F);
    SET_LAUNCH_CHECK();
    return SET_OK;
}

// ---------------------------------------------------------------------------------------------
// caption-encoder LSTM step (CaptionEncoderC editnet.py:333-338 ; nn.LSTM over a packed batch
// dcnet.py:233).  The reference sorts by length and shrinks the batch prefix; per row that is:
//   if t < len[b]:  (h,c) <- cell(x[b,pos], h, c);  H[b,pos] = h;  Mem[b,pos] = c
// with pos = t (forward) or len[b]-1-t (reverse direction of the packed BiLSTM).
// xg holds the hoisted input projection x W_x^T + b_x + b_h for every (b, position).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) encoder_pointwise_k(Slabs hh, const float* xg, long long ld_xg_row,
                                                           long long ld_xg_t, int t, const int64_t* lens,
                                                           int reverse, float* h, float* c, float* H, float* Mem,
                                                           long long ld_out_b, long long ld_out_t, int out_col0,
                                                           int B, int D, const float* b_extra) {
    const int per_row = D >> 2;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)B * per_row) return;
    const long long b = idx / per_row;
    const int j = (int)(idx - b * per_row) << 2;
    const int len = (int)lens[b];
    if (t >= len) return;
    const int pos = reverse ? (len - 1 - t) : t;
    const float* xr = xg + b * ld_xg_row + (long long)pos * ld_xg_t;
    f32x4 g[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) g[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
    slab_accum<4>(g, hh, b, j, D);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        g[q] += ld4(xr + q * D + j);
        if (b_extra) g[q] += ld4(b_extra + q * D + j);
    }
    const f32x4 cp = ld4(c + b * D + j);
    f32x4 cn, hn;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float ig = sigmoidf_(g[0][e]), fg = sigmoidf_(g[1][e]), gg = tanhf(g[2][e]), og = sigmoidf_(g[3][e]);
        cn[e] = fg * cp[e] + ig * gg;
        hn[e] = og * tanhf(cn[e]);
    }
    st4(c + b * D + j, cn);
    st4(h + b * D + j, hn);
    st4(H + b * ld_out_b + (long long)pos * ld_out_t + out_col0 + j, hn);
    if (Mem) st4(Mem + b * ld_out_b + (long long)pos * ld_out_t + out_col0 + j, cn);
}

int encoder_pointwise(Slabs hh, const float* xg, long long ld_xg_row, long long ld_xg_t, int t,
                      const int64_t* lens, int reverse, float* h, float* c, float* H, float* Mem,
                      long long ld_out_b, long long ld_out_t, int out_col0, int B, int D, const float* b_extra,
                      hipStream_t s) {
    if (D & 3) return SET_ERR_UNSUPPORTED;
    const long long n = (long long)B * (D >> 2);
    ProfScope ps("encoder_pointwise", s, 0.0, 4.0 * B * D * (4.0 * hh.n + 4.0 + 5.0));
    hipLaunchKernelGGL(encoder_pointwise_k, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, hh, xg, ld_xg_row,
                       ld_xg_t, t, lens, reverse, h, c, H, Mem, ld_out_b, ld_out_t, out_col0, B, D, b_extra);
    SET_LAUNCH_CHECK();
    return SET_OK;
}

// ---------------------------------------------------------------------------------------------
// mask[row] = (sum_d x[row, d]) != 0   (editnet.py:340, dcnet.py:239): one wave per row
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) rowsum_mask_k(const float* x, long long ld_row, int rows, int D, float* mask) {
    const int wave = (int)((blockIdx.x * (long long)blockDim.x + threadIdx.x) >> 6);
    const int lane = threadIdx.x & 63;
    if (wave >= rows) return;
    const float* p = x + (long long)wave * ld_row;
    float s = 0.f;
    for (int d = lane * 4; d < D; d += 256) {
        f32x4 v = ld4(p + d);
        s += (v[0] + v[1]) + (v[2] + v[3]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) mask[wave] = (s != 0.f) ? 1.f : 0.f;
}

int rowsum_mask(const float* x, long long ld_row, int rows, int D, float* mask, hipStream_t s) {
    if (D & 3) return SET_ERR_UNSUPPORTED;
    ProfScope ps("rowsum_mask", s, 0.0, 4.0 * rows * D);
    hipLaunchKernelGGL(rowsum_mask_k, dim3((unsigned)cdiv(rows, 4)), dim3(256), 0, s, x, ld_row, rows, D, mask);
    SET_LAUNCH_CHECK();
    return SET_OK;
}

int zero_f32(float* p, size_t n, hipStream_t s) {
    SET_HIP_TRY(hipMemsetAsync(p, 0, n * sizeof(float), s));
    return SET_OK;
}

}  // namespace set
