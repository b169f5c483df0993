// The tail of the training step (editnet.py:580-581, dcnet.py:399-400, editnet_rl.py:684-686):
//     torch.nn.utils.clip_grad_norm_(parameters, 0.25);  optimizer.step()        (torch.optim.Adam)
// as TWO launches over ALL parameters: (1) per-chunk sums of squares of every gradient, (2) every workgroup re-derives the
// global norm from those partials in a fixed order (deterministic, no atomics, no host round trip), forms the clipping
// coefficient and applies the Adam update of its chunk with the coefficient folded into the gradient read.  HBM traffic
// per step = read g (norm) + read g, p, m, v + write p, m, v: 8 x 4 bytes per parameter — the unfused sequence
// (norm, scale g in place, Adam) moves 10 x 4.
#include "set_common.h"

namespace set {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int OPT_MAX_TENSORS = 40;          // per launch (kernel-argument table); longer lists take several launches
constexpr int OPT_CHUNK = 16384;             // elements per workgroup: 16 float4 per thread and array

struct OptTensor {
    float* p;
    const float* g;
    float* m;
    float* v;
    long long n;
    float step_size;     // lr / (1 - beta1^step)
    float inv_bc2_sqrt;  // 1 / sqrt(1 - beta2^step)
    float omb1, beta2, omb2, eps, weight_decay;   // of the tensor's param group; omb = 1 - beta, rounded from double as torch does
    int first_chunk;     // index of this tensor's first chunk within the launch
};

struct OptTable {
    OptTensor t[OPT_MAX_TENSORS];
    int n;
    int chunk_base;      // global index of this launch's chunk 0 (partials of all launches share one array)
};

__device__ __forceinline__ int tensor_of_chunk(const OptTable& tab, int c) {
    int k = 0;
#pragma unroll 1
    for (int i = 1; i < tab.n; ++i) k = (tab.t[i].first_chunk <= c) ? i : k;   // uniform: scalar loop over the table
    return k;
}

__device__ __forceinline__ float block_sum(float s, float* red) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) red[wv] = s;
    __syncthreads();
    const float tot = (red[0] + red[1]) + (red[2] + red[3]);
    __syncthreads();
    return tot;
}

__global__ void __launch_bounds__(256) grad_sumsq_k(const OptTable tab, float* partials) {
    __shared__ float red[4];
    const int c = blockIdx.x;
    const OptTensor& T = tab.t[tensor_of_chunk(tab, c)];
    const long long base = (long long)(c - T.first_chunk) * OPT_CHUNK;
    const long long left = T.n - base;
    const int cnt = left < OPT_CHUNK ? (int)left : OPT_CHUNK;
    const float* g = T.g + base;
    float s = 0.f;
    const int n4 = cnt >> 2;
    for (int i = threadIdx.x; i < n4; i += 256) {
        const f32x4 x = reinterpret_cast<const f32x4*>(g)[i];
        s += (x[0] * x[0] + x[1] * x[1]) + (x[2] * x[2] + x[3] * x[3]);
    }
    for (int i = (n4 << 2) + threadIdx.x; i < cnt; i += 256) s += g[i] * g[i];
    const float tot = block_sum(s, red);
    if (threadIdx.x == 0) partials[tab.chunk_base + c] = tot;
}

// coefficient of clip_grad_norm_: min(1, max_norm / (||g|| + 1e-6)); every workgroup sums ALL partials in the same order
__device__ __forceinline__ float clip_coef(const float* partials, int n_partials, float max_norm, float* red, float* norm_out) {
    if (max_norm <= 0.f && !norm_out) return 1.f;
    float s = 0.f;
    for (int i = threadIdx.x; i < n_partials; i += 256) s += partials[i];
    const float norm = sqrtf(block_sum(s, red));
    if (norm_out && blockIdx.x == 0 && threadIdx.x == 0) *norm_out = norm;
    if (max_norm <= 0.f) return 1.f;
    const float c = max_norm / (norm + 1e-6f);
    return (c < 1.f || c != c) ? c : 1.f;   // torch.clamp(max=1) lets a NaN norm through to the gradients; so do we
}

template <bool SCALE_G>
__global__ void __launch_bounds__(256) clip_adam_k(const OptTable tab, const float* partials, int n_partials, float max_norm,
                                                   float* norm_out) {
    __shared__ float red[4];
    const float coef = clip_coef(partials, n_partials, max_norm, red, tab.chunk_base == 0 ? norm_out : nullptr);
    const int c = blockIdx.x;
    const OptTensor& T = tab.t[tensor_of_chunk(tab, c)];
    const long long base = (long long)(c - T.first_chunk) * OPT_CHUNK;
    const long long left = T.n - base;
    const int cnt = left < OPT_CHUNK ? (int)left : OPT_CHUNK;
    float* p = T.p + base;
    float* g = const_cast<float*>(T.g) + base;
    float* m = T.m + base;
    float* v = T.v + base;
    const float beta2 = T.beta2, eps = T.eps, weight_decay = T.weight_decay;
    const float ss = T.step_size, ib = T.inv_bc2_sqrt, omb1 = T.omb1, omb2 = T.omb2;
    const int n4 = cnt >> 2;
    for (int i = threadIdx.x; i < n4; i += 256) {
        f32x4 gg = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(g) + i);
        f32x4 pp = reinterpret_cast<f32x4*>(p)[i];
        f32x4 mm = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(m) + i);
        f32x4 vv = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(v) + i);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float x = gg[e] * coef;
            if (SCALE_G) gg[e] = x;
            x += weight_decay * pp[e];                         // torch.optim.Adam's L2 form (weight_decay = 0 in the reference)
            mm[e] += (x - mm[e]) * omb1;                        // exp_avg.lerp_(grad, 1 - beta1)
            vv[e] = vv[e] * beta2 + omb2 * x * x;               // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
            pp[e] -= ss * (mm[e] / (sqrtf(vv[e]) * ib + eps));  // param.addcdiv_(exp_avg, sqrt(v)/sqrt(bc2) + eps, -lr/bc1)
        }
        reinterpret_cast<f32x4*>(p)[i] = pp;
        __builtin_nontemporal_store(mm, reinterpret_cast<f32x4*>(m) + i);
        __builtin_nontemporal_store(vv, reinterpret_cast<f32x4*>(v) + i);
        if (SCALE_G) reinterpret_cast<f32x4*>(g)[i] = gg;
    }
    for (int i = (n4 << 2) + threadIdx.x; i < cnt; i += 256) {
        float x = g[i] * coef;
        if (SCALE_G) g[i] = x;
        x += weight_decay * p[i];
        const float mm = m[i] + (x - m[i]) * omb1;
        const float vv = v[i] * beta2 + omb2 * x * x;
        m[i] = mm;
        v[i] = vv;
        p[i] -= ss * (mm / (sqrtf(vv) * ib + eps));
    }
}

static long long chunks_of(long long n) { return (n + OPT_CHUNK - 1) / OPT_CHUNK; }

}  // namespace set

using namespace set;

extern "C" {

size_t set_clip_adam_workspace_bytes(int n, const int64_t* numel) {
    if (n <= 0 || !numel) return 0;
    long long c = 0;
    for (int i = 0; i < n; ++i) c += numel[i] > 0 ? chunks_of(numel[i]) : 0;
    return (size_t)c * sizeof(float) + 256;
}

int set_clip_adam_f32(int n, float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                      const int64_t* numel, const int64_t* step, const double* lr, const double* beta1, const double* beta2,
                      const double* eps, const double* weight_decay, float max_norm, int scale_grads, float* total_norm_out,
                      void* ws, size_t ws_bytes, void* stream) {
    if (n <= 0 || !params || !grads || !exp_avg || !exp_avg_sq || !numel || !lr || !step || !beta1 || !beta2 || !eps ||
        !weight_decay)
        return SET_ERR_ARG;
    long long total_chunks = 0;
    for (int i = 0; i < n; ++i) {
        if (numel[i] < 0 || step[i] < 1) return SET_ERR_ARG;
        if (!(beta1[i] >= 0.0 && beta1[i] < 1.0) || !(beta2[i] >= 0.0 && beta2[i] < 1.0) || !(eps[i] >= 0.0)) return SET_ERR_ARG;
        if (numel[i] == 0) continue;
        if (!params[i] || !grads[i] || !exp_avg[i] || !exp_avg_sq[i]) return SET_ERR_ARG;
        if (!aligned16(params[i]) || !aligned16(grads[i]) || !aligned16(exp_avg[i]) || !aligned16(exp_avg_sq[i]))
            return SET_ERR_UNSUPPORTED;
        total_chunks += chunks_of(numel[i]);
    }
    if (total_chunks == 0) return SET_OK;
    if (total_chunks > 0x7fffffffLL) return SET_ERR_UNSUPPORTED;
    if (!ws || !aligned16(ws) || ws_bytes < (size_t)total_chunks * sizeof(float)) return SET_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    float* partials = (float*)ws;
    const bool need_norm = max_norm > 0.f || total_norm_out;

    // the tensor list in launches of at most OPT_MAX_TENSORS table rows: first every sum-of-squares launch, then the updates
    for (int pass = need_norm ? 0 : 1; pass < 2; ++pass) {
        int i = 0, base = 0;
        while (i < n) {
            OptTable tab;
            tab.n = 0;
            tab.chunk_base = base;
            int c = 0;
            for (; i < n && tab.n < OPT_MAX_TENSORS; ++i) {
                if (numel[i] == 0) continue;
                OptTensor& T = tab.t[tab.n++];
                const double bc1 = 1.0 - pow(beta1[i], (double)step[i]);
                const double bc2 = 1.0 - pow(beta2[i], (double)step[i]);
                T.p = params[i], T.g = grads[i], T.m = exp_avg[i], T.v = exp_avg_sq[i], T.n = numel[i];
                T.step_size = (float)(lr[i] / bc1);
                T.inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
                T.omb1 = (float)(1.0 - beta1[i]), T.beta2 = (float)beta2[i], T.omb2 = (float)(1.0 - beta2[i]);
                T.eps = (float)eps[i], T.weight_decay = (float)weight_decay[i];
                T.first_chunk = c;
                c += (int)chunks_of(numel[i]);
            }
            if (tab.n == 0) break;
            if (pass == 0)
                hipLaunchKernelGGL(grad_sumsq_k, dim3(c), dim3(256), 0, st, tab, partials);
            else if (scale_grads)
                hipLaunchKernelGGL(clip_adam_k<true>, dim3(c), dim3(256), 0, st, tab, (const float*)partials, (int)total_chunks,
                                   need_norm ? max_norm : 0.f, total_norm_out);
            else
                hipLaunchKernelGGL(clip_adam_k<false>, dim3(c), dim3(256), 0, st, tab, (const float*)partials, (int)total_chunks,
                                   need_norm ? max_norm : 0.f, total_norm_out);
            base += c;
        }
    }
    SET_LAUNCH_CHECK();
    return SET_OK;
}

}  // extern "C"
