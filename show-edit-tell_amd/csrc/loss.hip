// Token cross-entropy of the training loops (editnet.py:571-577, dcnet.py:391-397):
//     scores  = pack_padded_sequence(scores, decode_lengths, batch_first=True).data
//     targets = pack_padded_sequence(caps_sorted[:, 1:], decode_lengths, batch_first=True).data
//     loss    = CrossEntropyLoss()(scores, targets)
// on the (B, T, V) scores where they lie (any batch / time strides): the packed rows are exactly the (b, t) with
// t < decode_lengths[b]; the batch is sorted by decreasing length, so row (b, t) is live iff b < live[t].  One workgroup
// per (t, b) row: a single pass keeps a running maximum and a rescaled sum of exponentials (one read of the row), the
// backward writes (softmax - onehot) * dloss straight into a (T, B, V4) buffer whose rows are padded to a multiple of
// 4 floats with zeros — the layout the fc weight / input gradient contractions read in 16-byte chunks — and zeros the
// rows of finished sequences.  HBM: forward reads the scores once, backward reads them once and writes the gradient.
#include "set_common.h"

namespace set {

constexpr int XE_MAX_T = 64;
struct XeLive { int n[XE_MAX_T]; };          // live[t] = number of sequences with decode length > t

__device__ __forceinline__ void wave_max_sum(float& m, float& s) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float m2 = __shfl_down(m, off, 64), s2 = __shfl_down(s, off, 64);
        const float mm = fmaxf(m, m2);
        s = s * __expf(m - mm) + s2 * __expf(m2 - mm);
        m = mm;
    }
}

__global__ void __launch_bounds__(256) xe_loss_fwd_k(const float* scores, long long sb, long long st, const int64_t* targets,
                                                     long long tb, long long tt, XeLive live, int B, int V, float* rowloss,
                                                     float* lse_out) {
    __shared__ float red_m[4], red_s[4];
    const int t = blockIdx.x / B, b = blockIdx.x - t * B;
    if (b >= live.n[t]) {
        if (threadIdx.x == 0) { rowloss[blockIdx.x] = 0.f; lse_out[blockIdx.x] = 0.f; }
        return;
    }
    const float* x = scores + b * sb + t * st;
    float m = -3.0e38f, s = 0.f;          // finite floor: lanes without an element merge as exp(0) * 0, never inf - inf
    for (int v = threadIdx.x; v < V; v += 256) {
        const float xv = x[v];
        if (xv > m) { s = s * __expf(m - xv) + 1.f; m = xv; }
        else s += __expf(xv - m);
    }
    wave_max_sum(m, s);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) { red_m[wv] = m; red_s[wv] = s; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float mm = fmaxf(fmaxf(red_m[0], red_m[1]), fmaxf(red_m[2], red_m[3]));
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) ss += red_s[i] * __expf(red_m[i] - mm);
        const float lse = mm + logf(ss);
        // an id outside [0, V) (corrupted caption, vocabulary mismatch): F.cross_entropy raises; without a host round trip the
        // loud form available here is a NaN loss — never a silently clamped label
        const long long tg = targets[b * tb + t * tt];
        rowloss[blockIdx.x] = (tg < 0 || tg >= V) ? __builtin_nanf("") : lse - x[tg];
        lse_out[blockIdx.x] = lse;
    }
}

// deterministic sum of n floats by one workgroup
__global__ void __launch_bounds__(256) sum_rows_k(const float* x, int n, float* out) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += x[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) *out = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ void __launch_bounds__(256) xe_loss_bwd_k(const float* scores, long long sb, long long st, const int64_t* targets,
                                                     long long tb, long long tt, XeLive live, int B, int V, int V4,
                                                     const float* lse, const float* dloss, float* grad) {
    const int t = blockIdx.x / B, b = blockIdx.x - t * B;
    float* g = grad + (long long)blockIdx.x * V4;
    if (b >= live.n[t]) {
        for (int v = threadIdx.x; v < V4; v += 256) g[v] = 0.f;
        return;
    }
    const float* x = scores + b * sb + t * st;
    const float l = lse[blockIdx.x], d = *dloss;
    const long long tg = targets[b * tb + t * tt];
    const bool bad = tg < 0 || tg >= V;                      // see the forward: the row's gradient is NaN, not a clamped label's
    for (int v = threadIdx.x; v < V4; v += 256) {
        float o = 0.f;
        if (v < V) o = bad ? __builtin_nanf("") : (__expf(x[v] - l) - (v == (int)tg ? 1.f : 0.f)) * d;
        g[v] = o;
    }
}

}  // namespace set

using namespace set;

extern "C" {

int set_xe_loss_f32(const float* scores, int64_t stride_b, int64_t stride_t, const int64_t* targets, int64_t tstride_b,
                    int64_t tstride_t, const int* live, int B, int T, int V, float* rowloss, float* lse, float* loss_sum,
                    void* stream) {
    if (!scores || !targets || !live || !rowloss || !lse || !loss_sum || B <= 0 || T <= 0 || V <= 0) return SET_ERR_ARG;
    if (T > XE_MAX_T) return SET_ERR_UNSUPPORTED;
    XeLive lv;
    for (int t = 0; t < XE_MAX_T; ++t) lv.n[t] = t < T ? live[t] : 0;
    for (int t = 0; t < T; ++t)
        if (lv.n[t] < 0 || lv.n[t] > B || (t > 0 && lv.n[t] > lv.n[t - 1])) return SET_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(xe_loss_fwd_k, dim3((unsigned)(T * B)), dim3(256), 0, s, scores, (long long)stride_b, (long long)stride_t,
                       targets, (long long)tstride_b, (long long)tstride_t, lv, B, V, rowloss, lse);
    hipLaunchKernelGGL(sum_rows_k, dim3(1), dim3(256), 0, s, (const float*)rowloss, T * B, loss_sum);
    SET_LAUNCH_CHECK();
    return SET_OK;
}

int set_xe_loss_bwd_f32(const float* scores, int64_t stride_b, int64_t stride_t, const int64_t* targets, int64_t tstride_b,
                        int64_t tstride_t, const int* live, int B, int T, int V, const float* lse, const float* dloss,
                        float* grad, int64_t ld_grad, void* stream) {
    if (!scores || !targets || !live || !lse || !dloss || !grad || B <= 0 || T <= 0 || V <= 0) return SET_ERR_ARG;
    if (T > XE_MAX_T) return SET_ERR_UNSUPPORTED;
    if (ld_grad != (((int64_t)V + 3) & ~(int64_t)3)) return SET_ERR_ARG;
    XeLive lv;
    for (int t = 0; t < XE_MAX_T; ++t) lv.n[t] = t < T ? live[t] : 0;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(xe_loss_bwd_k, dim3((unsigned)(T * B)), dim3(256), 0, s, scores, (long long)stride_b, (long long)stride_t,
                       targets, (long long)tstride_b, (long long)tstride_t, lv, B, V, (int)ld_grad, lse, dloss, grad);
    SET_LAUNCH_CHECK();
    return SET_OK;
}

}  // extern "C"
