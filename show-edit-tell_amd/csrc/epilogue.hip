// Step epilogues on the vocabulary row (one workgroup per batch row, HBM/L2-bound reduction):
//   greedy (editnet_rl.py:514-543, dcnet_rl.py:313-340): log_softmax, first arg-max, <end> -> 0,
//   `unfinished` latch, seq / seqLogprobs stores, early-break emulation, and the NEXT step's
//   embedding gather relu(E[it]) (editnet.py:300-304) fused in so the loop needs no extra launch.
#include "set_common.h"

namespace set {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float logit_at(const Slabs& s, const float* bias, long long row, int v) {
    const float* p = s.p + row * s.ld + v;
    float x = p[0];
    for (int i = 1; i < s.n; ++i) x += p[(long long)i * s.stride];
    return bias ? x + bias[v] : x;
}

// grid = B rows.  alive[t] counts rows still unfinished after step t (zeroed by the caller);
// once alive[t-1] == 0 the reference has left its loop (`break`, editnet_rl.py:546) and nothing
// more is written to seq / seq_logp.
// REG = true: the row (<= 4*256*GP_MAXQ logits) is read ONCE as float4 (all K-slabs + bias summed
// in registers), max / first-argmax and sum-exp are reduced from registers.
constexpr int GP_MAXQ = 12;

template <bool REG>
__global__ void __launch_bounds__(256) greedy_pick_k(Slabs logits, const float* bias, int V, int t, int max_len,
                                                     long long end_idx, long long* seq, float* seq_logp,
                                                     long long* it_buf, int* unfinished, int* alive,
                                                     const float* table, float* emb_out, int D) {
    __shared__ float s_val[4];
    __shared__ int s_idx[4];
    __shared__ float s_sum[4];
    __shared__ long long s_tok;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    f32x4 x[GP_MAXQ];
    if (REG) {
        const float* row = logits.p + (long long)b * logits.ld;
        // slab 0 (+ bias), then the remaining K-slabs in index order; GP_MAXQ independent float4 loads
        // are in flight per pass
        const bool bias4 = bias && !(V & 3) && ((reinterpret_cast<uintptr_t>(bias) & 15u) == 0);
        // the bias does not depend on the slabs: request it first (added last, as before)
        f32x4 bq[GP_MAXQ];
#pragma unroll
        for (int q = 0; q < GP_MAXQ; ++q) {
            const int v = (tid + 256 * q) * 4;
            bq[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (v < V) {
                if (bias4) {
                    bq[q] = *reinterpret_cast<const f32x4*>(bias + v);
                } else if (bias) {
                    bq[q][0] = bias[v];
                    if (v + 1 < V) bq[q][1] = bias[v + 1];
                    if (v + 2 < V) bq[q][2] = bias[v + 2];
                    if (v + 3 < V) bq[q][3] = bias[v + 3];
                }
            }
        }
#pragma unroll
        for (int q = 0; q < GP_MAXQ; ++q) {
            const int v = (tid + 256 * q) * 4;
            x[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (v < V) x[q] = *reinterpret_cast<const f32x4*>(row + v);
        }
        int i = 1;
        for (; i + 2 <= logits.n; i += 2) {          // two slabs (2 x GP_MAXQ loads) in flight, added in slab order
            f32x4 y[GP_MAXQ], z[GP_MAXQ];
#pragma unroll
            for (int q = 0; q < GP_MAXQ; ++q) {
                const int v = (tid + 256 * q) * 4;
                y[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
                z[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (v < V) {
                    y[q] = *reinterpret_cast<const f32x4*>(row + (long long)i * logits.stride + v);
                    z[q] = *reinterpret_cast<const f32x4*>(row + (long long)(i + 1) * logits.stride + v);
                }
            }
#pragma unroll
            for (int q = 0; q < GP_MAXQ; ++q) { x[q] += y[q]; x[q] += z[q]; }
        }
        for (; i < logits.n; ++i) {
            f32x4 y[GP_MAXQ];
#pragma unroll
            for (int q = 0; q < GP_MAXQ; ++q) {
                const int v = (tid + 256 * q) * 4;
                y[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (v < V) y[q] = *reinterpret_cast<const f32x4*>(row + (long long)i * logits.stride + v);
            }
#pragma unroll
            for (int q = 0; q < GP_MAXQ; ++q) x[q] += y[q];
        }
#pragma unroll
        for (int q = 0; q < GP_MAXQ; ++q) {
            const int v = (tid + 256 * q) * 4;
            if (bias) x[q] += bq[q];
#pragma unroll
            for (int e = 0; e < 4; ++e) if (v + e >= V) x[q][e] = -INFINITY;     // padding columns / rows past V
        }
#pragma unroll
        for (int q = 0; q < GP_MAXQ; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (x[q][e] > best) { best = x[q][e]; bi = (tid + 256 * q) * 4 + e; }   // ascending index per thread
    } else {
        for (int v = tid; v < V; v += 256) {
            const float xv = logit_at(logits, bias, b, v);
            if (xv > best) { best = xv; bi = v; }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o);
        const int oi = __shfl_xor(bi, o);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (lane == 0) { s_val[wave] = best; s_idx[wave] = bi; }
    __syncthreads();
    best = s_val[0]; bi = s_idx[0];
#pragma unroll
    for (int w = 1; w < 4; ++w)
        if (s_val[w] > best || (s_val[w] == best && s_idx[w] < bi)) { best = s_val[w]; bi = s_idx[w]; }
    float sum = 0.f;
    if (REG) {
#pragma unroll
        for (int q = 0; q < GP_MAXQ; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) sum += expf(x[q][e] - best);               // exp(-inf) == 0 for padding
    } else {
        for (int v = tid; v < V; v += 256) sum += expf(logit_at(logits, bias, b, v) - best);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    if (lane == 0) s_sum[wave] = sum;
    __syncthreads();
    if (tid == 0) {
        const float total = (s_sum[0] + s_sum[1]) + (s_sum[2] + s_sum[3]);
        const float logp = (best - best) - logf(total);         // log_softmax at the arg-max
        long long it = bi;
        if (it == end_idx) it = 0;
        int unf = (t == 0) ? (it > 0) : (unfinished[b] && it > 0);
        it = unf ? it : 0;
        const bool broken = (t > 0) && (alive[t - 1] == 0);
        if (t < max_len && !broken) {
            seq[(long long)b * max_len + t] = it;
            seq_logp[(long long)b * max_len + t] = logp;
        }
        unfinished[b] = unf;
        if (unf) atomicAdd(&alive[t], 1);
        it_buf[b] = it;
        s_tok = it;
    }
    __syncthreads();
    if (emb_out && table) {
        const long long tok = s_tok;
        for (int d = tid * 4; d < D; d += 1024) {
            f32x4 v = *reinterpret_cast<const f32x4*>(table + tok * D + d);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
            *reinterpret_cast<f32x4*>(emb_out + (long long)b * D + d) = v;
        }
    }
}

int greedy_pick(Slabs logits, const float* bias, int V, int t, int max_len, long long end_idx, long long* seq,
                float* seq_logp, long long* it, int* unfinished, int* alive, const float* table, float* emb_out,
                int D, int B, hipStream_t s) {
    if (B <= 0) return SET_OK;
    if (D & 3) return SET_ERR_UNSUPPORTED;
    ProfScope ps("greedy_pick", s, 0.0, 4.0 * B * (2.0 * V * logits.n + 2.0 * D));
    const bool reg = V <= 4 * 256 * GP_MAXQ && !(logits.ld & 3) && !(logits.stride & 3) && aligned16(logits.p);
    if (reg)
        hipLaunchKernelGGL(greedy_pick_k<true>, dim3(B), dim3(256), 0, s, logits, bias, V, t, max_len, end_idx, seq,
                           seq_logp, it, unfinished, alive, table, emb_out, D);
    else
        hipLaunchKernelGGL(greedy_pick_k<false>, dim3(B), dim3(256), 0, s, logits, bias, V, t, max_len, end_idx, seq,
                           seq_logp, it, unfinished, alive, table, emb_out, D);
    SET_LAUNCH_CHECK();
    return SET_OK;
}

// it[:] = value; unfinished[:] = 1; alive[0..n_alive) = 0
__global__ void __launch_bounds__(256) set_tokens_k(long long* it, long long value, int* unfinished, int* alive,
                                                    int n_alive, int B) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B) { it[i] = value; unfinished[i] = 1; }
    if (i < n_alive) alive[i] = 0;
}

int set_tokens(long long* it, long long value, int* unfinished, int* alive, int n_alive, int B, hipStream_t s) {
    const int n = B > n_alive ? B : n_alive;
    hipLaunchKernelGGL(set_tokens_k, dim3(cdiv(n, 256)), dim3(256), 0, s, it, value, unfinished, alive, n_alive, B);
    SET_LAUNCH_CHECK();
    return SET_OK;
}

__global__ void __launch_bounds__(256) iota_i64_k(long long* p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = i;
}

int iota_i64(long long* p, int n, hipStream_t s) {
    hipLaunchKernelGGL(iota_i64_k, dim3(cdiv(n, 256)), dim3(256), 0, s, p, n);
    SET_LAUNCH_CHECK();
    return SET_OK;
}

}  // namespace set
