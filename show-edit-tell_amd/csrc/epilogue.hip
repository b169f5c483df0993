// Step epilogues on the vocabulary row (one workgroup per batch row, HBM/L2-bound reduction):
//   greedy (editnet_rl.py:514-543, dcnet_rl.py:313-340): log_softmax, first arg-max, <end> -> 0,
//   `unfinished` latch, seq / seqLogprobs stores, early-break emulation, and the NEXT step's
//   embedding gather relu(E[it]) (editnet.py:300-304) fused in so the loop needs no extra launch.
#include "set_common.h"
#include "philox.h"

namespace set {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float logit_at(const Slabs& s, const float* bias, long long row, int v) {
    const float* p = s.p + row * s.ld + v;
    float x = p[0];
    for (int i = 1; i < s.n; ++i) x += p[(long long)i * s.stride];
    return bias ? x + bias[v] : x;
}

// ---- LstmTail (set_common.h): the next timestep's attention-LSTM cell of this row, finished by the workgroup that chose
// the row's word.  Arithmetic and operand order are lstm_pointwise_k's (slab 0, 1, ..., pre, table row; c' = f c + i g,
// h' = o tanh c').  tail_fetch requests everything that does not depend on the word (the first two slabs, pre, c) so the
// latency sits under the logits fetch; tail_finish adds the table row and stores.
__device__ __forceinline__ float tail_sigm(float x) { return 1.f / (1.f + expf(-x)); }
struct TailRegs {
    f32x4 s0[4], s1[4], pre[4], c;
};
__device__ __forceinline__ void tail_fetch(const LstmTail& L, int b, int j, TailRegs& r) {
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    const float* p = L.g0.p + (long long)b * L.g0.ld + j;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        r.s0[q] = L.g0.n > 0 ? *reinterpret_cast<const f32x4*>(p + q * L.D) : z;
        r.s1[q] = L.g0.n > 1 ? *reinterpret_cast<const f32x4*>(p + L.g0.stride + q * L.D) : z;
        r.pre[q] = L.pre ? *reinterpret_cast<const f32x4*>(L.pre + (long long)b * L.ldpre + q * L.D + j) : z;
    }
    r.c = *reinterpret_cast<const f32x4*>(L.c_in + (long long)b * L.D + j);
}
struct TailRow {
    f32x4 xt[4];
};
__device__ __forceinline__ void tail_row_fetch(const LstmTail& L, int j, const float* trow, TailRow& w) {
#pragma unroll
    for (int q = 0; q < 4; ++q) w.xt[q] = *reinterpret_cast<const f32x4*>(trow + q * L.D + j);
}
__device__ __forceinline__ void tail_finish(const LstmTail& L, int b, int j, const TailRow& w, const TailRegs& r) {
    f32x4 g[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        g[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (L.g0.n > 0) g[q] += r.s0[q];
        if (L.g0.n > 1) g[q] += r.s1[q];
    }
    for (int i = 2; i < L.g0.n; i += 2) {            // the remaining partials two at a time (requested together, added in order)
        const float* p = L.g0.p + (long long)i * L.g0.stride + (long long)b * L.g0.ld + j;
        const bool two = i + 1 < L.g0.n;
        f32x4 v0[4], v1[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            v0[q] = *reinterpret_cast<const f32x4*>(p + q * L.D);
            if (two) v1[q] = *reinterpret_cast<const f32x4*>(p + L.g0.stride + q * L.D);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) { g[q] += v0[q]; if (two) g[q] += v1[q]; }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (L.pre) g[q] += r.pre[q];
        g[q] += w.xt[q];
    }
    f32x4 cn, hn;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float ig = tail_sigm(g[0][e]), fg = tail_sigm(g[1][e]), gg = tanhf(g[2][e]);
        const float og = tail_sigm(g[3][e]);
        cn[e] = fg * r.c[e] + ig * gg;
        hn[e] = og * tanhf(cn[e]);
    }
    *reinterpret_cast<f32x4*>(L.c_out + (long long)b * L.D + j) = cn;
    *reinterpret_cast<f32x4*>(L.h_out + (long long)b * L.D + j) = hn;
}
__device__ __forceinline__ const float* tail_row(const LstmTail& L, long long tok) {
    tok = tok < 0 ? 0 : (tok >= L.nrows ? L.nrows - 1 : tok);       // clamped like RowGather::row
    return L.tab + tok * L.ld_tab + L.col0;
}

static bool tail_ok(const LstmTail& L) {
    return L.D > 0 && !(L.D & 3) && L.g0.p && L.g0.n >= 1 && !(L.g0.ld & 3) && !(L.g0.stride & 3) && aligned16(L.g0.p) &&
           L.tab && !(L.ld_tab & 3) && !(L.col0 & 3) && aligned16(L.tab) && L.nrows > 0 && L.c_in && L.c_out && L.h_out &&
           aligned16(L.c_in) && aligned16(L.c_out) && aligned16(L.h_out) && (!L.pre || (aligned16(L.pre) && !(L.ldpre & 3)));
}

// grid = B rows.  alive[t] counts rows still unfinished after step t (zeroed by the caller);
// once alive[t-1] == 0 the reference has left its loop (`break`, editnet_rl.py:546) and nothing
// more is written to seq / seq_logp.
// REG = true: the row (<= 4*256*GP_MAXQ logits) is read ONCE as float4 (all K-slabs + bias summed
// in registers), max / first-argmax and sum-exp are reduced from registers.
constexpr int GP_MAXQ = 12;

template <bool REG, bool TAIL>
__global__ void SET_VGPR_CAP __launch_bounds__(256) greedy_pick_k(Slabs logits, const float* bias, int V, int t, int max_len,
                                                     long long end_idx, long long* seq, float* seq_logp,
                                                     long long* it_buf, int* unfinished, int* alive,
                                                     const float* table, float* emb_out, int D, const LstmTail tail,
                                                     const int* row_limit) {
    __shared__ float s_val[4];
    __shared__ int s_idx[4];
    __shared__ float s_sum[4];
    __shared__ long long s_tok;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // bookkeeping words of the serial tail (thread 0): requested now, under the row fetch, instead of as a dependent round
    // trip after the reductions.  (The loop-left test of set_common.h RowGate is NOT made here: a dependent load at the top of
    // every 10-us kernel of the chain costs more — 2 % of a single-stream decode when all seven kernels of a timestep make
    // it — than these short kernels cost after the break; the three GEMM launches and the attention launch make it.)
    int unf_prev = 1, alive_prev = 1;
    if (t > 0) {
        if (TAIL || tid == 0) unf_prev = unfinished[b];          // (TAIL: every thread derives the word right after the arg-max)
        if (tid == 0) alive_prev = alive[t - 1];
    }
    TailRegs tr;
    const bool tail0 = TAIL && tid * 4 < tail.D;
    if (TAIL) { if (tail0) tail_fetch(tail, b, tid * 4, tr); }
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
    if (bi == 0x7fffffff) { bi = 0; best = __builtin_nanf(""); }   // no comparison succeeded: the row is all NaN.  Word 0 and a NaN log-prob, never an out-of-range gather
    // set_decode_row_limits: row b's caption is at most row_limit[b] words long — at that timestep the loop ends it (<end> is
    // taken whatever the scores say; the recorded log-prob stays the arg-max's)
    if (row_limit && t + 1 >= row_limit[b]) bi = (int)end_idx;
    // the word follows from the arg-max and the row's latch alone: with a tail, every thread requests its piece of the
    // token-table row NOW, so that round trip runs under the sum-exp pass instead of after the serial bookkeeping
    TailRow tw;
    if (TAIL) {
        long long it = bi;
        if (it == end_idx) it = 0;
        const int unf = (t == 0) ? (it > 0) : (unf_prev && it > 0);
        it = unf ? it : 0;
        if (tail0) tail_row_fetch(tail, tid * 4, tail_row(tail, it), tw);
    }
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
        int unf = (t == 0) ? (it > 0) : (unf_prev && it > 0);
        it = unf ? it : 0;
        const bool broken = (t > 0) && (alive_prev == 0);
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
    if (TAIL) {
        if (tail0) tail_finish(tail, b, tid * 4, tw, tr);
        const float* trow = tail_row(tail, s_tok);
        for (int j = tid * 4 + 1024; j < tail.D; j += 1024) {
            TailRegs r2;
            TailRow w2;
            tail_fetch(tail, b, j, r2);
            tail_row_fetch(tail, j, trow, w2);
            tail_finish(tail, b, j, w2, r2);
        }
    }
}

int greedy_pick(Slabs logits, const float* bias, int V, int t, int max_len, long long end_idx, long long* seq,
                float* seq_logp, long long* it, int* unfinished, int* alive, const float* table, float* emb_out,
                int D, int B, hipStream_t s, const LstmTail* tail) {
#if defined(SET_EXP_SKIP_POINTWISE) || defined(SET_EXP_SKIP_PICK)      // diagnostic build (EXPERIMENTS 5.7): the launch is dropped, results are garbage
    return SET_OK;
#endif
    if (B <= 0) return SET_OK;
    if (D & 3) return SET_ERR_UNSUPPORTED;
    if (tail && !tail_ok(*tail)) return SET_ERR_ARG;
    const LstmTail tl = tail ? *tail : LstmTail();
    ProfScope ps("greedy_pick", s, 0.0,
                 4.0 * B * (2.0 * V * logits.n + 2.0 * D + (tail ? tl.D * (4.0 * (tl.g0.n + 2) + 3.0) : 0.0)));
    const bool reg = V <= 4 * 256 * GP_MAXQ && !(logits.ld & 3) && !(logits.stride & 3) && aligned16(logits.p);
#define SET_PICK_LAUNCH(REG, TAIL)                                                                                    \
    hipLaunchKernelGGL((greedy_pick_k<REG, TAIL>), dim3(B), dim3(256), 0, s, logits, bias, V, t, max_len, end_idx, seq, \
                       seq_logp, it, unfinished, alive, table, emb_out, D, tl, g_row_limit)
    if (reg) { if (tail) SET_PICK_LAUNCH(true, true); else SET_PICK_LAUNCH(true, false); }
    else { if (tail) SET_PICK_LAUNCH(false, true); else SET_PICK_LAUNCH(false, false); }
#undef SET_PICK_LAUNCH
    SET_LAUNCH_CHECK();
    return SET_OK;
}

// ---------------------------------------------------------------------------------------------
// Multinomial sampling epilogue (editnet_rl.py:521-543, dcnet_rl.py:320-340): the sampling twin of
// greedy_pick_k.  prob = exp(log_softmax(logits)); it ~ Categorical(prob); seqLogprobs = logprobs[it];
// then the same <end> -> 0 / `unfinished` latch / early-break bookkeeping and next-embedding gather.
//
// Device RNG: Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11), counter
// = (row, timestep, offset_lo, offset_hi), key = (seed_lo, seed_hi): one independent 24-bit uniform per
// (seed, offset, row, timestep), reproducible and independent of launch geometry.
// Draw: inverse CDF over a FIXED enumeration of the vocabulary (thread-major: thread tid owns the words
// (tid + 256 q) * 4 + e, q = 0.., e = 0..3, in that order; any fixed enumeration samples the same categorical
// distribution).  One block-wide inclusive scan of the per-thread probability mass locates the owning
// thread, which then walks its <= 48 words.  u = (r >> 8) * 2^-24 lies in [0, 1 - 2^-24], so u * total
// rounds strictly below total and exactly one thread owns the target.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float sample_uniform(unsigned long long seed, unsigned long long offset, int row, int t) {
    uint32_t c[4] = {(uint32_t)row, (uint32_t)t, (uint32_t)offset, (uint32_t)(offset >> 32)};
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    return (float)(c[0] >> 8) * (1.0f / 16777216.0f);
}

__global__ void __launch_bounds__(64) philox_fill_k(uint32_t* out, int n, unsigned long long seed,
                                                    unsigned long long offset) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;      // element i -> counter (i, 0, offset): 4 words each
    if (i >= n) return;
    uint32_t c[4] = {(uint32_t)i, 0u, (uint32_t)offset, (uint32_t)(offset >> 32)};
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    out[4 * i] = c[0]; out[4 * i + 1] = c[1]; out[4 * i + 2] = c[2]; out[4 * i + 3] = c[3];
}

int philox_fill(uint32_t* out, int n, unsigned long long seed, unsigned long long offset, hipStream_t s) {
    if (n <= 0) return SET_OK;
    hipLaunchKernelGGL(philox_fill_k, dim3(cdiv(n, 64)), dim3(64), 0, s, out, n, seed, offset);
    SET_LAUNCH_CHECK();
    return SET_OK;
}

struct SampleOut {
    long long* raw_ids;   // (B) sampled word BEFORE the <end> -> 0 rewriting; -1 once the loop has been left
    float* lse;           // (B) log-sum-exp of the row (for the backward of the gathered log-prob)
    float* step_logp;     // (B) this step's log-prob (0 once the loop has been left)
};

template <bool REG, bool TAIL>
__global__ void __launch_bounds__(256) sample_pick_k(Slabs logits, const float* bias, int V, int t, int max_len,
                                                     long long end_idx, long long* seq, float* seq_logp,
                                                     long long* it_buf, int* unfinished, int* alive,
                                                     const float* table, float* emb_out, int D,
                                                     unsigned long long seed, unsigned long long offset,
                                                     SampleOut so, const LstmTail tail) {
    __shared__ float s_red[4];
    __shared__ float s_scan[4];
    __shared__ float s_max;
    __shared__ long long s_tok;
    __shared__ int s_pick;
    __shared__ float s_pick_x;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    f32x4 x[GP_MAXQ];
    float best = -INFINITY;
    if (REG) {
        const float* row = logits.p + (long long)b * logits.ld;
#pragma unroll
        for (int q = 0; q < GP_MAXQ; ++q) {
            const int v = (tid + 256 * q) * 4;
            x[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (v < V) x[q] = *reinterpret_cast<const f32x4*>(row + v);
        }
        for (int i = 1; i < logits.n; ++i) {             // K-slabs in index order, as greedy_pick_k
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
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (v + e < V) { if (bias) x[q][e] += bias[v + e]; }
                else x[q][e] = -INFINITY;
                best = fmaxf(best, x[q][e]);
            }
        }
    } else {
        for (int v = tid; v < V; v += 256) best = fmaxf(best, logit_at(logits, bias, b, v));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) best = fmaxf(best, __shfl_xor(best, o));
    if (lane == 0) s_red[wave] = best;
    if (tid == 0) s_pick = -1;
    __syncthreads();
    best = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
    // per-thread probability mass (unnormalised), then inclusive scan over the threads
    float mass = 0.f;
    if (REG) {
#pragma unroll
        for (int q = 0; q < GP_MAXQ; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) mass += expf(x[q][e] - best);                  // exp(-inf) == 0
    } else {
        for (int v = tid; v < V; v += 256) mass += expf(logit_at(logits, bias, b, v) - best);
    }
    float incl = mass;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const float up = __shfl_up(incl, o);
        if (lane >= o) incl += up;
    }
    if (lane == 63) s_scan[wave] = incl;
    __syncthreads();
    float wave_base = 0.f;
#pragma unroll
    for (int w = 0; w < 3; ++w) if (w < wave) wave_base += s_scan[w];
    incl += wave_base;
    const float total = ((s_scan[0] + s_scan[1]) + s_scan[2]) + s_scan[3];
    const float target = sample_uniform(seed, offset, b, t) * total;
    // the owner is the first thread whose inclusive mass exceeds the target (monotone scan; `excl` of a thread
    // is recomputed from its own incl, so neighbours may disagree by an ulp: ownership is decided on incl only)
    float prev_incl = __shfl_up(incl, 1);
    if (lane == 0) prev_incl = wave_base;
    const bool owner = (target < incl) && !(target < prev_incl);
    if (owner) {
        float c = prev_incl;
        int pick = -1;
        float pick_x = 0.f;
        if (REG) {
#pragma unroll
            for (int q = 0; q < GP_MAXQ; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float pe = expf(x[q][e] - best);
                    c += pe;
                    if (pick < 0 && pe > 0.f && target < c) { pick = (tid + 256 * q) * 4 + e; pick_x = x[q][e]; }
                }
            if (pick < 0) {                      // rounding left the target at the very end of this thread's span
#pragma unroll
                for (int q = 0; q < GP_MAXQ; ++q)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (x[q][e] > -INFINITY && expf(x[q][e] - best) > 0.f) { pick = (tid + 256 * q) * 4 + e; pick_x = x[q][e]; }
            }
        } else {
            int last = -1;
            float last_x = 0.f;
            for (int v = tid; v < V; v += 256) {
                const float xv = logit_at(logits, bias, b, v);
                const float pe = expf(xv - best);
                if (pe > 0.f) { last = v; last_x = xv; }
                c += pe;
                if (pe > 0.f && target < c) { pick = v; pick_x = xv; break; }
            }
            if (pick < 0) { pick = last; pick_x = last_x; }
        }
        s_pick = pick;
        s_pick_x = pick_x;
    }
    __syncthreads();
    if (tid == 0) {
        int pick = s_pick;
        float px = s_pick_x;
        if (pick < 0) {                          // cannot happen for finite logits; keep the row well defined
            pick = 0;
            px = REG ? best : logit_at(logits, bias, b, 0);
        }
        const float lse = best + logf(total);
        const float logp = (px - best) - logf(total);
        long long it = pick;
        if (it == end_idx) it = 0;
        int unf = (t == 0) ? (it > 0) : (unfinished[b] && it > 0);
        it = unf ? it : 0;
        const bool broken = (t > 0) && (alive[t - 1] == 0);
        if (t < max_len && !broken) {
            seq[(long long)b * max_len + t] = it;
            if (seq_logp) seq_logp[(long long)b * max_len + t] = logp;
        }
        if (so.raw_ids) so.raw_ids[b] = broken ? -1 : (long long)pick;
        if (so.lse) so.lse[b] = lse;
        if (so.step_logp) so.step_logp[b] = broken ? 0.f : logp;
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
    if (TAIL) {
        const float* trow = tail_row(tail, s_tok);
        for (int j = tid * 4; j < tail.D; j += 1024) {
            TailRegs r2;
            TailRow w2;
            tail_fetch(tail, b, j, r2);
            tail_row_fetch(tail, j, trow, w2);
            tail_finish(tail, b, j, w2, r2);
        }
    }
}

int sample_pick(Slabs logits, const float* bias, int V, int t, int max_len, long long end_idx, long long* seq,
                float* seq_logp, long long* it, int* unfinished, int* alive, const float* table, float* emb_out, int D,
                int B, unsigned long long seed, unsigned long long offset, long long* raw_ids, float* lse,
                float* step_logp, hipStream_t s, const LstmTail* tail) {
    if (B <= 0) return SET_OK;
    if (D & 3) return SET_ERR_UNSUPPORTED;
    if (tail && !tail_ok(*tail)) return SET_ERR_ARG;
    const LstmTail tl = tail ? *tail : LstmTail();
    ProfScope ps("sample_pick", s, 0.0,
                 4.0 * B * (1.0 * V * logits.n + 2.0 * D + (tail ? tl.D * (4.0 * (tl.g0.n + 2) + 3.0) : 0.0)));
    const bool reg = V <= 4 * 256 * GP_MAXQ && !(logits.ld & 3) && !(logits.stride & 3) && aligned16(logits.p);
    SampleOut so{raw_ids, lse, step_logp};
#define SET_PICK_LAUNCH(REG, TAIL)                                                                                    \
    hipLaunchKernelGGL((sample_pick_k<REG, TAIL>), dim3(B), dim3(256), 0, s, logits, bias, V, t, max_len, end_idx, seq, \
                       seq_logp, it, unfinished, alive, table, emb_out, D, seed, offset, so, tl)
    if (reg) { if (tail) SET_PICK_LAUNCH(true, true); else SET_PICK_LAUNCH(true, false); }
    else { if (tail) SET_PICK_LAUNCH(false, true); else SET_PICK_LAUNCH(false, false); }
#undef SET_PICK_LAUNCH
    SET_LAUNCH_CHECK();
    return SET_OK;
}

// d logits[b, v] = g[b] * (1[v == id_b] - softmax(logits[b])[v])   (gradient of the gathered log-softmax;
// rows with id < 0 — steps after the loop was left — get zeros)
__global__ void __launch_bounds__(256) sample_logp_bwd_k(const float* logits, long long ld, const float* lse,
                                                         const long long* ids, const float* g, float* dlogits,
                                                         long long ldd, int V) {
    const int b = blockIdx.x;
    const long long id = ids[b];
    const float gb = g[b], l = lse[b];
    for (int v = threadIdx.x; v < V; v += 256) {
        float d = 0.f;
        if (id >= 0) d = gb * ((v == id ? 1.f : 0.f) - expf(logits[(long long)b * ld + v] - l));
        dlogits[(long long)b * ldd + v] = d;
    }
}

int sample_logp_bwd(const float* logits, long long ld, const float* lse, const long long* ids, const float* g,
                    float* dlogits, long long ldd, int B, int V, hipStream_t s) {
    if (B <= 0) return SET_OK;
    hipLaunchKernelGGL(sample_logp_bwd_k, dim3(B), dim3(256), 0, s, logits, ld, lse, ids, g, dlogits, ldd, V);
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
