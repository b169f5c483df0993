// Pointwise helpers of the whole-sequence training node (show-edit-tell_amd/xe_sequence.py): the reference's three
// nn.Dropout sites (editnet.py:299-302 embedding, :441 region embedding, :546 before fc) and the row-block packing that
// builds the LSTM input rows [emb | final_hidden | h2 | mean] / [h1 | attend_cap | attend_img] (editnet.py:523,541)
// straight inside the per-sequence operand logs the time-batched weight-gradient contractions read.
#include "set_common.h"
#include "philox.h"

namespace set {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// y = x * keep / (1 - p), keep ~ Bernoulli(1 - p): one Philox call -> four 24-bit uniforms -> four adjacent columns.
// counter = (row, column group, offset_lo, offset_hi), key = seed: independent of the launch geometry.
__global__ void __launch_bounds__(256) dropout_k(const float* x, long long ldx, float* y, long long ldy, int rows, int cols4,
                                                 float p, float scale, unsigned long long seed, unsigned long long offset) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)rows * cols4) return;
    const int r = (int)(i / cols4), c = (int)(i - (long long)r * cols4);
    uint32_t k[4] = {(uint32_t)r, (uint32_t)c, (uint32_t)offset, (uint32_t)(offset >> 32)};
    philox4x32_10(k, (uint32_t)seed, (uint32_t)(seed >> 32));
    const f32x4 v = *reinterpret_cast<const f32x4*>(x + r * ldx + 4 * c);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = ((float)(k[e] >> 8) * (1.0f / 16777216.0f) >= p) ? v[e] * scale : 0.f;
    *reinterpret_cast<f32x4*>(y + r * ldy + 4 * c) = o;
}

// dx (+)= dy * (y != 0 ? scale : 0): y is the dropout OUTPUT.  y == 0 means "dropped" or "the input was exactly 0"; for
// the two sites that follow a ReLU (embedding, region embedding) the second case carries no gradient anyway, so the
// same kernel also applies the ReLU's derivative there (scale = 1, eval mode: plain ReLU backward).
__global__ void __launch_bounds__(256) dropout_bwd_k(const float* dy, long long lddy, const float* y, long long ldy, float* dx,
                                                     long long ldx, int rows, int cols4, float scale, int accumulate) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)rows * cols4) return;
    const int r = (int)(i / cols4), c = (int)(i - (long long)r * cols4);
    const f32x4 g = *reinterpret_cast<const f32x4*>(dy + r * lddy + 4 * c);
    const f32x4 yy = *reinterpret_cast<const f32x4*>(y + r * ldy + 4 * c);
    float* o = dx + r * ldx + 4 * c;
    f32x4 v = accumulate ? *reinterpret_cast<const f32x4*>(o) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] += yy[e] != 0.f ? g[e] * scale : 0.f;
    *reinterpret_cast<f32x4*>(o) = v;
}

// dx (+)= dy * keep / (1 - p) with the keep mask REGENERATED from (seed, offset) — the backward of dropout_k for a site
// that does not follow a ReLU (h2 before fc, editnet.py:545): an input that is exactly 0 and kept still passes its
// gradient, which the zero pattern of the output cannot tell from "dropped".
__global__ void __launch_bounds__(256) dropout_bwd_philox_k(const float* dy, long long lddy, float* dx, long long ldx, int rows,
                                                            int cols4, float p, float scale, unsigned long long seed,
                                                            unsigned long long offset, int accumulate) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)rows * cols4) return;
    const int r = (int)(i / cols4), c = (int)(i - (long long)r * cols4);
    uint32_t k[4] = {(uint32_t)r, (uint32_t)c, (uint32_t)offset, (uint32_t)(offset >> 32)};
    philox4x32_10(k, (uint32_t)seed, (uint32_t)(seed >> 32));
    const f32x4 g = *reinterpret_cast<const f32x4*>(dy + r * lddy + 4 * c);
    float* o = dx + r * ldx + 4 * c;
    f32x4 v = accumulate ? *reinterpret_cast<const f32x4*>(o) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] += ((float)(k[e] >> 8) * (1.0f / 16777216.0f) >= p) ? g[e] * scale : 0.f;
    *reinterpret_cast<f32x4*>(o) = v;
}

// The same dropout of ONE operand for `steps` timesteps in one launch: y[t] = dropout(x) with the counters of timestep t
// (offset + t: rng.offset(site, t) keeps t in the low bits), bit for bit what `steps` launches of dropout_k write.  The
// region embedding of the train-mode forward (editnet.py:441, fresh mask per timestep) is the caller: 19 launches of
// 10 us become one that is bound by its 358 MB of stores.
__global__ void __launch_bounds__(256) dropout_steps_k(const float* x, long long ldx, float* y, long long ldy, long long y_step,
                                                       int rows, int cols4, int steps, float p, float scale,
                                                       unsigned long long seed, unsigned long long offset) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)rows * cols4) return;
    const int r = (int)(i / cols4), c = (int)(i - (long long)r * cols4);
    const f32x4 v = *reinterpret_cast<const f32x4*>(x + r * ldx + 4 * c);
    for (int t = 0; t < steps; ++t) {
        const unsigned long long off = offset + (unsigned long long)t;
        uint32_t k[4] = {(uint32_t)r, (uint32_t)c, (uint32_t)off, (uint32_t)(off >> 32)};
        philox4x32_10(k, (uint32_t)seed, (uint32_t)(seed >> 32));
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = ((float)(k[e] >> 8) * (1.0f / 16777216.0f) >= p) ? v[e] * scale : 0.f;
        *reinterpret_cast<f32x4*>(y + (long long)t * y_step + r * ldy + 4 * c) = o;
    }
}

// dx (+)= sum_t dy[t] * (y[t] != 0 ? scale : 0), t ascending: what `steps` accumulating launches of dropout_bwd_k leave in
// dx, bit for bit, with dx read and written once
__global__ void __launch_bounds__(256) dropout_bwd_steps_k(const float* dy, long long lddy, long long dy_step, const float* y,
                                                           long long ldy, long long y_step, float* dx, long long ldx, int rows,
                                                           int cols4, int steps, float scale, int accumulate) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)rows * cols4) return;
    const int r = (int)(i / cols4), c = (int)(i - (long long)r * cols4);
    float* o = dx + r * ldx + 4 * c;
    f32x4 v = accumulate ? *reinterpret_cast<const f32x4*>(o) : (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int t = 0; t < steps; ++t) {
        const f32x4 g = *reinterpret_cast<const f32x4*>(dy + (long long)t * dy_step + r * lddy + 4 * c);
        const f32x4 yy = *reinterpret_cast<const f32x4*>(y + (long long)t * y_step + r * ldy + 4 * c);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += yy[e] != 0.f ? g[e] * scale : 0.f;
    }
    *reinterpret_cast<f32x4*>(o) = v;
}

// EmbeddingC.forward in train mode (editnet.py:299-302): relu(table[ids]) followed by the dropout of dropout_k (same
// counters: the result equals set_embed_relu_f32 + set_dropout_f32 in place), one launch
__global__ void __launch_bounds__(256) embed_relu_dropout_k(const float* table, const int64_t* ids, long long ids_stride,
                                                            float* out, long long ldo, int n, int D4, int V, float p,
                                                            float scale, unsigned long long seed, unsigned long long offset) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)n * D4) return;
    const int r = (int)(i / D4), c = (int)(i - (long long)r * D4);
    long long id = ids[r * ids_stride];
    id = id < 0 ? 0 : (id >= V ? V - 1 : id);
    const f32x4 v = *reinterpret_cast<const f32x4*>(table + id * (4LL * D4) + 4 * c);
    uint32_t k[4] = {(uint32_t)r, (uint32_t)c, (uint32_t)offset, (uint32_t)(offset >> 32)};
    philox4x32_10(k, (uint32_t)seed, (uint32_t)(seed >> 32));
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float x = v[e] > 0.f ? v[e] : 0.f;
        o[e] = ((float)(k[e] >> 8) * (1.0f / 16777216.0f) >= p) ? x * scale : 0.f;
    }
    *reinterpret_cast<f32x4*>(out + r * ldo + 4 * c) = o;
}

struct PackArgs {
    const float* src[4];
    long long ld[4];
    int c4_end[4];       // cumulative column groups (of 4 floats)
    int nseg;
};
__global__ void __launch_bounds__(256) pack_k(float* dst, long long ldd, int rows, int cols4, PackArgs a, int accumulate) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)rows * cols4) return;
    const int r = (int)(i / cols4), c = (int)(i - (long long)r * cols4);
    int s = 0, c0 = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k)
        if (k + 1 < a.nseg && c >= a.c4_end[k]) { s = k + 1; c0 = a.c4_end[k]; }
    const f32x4 v = *reinterpret_cast<const f32x4*>(a.src[s] + r * a.ld[s] + 4 * (c - c0));
    float* o = dst + r * ldd + 4 * c;
    *reinterpret_cast<f32x4*>(o) = accumulate ? *reinterpret_cast<const f32x4*>(o) + v : v;
}

// column sums of a (rows, cols) matrix (bias gradients over all (t, b) rows of a sequence): pass 1 reduces row chunks
// into partial rows, pass 2 adds the partials in chunk order (deterministic, no atomics).
constexpr int COLSUM_CHUNKS = 64;
__global__ void __launch_bounds__(256) colsum_partial_k(const float* x, long long ld, int rows, int cols4, float* part) {
    __shared__ f32x4 s_red[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const int chunk = blockIdx.y, per = (rows + COLSUM_CHUNKS - 1) / COLSUM_CHUNKS;
    const int r0 = chunk * per, r1 = r0 + per < rows ? r0 + per : rows;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (c < cols4)
        for (int r = r0 + wave; r < r1; r += 4) acc += *reinterpret_cast<const f32x4*>(x + r * ld + 4 * c);
    s_red[wave][lane] = acc;
    __syncthreads();
    if (wave == 0 && c < cols4)
        *reinterpret_cast<f32x4*>(part + ((long long)chunk * cols4 + c) * 4) = (s_red[0][lane] + s_red[1][lane]) + (s_red[2][lane] + s_red[3][lane]);
}
__global__ void __launch_bounds__(256) colsum_final_k(const float* part, int cols4, float* out, int accumulate) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= cols4) return;
    f32x4 acc = accumulate ? *reinterpret_cast<const f32x4*>(out + 4 * c) : (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < COLSUM_CHUNKS; ++k) acc += *reinterpret_cast<const f32x4*>(part + ((long long)k * cols4 + c) * 4);
    *reinterpret_cast<f32x4*>(out + 4 * c) = acc;
}


// ---- all-timestep forms for the teacher-forced training loop (train_loop.hip): the words of every timestep are known before
// the loop and the dropped-out h2 is only read after it, so the embedding and the output dropout of T timesteps are ONE
// launch each (grid.y = t; timestep t covers its first bt[t] rows, with the counters of the per-step launches: same bits),
// and the two operand logs a timestep packs are one launch
struct StepRows { int bt[SET_STEPS_MAX]; };
__global__ void __launch_bounds__(256) embed_relu_dropout_steps_k(const float* table, const int64_t* ids, long long ids_step,
                                                                  long long ids_stride, float* out, long long out_step,
                                                                  long long ldo, const StepRows R, int D4, int V, float p, float scale,
                                                                  unsigned long long seed, unsigned long long offset) {
    const int t = blockIdx.y;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)R.bt[t] * D4) return;
    const int r = (int)(i / D4), c = (int)(i - (long long)r * D4);
    long long id = ids[t * ids_step + r * ids_stride];
    id = id < 0 ? 0 : (id >= V ? V - 1 : id);
    const f32x4 v = *reinterpret_cast<const f32x4*>(table + id * (4LL * D4) + 4 * c);
    f32x4 o;
    if (p > 0.f) {
        const unsigned long long off = offset + (unsigned long long)t;
        uint32_t k[4] = {(uint32_t)r, (uint32_t)c, (uint32_t)off, (uint32_t)(off >> 32)};
        philox4x32_10(k, (uint32_t)seed, (uint32_t)(seed >> 32));
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float x = v[e] > 0.f ? v[e] : 0.f;
            o[e] = ((float)(k[e] >> 8) * (1.0f / 16777216.0f) >= p) ? x * scale : 0.f;
        }
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = v[e] > 0.f ? v[e] : 0.f;
    }
    *reinterpret_cast<f32x4*>(out + t * out_step + r * ldo + 4 * c) = o;
}
__global__ void __launch_bounds__(256) dropout_xsteps_k(const float* x, long long x_step, long long ldx, float* y, long long y_step,
                                                        long long ldy, const StepRows R, int cols4, float p, float scale,
                                                        unsigned long long seed, unsigned long long offset) {
    const int t = blockIdx.y;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)R.bt[t] * cols4) return;
    const int r = (int)(i / cols4), c = (int)(i - (long long)r * cols4);
    const unsigned long long off = offset + (unsigned long long)t;
    uint32_t k[4] = {(uint32_t)r, (uint32_t)c, (uint32_t)off, (uint32_t)(off >> 32)};
    philox4x32_10(k, (uint32_t)seed, (uint32_t)(seed >> 32));
    const f32x4 v = *reinterpret_cast<const f32x4*>(x + t * x_step + r * ldx + 4 * c);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = ((float)(k[e] >> 8) * (1.0f / 16777216.0f) >= p) ? v[e] * scale : 0.f;
    *reinterpret_cast<f32x4*>(y + t * y_step + r * ldy + 4 * c) = o;
}
__device__ __forceinline__ void pack_body(const long long i, float* dst, long long ldd, int rows, int cols4, const PackArgs& a) {
    if (i >= (long long)rows * cols4) return;
    const int r = (int)(i / cols4), c = (int)(i - (long long)r * cols4);
    int s = 0, c0 = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k)
        if (k + 1 < a.nseg && c >= a.c4_end[k]) { s = k + 1; c0 = a.c4_end[k]; }
    *reinterpret_cast<f32x4*>(dst + r * ldd + 4 * c) = *reinterpret_cast<const f32x4*>(a.src[s] + r * a.ld[s] + 4 * (c - c0));
}
__global__ void __launch_bounds__(256) pack2_k(float* dst0, long long ldd0, int cols4_0, PackArgs a0, int nblk0, float* dst1,
                                               long long ldd1, int cols4_1, PackArgs a1, int rows) {
    if ((int)blockIdx.x < nblk0) pack_body((long long)blockIdx.x * 256 + threadIdx.x, dst0, ldd0, rows, cols4_0, a0);
    else pack_body((long long)(blockIdx.x - nblk0) * 256 + threadIdx.x, dst1, ldd1, rows, cols4_1, a1);
}

// Grouped form: the ~16 bias gradients of one training step (every one a column sum over the same (t, b) rows) as TWO
// launches instead of 32.  Problem j owns the blocks [blk0[j], blk0[j+1]) of pass 1 (column tiles x row chunks; the number of
// chunks grows with the rows, so the (T B R, A) projection gradient fills the chip instead of 128 workgroups) and the
// float4 columns [c0[j], c0[j+1]) of pass 2; up to two destinations per problem (two biases fed by the same gradient).
struct ColsumGroup {
    const float* x[SET_COLSUM_MAX]; long long ld[SET_COLSUM_MAX]; float* out[SET_COLSUM_MAX]; float* out2[SET_COLSUM_MAX];
    int rows[SET_COLSUM_MAX], cols4[SET_COLSUM_MAX], chunks[SET_COLSUM_MAX];
    int blk0[SET_COLSUM_MAX + 1], c0[SET_COLSUM_MAX + 1];
    long long p0[SET_COLSUM_MAX];           // float4 offset of the problem's partials
    unsigned acc, acc2;
    int n;
};
__global__ void __launch_bounds__(256) colsum_group_partial_k(const ColsumGroup G, float* part) {
    __shared__ f32x4 s_red[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int j = 0;
    for (int k = 1; k < G.n; ++k)
        if ((int)blockIdx.x >= G.blk0[k]) j = k;
    const int local = blockIdx.x - G.blk0[j], cols4 = G.cols4[j], rows = G.rows[j], tiles = (cols4 + 63) >> 6;
    const int chunk = local / tiles, c = (local - chunk * tiles) * 64 + lane;
    const int per = (rows + G.chunks[j] - 1) / G.chunks[j];
    const int r0 = chunk * per, r1 = r0 + per < rows ? r0 + per : rows;
    const float* x = G.x[j];
    const long long ld = G.ld[j];
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (c < cols4) {
        int r = r0 + wave;
        for (; r + 12 < r1; r += 16) {          // four rows of this wave in flight, added in row order
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(x + r * ld + 4 * c);
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(x + (r + 4) * ld + 4 * c);
            const f32x4 v2 = *reinterpret_cast<const f32x4*>(x + (r + 8) * ld + 4 * c);
            const f32x4 v3 = *reinterpret_cast<const f32x4*>(x + (r + 12) * ld + 4 * c);
            acc += v0; acc += v1; acc += v2; acc += v3;
        }
        for (; r < r1; r += 4) acc += *reinterpret_cast<const f32x4*>(x + r * ld + 4 * c);
    }
    s_red[wave][lane] = acc;
    __syncthreads();
    if (wave == 0 && c < cols4)
        *reinterpret_cast<f32x4*>(part + (G.p0[j] + (long long)chunk * cols4 + c) * 4) =
            (s_red[0][lane] + s_red[1][lane]) + (s_red[2][lane] + s_red[3][lane]);
}
// pass 2: a workgroup = 64 float4 columns x 4 slices of the chunks (slice s adds chunks s, s + 4, ... in order; the four
// slice sums are added in slice order) — the partials of a column are 64-256 rows, one thread walking them alone is latency
__global__ void __launch_bounds__(256) colsum_group_final_k(const ColsumGroup G, const float* part) {
    __shared__ f32x4 s_red[4][64];
    const int lane = threadIdx.x & 63, slice = threadIdx.x >> 6;
    const int g = blockIdx.x * 64 + lane;
    const bool on = g < G.c0[G.n];
    int j = 0;
    for (int k = 1; k < G.n; ++k)
        if (g >= G.c0[k]) j = k;
    const int c = g - G.c0[j], cols4 = G.cols4[j], nk = G.chunks[j];
    f32x4 sum = {0.f, 0.f, 0.f, 0.f};
    if (on) {
        const float* p = part + (G.p0[j] + c) * 4;
        int k = slice;
        for (; k + 12 < nk; k += 16) {
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(p + (long long)k * cols4 * 4);
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(p + (long long)(k + 4) * cols4 * 4);
            const f32x4 v2 = *reinterpret_cast<const f32x4*>(p + (long long)(k + 8) * cols4 * 4);
            const f32x4 v3 = *reinterpret_cast<const f32x4*>(p + (long long)(k + 12) * cols4 * 4);
            sum += v0; sum += v1; sum += v2; sum += v3;
        }
        for (; k < nk; k += 4) sum += *reinterpret_cast<const f32x4*>(p + (long long)k * cols4 * 4);
    }
    s_red[slice][lane] = sum;
    __syncthreads();
    if (slice != 0 || !on) return;
    sum = (s_red[0][lane] + s_red[1][lane]) + (s_red[2][lane] + s_red[3][lane]);
    float* o = G.out[j] + 4 * c;
    *reinterpret_cast<f32x4*>(o) = ((G.acc >> j) & 1u) ? *reinterpret_cast<const f32x4*>(o) + sum : sum;
    if (G.out2[j]) {
        float* o2 = G.out2[j] + 4 * c;
        *reinterpret_cast<f32x4*>(o2) = ((G.acc2 >> j) & 1u) ? *reinterpret_cast<const f32x4*>(o2) + sum : sum;
    }
}
static int pack_args(PackArgs* a, int nseg, const float* const* src, const int64_t* ld, const int* cols, int* c4_out) {
    int c4 = 0;
    for (int i = 0; i < 4; ++i) {
        if (i < nseg) {
            if (!src[i] || cols[i] <= 0) return SET_ERR_ARG;
            if ((cols[i] & 3) || (ld[i] & 3) || !aligned16(src[i])) return SET_ERR_UNSUPPORTED;
            c4 += cols[i] >> 2;
            a->src[i] = src[i]; a->ld[i] = ld[i];
        } else { a->src[i] = nullptr; a->ld[i] = 0; }
        a->c4_end[i] = c4;
    }
    a->nseg = nseg;
    *c4_out = c4;
    return SET_OK;
}

int embed_relu_dropout_steps(const float* table, const int64_t* ids, long long ids_step, long long ids_stride, float* out,
                             long long out_step, long long ldo, const int* bts, int T, int B, int D, int V, float p, uint64_t seed,
                             uint64_t offset, hipStream_t st) {
    if (!table || !ids || !out || !bts || T <= 0 || T > SET_STEPS_MAX || B <= 0 || D <= 0 || V <= 0 || !(p >= 0.f) || !(p < 1.f))
        return SET_ERR_ARG;
    if ((D & 3) || (ldo & 3) || (out_step & 3) || !aligned16(table) || !aligned16(out)) return SET_ERR_UNSUPPORTED;
    StepRows R{};
    for (int t = 0; t < T; ++t) R.bt[t] = bts[t] < 0 ? 0 : (bts[t] > B ? B : bts[t]);
    const long long n = (long long)B * (D >> 2);
    hipLaunchKernelGGL(embed_relu_dropout_steps_k, dim3((unsigned)((n + 255) / 256), T), dim3(256), 0, st, table, ids, ids_step,
                       ids_stride, out, out_step, ldo, R, D >> 2, V, p, 1.0f / (1.0f - p), (unsigned long long)seed,
                       (unsigned long long)offset);
    SET_LAUNCH_CHECK();
    return SET_OK;
}

int dropout_xsteps(const float* x, long long x_step, long long ldx, float* y, long long y_step, long long ldy, const int* bts, int T,
                   int B, int cols, float p, uint64_t seed, uint64_t offset, hipStream_t st) {
    if (!x || !y || !bts || T <= 0 || T > SET_STEPS_MAX || B <= 0 || cols <= 0 || !(p >= 0.f) || !(p < 1.f)) return SET_ERR_ARG;
    if ((cols & 3) || (ldx & 3) || (ldy & 3) || (x_step & 3) || (y_step & 3) || !aligned16(x) || !aligned16(y)) return SET_ERR_UNSUPPORTED;
    StepRows R{};
    for (int t = 0; t < T; ++t) R.bt[t] = bts[t] < 0 ? 0 : (bts[t] > B ? B : bts[t]);
    const long long n = (long long)B * (cols >> 2);
    hipLaunchKernelGGL(dropout_xsteps_k, dim3((unsigned)((n + 255) / 256), T), dim3(256), 0, st, x, x_step, ldx, y, y_step, ldy, R,
                       cols >> 2, p, 1.0f / (1.0f - p), (unsigned long long)seed, (unsigned long long)offset);
    SET_LAUNCH_CHECK();
    return SET_OK;
}

int pack2(float* dst0, long long ldd0, int nseg0, const float* const* src0, const int64_t* ld0, const int* cols0, float* dst1,
          long long ldd1, int nseg1, const float* const* src1, const int64_t* ld1, const int* cols1, int rows, hipStream_t st) {
    if (!dst0 || !dst1 || rows <= 0 || nseg0 <= 0 || nseg0 > 4 || nseg1 <= 0 || nseg1 > 4) return SET_ERR_ARG;
    if ((ldd0 & 3) || (ldd1 & 3) || !aligned16(dst0) || !aligned16(dst1)) return SET_ERR_UNSUPPORTED;
    PackArgs a0, a1;
    int c0 = 0, c1 = 0;
    SET_TRY(pack_args(&a0, nseg0, src0, ld0, cols0, &c0));
    SET_TRY(pack_args(&a1, nseg1, src1, ld1, cols1, &c1));
    const int n0 = (int)(((long long)rows * c0 + 255) / 256), n1 = (int)(((long long)rows * c1 + 255) / 256);
    hipLaunchKernelGGL(pack2_k, dim3(n0 + n1), dim3(256), 0, st, dst0, ldd0, c0, a0, n0, dst1, ldd1, c1, a1, rows);
    SET_LAUNCH_CHECK();
    return SET_OK;
}

static inline int colsum_group_chunks(int rows) {
    const int c = (rows + 255) / 256;
    return c < COLSUM_CHUNKS ? COLSUM_CHUNKS : c > 256 ? 256 : c;
}

}  // namespace set

using namespace set;

extern "C" {

int set_dropout_f32(const float* x, int64_t ldx, float* y, int64_t ldy, int rows, int cols, float p, uint64_t seed,
                    uint64_t offset, void* stream) {
    if (!x || !y || rows <= 0 || cols <= 0 || !(p >= 0.f) || !(p < 1.f)) return SET_ERR_ARG;
    if ((cols & 3) || (ldx & 3) || (ldy & 3) || !aligned16(x) || !aligned16(y)) return SET_ERR_UNSUPPORTED;
    const long long n = (long long)rows * (cols >> 2);
    hipLaunchKernelGGL(dropout_k, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, (long long)ldx, y,
                       (long long)ldy, rows, cols >> 2, p, 1.0f / (1.0f - p), (unsigned long long)seed,
                       (unsigned long long)offset);
    SET_LAUNCH_CHECK();
    return SET_OK;
}

int set_dropout_steps_f32(const float* x, int64_t ldx, float* y, int64_t ldy, int64_t y_step, int rows, int cols, int steps,
                          float p, uint64_t seed, uint64_t offset, void* stream) {
    if (!x || !y || rows <= 0 || cols <= 0 || steps <= 0 || !(p >= 0.f) || !(p < 1.f)) return SET_ERR_ARG;
    if ((cols & 3) || (ldx & 3) || (ldy & 3) || (y_step & 3) || !aligned16(x) || !aligned16(y)) return SET_ERR_UNSUPPORTED;
    const long long n = (long long)rows * (cols >> 2);
    hipLaunchKernelGGL(dropout_steps_k, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, (long long)ldx, y,
                       (long long)ldy, (long long)y_step, rows, cols >> 2, steps, p, 1.0f / (1.0f - p), (unsigned long long)seed,
                       (unsigned long long)offset);
    SET_LAUNCH_CHECK();
    return SET_OK;
}

int set_dropout_bwd_steps_f32(const float* dy, int64_t lddy, int64_t dy_step, const float* y, int64_t ldy, int64_t y_step,
                              float* dx, int64_t ldx, int rows, int cols, int steps, float scale, int accumulate,
                              void* stream) {
    if (!dy || !y || !dx || rows <= 0 || cols <= 0 || steps <= 0) return SET_ERR_ARG;
    if ((cols & 3) || (lddy & 3) || (ldy & 3) || (ldx & 3) || (dy_step & 3) || (y_step & 3) || !aligned16(dy) || !aligned16(y) ||
        !aligned16(dx))
        return SET_ERR_UNSUPPORTED;
    const long long n = (long long)rows * (cols >> 2);
    hipLaunchKernelGGL(dropout_bwd_steps_k, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dy,
                       (long long)lddy, (long long)dy_step, y, (long long)ldy, (long long)y_step, dx, (long long)ldx, rows,
                       cols >> 2, steps, scale, accumulate);
    SET_LAUNCH_CHECK();
    return SET_OK;
}

int set_embed_relu_dropout_f32(const float* table, const int64_t* ids, int64_t ids_stride, float* out, int64_t ldo, int n,
                               int D, int V, float p, uint64_t seed, uint64_t offset, void* stream) {
    if (!table || !ids || !out || n < 0 || D <= 0 || V <= 0 || !(p >= 0.f) || !(p < 1.f)) return SET_ERR_ARG;
    if ((D & 3) || (ldo & 3) || !aligned16(table) || !aligned16(out)) return SET_ERR_UNSUPPORTED;
    if (n == 0) return SET_OK;
    const long long t = (long long)n * (D >> 2);
    hipLaunchKernelGGL(embed_relu_dropout_k, dim3((unsigned)((t + 255) / 256)), dim3(256), 0, (hipStream_t)stream, table, ids,
                       (long long)ids_stride, out, (long long)ldo, n, D >> 2, V, p, 1.0f / (1.0f - p), (unsigned long long)seed,
                       (unsigned long long)offset);
    SET_LAUNCH_CHECK();
    return SET_OK;
}

int set_dropout_bwd_f32(const float* dy, int64_t lddy, const float* y, int64_t ldy, float* dx, int64_t ldx, int rows,
                        int cols, float scale, int accumulate, void* stream) {
    if (!dy || !y || !dx || rows <= 0 || cols <= 0) return SET_ERR_ARG;
    if ((cols & 3) || (lddy & 3) || (ldy & 3) || (ldx & 3) || !aligned16(dy) || !aligned16(y) || !aligned16(dx))
        return SET_ERR_UNSUPPORTED;
    const long long n = (long long)rows * (cols >> 2);
    hipLaunchKernelGGL(dropout_bwd_k, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dy,
                       (long long)lddy, y, (long long)ldy, dx, (long long)ldx, rows, cols >> 2, scale, accumulate);
    SET_LAUNCH_CHECK();
    return SET_OK;
}

int set_dropout_bwd_philox_f32(const float* dy, int64_t lddy, float* dx, int64_t ldx, int rows, int cols, float p,
                               uint64_t seed, uint64_t offset, int accumulate, void* stream) {
    if (!dy || !dx || rows <= 0 || cols <= 0 || !(p >= 0.f) || !(p < 1.f)) return SET_ERR_ARG;
    if ((cols & 3) || (lddy & 3) || (ldx & 3) || !aligned16(dy) || !aligned16(dx)) return SET_ERR_UNSUPPORTED;
    const long long n = (long long)rows * (cols >> 2);
    hipLaunchKernelGGL(dropout_bwd_philox_k, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dy,
                       (long long)lddy, dx, (long long)ldx, rows, cols >> 2, p, 1.0f / (1.0f - p), (unsigned long long)seed,
                       (unsigned long long)offset, accumulate);
    SET_LAUNCH_CHECK();
    return SET_OK;
}

/* mask[r] = (sum of row r of x != 0): the adaptive model's data-derived region mask (editnet_adaptive.py:449-453) */
int set_rowsum_mask_f32(const float* x, int64_t ld, int rows, int cols, float* mask, void* stream) {
    if (!x || !mask || rows <= 0 || cols <= 0) return SET_ERR_ARG;
    if ((ld & 3) || !aligned16(x)) return SET_ERR_UNSUPPORTED;
    return rowsum_mask(x, ld, rows, cols, mask, (hipStream_t)stream);
}

int set_pack_f32(float* dst, int64_t ldd, int rows, int nseg, const float* const* src, const int64_t* ld, const int* cols,
                 int accumulate, void* stream) {
    if (!dst || !src || !ld || !cols || rows <= 0 || nseg <= 0 || nseg > 4) return SET_ERR_ARG;
    PackArgs a;
    int c4 = 0;
    for (int i = 0; i < 4; ++i) {
        if (i < nseg) {
            if (!src[i] || cols[i] <= 0) return SET_ERR_ARG;
            if ((cols[i] & 3) || (ld[i] & 3) || !aligned16(src[i])) return SET_ERR_UNSUPPORTED;
            c4 += cols[i] >> 2;
            a.src[i] = src[i]; a.ld[i] = ld[i];
        } else { a.src[i] = nullptr; a.ld[i] = 0; }
        a.c4_end[i] = c4;
    }
    a.nseg = nseg;
    if ((ldd & 3) || !aligned16(dst)) return SET_ERR_UNSUPPORTED;
    const long long n = (long long)rows * c4;
    hipLaunchKernelGGL(pack_k, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dst, (long long)ldd, rows,
                       c4, a, accumulate);
    SET_LAUNCH_CHECK();
    return SET_OK;
}

size_t set_colsum_workspace_bytes(int cols) { return cols > 0 ? (size_t)COLSUM_CHUNKS * round_up((size_t)cols, 4) * sizeof(float) + 256 : 0; }

int set_colsum_f32(const float* x, int64_t ld, int rows, int cols, float* out, int accumulate, void* ws, size_t ws_bytes,
                   void* stream) {
    if (!x || !out || rows <= 0 || cols <= 0) return SET_ERR_ARG;
    if ((cols & 3) || (ld & 3) || !aligned16(x) || !aligned16(out)) return SET_ERR_UNSUPPORTED;
    if (!ws || !aligned16(ws) || ws_bytes < set_colsum_workspace_bytes(cols) - 256) return SET_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int cols4 = cols >> 2;
    hipLaunchKernelGGL(colsum_partial_k, dim3(cdiv(cols4, 64), COLSUM_CHUNKS), dim3(256), 0, st, x, (long long)ld, rows, cols4,
                       (float*)ws);
    hipLaunchKernelGGL(colsum_final_k, dim3(cdiv(cols4, 256)), dim3(256), 0, st, (const float*)ws, cols4, out, accumulate);
    SET_LAUNCH_CHECK();
    return SET_OK;
}

size_t set_colsum_group_workspace_bytes(const SetColsumDesc* d, int n) {
    if (!d || n <= 0) return 0;
    size_t b = 256;
    for (int j = 0; j < n; ++j)
        if (d[j].rows > 0 && d[j].cols > 0) b += (size_t)colsum_group_chunks(d[j].rows) * round_up((size_t)d[j].cols, 4) * sizeof(float);
    return b;
}

int set_colsum_group_f32(const SetColsumDesc* d, int n, void* ws, size_t ws_bytes, void* stream) {
    if (!d || n <= 0 || n > SET_COLSUM_MAX) return SET_ERR_ARG;
    if (!ws || !aligned16(ws) || ws_bytes < set_colsum_group_workspace_bytes(d, n) - 256) return SET_ERR_WORKSPACE;
    ColsumGroup G{};
    G.n = n;
    long long p = 0;
    int blk = 0, c0 = 0;
    for (int j = 0; j < n; ++j) {
        const SetColsumDesc& e = d[j];
        if (!e.x || !e.out || e.rows <= 0 || e.cols <= 0) return SET_ERR_ARG;
        if ((e.cols & 3) || (e.ld & 3) || e.ld < e.cols || !aligned16(e.x) || !aligned16(e.out) || (e.out2 && !aligned16(e.out2)))
            return SET_ERR_UNSUPPORTED;
        G.x[j] = e.x; G.ld[j] = e.ld; G.out[j] = e.out; G.out2[j] = e.out2; G.rows[j] = e.rows; G.cols4[j] = e.cols >> 2;
        G.chunks[j] = colsum_group_chunks(e.rows);
        if (e.accumulate) G.acc |= 1u << j;
        if (e.accumulate2) G.acc2 |= 1u << j;
        G.blk0[j] = blk; G.c0[j] = c0; G.p0[j] = p;
        blk += cdiv(G.cols4[j], 64) * G.chunks[j];
        c0 += G.cols4[j];
        p += (long long)G.chunks[j] * G.cols4[j];
    }
    G.blk0[n] = blk; G.c0[n] = c0;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(colsum_group_partial_k, dim3(blk), dim3(256), 0, st, G, (float*)ws);
    hipLaunchKernelGGL(colsum_group_final_k, dim3(cdiv(c0, 64)), dim3(256), 0, st, G, (const float*)ws);
    SET_LAUNCH_CHECK();
    return SET_OK;
}

}  // extern "C"
