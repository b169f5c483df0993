// Internal helpers shared by the HIP translation units of libset_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/set_hip.h"

// experiment (EXPERIMENTS 5.7): -DSET_EXP_VGPR_CAP=w asks for w waves per SIMD (w = 4: at most 128 registers) of the short decode kernels (co-residency with the
// GEMM's waves when several batches are in flight)
#ifdef SET_EXP_VGPR_CAP
#define SET_VGPR_CAP __attribute__((amdgpu_waves_per_eu(SET_EXP_VGPR_CAP, 8)))
#else
#define SET_VGPR_CAP
#endif
#ifdef SET_EXP_VGPR_CAP_ATT
#define SET_VGPR_CAP_ATT __attribute__((amdgpu_waves_per_eu(SET_EXP_VGPR_CAP_ATT, 8)))
#else
#define SET_VGPR_CAP_ATT SET_VGPR_CAP
#endif

namespace set {

// ---------------------------------------------------------------------------------------------
// error plumbing: HIP failures are recorded per thread and surfaced as SET_ERR_HIP
// ---------------------------------------------------------------------------------------------
extern thread_local int g_last_hip_error;
inline int hip_fail(hipError_t e) { g_last_hip_error = (int)e; return SET_ERR_HIP; }

#define SET_HIP_TRY(expr)                                   \
    do {                                                    \
        hipError_t _e = (expr);                             \
        if (_e != hipSuccess) return ::set::hip_fail(_e);   \
    } while (0)
#define SET_TRY(expr)                  \
    do {                               \
        int _rc = (expr);              \
        if (_rc != SET_OK) return _rc; \
    } while (0)
#define SET_LAUNCH_CHECK() SET_HIP_TRY(hipGetLastError())

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
inline size_t round_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// bump allocator over the caller-supplied workspace (256-byte granules)
struct Carver {
    char* base;
    size_t off = 0;
    explicit Carver(void* p) : base(static_cast<char*>(p)) {}
    template <typename T>
    T* take(size_t n) {
        T* r = reinterpret_cast<T*>(base + off);
        off += round_up(n * sizeof(T), 256);
        return r;
    }
};

// ---------------------------------------------------------------------------------------------
// grouped skinny GEMM  C[M,N] (+slabs) = A[M,K] * W[N,K]^T   (gemm_f32.hip)
//   fp32 in / fp32 accumulate on v_mfma_f32_32x32x2_f32; K may be the concatenation of up to
//   three (A,W) column segments that live in different tensors (weights are used in place).
// ---------------------------------------------------------------------------------------------
constexpr int GEMM_BK = 32;
constexpr int GEMM_MAX_SEG = 3;
constexpr int GEMM_MAX_TASKS = 6;
constexpr int GEMM_MAX_KSPLIT = 8;

struct GemmSeg {
    const float* A;   // (M, K) rows, leading stride lda
    const float* W;   // (N, K) rows, leading stride ldw
    long long lda, ldw;
    int K;            // multiple of GEMM_BK
};

struct GemmProb {
    GemmSeg seg[GEMM_MAX_SEG];
    int nseg = 0;
    float* C = nullptr;          // slab s at C + s*slab_stride; (M,N) rows with stride ldc
    long long ldc = 0, slab_stride = 0;
    const float* bias = nullptr; // fused only when ksplit == 1
    int M = 0, N = 0;
    int act = SET_ACT_NONE;      // fused only when ksplit == 1
    int ksplit = 1;              // filled by plan_ksplit or by the caller
    int max_ksplit = GEMM_MAX_KSPLIT;
    int bm_hint = 0;             // 0: row tile chosen from M (gemm_tile_m); 64 / 128: forced for the whole launch (probs[0] decides)
    void add(const float* A, long long lda, const float* W, long long ldw, int K) {
        seg[nseg++] = GemmSeg{A, W, lda, ldw, K};
    }
    int ktiles() const { int k = 0; for (int i = 0; i < nseg; ++i) k += seg[i].K / GEMM_BK; return k; }
};

// choose per-problem ksplit so that the whole group launches about `target_wgs` workgroups of
// similar length; problems with max_ksplit == 1 keep a fused bias/activation epilogue
void plan_ksplit(GemmProb* probs, int n, int target_wgs);
int gemm_group(const GemmProb* probs, int n, hipStream_t stream, const char* tag = nullptr);
// gemm_gen.hip: C (+)= a . b^T with either operand stored k-major or k-minor (training backward)
int gemm_gen(const float* A, long long lda, int a_kminor, const float* B, long long ldb, int b_kminor, float* C,
             long long ldc, int M, int N, int K, int accumulate, void* ws, size_t ws_bytes, hipStream_t s);
int gemm_gen_group(const SetGemmDesc* d, int n, int a_kminor, int b_kminor, void* ws, size_t ws_bytes, hipStream_t s,
                   SetSlabSrc* slabs_out = nullptr);   // slabs_out: split problems keep their partials (no reduction launch)
int gemm_tile_m(int M);   // BM the launcher will pick for M rows

// ---------------------------------------------------------------------------------------------
// Row gate of the free-running decode loops (editnet_rl.py:529-547, dcnet_rl.py:326-344).  The host enqueues all
// max_len + 1 timesteps without ever reading the device, so the way the reference stops doing work is decided ON
// the device, by the kernels of a timestep:
//   alive_prev   *alive_prev == 0: every row had finished after the previous timestep — the reference has left its loop
//                (`if unfinished.sum() == 0: break`); the kernel returns at once.  Outputs are unchanged (nothing was
//                written after the break before either); always on in the fused loops, for the kernels that carry a
//                timestep's time (the three grouped GEMM launches, which get the pointer as a preloaded scalar argument,
//                and the attention launch); the short pointwise / pick kernels skip the test (a dependent load at their
//                top costs more than they do after the break).
// The launchers read the current gate from a thread-local that the loop sets around each timestep (RowGateScope).
// ---------------------------------------------------------------------------------------------
struct RowGate {
    const int* alive_prev = nullptr;
    __device__ __forceinline__ bool loop_left() const { return alive_prev && *alive_prev == 0; }
};
extern thread_local RowGate g_row_gate;
struct RowGateScope {
    RowGate prev;
    explicit RowGateScope(const RowGate& g) : prev(g_row_gate) { g_row_gate = g; }
    ~RowGateScope() { g_row_gate = prev; }
};
extern thread_local const int* g_row_limit;   // set_decode_row_limits(): per-row cap on the caption length of the greedy loops (or NULL)

// ---------------------------------------------------------------------------------------------
// Addends of a gradient that are still split-K partials (include/set_hip.h SetSlabSrc): the consumer sums them while it
// loads, so the products that feed it need no reduction launch.  Per entry the partials are added in slab order, the
// entries in list order.
// ---------------------------------------------------------------------------------------------
struct SrcList {
    SetSlabSrc s[SET_MAX_SRC];
    int n = 0;
#if defined(__HIPCC__)
    typedef float v4 __attribute__((ext_vector_type(4)));
    // base value (or 0) + every addend at (row m, columns j .. j + 3).  Four partials of every entry are requested before any
    // is added (a loop of "load, add" with a run-time trip count waits for every load in turn: the consumers took 2 - 8 us
    // longer than the reduction launch they replace); the sums keep slab order within an entry and list order between entries
    __device__ __forceinline__ v4 load4(const float* base, long long base_off, long long m, int j) const {
        v4 v = base ? *reinterpret_cast<const v4*>(base + base_off) : (v4){0.f, 0.f, 0.f, 0.f};
        if (n <= 0) return v;
        const v4 z = {0.f, 0.f, 0.f, 0.f};
        v4 acc[SET_MAX_SRC];
        const float* q[SET_MAX_SRC];
        int cnt[SET_MAX_SRC], maxn = 0;
#pragma unroll
        for (int i = 0; i < SET_MAX_SRC; ++i) {
            const bool on = i < n && s[i].nslab > 0 && m < s[i].rows;
            cnt[i] = on ? s[i].nslab : 0;
            q[i] = on ? s[i].p + m * s[i].ld + j : nullptr;
            acc[i] = z;
            maxn = cnt[i] > maxn ? cnt[i] : maxn;
        }
        for (int k0 = 0; k0 < maxn; k0 += 4) {
            v4 x[SET_MAX_SRC][4];
#pragma unroll
            for (int i = 0; i < SET_MAX_SRC; ++i)
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    x[i][u] = (k0 + u < cnt[i]) ? *reinterpret_cast<const v4*>(q[i] + (long long)(k0 + u) * s[i].slab_stride) : z;
#pragma unroll
            for (int i = 0; i < SET_MAX_SRC; ++i)
#pragma unroll
                for (int u = 0; u < 4; ++u) acc[i] += x[i][u];
        }
#pragma unroll
        for (int i = 0; i < SET_MAX_SRC; ++i) v += acc[i];
        return v;
    }
#endif
};
int make_src_list(const SetSlabSrc* src, int n, SrcList* out);     // backward.hip: validates (alignment, counts) and copies
int copy_lstm_segs(const SetEditNetWeights* w, int nseg, const float* const* xs, const int64_t* lds, const int* Ks, const float* h2,
                   const float* c2, const float* c_memory, float* h_out, float* c_out, float* gates_out, float* cnew_out,
                   float* cg_out, int M, int D, void* ws, size_t ws_bytes, void* stream);      // ops.hip
// all-timestep / merged pointwise launches of the teacher-forced training loop (train_seq.hip; called by train_loop.hip):
// bit for bit what the per-timestep entry points write
constexpr int SET_STEPS_MAX = 64;
int embed_relu_dropout_steps(const float* table, const int64_t* ids, long long ids_step, long long ids_stride, float* out,
                             long long out_step, long long ldo, const int* bts, int T, int B, int D, int V, float p, uint64_t seed,
                             uint64_t offset, hipStream_t st);
int dropout_xsteps(const float* x, long long x_step, long long ldx, float* y, long long y_step, long long ldy, const int* bts, int T,
                   int B, int cols, float p, uint64_t seed, uint64_t offset, hipStream_t st);
int pack2(float* dst0, long long ldd0, int nseg0, const float* const* src0, const int64_t* ld0, const int* cols0, float* dst1,
          long long ldd1, int nseg1, const float* const* src1, const int64_t* ld1, const int* cols1, int rows, hipStream_t st);
// merged launches of the training timestep loop (backward.hip; called by train_loop.hip): LSTM gate backward + SelectC
// backward, visual attention backward + context gating backward — same results as the separate entry points
int lstm_gates_select_bwd_src(const float* dcn_base, const SetSlabSrc* src, const float* do_pre, const float* gates,
                              const float* c_prev, float* dgates, float* dc_prev, const float* dsel_base, const SetSlabSrc* ssrc,
                              const float* Mem, const float* alpha, float* dM, float* dalpha, int M, int T, int D, int acc_dM,
                              hipStream_t st);
int attention_pair_bwd_src(const SetSlabSrc* vsrc, const float* alpha_v, const float* X, const float* att1_v, const float* att2_v,
                           const float* wfull_v, float* datt1_v, float* datt2_v, float* dwf_v, float* de_v, int R, int F, int acc_v,
                           const SetSlabSrc* csrc, float* dctx_out, const float* dalpha_ext, const float* alpha_c, const float* H,
                           const float* att1_c, const float* att2_c, const float* wfull_c, float* datt1_c, float* datt2_c,
                           float* dwf_c, float* de_c, int Tc, int D, int acc_c, int M, int A, long long ld_datt2, hipStream_t st);
int attention_ctxgate_bwd_src(const SetSlabSrc* src, const float* alpha, const float* values, const float* att1, const float* att2,
                              const float* w_full, float* datt1, float* datt2, float* dwfull_part, float* de, int M, int L, int Dv,
                              int A, int acc_datt1, long long ld_datt2, const SetSlabSrc* csrc, const float* zt, const float* s,
                              const float* t, float* dz, float* ds, float* dt, long long ld_out, int D, hipStream_t st);

// ---------------------------------------------------------------------------------------------
// slab views consumed by the pointwise kernels: value(m,n) = sum_s p[s*stride + m*ld + n]
// ---------------------------------------------------------------------------------------------
struct Slabs {
    const float* p;
    long long stride;
    long long ld;
    int n;
};
inline Slabs slabs_of(const GemmProb& g) { return Slabs{g.C, g.slab_stride, g.ldc, g.ksplit}; }
// sum over the slabs of the float4 at element offset `off` of every slab, added in slab order (the deterministic order every
// consumer uses).  Up to four slabs are REQUESTED together whatever the count: a `for (i < n) acc += load` loop with a run-time
// trip count is n dependent round trips (the add of iteration i waits for its load before iteration i + 1 issues), which put
// 3-6 serial L2 latencies at the top of kernels that are otherwise one or two round trips long.
typedef float slab_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ slab_f32x4 slab_sum4_at(const Slabs& s, long long off) {
    slab_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const float* p = s.p + off;
    for (int i = 0; i < s.n; i += 4) {
        const int r = s.n - i;
        const slab_f32x4 z = {0.f, 0.f, 0.f, 0.f};
        const slab_f32x4 v0 = *reinterpret_cast<const slab_f32x4*>(p + (long long)i * s.stride);
        const slab_f32x4 v1 = r > 1 ? *reinterpret_cast<const slab_f32x4*>(p + (long long)(i + 1) * s.stride) : z;
        const slab_f32x4 v2 = r > 2 ? *reinterpret_cast<const slab_f32x4*>(p + (long long)(i + 2) * s.stride) : z;
        const slab_f32x4 v3 = r > 3 ? *reinterpret_cast<const slab_f32x4*>(p + (long long)(i + 3) * s.stride) : z;
        acc += v0;
        if (r > 1) acc += v1;
        if (r > 2) acc += v2;
        if (r > 3) acc += v3;
    }
    return acc;
}

// prof.hip: RAII timing scope around a kernel launch (no-op unless set_profile_enable(1))
struct ProfScope {
    int idx;
    hipStream_t st;
    ProfScope(const char* tag, hipStream_t s, double flops = 0.0, double bytes = 0.0);
    ~ProfScope();
};

// editnet.hip (shared host helpers)
int env_int(const char* name, int dflt);
int gemm_target_wgs();
GemmProb slab_prob(float* slab, int M, int N, int Bmax);
GemmProb direct_prob(float* out, long long ldo, int M, int N, const float* bias, int act);
int editnet_encoder(const SetEditNetWeights* w, const int64_t* seq, const int64_t* lens, float* H, float* Mem,
                    float* final_hidden, float* mask, int B, int T, int D, int V, float* emb_seq, float* xg,
                    float* enc_h, float* enc_c, float* s_enc, float* s_aff, hipStream_t st, int* order = nullptr,
                    void* enc_bar = nullptr);
// encoder_persistent.hip: the whole encoder recurrence in one weights-stationary launch with grid barriers
size_t persistent_encoder_bar_bytes();
bool persistent_encoder_ok(int B, int D, int T);
struct PEncDirHost { const float* w_hh; const float* xg; const float* b_extra; float* hbuf0; float* hbuf1; int out_col0; int reverse; };
int persistent_encoder_dirs(const PEncDirHost* dirs, int ndir, long long ld_xg_row, long long ld_xg_t, const int64_t* lens,
                            const int64_t* seq, int seq_T, int seq_V, float* H, float* Mem, long long ld_out_b,
                            long long ld_out_t, const int* perm, const int* nactive, void* bar, int B, int D, int T,
                            hipStream_t s);
int persistent_encoder(const float* w_hh, const float* xg, long long ld_xg_row, long long ld_xg_t, const float* b_extra,
                       const int64_t* lens, const int64_t* seq, int seq_T, int seq_V, float* hbuf0, float* hbuf1, float* H,
                       float* Mem, long long ld_out_b, long long ld_out_t, const int* perm, const int* nactive, void* bar,
                       int B, int D, int T, hipStream_t s);

struct PDecTeacher { const int64_t* caps; long long caps_stride; float* predictions; const int* host_decode_lengths; };   // teacher-forced mode
struct PDecBeam { int* hist_par; int64_t* hist_word; float* best_score; int64_t* best_word; int* result; };                // beam mode (one image, rows = hypotheses)
// decode_persistent.hip: the greedy loop of a small batch as one launch with grid barriers
constexpr int PDEC_MAXB = 8;          // rows of the <= 8-row persistent decode kernels
constexpr int PDW_MAXB = 16;          // rows of the wide EditNet variant (decode_persistent_wide.hip): one full 16-row MFMA tile
size_t dcnet_persistent_xbytes(int B, int D, int A);
bool dcnet_persistent_ok(const SetDcnetDims* d, int max_len);
int dcnet_persistent_greedy(const SetDcnetWeights* w, const SetDcnetDims* d, const float* pre1, const float* att1_c,
                            const float* mask, const float* pc, void* xbuf, long long* it, int* unfinished, int* alive,
                            long long start_idx, long long end_idx, int max_len, long long* seq, float* seq_logp,
                            hipStream_t s, const PDecTeacher* teach = nullptr);

size_t editnet_persistent_xbytes(int B, int D, int A);
bool editnet_persistent_ok(const SetEditNetDims* d, int max_len);
bool editnet_persistent_wide_ok(int B, int D, int A, int T, int R, int V);    // decode_persistent_wide.hip
bool persistent_disabled();                                                   // encoder_persistent.hip: a persistent launch timed out earlier
int editnet_persistent_greedy(const SetEditNetWeights* w, const SetEditNetDims* d, const float* pre1, const float* att1,
                              const float* att1_c, const float* mask, const float* capP, const float* memQ, const float* Mem,
                              const float* pv, void* xbuf, long long* it, int* unfinished, int* alive, long long start_idx,
                              long long end_idx, int max_len, long long* seq, float* seq_logp, hipStream_t s,
                              const PDecTeacher* teach = nullptr, const PDecBeam* beam = nullptr);

// a per-row gathered addend: value(m, n) = tab[ids[m*id_stride]*ld + col0 + n]   (tab == NULL: none)
struct RowGather {
    const float* tab = nullptr;
    const long long* ids = nullptr;
    long long id_stride = 1, ld = 0;
    int col0 = 0;
    int nrows = 1;          // table rows (V): ids are clamped to [0, nrows) like embed_relu_k does (no OOB read)
    __host__ __device__ const float* row(long long m) const {
        long long id = ids[m * id_stride];
        id = id < 0 ? 0 : (id >= nrows ? nrows - 1 : id);
        return tab + id * ld + col0;
    }
};

// pointwise.hip
int lstm_pointwise(Slabs g0, Slabs g1, Slabs g2, const float* pre, long long ldpre, const float* b0,
                   const float* b1, const float* c_in, float* c_out, float* h_out, float* ogate_out,
                   int M, int D, hipStream_t s, RowGather gt = RowGather(), float* gates_out = nullptr);
int context_gate_pointwise(Slabs cg_a, Slabs cg_b, const float* cg_bias, Slabs sc, const float* sc_bias,
                           Slabs tc, const float* tc_bias, float* out, int M, int D, hipStream_t s,
                           float* zt_out = nullptr, float* s_out = nullptr, float* t_out = nullptr,
                           RowGather gz = RowGather(), RowGather gt = RowGather());
int copy_gate_pointwise(Slabs gn, const float* bn, Slabs gm, const float* bm, const float* c_new,
                        const float* sel, const float* ogate, float* c_out, float* h_out, int M, int D,
                        hipStream_t s, float* cg_out = nullptr);
int reduce_bias_act(Slabs in, const float* b0, const float* b1, float* out, long long ldo, int M, int N,
                    int act, hipStream_t s);
int embed_relu(const float* table, const int64_t* ids, long long ids_stride, float* out, long long ldo,
               int n, int D, int V, hipStream_t s);
int mean_regions(const float* X, float* out, int B, int R, int F, hipStream_t s);
int encoder_pointwise(Slabs hh, const float* xg, long long ld_xg_row, long long ld_xg_t, int t,
                      const int64_t* lens, int reverse, float* h, float* c, float* H, float* Mem,
                      long long ld_out_b, long long ld_out_t, int out_col0, int B, int D, const float* b_extra,
                      hipStream_t s);
int rowsum_mask(const float* x, long long ld_row, int rows, int D, float* mask, hipStream_t s);
int zero_f32(float* p, size_t n, hipStream_t s);
int zero_runs(float* const* p, const size_t* n, int cnt, hipStream_t st);   // editnet.hip: adjacent buffers share one fill

// attention.hip
int caption_attention(const float* att1_c, Slabs att2_c, const float* dec_bias, const float* w_full,
                      const float* b_full, const float* mask, const float* H, const float* Mem,
                      float* ctx, float* sel, float* alpha_out, int M, int T, int Dh, int A,
                      hipStream_t s, float* att2_out = nullptr);
int visual_attention(const float* att1, Slabs att2, const float* dec_bias, const float* w_full,
                     const float* b_full, const float* X, const float* rmask, float* ctx,
                     float* alpha_out, int M, int R, int F, int A, hipStream_t s);
// hoisted-projection operands of the caption role (attention.hip CapAttArgs): P (B,T,2Dh), Q (B,T,Dh) or NULL
struct CapHoist {
    const float* P = nullptr; const float* Q = nullptr; float* cmem_out = nullptr; float* gated_out = nullptr;
    Slabs cg_ab{nullptr, 0, 0, 0}, tc{nullptr, 0, 0, 0}; RowGather gz, gtc;
    const float *b_gate = nullptr, *b_sc = nullptr, *b_tc = nullptr;
};
int step_attention(const float* att1, Slabs att2, const float* v_dec_bias, const float* v_w_full, const float* v_b_full,
                   const float* X, const float* rmask, float* v_ctx, float* v_alpha, int R, int F,
                   const float* att1_c, Slabs att2_c, const float* c_dec_bias, const float* c_w_full,
                   const float* c_b_full, const float* mask, const float* H, const float* Mem, float* c_ctx, float* sel,
                   float* c_alpha, int T, int Dh, int A, int M, hipStream_t s, const CapHoist* hoist = nullptr,
                   float* v_att2_out = nullptr, float* c_att2_out = nullptr);
int region_masks(const float* X, const float* fe, float* rmask, int B, int R, int F, int D, hipStream_t s);
int select_rows(const float* Mem, const float* alpha, float* sel, int M, int T, int D, hipStream_t s);

// gemm_fused.hip: small-tile GEMMs with fused pointwise epilogues (no slabs)
int fused_context_gate(const float* ctx, const float* w_gate_ctx, long long ld_gate, const float* w_sc, Slabs cg_ab,
                       Slabs tc, const float* b_gate, const float* b_sc, const float* b_tc, float* out, int M, int D,
                       hipStream_t s, RowGather gz = RowGather(), RowGather gtc = RowGather());
int fused_copy_gate(const float* c_new, const float* sel, const float* ogate, const float* w_cnew, const float* w_cmem,
                    const float* b_cnew, const float* b_cmem, float* c_out, float* h_out, int M, int D, hipStream_t s);
int fused_encoder_step_train(const float* h_in, const float* c_in, float* h_out, float* c_out, const float* w_hh,
                             const float* xg, long long ld_xg_row, long long ld_xg_t, const float* b_hh, const int64_t* lens,
                             int t, float* H, float* Mem, float* Hprev, long long ld_out_b, long long ld_out_t, int out_col0,
                             float* gates, int B, int D, hipStream_t s);
int fused_copy_gate_pre(const float* c_new, const float* sel, const float* cmem_pre, const float* ogate,
                        const float* w_cnew, const float* b_cnew, const float* b_cmem, float* c_out, float* h_out, int M,
                        int D, hipStream_t s);
int fused_encoder_step(const float* h_in, float* h_out, float* c, const float* w_hh, const float* xg,
                       long long ld_xg_row, long long ld_xg_t, const float* b_extra, const int64_t* lens, int t,
                       int reverse, float* H, float* Mem, long long ld_out_b, long long ld_out_t, int out_col0, int B,
                       int D, hipStream_t s, const int64_t* seq = nullptr, int seq_T = 0, int seq_V = 0, const int* perm = nullptr,
                       const int* nactive = nullptr);
int encoder_order(const int64_t* lens, int B, int T, int* perm, int* nactive, hipStream_t s);

// epilogue.hip
// Optional tail of the pick kernels (free-running loops with the token table): the NEXT timestep's attention-LSTM cell
// update of the row.  Its gate products [h2 | h1] W were launched together with this timestep's fc; the only operand that
// waits for the pick is the token-table row, so the workgroup that has just chosen the row's word finishes the cell here
// and the loop needs no lstm_pointwise launch between the pick and phase B.  Same operand order as lstm_pointwise_k
// (slabs, pre, table row).
struct LstmTail {
    Slabs g0{nullptr, 0, 0, 0};              // gate pre-activation slabs (rows, 4D)
    const float* pre = nullptr;              // hoisted loop-invariant columns + biases (rows, ldpre)
    long long ldpre = 0;
    const float* tab = nullptr;              // token table: row `word`, columns col0 .. col0 + 4D
    long long ld_tab = 0;
    int col0 = 0, nrows = 0;
    const float* c_in = nullptr;
    float* c_out = nullptr;
    float* h_out = nullptr;
    int D = 0;
};
int greedy_pick(Slabs logits, const float* bias, int V, int t, int max_len, long long end_idx,
                long long* seq, float* seq_logp, long long* it, int* unfinished, int* alive,
                const float* table, float* emb_out, int D, int B, hipStream_t s, const LstmTail* tail = nullptr);
int sample_pick(Slabs logits, const float* bias, int V, int t, int max_len, long long end_idx, long long* seq,
                float* seq_logp, long long* it, int* unfinished, int* alive, const float* table, float* emb_out, int D,
                int B, unsigned long long seed, unsigned long long offset, long long* raw_ids, float* lse,
                float* step_logp, hipStream_t s, const LstmTail* tail = nullptr);
int sample_logp_bwd(const float* logits, long long ld, const float* lse, const long long* ids, const float* g,
                    float* dlogits, long long ldd, int B, int V, hipStream_t s);
int philox_fill(uint32_t* out, int n, unsigned long long seed, unsigned long long offset, hipStream_t s);
int iota_i64(long long* p, int n, hipStream_t s);
int set_tokens(long long* it, long long value, int* unfinished, int* alive, int n_alive, int B,
               hipStream_t s);

}  // namespace set
