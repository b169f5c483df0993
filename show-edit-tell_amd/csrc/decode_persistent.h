// Shared pieces of the persistent small-batch decode kernels (decode_persistent.hip: DCNet, decode_persistent_wide.hip:
// EditNet): constants, the MFMA GEMV tile helpers and the diagnostic time stamps.
#pragma once
#include <cstdio>
#include "set_common.h"
#include "grid_barrier.h"

namespace set {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef const f32x4 __attribute__((address_space(1)))* gptr4;

// (set_common.h: PDEC_MAXB = 8 rows in the DCNet kernel, PDW_MAXB = 16 — one full 16-row MFMA tile — in the EditNet kernel)
constexpr int PDEC_TREG = 20;      // DCNet, B <= 4: previous-caption positions whose hoisted attention rows a wave keeps in registers
constexpr int PDEC_TMAX = 32;      // previous-caption positions held in registers by the Pc gather
constexpr int PDEC_KB = 16;        // 16-wide k-blocks per wave and gate tile: D = 1024 -> K quarter 256
constexpr int PDEC_THREADS = 256;
constexpr int PDEC_FC_TILES = 3;   // 16-row fc tiles per workgroup: up to 48 vocabulary rows

// diagnostic (SET_PDEC_STAMPS=1): 100-MHz time stamps of workgroup SET_PDEC_STAMP_WG, 24 per timestep
constexpr int PD_STAMPS = 24, PD_STAMP_STEPS = 64;
#define PD_STAMP(i) if (P.stamps && t < PD_STAMP_STEPS && (int)blockIdx.x == P.stamp_wg && threadIdx.x == 0) P.stamps[t * PD_STAMPS + (i)] = __builtin_amdgcn_s_memrealtime()

__device__ __forceinline__ float pd_sigm(float x) { return 1.f / (1.f + expf(-x)); }
// tanh of the attention scores: 1 - 2 / (1 + e^(2x)) on the hardware exp2 / rcp (absolute error ~2e-7; saturates to +-1 for
// large |x| without a branch).  -DSET_PDEC_TANHF: libm's tanhf, ~10x the instructions (160 per lane and row every timestep)
__device__ __forceinline__ float pd_tanh(float x) {
#ifdef SET_PDEC_TANHF
    return tanhf(x);
#else
    const float e = __builtin_amdgcn_exp2f(x * 2.885390081777927f);      // e^(2x)
    return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + e);
#endif
}
// this lane's share of one attention score: sum over its 8 columns of w * tanh(x), two columns per packed instruction
// (v_pk_mul / v_pk_add / v_pk_fma; the exp2 / rcp pairs stay scalar).  -DSET_PDEC_TANHF: the libm form, column by column.
typedef float pd_f32x2 __attribute__((ext_vector_type(2)));
typedef float pd_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float pd_score8(const pd_f32x4& x0, const pd_f32x4& x1, const pd_f32x4& w0, const pd_f32x4& w1) {
#ifdef SET_PDEC_TANHF
    float s = w0[0] * tanhf(x0[0]) + w0[1] * tanhf(x0[1]) + w0[2] * tanhf(x0[2]) + w0[3] * tanhf(x0[3]);
    s += w1[0] * tanhf(x1[0]) + w1[1] * tanhf(x1[1]) + w1[2] * tanhf(x1[2]) + w1[3] * tanhf(x1[3]);
    return s;
#else
    const pd_f32x2 c2 = {2.885390081777927f, 2.885390081777927f}, one = {1.f, 1.f}, mtwo = {-2.f, -2.f};
    pd_f32x2 acc = {0.f, 0.f};
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const pd_f32x2 x = p < 2 ? (pd_f32x2){x0[2 * p], x0[2 * p + 1]} : (pd_f32x2){x1[2 * p - 4], x1[2 * p - 3]};
        const pd_f32x2 w = p < 2 ? (pd_f32x2){w0[2 * p], w0[2 * p + 1]} : (pd_f32x2){w1[2 * p - 4], w1[2 * p - 3]};
        const pd_f32x2 a = x * c2;
        const pd_f32x2 d = (pd_f32x2){__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y)} + one;      // 1 + e^(2x)
        const pd_f32x2 r = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
        acc += w * (r * mtwo + one);                                                                      // w * tanh(x)
    }
    return acc.x + acc.y;
#endif
}
__device__ __forceinline__ float pd_wsum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float pd_wmax(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// ---- wave reductions on DPP: quad swaps, half-row mirror, row mirror leave every lane of a 16-lane row with the row's
// result; the four rows are then combined in a fixed order from four v_readlane
template <int CTRL>
__device__ __forceinline__ float pw_dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
template <int CTRL>
__device__ __forceinline__ int pw_dppi(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false); }
__device__ __forceinline__ float pw_lane(float v, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l)); }
__device__ __forceinline__ float pw_wsum(float v) {
    v += pw_dpp<0xB1>(v);            // quad_perm [1,0,3,2]
    v += pw_dpp<0x4E>(v);            // quad_perm [2,3,0,1]
    v += pw_dpp<0x141>(v);           // row_half_mirror
    v += pw_dpp<0x140>(v);           // row_mirror
    return ((pw_lane(v, 0) + pw_lane(v, 16)) + pw_lane(v, 32)) + pw_lane(v, 48);
}
__device__ __forceinline__ float pw_wmax(float v) {
    v = fmaxf(v, pw_dpp<0xB1>(v));
    v = fmaxf(v, pw_dpp<0x4E>(v));
    v = fmaxf(v, pw_dpp<0x141>(v));
    v = fmaxf(v, pw_dpp<0x140>(v));
    return fmaxf(fmaxf(pw_lane(v, 0), pw_lane(v, 16)), fmaxf(pw_lane(v, 32), pw_lane(v, 48)));
}
// (largest value, smallest index among equals): torch.max's first-index rule; a NaN never wins a comparison
__device__ __forceinline__ void pw_wargmax(float& best, int& bi) {
#define PW_STEP(CTRL)                                                                  \
    {                                                                                  \
        const float ob = pw_dpp<CTRL>(best);                                           \
        const int oi = pw_dppi<CTRL>(bi);                                              \
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }              \
    }
    PW_STEP(0xB1) PW_STEP(0x4E) PW_STEP(0x141) PW_STEP(0x140)
#undef PW_STEP
    float b = pw_lane(best, 0);
    int i = __builtin_amdgcn_readlane(bi, 0);
#pragma unroll
    for (int l = 16; l < 64; l += 16) {
        const float ob = pw_lane(best, l);
        const int oi = __builtin_amdgcn_readlane(bi, l);
        if (ob > b || (ob == b && oi < i)) { b = ob; i = oi; }
    }
    best = b; bi = i;
}


template <int KB>
__device__ __forceinline__ void pd_load(f32x4 (&w)[KB], const float* p) {
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) w[kb] = *(gptr4)(p + 16 * kb);
}
template <int KB>
__device__ __forceinline__ void pd_load_if(f32x4 (&w)[KB], const float* p, bool valid) {
    if (valid) {
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) w[kb] = *(gptr4)(p + 16 * kb);
    } else {
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) w[kb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
}
// acc (16 batch rows x 16 weight rows) += act[rows, this wave's k range] . W^T; the activation operand comes from LDS
template <int KB>
__device__ __forceinline__ void pd_mma(f32x4& acc, const f32x4 (&w)[KB], const float* sact) {
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(sact + 16 * kb);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], w[kb][j], acc, 0, 0, 0);
    }
}

// ---- EditNet: arguments of decode_persistent_wide.hip (1 .. 16 rows; greedy, teacher-forced and beam mode)
constexpr int PDEC_RREG = 36;      // image regions whose hoisted x2h products a thread keeps in registers

struct PDecEditArgs {
    // weights (nn.Linear layout, used in place)
    const float* al_wih; long long ld_ih;        // attention_lstm.weight_ih (4D, 3D + F): columns [2D, 3D) = h2
    const float* al_whh;
    const float* cl_h2h_w;
    const float* cl_x2h_w; long long ld_x2h;     // copy_lstm.x2h (4D, 2D + F): [0, D) h1, [D, 2D) attend_cap
    const float *cl_x2h_b, *cl_h2h_b;
    const float *ca_gate_w, *ca_gate_b;          // (D, 3D): columns [D, 2D) = h1
    const float *ca_tc_w, *ca_tc_b;              // (D, 2D): columns [D, 2D) = h1
    const float* ca_sc_b;
    const float *ca_dec_w, *ca_dec_b, *ca_full_w, *ca_full_b;
    const float *va_dec_w, *va_dec_b, *va_full_w, *va_full_b;
    const float *cl_cnew_w, *cl_cnew_b, *cl_cmem_b;
    const float *fc_w, *fc_b;
    const float* tok_table; long long ld_tab;    // (V, 10D): [0, 4D) gates, [4D, 5D) tc_affine, [5D, 6D) context_gate
    // per sequence (prologue outputs)
    const float* pre1;                           // (B, 4D)
    const float* att1;                           // (B, R, A) features_att(relu(att_embed(X)))
    const float* att1_c;                         // (B, T, A)
    const float* mask;                           // (B, T)
    const float* capP;                           // (B, T, 2D) [context_gate.W[:, 2D:] H | sc_affine.W H]
    const float* memQ;                           // (B, T, D)  gate_cmem.W Mem
    const float* Mem;                            // (B, T, D)
    const float* pv;                             // (B, R, 4D) X x2h[:, 2D:]^T
    // exchange buffers (flag-in-data words, zero-filled before the launch)
    void *x_h1, *x_a2, *x_gt, *x_vs, *x_cn, *x_h2, *x_fc;
    void* x_cs;                                  // wide variant (decode_persistent_wide.hip): caption scores (B, TMAX)
    long long* it; int* unfinished; int* alive;
    long long* seq; float* seq_logp;
    unsigned* status; unsigned* fault; unsigned spin_limit; int test_stall;
    int B, D, T, R, A, V, max_len, rpw;
    long long start_idx, end_idx;
    // teacher-forced mode (set_editnet_xe_forward, editnet.py:505-546): words from caps, scores of the first bt rows written
    // out, no pick and no sixth exchange
    const long long* caps; long long caps_stride;
    float* predictions; long long ld_pred_b;     // (B, maxT, V)
    int dlen[PDW_MAXB];                          // decode lengths, descending
    int stamp_wg;
    unsigned long long* stamps;
    // beam mode of the wide variant (set_editnet_beam_persistent): the rows are the k hypotheses of ONE image
    void* x_fcb;                                 // (B, G, 12) per-slice (max, sum exp, 4 x (score, word)) words
    int* bm_hist_par;                            // (max_len, 4) parent slot of every slot after every pick
    long long* bm_hist_word;                     // (max_len, 4) word appended to every slot at every pick
    float* bm_best_score;                        // [1] best completed hypothesis (-inf: none)
    long long* bm_best_word;                     // [1] its last word (<end>)
    int* bm_result;                              // [4] pick index and parent slot of the best completed hypothesis, k_left, picks made
};

// the wide variant's launch (decode_persistent_wide.hip); P is complete except for the exchange pointers it lays out itself
int editnet_persistent_wide_launch(PDecEditArgs& P, void* xbuf, PersistentGuard& guard, hipStream_t s, bool* unsupported, bool beam = false);
size_t editnet_persistent_wide_xbytes(int B, int D, int A);
bool editnet_persistent_wide_ok(int B, int D, int A, int T, int R, int V);

// host side of the stamps: buffer for a launch (or NULL) and the report after it
inline int pd_stamps_begin(unsigned long long** out, int* wg, hipStream_t s) {
    static const int on = env_int("SET_PDEC_STAMPS", 0);
    static unsigned long long* d_stamps = nullptr;
    *out = nullptr; *wg = 0;
    if (!on) return SET_OK;
    if (!d_stamps) SET_HIP_TRY(hipMalloc((void**)&d_stamps, sizeof(unsigned long long) * PD_STAMPS * PD_STAMP_STEPS));
    SET_HIP_TRY(hipMemsetAsync(d_stamps, 0, sizeof(unsigned long long) * PD_STAMPS * PD_STAMP_STEPS, s));
    *out = d_stamps; *wg = env_int("SET_PDEC_STAMP_WG", 0);
    return SET_OK;
}
inline int pd_stamps_report(const unsigned long long* d_stamps, int wg, int last, int max_len, hipStream_t s) {
    if (!d_stamps) return SET_OK;
    static unsigned long long h[PD_STAMPS * PD_STAMP_STEPS];
    SET_HIP_TRY(hipStreamSynchronize(s));
    SET_HIP_TRY(hipMemcpy(h, d_stamps, sizeof(h), hipMemcpyDeviceToHost));
    double acc[PD_STAMPS] = {0}, tot = 0;
    int n = 0;
    for (int t = 1; t < max_len && t < PD_STAMP_STEPS && h[t * PD_STAMPS + last]; ++t, ++n) {
        for (int i = 0; i < last; ++i) acc[i] += (double)(h[t * PD_STAMPS + i + 1] - h[t * PD_STAMPS + i]) * 0.01;
        tot += (double)(h[t * PD_STAMPS + last] - h[t * PD_STAMPS]) * 0.01;
    }
    if (n) {
        fprintf(stderr, "pdec stamps (us, mean of %d steps, wg %d):", n, wg);
        for (int i = 0; i < last; ++i) fprintf(stderr, " %d-%d:%.2f", i, i + 1, acc[i] / n);
        if (env_int("SET_PDEC_STAMPS", 0) > 1) {                 // 2: also every stamp relative to the timestep's first one
            fprintf(stderr, "  | at:");
            for (int i = 1; i <= last; ++i) {
                double a = 0;
                for (int t = 1; t <= n; ++t) a += (double)(h[t * PD_STAMPS + i] - h[t * PD_STAMPS]) * 0.01;
                fprintf(stderr, " %d=%.2f", i, a / n);
            }
        }
        fprintf(stderr, "  step %.2f  (step-to-step %.2f)\n", tot / n, n > 1 ? (double)(h[n * PD_STAMPS] - h[PD_STAMPS]) * 0.01 / (n - 1) : 0.0);
    }
    return SET_OK;
}

}  // namespace set
