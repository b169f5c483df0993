// Operator-level C ABI: the sub-module `forward`s that the reference's beam search and ensemble
// evaluation call directly (editnet.py:645-653, eval/eval xe/eval_full.py:133-149).  Each is a
// composition of the same kernels the fused step uses; nothing is hoisted here unless the caller
// passes the hoisted tensor (att1 / att1_c), so semantics match the reference call for call.
#include "set_common.h"

using namespace set;

namespace {
constexpr size_t KS = GEMM_MAX_KSPLIT;
size_t fbytes(size_t n) { return round_up(n * sizeof(float), 256); }
}  // namespace

extern "C" {

// ------------------------------------------------------------------------------- nn.Linear
size_t set_linear_workspace_bytes(int M, int N, int K) {
    (void)K;
    if (M <= 0 || N <= 0) return 0;
    return fbytes(KS * (size_t)M * N) + 256;
}

int set_linear_f32(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* bias, float* y, int64_t ldy,
                   int M, int N, int K, int act, void* ws, size_t ws_bytes, void* stream) {
    if (!x || !w || !y || M <= 0 || N <= 0 || K <= 0) return SET_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    GemmProb p = slab_prob((float*)ws, M, N, M);
    p.add(x, ldx, w, ldw, K);
    plan_ksplit(&p, 1, gemm_target_wgs());
    if (p.ksplit == 1) {
        GemmProb q = direct_prob(y, ldy, M, N, bias, act);
        q.add(x, ldx, w, ldw, K);
        return gemm_group(&q, 1, st);
    }
    if (!ws || !aligned16(ws) || ws_bytes < fbytes((size_t)p.ksplit * M * N)) return SET_ERR_WORKSPACE;
    SET_TRY(gemm_group(&p, 1, st));
    return reduce_bias_act(slabs_of(p), bias, nullptr, y, ldy, M, N, act, st);
}

// ------------------------------------------------------------------------------- EmbeddingC
int set_embed_relu_f32(const float* table, const int64_t* ids, int64_t ids_stride, float* out, int64_t ldo, int n,
                       int D, int V, void* stream) {
    if (!table || !ids || !out || n < 0 || D <= 0 || V <= 0) return SET_ERR_ARG;
    return embed_relu(table, ids, ids_stride, out, ldo, n, D, V, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------- nn.LSTMCell / LSTMCellC
size_t set_lstm_cell_workspace_bytes(int M, int D, int Kx) {
    (void)Kx;
    if (M <= 0 || D <= 0) return 0;
    return fbytes(KS * (size_t)M * 4 * D) + 256;
}

static int lstm_cell_impl(const float* x, int64_t ldx, int Kx, const float* h, const float* c, const float* w_ih,
                          int64_t ld_wih, const float* w_hh, const float* b_ih, const float* b_hh, float* h_out,
                          float* c_out, float* gates_out, int M, int D, void* ws, size_t ws_bytes, void* stream) {
    if (!x || !h || !c || !w_ih || !w_hh || !h_out || !c_out || M <= 0 || D <= 0 || Kx <= 0) return SET_ERR_ARG;
    if (!ws || !aligned16(ws) || ws_bytes < set_lstm_cell_workspace_bytes(M, D, Kx) - 256) return SET_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    GemmProb p = slab_prob((float*)ws, M, 4 * D, M);
    p.add(x, ldx, w_ih, ld_wih, Kx);
    p.add(h, D, w_hh, D, D);
    plan_ksplit(&p, 1, gemm_target_wgs());
    SET_TRY(gemm_group(&p, 1, st));
    const Slabs none{nullptr, 0, 0, 0};
    return lstm_pointwise(slabs_of(p), none, none, nullptr, 0, b_ih, b_hh, c, c_out, h_out, nullptr, M, D, st,
                          RowGather(), gates_out);
}

int set_lstm_cell_f32(const float* x, int64_t ldx, int Kx, const float* h, const float* c, const float* w_ih,
                      int64_t ld_wih, const float* w_hh, const float* b_ih, const float* b_hh, float* h_out,
                      float* c_out, int M, int D, void* ws, size_t ws_bytes, void* stream) {
    return lstm_cell_impl(x, ldx, Kx, h, c, w_ih, ld_wih, w_hh, b_ih, b_hh, h_out, c_out, nullptr, M, D, ws, ws_bytes,
                          stream);
}

int set_lstm_cell_train_f32(const float* x, int64_t ldx, int Kx, const float* h, const float* c, const float* w_ih,
                            int64_t ld_wih, const float* w_hh, const float* b_ih, const float* b_hh, float* h_out,
                            float* c_out, float* gates_out, int M, int D, void* ws, size_t ws_bytes, void* stream) {
    if (!gates_out) return SET_ERR_ARG;
    return lstm_cell_impl(x, ldx, Kx, h, c, w_ih, ld_wih, w_hh, b_ih, b_hh, h_out, c_out, gates_out, M, D, ws, ws_bytes,
                          stream);
}

// nn.LSTMCell whose input is a concatenation with loop-invariant column blocks (editnet.py:523 [emb | final_hidden | h2 |
// image_mean], dcnet.py:336 [emb | final_hidden | h2]): the caller contracts the invariant blocks ONCE per sequence
// (`pre` (M,4D) = their products + both biases) and passes only the per-step blocks as up to two (x, W column block)
// segments — half of the attention LSTM's contraction length, and the X-side weight gradient of the invariant columns
// becomes (sum_t dgates) x (invariant input) instead of a product over all T*B rows.
int set_lstm_cell_pre_train_f32(const float* x0, int64_t ld_x0, const float* w0, int64_t ld_w0, int K0, const float* x1,
                                int64_t ld_x1, const float* w1, int64_t ld_w1, int K1, const float* h, const float* w_hh,
                                const float* pre, int64_t ld_pre, const float* c, float* h_out, float* c_out,
                                float* gates_out, int M, int D, void* ws, size_t ws_bytes, void* stream) {
    if (!x0 || !w0 || !h || !w_hh || !pre || !c || !h_out || !c_out || !gates_out || M <= 0 || D <= 0 || K0 <= 0)
        return SET_ERR_ARG;
    if (x1 && (!w1 || K1 <= 0)) return SET_ERR_ARG;
    if (!ws || !aligned16(ws) || ws_bytes < set_lstm_cell_workspace_bytes(M, D, K0) - 256) return SET_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    GemmProb p = slab_prob((float*)ws, M, 4 * D, M);
    p.add(x0, ld_x0, w0, ld_w0, K0);
    if (x1) p.add(x1, ld_x1, w1, ld_w1, K1);
    p.add(h, D, w_hh, D, D);
    plan_ksplit(&p, 1, gemm_target_wgs());
    SET_TRY(gemm_group(&p, 1, st));
    const Slabs none{nullptr, 0, 0, 0};
    return lstm_pointwise(slabs_of(p), none, none, pre, ld_pre, nullptr, nullptr, c, c_out, h_out, nullptr, M, D, st,
                          RowGather(), gates_out);
}

// ------------------------------------------------------------------------------- CaptionAttentionC
size_t set_caption_attention_workspace_bytes(int M, int T, int Dh, int A) {
    if (M <= 0 || T <= 0 || Dh <= 0 || A <= 0) return 0;
    return fbytes((size_t)M * T * A) + fbytes(KS * (size_t)M * A) + 4 * fbytes(KS * (size_t)M * Dh) +
           fbytes((size_t)M * Dh) + 256;
}

static int caption_attention_impl(const SetEditNetWeights* w, const float* H, const float* att1_c, const float* h1,
                                  const float* word, const float* mask, float* gated, float* alpha_c, float* ctx_out,
                                  float* zt_out, float* s_out, float* t_out, int M, int T, int Dh, int D, int A,
                                  void* ws, size_t ws_bytes, void* stream, float* att2_out = nullptr) {
    if (!w || !H || !h1 || !mask || !gated || M <= 0 || T <= 0 || Dh <= 0 || D <= 0 || A <= 0) return SET_ERR_ARG;
    if (!ws || !aligned16(ws) || ws_bytes < set_caption_attention_workspace_bytes(M, T, Dh, A) - 256)
        return SET_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int tgt = gemm_target_wgs();
    const bool gating = w->ca_gate_w != nullptr;
    if (gating && (!word || Dh != D)) return SET_ERR_ARG;
    Carver cv(ws);
    float* a1 = cv.take<float>((size_t)M * T * A);
    float* s_att2 = cv.take<float>(KS * (size_t)M * A);
    float* s_tc = cv.take<float>(KS * (size_t)M * Dh);
    float* s_cga = cv.take<float>(KS * (size_t)M * Dh);
    float* s_cgb = cv.take<float>(KS * (size_t)M * Dh);
    float* s_sc = cv.take<float>(KS * (size_t)M * Dh);
    float* ctx = cv.take<float>((size_t)M * Dh);
    if (ctx_out) ctx = ctx_out;
    if (!att1_c) {                                                     // editnet.py:370 / dcnet.py:261
        GemmProb p = direct_prob(a1, A, M * T, A, w->ca_feat_b, SET_ACT_NONE);
        p.add(H, Dh, w->ca_feat_w, Dh, Dh);
        SET_TRY(gemm_group(&p, 1, st));
        att1_c = a1;
    }
    GemmProb b[3];
    int nb = 1;
    b[0] = slab_prob(s_att2, M, A, M);
    b[0].add(h1, D, w->ca_dec_w, D, D);
    if (gating) {
        b[1] = slab_prob(s_tc, M, D, M);
        b[1].add(word, D, w->ca_tc_w, 2 * D, D);
        b[1].add(h1, D, w->ca_tc_w + D, 2 * D, D);
        b[2] = slab_prob(s_cga, M, D, M);
        b[2].add(word, D, w->ca_gate_w, 3 * D, D);
        b[2].add(h1, D, w->ca_gate_w + D, 3 * D, D);
        nb = 3;
    }
    plan_ksplit(b, nb, tgt);
    SET_TRY(gemm_group(b, nb, st));
    SET_TRY(caption_attention(att1_c, slabs_of(b[0]), w->ca_dec_b, w->ca_full_w, w->ca_full_b, mask, H, nullptr,
                              gating ? ctx : gated, nullptr, alpha_c, M, T, Dh, A, st, att2_out));
    if (!gating) return SET_OK;
    GemmProb c[2];
    c[0] = slab_prob(s_cgb, M, D, M);
    c[0].add(ctx, D, w->ca_gate_w + 2 * D, 3 * D, D);
    c[1] = slab_prob(s_sc, M, D, M);
    c[1].add(ctx, D, w->ca_sc_w, D, D);
    plan_ksplit(c, 2, tgt);
    SET_TRY(gemm_group(c, 2, st));
    return context_gate_pointwise(slabs_of(b[2]), slabs_of(c[0]), w->ca_gate_b, slabs_of(c[1]), w->ca_sc_b,
                                  slabs_of(b[1]), w->ca_tc_b, gated, M, D, st, zt_out, s_out, t_out);
}

int set_caption_attention_f32(const SetEditNetWeights* w, const float* H, const float* att1_c, const float* h1,
                              const float* word, const float* mask, float* gated, float* alpha_c, int M, int T,
                              int Dh, int D, int A, void* ws, size_t ws_bytes, void* stream) {
    return caption_attention_impl(w, H, att1_c, h1, word, mask, gated, alpha_c, nullptr, nullptr, nullptr, nullptr, M, T,
                                  Dh, D, A, ws, ws_bytes, stream);
}

// DCNet's un-gated attention (dcnet.py:254-270) for its training node: also emits the decoder-side projection it scored
// with (att2_out (M,A), what set_attention_bwd_f32 takes), so the backward does not recompute it
int set_caption_attention_att2_f32(const SetEditNetWeights* w, const float* H, const float* att1_c, const float* h1,
                                   const float* mask, float* ctx, float* alpha_c, float* att2_out, int M, int T, int Dh,
                                   int D, int A, void* ws, size_t ws_bytes, void* stream) {
    if (!w || w->ca_gate_w || !att1_c || !alpha_c || !att2_out) return SET_ERR_ARG;
    return caption_attention_impl(w, H, att1_c, h1, nullptr, mask, ctx, alpha_c, nullptr, nullptr, nullptr, nullptr, M, T, Dh,
                                  D, A, ws, ws_bytes, stream, att2_out);
}

int set_caption_attention_train_f32(const SetEditNetWeights* w, const float* H, const float* att1_c, const float* h1,
                                    const float* word, const float* mask, float* gated, float* alpha_c, float* ctx,
                                    float* zt, float* s, float* t, int M, int T, int Dh, int D, int A, void* ws,
                                    size_t ws_bytes, void* stream) {
    if (!ctx || !zt || !s || !t || !alpha_c || !att1_c) return SET_ERR_ARG;
    return caption_attention_impl(w, H, att1_c, h1, word, mask, gated, alpha_c, ctx, zt, s, t, M, T, Dh, D, A, ws, ws_bytes,
                                  stream);
}

// ------------------------------------------------------------------------------- VisualAttentionC
size_t set_visual_attention_workspace_bytes(int M, int R, int F, int D, int A) {
    (void)F;
    if (M <= 0 || R <= 0 || D <= 0 || A <= 0) return 0;
    return fbytes((size_t)M * R * D) + fbytes((size_t)M * R * A) + fbytes(KS * (size_t)M * A) + fbytes((size_t)M * R) +
           256;
}

int set_visual_attention_f32(const SetEditNetWeights* w, const float* X, const float* att1, const float* h1,
                             float* ctx, float* alpha, int M, int R, int F, int D, int A, int adaptive, void* ws,
                             size_t ws_bytes, void* stream) {
    if (!w || !X || !h1 || !ctx || M <= 0 || R <= 0 || F <= 0 || D <= 0 || A <= 0) return SET_ERR_ARG;
    if (!ws || !aligned16(ws) || ws_bytes < set_visual_attention_workspace_bytes(M, R, F, D, A) - 256)
        return SET_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    Carver cv(ws);
    float* fe = cv.take<float>((size_t)M * R * D);
    float* a1 = cv.take<float>((size_t)M * R * A);
    float* s_att2 = cv.take<float>(KS * (size_t)M * A);
    float* rmask = cv.take<float>((size_t)M * R);
    if (!att1 || adaptive) {
        GemmProb p = direct_prob(fe, D, M * R, D, w->va_emb_b, SET_ACT_RELU);        // editnet.py:441
        p.add(X, F, w->va_emb_w, F, F);
        SET_TRY(gemm_group(&p, 1, st));
        if (adaptive) SET_TRY(region_masks(X, fe, rmask, M, R, F, D, st));
    }
    if (!att1) {
        GemmProb q = direct_prob(a1, A, M * R, A, w->va_feat_b, SET_ACT_NONE);       // editnet.py:442
        q.add(fe, D, w->va_feat_w, D, D);
        SET_TRY(gemm_group(&q, 1, st));
        att1 = a1;
    }
    GemmProb b = slab_prob(s_att2, M, A, M);
    b.add(h1, D, w->va_dec_w, D, D);
    plan_ksplit(&b, 1, gemm_target_wgs());
    SET_TRY(gemm_group(&b, 1, st));
    return visual_attention(att1, slabs_of(b), w->va_dec_b, w->va_full_w, w->va_full_b, X, adaptive ? rmask : nullptr,
                            ctx, alpha, M, R, F, A, st);
}

// masked variant for the grad-enabled adaptive path: att1 and the region mask (att_embed(X).sum(2) != 0,
// editnet_adaptive.py:449-453) are supplied by the caller, nothing is recomputed
int set_visual_attention_masked_f32(const SetEditNetWeights* w, const float* X, const float* att1, const float* rmask,
                                    const float* h1, float* ctx, float* alpha, int M, int R, int F, int D, int A,
                                    void* ws, size_t ws_bytes, void* stream) {
    if (!w || !X || !att1 || !h1 || !ctx || M <= 0 || R <= 0 || F <= 0 || D <= 0 || A <= 0) return SET_ERR_ARG;
    if (!ws || !aligned16(ws) || ws_bytes < fbytes(KS * (size_t)M * A)) return SET_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    Carver cv(ws);
    float* s_att2 = cv.take<float>(KS * (size_t)M * A);
    GemmProb b = slab_prob(s_att2, M, A, M);
    b.add(h1, D, w->va_dec_w, D, D);
    plan_ksplit(&b, 1, gemm_target_wgs());
    SET_TRY(gemm_group(&b, 1, st));
    return visual_attention(att1, slabs_of(b), w->va_dec_b, w->va_full_w, w->va_full_b, X, rmask, ctx, alpha, M, R, F,
                            A, st);
}

// ------------------------------------------------------------------------------- both attentions + SelectC, one timestep
// The training node's per-step attention block (editnet.py:534-540): cap_decoder_att / decoder_att / tc_affine /
// context_gate[word,h1] in ONE grouped launch, both attention roles and the hard selection in ONE launch
// (step_attention_k), the context side of the gate in one launch + its pointwise — 4 launches instead of the 7 of
// set_caption_attention_train_f32 + set_visual_attention_masked_f32 + set_select_f32 called one after the other.
size_t set_editnet_attentions_workspace_bytes(int M, int D, int A) {
    if (M <= 0 || D <= 0 || A <= 0) return 0;
    return 2 * fbytes(KS * (size_t)M * A) + 4 * fbytes(KS * (size_t)M * D) + 256;
}

int set_editnet_attentions_train_f32(const SetEditNetWeights* w, const float* H, const float* att1_c, const float* mask,
                                     const float* Mem, const float* X, const float* att1, const float* rmask,
                                     const float* h1, const float* word, float* gated, float* alpha_c, float* ctx,
                                     float* zt, float* s, float* t, float* sel, float* attend_img, float* alpha_v,
                                     float* att2_c_out, float* att2_v_out, int M, int T, int R, int F, int D, int A, void* ws,
                                     size_t ws_bytes, void* stream) {
    if (!w || !H || !att1_c || !mask || !Mem || !X || !att1 || !h1 || !word || !gated || !alpha_c || !ctx || !zt || !s ||
        !t || !sel || !attend_img || !alpha_v || M <= 0 || T <= 0 || R <= 0 || F <= 0 || D <= 0 || A <= 0)
        return SET_ERR_ARG;
    if (!ws || !aligned16(ws) || ws_bytes < set_editnet_attentions_workspace_bytes(M, D, A) - 256) return SET_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int tgt = gemm_target_wgs();
    Carver cv(ws);
    float* s_a2c = cv.take<float>(KS * (size_t)M * A);
    float* s_a2v = cv.take<float>(KS * (size_t)M * A);
    float* s_tc = cv.take<float>(KS * (size_t)M * D);
    float* s_cga = cv.take<float>(KS * (size_t)M * D);
    float* s_cgb = cv.take<float>(KS * (size_t)M * D);
    float* s_sc = cv.take<float>(KS * (size_t)M * D);
    GemmProb b[4];
    b[0] = slab_prob(s_a2c, M, A, M);
    b[0].add(h1, D, w->ca_dec_w, D, D);
    b[1] = slab_prob(s_a2v, M, A, M);
    b[1].add(h1, D, w->va_dec_w, D, D);
    b[2] = slab_prob(s_tc, M, D, M);
    b[2].add(word, D, w->ca_tc_w, 2 * D, D);
    b[2].add(h1, D, w->ca_tc_w + D, 2 * D, D);
    b[3] = slab_prob(s_cga, M, D, M);
    b[3].add(word, D, w->ca_gate_w, 3 * D, D);
    b[3].add(h1, D, w->ca_gate_w + D, 3 * D, D);
    plan_ksplit(b, 4, tgt);
    SET_TRY(gemm_group(b, 4, st, "gemm:train att2,tc,cg"));
    SET_TRY(step_attention(att1, slabs_of(b[1]), w->va_dec_b, w->va_full_w, w->va_full_b, X, rmask, attend_img, alpha_v, R, F,
                           att1_c, slabs_of(b[0]), w->ca_dec_b, w->ca_full_w, w->ca_full_b, mask, H, Mem, ctx, sel, alpha_c,
                           T, D, A, M, st, nullptr, att2_v_out, att2_c_out));
    GemmProb c[2];
    c[0] = slab_prob(s_cgb, M, D, M);
    c[0].add(ctx, D, w->ca_gate_w + 2 * D, 3 * D, D);
    c[1] = slab_prob(s_sc, M, D, M);
    c[1].add(ctx, D, w->ca_sc_w, D, D);
    plan_ksplit(c, 2, tgt);
    SET_TRY(gemm_group(c, 2, st, "gemm:train cg_c,sc"));
    return context_gate_pointwise(slabs_of(b[3]), slabs_of(c[0]), w->ca_gate_b, slabs_of(c[1]), w->ca_sc_b, slabs_of(b[2]),
                                  w->ca_tc_b, gated, M, D, st, zt, s, t);
}

// ------------------------------------------------------------------------------- SelectC
int set_select_f32(const float* Mem, const float* alpha_c, float* sel, int M, int T, int D, void* stream) {
    if (!Mem || !alpha_c || !sel || M <= 0 || T <= 0 || D <= 0) return SET_ERR_ARG;
    return select_rows(Mem, alpha_c, sel, M, T, D, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------- CopyLSTMCellC
size_t set_copy_lstm_workspace_bytes(int M, int D, int Kx) {
    (void)Kx;
    if (M <= 0 || D <= 0) return 0;
    return 2 * fbytes(KS * (size_t)M * 4 * D) + 2 * fbytes(KS * (size_t)M * D) + 2 * fbytes((size_t)M * D) + 256;
}

}  // extern "C"
namespace set {
// the input rows x = [x_0 | x_1 | ...] given as 1..3 column segments (segment i: K[i] columns, row stride ld[i]) contracted
// against the matching column blocks of x2h.weight — the training loop feeds [h1 | gated | attend_img] from where the
// producing kernels left them instead of packing a row first
int copy_lstm_segs(const SetEditNetWeights* w, int nseg, const float* const* xs, const int64_t* lds, const int* Ks, const float* h2,
                   const float* c2, const float* c_memory, float* h_out, float* c_out, float* gates_out, float* cnew_out,
                   float* cg_out, int M, int D, void* ws, size_t ws_bytes, void* stream) {
    if (!w || !xs || !lds || !Ks || nseg <= 0 || nseg > 3 || !h2 || !c2 || !c_memory || !h_out || !c_out || M <= 0 || D <= 0)
        return SET_ERR_ARG;
    int Kx = 0;
    for (int i = 0; i < nseg; ++i) {
        if (!xs[i] || Ks[i] <= 0 || lds[i] < Ks[i]) return SET_ERR_ARG;
        Kx += Ks[i];
    }
    if (!ws || !aligned16(ws) || ws_bytes < set_copy_lstm_workspace_bytes(M, D, Kx) - 256) return SET_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int tgt = gemm_target_wgs();
    Carver cv(ws);
    float* s_g = cv.take<float>(KS * (size_t)M * 4 * D);
    float* s_g2 = cv.take<float>(KS * (size_t)M * 4 * D);
    float* s_m = cv.take<float>(KS * (size_t)M * D);
    float* s_n = cv.take<float>(KS * (size_t)M * D);
    float* c_new = cv.take<float>((size_t)M * D);
    float* ogate = cv.take<float>((size_t)M * D);
    if (cnew_out) c_new = cnew_out;
    // a problem holds GEMM_MAX_SEG = 3 segments: [x | h2] fits one problem when x is one or two segments; three x segments
    // make h2 W_hh^T its own problem whose partials the pointwise kernel adds (same launch either way)
    const bool own = nseg + 1 > GEMM_MAX_SEG;
    GemmProb a[3];
    a[0] = slab_prob(s_g, M, 4 * D, M);
    for (int i = 0, k0 = 0; i < nseg; k0 += Ks[i], ++i) a[0].add(xs[i], lds[i], w->cl_x2h_w + k0, Kx, Ks[i]);
    a[1] = slab_prob(s_m, M, D, M);
    a[1].add(c_memory, D, w->cl_cmem_w, D, D);
    if (own) {
        a[2] = slab_prob(s_g2, M, 4 * D, M);
        a[2].add(h2, D, w->cl_h2h_w, D, D);
    } else {
        a[0].add(h2, D, w->cl_h2h_w, D, D);
    }
    plan_ksplit(a, own ? 3 : 2, tgt);
    SET_TRY(gemm_group(a, own ? 3 : 2, st));
    const Slabs none{nullptr, 0, 0, 0};
    SET_TRY(lstm_pointwise(slabs_of(a[0]), own ? slabs_of(a[2]) : none, none, nullptr, 0, w->cl_x2h_b, w->cl_h2h_b, c2, c_new,
                           nullptr, ogate, M, D, st, RowGather(), gates_out));
    GemmProb e = slab_prob(s_n, M, D, M);
    e.add(c_new, D, w->cl_cnew_w, D, D);
    plan_ksplit(&e, 1, tgt);
    SET_TRY(gemm_group(&e, 1, st));
    return copy_gate_pointwise(slabs_of(e), w->cl_cnew_b, slabs_of(a[1]), w->cl_cmem_b, c_new, c_memory, ogate, c_out,
                               h_out, M, D, st, cg_out);
}
}  // namespace set
extern "C" {
static int copy_lstm_impl(const SetEditNetWeights* w, const float* x, int64_t ldx, int Kx, const float* h2,
                          const float* c2, const float* c_memory, float* h_out, float* c_out, float* gates_out,
                          float* cnew_out, float* cg_out, int M, int D, void* ws, size_t ws_bytes, void* stream) {
    if (!x || Kx <= 0) return SET_ERR_ARG;
    return copy_lstm_segs(w, 1, &x, &ldx, &Kx, h2, c2, c_memory, h_out, c_out, gates_out, cnew_out, cg_out, M, D, ws, ws_bytes,
                          stream);
}

int set_copy_lstm_f32(const SetEditNetWeights* w, const float* x, int64_t ldx, int Kx, const float* h2,
                      const float* c2, const float* c_memory, float* h_out, float* c_out, int M, int D, void* ws,
                      size_t ws_bytes, void* stream) {
    return copy_lstm_impl(w, x, ldx, Kx, h2, c2, c_memory, h_out, c_out, nullptr, nullptr, nullptr, M, D, ws, ws_bytes,
                          stream);
}

int set_copy_lstm_train_f32(const SetEditNetWeights* w, const float* x, int64_t ldx, int Kx, const float* h2,
                            const float* c2, const float* c_memory, float* h_out, float* c_out, float* gates,
                            float* c_new, float* cg, int M, int D, void* ws, size_t ws_bytes, void* stream) {
    if (!gates || !c_new || !cg) return SET_ERR_ARG;
    return copy_lstm_impl(w, x, ldx, Kx, h2, c2, c_memory, h_out, c_out, gates, c_new, cg, M, D, ws, ws_bytes, stream);
}

// ------------------------------------------------------------------------------- CaptionEncoderC
size_t set_caption_encoder_workspace_bytes(int B, int T, int D) {
    if (B <= 0 || T <= 0 || D <= 0) return 0;
    return fbytes((size_t)B * T * D) + fbytes((size_t)B * T * 4 * D) + 2 * fbytes((size_t)B * D) +
           fbytes(KS * (size_t)B * 4 * D) + fbytes(KS * (size_t)B * D) + 256;
}

int set_caption_encoder_f32(const SetEditNetWeights* w, const int64_t* seq, const int64_t* seq_len, float* H,
                            float* Mem, float* final_hidden, float* mask, int B, int T, int D, int V, void* ws,
                            size_t ws_bytes, void* stream) {
    if (!w || !seq || !seq_len || !H || !Mem || !final_hidden || !mask || B <= 0 || T <= 0 || D <= 0) return SET_ERR_ARG;
    if (D % GEMM_BK) return SET_ERR_UNSUPPORTED;
    if (!ws || !aligned16(ws) || ws_bytes < set_caption_encoder_workspace_bytes(B, T, D) - 256) return SET_ERR_WORKSPACE;
    Carver cv(ws);
    float* emb_seq = cv.take<float>((size_t)B * T * D);
    float* xg = cv.take<float>((size_t)B * T * 4 * D);
    float* enc_h = cv.take<float>((size_t)B * D);
    float* enc_c = cv.take<float>((size_t)B * D);
    float* s_enc = cv.take<float>(KS * (size_t)B * 4 * D);
    float* s_aff = cv.take<float>(KS * (size_t)B * D);
    return editnet_encoder(w, seq, seq_len, H, Mem, final_hidden, mask, B, T, D, V, emb_seq, xg, enc_h, enc_c, s_enc,
                           s_aff, (hipStream_t)stream);
}

// ---- multinomial sampling epilogue as an operator (editnet_rl.py:521-543 / dcnet_rl.py:320-340)
int set_sample_pick_f32(const float* logits, int64_t ld_logits, int B, int V, int t, int max_len, int64_t end_idx,
                        uint64_t seed, uint64_t offset, int64_t* seq, int64_t* it, int32_t* unfinished, int32_t* alive,
                        int64_t* raw_ids, float* lse, float* step_logp, void* stream) {
    if (!logits || !seq || !it || !unfinished || !alive || B <= 0 || V <= 0 || t < 0 || max_len <= 0 || ld_logits < V)
        return SET_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (t == 0) SET_TRY(set_tokens((long long*)it, 0, unfinished, alive, max_len + 2, B, st));
    Slabs lg{logits, 0, ld_logits, 1};
    return sample_pick(lg, nullptr, V, t, max_len, end_idx, (long long*)seq, nullptr, (long long*)it, unfinished, alive,
                       nullptr, nullptr, 4, B, seed, offset, (long long*)raw_ids, lse, step_logp, st);
}

int set_sample_logp_bwd_f32(const float* logits, int64_t ld_logits, const float* lse, const int64_t* raw_ids,
                            const float* g, float* dlogits, int64_t ld_dlogits, int B, int V, void* stream) {
    if (!logits || !lse || !raw_ids || !g || !dlogits || B <= 0 || V <= 0) return SET_ERR_ARG;
    return sample_logp_bwd(logits, ld_logits, lse, (const long long*)raw_ids, g, dlogits, ld_dlogits, B, V,
                           (hipStream_t)stream);
}

int set_philox4x32(uint32_t* out, int n, uint64_t seed, uint64_t offset, void* stream) {
    if (!out || n < 0) return SET_ERR_ARG;
    return philox_fill(out, n, seed, offset, (hipStream_t)stream);
}

}  // extern "C"
