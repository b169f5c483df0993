// Host-side sequencing of the DCNet (DAE) decode path + its C ABI.  Text-only denoising
// auto-encoder (reference dcnet.py:273-350, dcnet_rl.py:256-346): BiLSTM caption encoder
// (nn.LSTM on a packed batch), additive caption attention, two nn.LSTMCell updates, fc.
// Four grouped-GEMM launches per timestep:
//   A  gates1 = [emb|h2|h1] x [Wih_emb|Wih_h2|Whh] (+ hoisted final_hidden columns and biases)
//      + language_lstm Whh(h2)                                   -> LSTM pointwise (h1, c1)
//   B  att2_c = cap_decoder_att(h1), language_lstm Wih[:, :D](h1) -> caption attention (attend_cap)
//   C  language_lstm Wih[:, D:](attend_cap)                       -> LSTM pointwise (h2, c2)
//   D  fc(h2)
// Inference-time token table (w->tok_table, (V, 4D + 8C), set_dcnet_build_token_table): the contractions whose only input
// is a token — attention_lstm.W_ih[:, :E] relu(E[v]) and the two encoder input projections W_ih relu(E[v]) + b_ih — become
// row gathers; phase A then no longer sees the current token, so fc(h2_t) and phase A of timestep t+1 ride ONE launch
// (the same F/A merge as csrc/editnet.hip) and the prologue skips its 21.5-GFLOP input projection.
#include <cstring>
#include "set_common.h"

namespace set {

struct DcnetWs {
    float *enc, *final_hidden, *mask, *att1_c, *pre1;
    float *h1, *c1, *h2, *c2, *emb, *attend_cap, *alpha_c, *logits;
    long long* it;
    int *unfinished, *alive;
    float *sA0, *sA1, *sB0, *sB1, *sC0, *sF0;
    float *hf, *cf, *hb, *cb, *xg_f, *xg_b, *emb_seq, *s_ef, *s_eb, *s_cat, *s_pre;
    int* enc_order;                   // [perm (B) | nactive (T)] of the length-ordered persistent encoder
    char* enc_bar;                    // its barrier words
    float* pd_pc;                     // persistent small-batch decode (decode_persistent.hip): hoisted context products ...
    char* pd_x;                       // ... and its exchange region
    size_t bytes;
};

static int check_dims(const SetDcnetDims* d) {
    if (!d) return SET_ERR_ARG;
    if (d->B <= 0 || d->T <= 0 || d->D <= 0 || d->A <= 0 || d->C <= 0 || d->E <= 0 || d->V <= 0 || d->maxT <= 0)
        return SET_ERR_ARG;
    if ((d->D % GEMM_BK) || (d->A % GEMM_BK) || (d->C % GEMM_BK) || (d->E % GEMM_BK)) return SET_ERR_UNSUPPORTED;
    // attention_lstm input = [emb (E) | final_hidden (2C) | h2 (D)] must be 3E wide; language_lstm
    // input = [h1 (D) | attend_cap (2C)] must be 2E wide (dcnet.py:286-287,336-345)
    if (d->E + 2 * d->C + d->D != 3 * d->E || d->D + 2 * d->C != 2 * d->E) return SET_ERR_UNSUPPORTED;
    if (d->A > 512 || d->T > 256) return SET_ERR_UNSUPPORTED;
    return SET_OK;
}

static DcnetWs carve(const SetDcnetDims* d, void* base) {
    DcnetWs w;
    Carver c(base);
    const size_t B = d->B, T = d->T, D = d->D, A = d->A, C = d->C, E = d->E, V = d->V, KS = GEMM_MAX_KSPLIT;
    const size_t Vp = round_up(V, 64);
    w.enc = c.take<float>(B * T * 2 * C);
    w.final_hidden = c.take<float>(B * 2 * C);
    w.mask = c.take<float>(B * T);
    w.att1_c = c.take<float>(B * T * A);
    w.pre1 = c.take<float>(B * 4 * D);
    w.h1 = c.take<float>(B * D);
    w.c1 = c.take<float>(B * D);
    w.h2 = c.take<float>(B * D);
    w.c2 = c.take<float>(B * D);
    w.emb = c.take<float>(B * E);
    w.attend_cap = c.take<float>(B * 2 * C);
    w.alpha_c = c.take<float>(B * T);
    w.logits = c.take<float>(B * Vp);
    w.it = c.take<long long>(B);
    w.unfinished = c.take<int>(B);
    w.alive = c.take<int>(d->maxT + 2);
    w.sA0 = c.take<float>(KS * B * 4 * D);
    w.sA1 = c.take<float>(KS * B * 4 * D);
    w.sB0 = c.take<float>(KS * B * A);
    w.sB1 = c.take<float>(KS * B * 4 * D);
    w.sC0 = c.take<float>(KS * B * 4 * D);
    w.sF0 = c.take<float>(KS * B * Vp);
    w.hf = c.take<float>(B * C);
    w.cf = c.take<float>(B * C);
    w.hb = c.take<float>(B * C);
    w.cb = c.take<float>(B * C);
    w.xg_f = c.take<float>(B * T * 4 * C);
    w.xg_b = c.take<float>(B * T * 4 * C);
    w.emb_seq = c.take<float>(B * T * E);
    w.s_ef = c.take<float>(KS * B * 4 * C);
    w.s_eb = c.take<float>(KS * B * 4 * C);
    w.s_cat = c.take<float>(KS * B * 2 * C);
    w.s_pre = c.take<float>(KS * B * 4 * D);
    w.enc_order = c.take<int>(B + T);
    w.enc_bar = c.take<char>(persistent_encoder_bar_bytes());
    {
        const size_t pb = B <= (size_t)PDEC_MAXB ? B : 0;            // only small batches take the persistent decode
        w.pd_pc = c.take<float>(pb * T * 4 * D);
        w.pd_x = c.take<char>(dcnet_persistent_xbytes((int)B, (int)D, (int)A));
    }
    w.bytes = c.off;
    return w;
}

static int begin_impl(const SetDcnetWeights* w, const SetDcnetDims* d, const int64_t* prev, const int64_t* prevlen,
                      DcnetWs& ws, hipStream_t st) {
    const int B = d->B, T = d->T, D = d->D, A = d->A, C = d->C, E = d->E;
    const int tgt = gemm_target_wgs();
    // ---- CaptionEncoder (dcnet.py:220-243): packed BiLSTM == per-row masked recurrences
    const bool fused = (C % 128 == 0) && env_int("SET_NO_FUSED", 0) == 0;
    const bool tab = w->tok_table != nullptr && fused && (D % 64 == 0);
    const long long ldt = 4LL * D + 8LL * C;                 // token-table row stride
    if (!tab) SET_TRY(embed_relu(w->embed, prev, 1, ws.emb_seq, E, B * T, E, d->V, st));
    if (!tab) {
        GemmProb p[2];
        p[0] = direct_prob(ws.xg_f, 4 * C, B * T, 4 * C, w->enc_bih_f, SET_ACT_NONE);
        p[0].add(ws.emb_seq, E, w->enc_wih_f, E, E);
        p[1] = direct_prob(ws.xg_b, 4 * C, B * T, 4 * C, w->enc_bih_b, SET_ACT_NONE);
        p[1].add(ws.emb_seq, E, w->enc_wih_b, E, E);
        SET_TRY(gemm_group(p, 2, st, "gemm:enc x2h"));
    }
    SET_TRY(zero_f32(ws.enc, (size_t)B * T * 2 * C, st));
    {
        float* zp[4] = {ws.hf, ws.cf, ws.hb, ws.cb};
        const size_t zn[4] = {(size_t)B * C, (size_t)B * C, (size_t)B * C, (size_t)B * C};
        SET_TRY(zero_runs(zp, zn, 4, st));
    }
    float *hf_cur = ws.hf, *hf_nxt = ws.s_ef, *hb_cur = ws.hb, *hb_nxt = ws.s_eb;   // ping-pong (slab regions are free)
    // both directions as ONE weights-stationary launch with grid barriers (encoder_persistent.hip) when the shape allows it
    // (small batches by default): 2 T launches of ~14 us become T barrier steps of ~6 us
    bool persistent = fused && B <= 4096 && persistent_encoder_ok(B, C, T);
    if (persistent) {
        int* perm = ws.enc_order; int* nactive = ws.enc_order + B;
        SET_TRY(encoder_order(prevlen, B, T, perm, nactive, st));
        PEncDirHost dirs[2];
        if (tab) {
            dirs[0] = PEncDirHost{w->enc_whh_f, w->tok_table + 4 * D, w->enc_bhh_f, hf_cur, hf_nxt, 0, 0};
            dirs[1] = PEncDirHost{w->enc_whh_b, w->tok_table + 4 * D + 4 * C, w->enc_bhh_b, hb_cur, hb_nxt, C, 1};
        } else {
            dirs[0] = PEncDirHost{w->enc_whh_f, ws.xg_f, w->enc_bhh_f, hf_cur, hf_nxt, 0, 0};
            dirs[1] = PEncDirHost{w->enc_whh_b, ws.xg_b, w->enc_bhh_b, hb_cur, hb_nxt, C, 1};
        }
        const int rc = tab ? persistent_encoder_dirs(dirs, 2, ldt, 0, prevlen, prev, T, d->V, ws.enc, nullptr, (long long)T * 2 * C,
                                                     2 * C, perm, nactive, ws.enc_bar, B, C, T, st)
                           : persistent_encoder_dirs(dirs, 2, (long long)T * 4 * C, 4 * C, prevlen, nullptr, 0, 0, ws.enc, nullptr,
                                                     (long long)T * 2 * C, 2 * C, perm, nactive, ws.enc_bar, B, C, T, st);
        if (rc == SET_OK) {
            if (T & 1) { float* tmp = hf_cur; hf_cur = hf_nxt; hf_nxt = tmp; tmp = hb_cur; hb_cur = hb_nxt; hb_nxt = tmp; }
        } else if (rc == SET_ERR_UNSUPPORTED) persistent = false;
        else return rc;
    }
    for (int t = 0; t < T && !persistent; ++t) {
        if (fused && tab) {                   // x W_ih^T + b_ih of every word is a row of the token table
            SET_TRY(fused_encoder_step(hf_cur, hf_nxt, ws.cf, w->enc_whh_f, w->tok_table + 4 * D, ldt, 0, w->enc_bhh_f,
                                       prevlen, t, 0, ws.enc, nullptr, (long long)T * 2 * C, 2 * C, 0, B, C, st, prev, T, d->V));
            SET_TRY(fused_encoder_step(hb_cur, hb_nxt, ws.cb, w->enc_whh_b, w->tok_table + 4 * D + 4 * C, ldt, 0,
                                       w->enc_bhh_b, prevlen, t, 1, ws.enc, nullptr, (long long)T * 2 * C, 2 * C, C, B, C, st,
                                       prev, T, d->V));
            float* tmp = hf_cur; hf_cur = hf_nxt; hf_nxt = tmp;
            tmp = hb_cur; hb_cur = hb_nxt; hb_nxt = tmp;
            continue;
        }
        if (fused) {
            SET_TRY(fused_encoder_step(hf_cur, hf_nxt, ws.cf, w->enc_whh_f, ws.xg_f, (long long)T * 4 * C, 4 * C,
                                       w->enc_bhh_f, prevlen, t, 0, ws.enc, nullptr, (long long)T * 2 * C, 2 * C, 0, B, C,
                                       st));
            SET_TRY(fused_encoder_step(hb_cur, hb_nxt, ws.cb, w->enc_whh_b, ws.xg_b, (long long)T * 4 * C, 4 * C,
                                       w->enc_bhh_b, prevlen, t, 1, ws.enc, nullptr, (long long)T * 2 * C, 2 * C, C, B, C,
                                       st));
            float* tmp = hf_cur; hf_cur = hf_nxt; hf_nxt = tmp;
            tmp = hb_cur; hb_cur = hb_nxt; hb_nxt = tmp;
            continue;
        }
        GemmProb p[2];
        p[0] = slab_prob(ws.s_ef, B, 4 * C, B);
        p[0].add(ws.hf, C, w->enc_whh_f, C, C);
        p[1] = slab_prob(ws.s_eb, B, 4 * C, B);
        p[1].add(ws.hb, C, w->enc_whh_b, C, C);
        plan_ksplit(p, 2, tgt);
        if (t > 0) SET_TRY(gemm_group(p, 2, st, "gemm:enc h2h"));
        Slabs f = slabs_of(p[0]), b = slabs_of(p[1]);
        if (t == 0) { f.n = 0; b.n = 0; }
        SET_TRY(encoder_pointwise(f, ws.xg_f, (long long)T * 4 * C, 4 * C, t, prevlen, 0, ws.hf, ws.cf, ws.enc, nullptr,
                                  (long long)T * 2 * C, 2 * C, 0, B, C, w->enc_bhh_f, st));
        SET_TRY(encoder_pointwise(b, ws.xg_b, (long long)T * 4 * C, 4 * C, t, prevlen, 1, ws.hb, ws.cb, ws.enc, nullptr,
                                  (long long)T * 2 * C, 2 * C, C, B, C, w->enc_bhh_b, st));
    }
    {
        GemmProb p = slab_prob(ws.s_cat, B, 2 * C, B);                 // tanh(concat([h_fwd, h_bwd])) dcnet.py:241-242
        p.add(hf_cur, C, w->enc_cat_w, 2 * C, C);
        p.add(hb_cur, C, w->enc_cat_w + C, 2 * C, C);
        plan_ksplit(&p, 1, tgt);
        SET_TRY(gemm_group(&p, 1, st));
        SET_TRY(reduce_bias_act(slabs_of(p), w->enc_cat_b, nullptr, ws.final_hidden, 2 * C, B, 2 * C, SET_ACT_TANH, st));
    }
    SET_TRY(rowsum_mask(ws.enc, 2 * C, B * T, 2 * C, ws.mask, st));                    // dcnet.py:239
    {
        // small batches (persistent decode, decode_persistent.hip): the context half of language_lstm's input product is
        // linear in the attention weights — Pc = enc W_ih[:, D:]^T (B, T, 4D) rides this launch (same operand enc)
        GemmProb p[2];
        p[0] = direct_prob(ws.att1_c, A, B * T, A, w->ca_feat_b, SET_ACT_NONE);            // dcnet.py:261
        p[0].add(ws.enc, 2 * C, w->ca_feat_w, 2 * C, 2 * C);
        const bool pc = dcnet_persistent_ok(d, 1);
        if (pc) {
            p[1] = direct_prob(ws.pd_pc, 4LL * D, B * T, 4 * D, nullptr, SET_ACT_NONE);
            p[1].add(ws.enc, 2 * C, w->ll_wih + D, 2 * E, 2 * C);
        }
        SET_TRY(gemm_group(p, pc ? 2 : 1, st, "gemm:pro cap_features_att"));
    }
    {
        GemmProb p = slab_prob(ws.s_pre, B, 4 * D, B);                 // final_hidden columns of attention_lstm
        p.add(ws.final_hidden, 2 * C, w->al_wih + E, 3 * E, 2 * C);
        plan_ksplit(&p, 1, tgt);
        SET_TRY(gemm_group(&p, 1, st));
        SET_TRY(reduce_bias_act(slabs_of(p), w->al_bih, w->al_bhh, ws.pre1, 4 * D, B, 4 * D, SET_ACT_NONE, st));
    }
    {
        float* zp[4] = {ws.h1, ws.c1, ws.h2, ws.c2};
        const size_t zn[4] = {(size_t)B * D, (size_t)B * D, (size_t)B * D, (size_t)B * D};
        SET_TRY(zero_runs(zp, zn, 4, st));
    }
    return SET_OK;
}

static bool table_active(const SetDcnetWeights* w, const SetDcnetDims* d) {
    return w->tok_table != nullptr && (d->C % 128 == 0) && (d->D % 64 == 0) && env_int("SET_NO_FUSED", 0) == 0;
}

static void build_phase_a(const SetDcnetWeights* w, const SetDcnetDims* d, DcnetWs& ws, int bt, bool tab, GemmProb a[2]) {
    const int B = d->B, D = d->D, C = d->C, E = d->E;
    a[0] = slab_prob(ws.sA0, bt, 4 * D, B);
    if (!tab) a[0].add(ws.emb, E, w->al_wih, 3 * E, E);
    a[0].add(ws.h2, D, w->al_wih + E + 2 * C, 3 * E, D);
    a[0].add(ws.h1, D, w->al_whh, D, D);
    a[1] = slab_prob(ws.sA1, bt, 4 * D, B);
    a[1].add(ws.h2, D, w->ll_whh, D, D);
}

// tok_ids / tok_stride: the tokens of THIS timestep (table gather).  a_pre / a_next / bt_next / logits_biased: the F/A
// merge, exactly as in csrc/editnet.hip step_impl.
static int step_impl(const SetDcnetWeights* w, const SetDcnetDims* d, int bt, DcnetWs& ws, float* dst, long long ld_dst,
                     Slabs* logits_out, hipStream_t st, const long long* tok_ids = nullptr, long long tok_stride = 1,
                     const GemmProb* a_pre = nullptr, GemmProb* a_next = nullptr, bool* logits_biased = nullptr,
                     int bt_next = -1, bool a_done = false) {
    const int B = d->B, T = d->T, D = d->D, A = d->A, C = d->C, E = d->E, V = d->V;
    const int tgt = gemm_target_wgs();
    const Slabs none{nullptr, 0, 0, 0};
    const bool tab = table_active(w, d) && tok_ids != nullptr;
    RowGather g_gates;
    if (tab) g_gates = RowGather{w->tok_table, tok_ids, tok_stride, 4LL * D + 8LL * C, 0, V};
    GemmProb a[2];
    if (a_pre) {
        a[0] = a_pre[0]; a[1] = a_pre[1];
    } else {
        build_phase_a(w, d, ws, bt, tab, a);
        plan_ksplit(a, 2, tgt);
        SET_TRY(gemm_group(a, 2, st, "gemm:A gates1+h2h"));
    }
    if (!(a_done && a_pre && tab))       // a_done: the previous pick finished this cell (LstmTail, as in csrc/editnet.hip)
        SET_TRY(lstm_pointwise(slabs_of(a[0]), none, none, ws.pre1, 4 * D, nullptr, nullptr, ws.c1, ws.c1, ws.h1, nullptr,
                               bt, D, st, g_gates));
    GemmProb b[2];
    b[0] = slab_prob(ws.sB0, bt, A, B);
    b[0].add(ws.h1, D, w->ca_dec_w, D, D);
    b[1] = slab_prob(ws.sB1, bt, 4 * D, B);
    b[1].add(ws.h1, D, w->ll_wih, 2 * E, D);
    plan_ksplit(b, 2, tgt);
    SET_TRY(gemm_group(b, 2, st, "gemm:B att2_c,ll_h1"));
    SET_TRY(caption_attention(ws.att1_c, slabs_of(b[0]), w->ca_dec_b, w->ca_full_w, w->ca_full_b, ws.mask, ws.enc,
                              nullptr, ws.attend_cap, nullptr, ws.alpha_c, bt, T, 2 * C, A, st));
    GemmProb c = slab_prob(ws.sC0, bt, 4 * D, B);
    c.add(ws.attend_cap, 2 * C, w->ll_wih + D, 2 * E, 2 * C);
    plan_ksplit(&c, 1, tgt);
    SET_TRY(gemm_group(&c, 1, st, "gemm:C ll_ctx"));
    SET_TRY(lstm_pointwise(slabs_of(a[1]), slabs_of(b[1]), slabs_of(c), nullptr, 0, w->ll_bih, w->ll_bhh, ws.c2, ws.c2,
                           ws.h2, nullptr, bt, D, st));
    const long long Vp = (long long)round_up((size_t)V, 64);
    GemmProb f = slab_prob(ws.sF0, bt, V, B);
    f.ldc = Vp;
    f.slab_stride = (long long)B * Vp;
    f.add(ws.h2, D, w->fc_w, D, D);
    if (logits_biased) *logits_biased = false;
    if (a_next && tab) {
        GemmProb fa[3];
        fa[0] = f;
        build_phase_a(w, d, ws, bt_next > 0 ? bt_next : bt, true, fa + 1);
        plan_ksplit(fa, 3, tgt);
        if (fa[0].ksplit == 1) {             // unsplit fc: write the scores once, bias fused (straight into dst when given)
            fa[0].C = dst ? dst : ws.logits; fa[0].slab_stride = 0; fa[0].bias = w->fc_b;
            fa[0].ldc = dst ? ld_dst : Vp;
            if (logits_biased) *logits_biased = true;
        }
        SET_TRY(gemm_group(fa, 3, st, "gemm:F fc + next A"));
        if (dst && fa[0].ksplit > 1)
            SET_TRY(reduce_bias_act(slabs_of(fa[0]), w->fc_b, nullptr, dst, ld_dst, bt, V, SET_ACT_NONE, st));
        a_next[0] = fa[1]; a_next[1] = fa[2];
        if (logits_out) *logits_out = slabs_of(fa[0]);
        return SET_OK;
    }
    plan_ksplit(&f, 1, tgt);
    if (dst && f.ksplit == 1) {
        f.C = dst; f.ldc = ld_dst; f.bias = w->fc_b; f.slab_stride = 0;
        SET_TRY(gemm_group(&f, 1, st, "gemm:F fc"));
    } else {
        SET_TRY(gemm_group(&f, 1, st, "gemm:F fc"));
        if (dst) SET_TRY(reduce_bias_act(slabs_of(f), w->fc_b, nullptr, dst, ld_dst, bt, V, SET_ACT_NONE, st));
    }
    if (logits_out) *logits_out = slabs_of(f);
    return SET_OK;
}

}  // namespace set

using namespace set;

extern "C" {

size_t set_dcnet_workspace_bytes(const SetDcnetDims* d) {
    if (check_dims(d) != SET_OK) return 0;
    return carve(d, nullptr).bytes + 256;
}

static int prep(const SetDcnetDims* d, void* ws, size_t ws_bytes, DcnetWs* out) {
    SET_TRY(check_dims(d));
    if (!ws || !aligned16(ws)) return SET_ERR_ARG;
    *out = carve(d, ws);
    if (out->bytes > ws_bytes) return SET_ERR_WORKSPACE;
    return SET_OK;
}

int set_dcnet_begin(const SetDcnetWeights* w, const SetDcnetDims* d, const int64_t* prev, const int64_t* prevlen,
                    void* ws, size_t ws_bytes, void* stream) {
    if (!w || !prev || !prevlen) return SET_ERR_ARG;
    DcnetWs W;
    SET_TRY(prep(d, ws, ws_bytes, &W));
    return begin_impl(w, d, prev, prevlen, W, (hipStream_t)stream);
}

int set_dcnet_step(const SetDcnetWeights* w, const SetDcnetDims* d, const int64_t* tokens, int64_t tokens_stride,
                   int bt, float* logits, int64_t ld_logits, void* ws, size_t ws_bytes, void* stream) {
    if (!w || !logits) return SET_ERR_ARG;
    DcnetWs W;
    SET_TRY(prep(d, ws, ws_bytes, &W));
    if (bt <= 0 || bt > d->B || ld_logits < d->V) return SET_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (tokens) SET_TRY(embed_relu(w->embed, tokens, tokens_stride, W.emb, d->E, bt, d->E, d->V, st));
    return step_impl(w, d, bt, W, logits, ld_logits, nullptr, st, tokens ? (const long long*)tokens : W.it,
                     tokens ? (long long)tokens_stride : 1);
}

int set_dcnet_greedy_pick(const SetDcnetWeights* w, const SetDcnetDims* d, const float* logits, int64_t ld_logits,
                          int t, int64_t end_idx, int64_t* seq, float* seq_logp, int max_len, void* ws,
                          size_t ws_bytes, void* stream) {
    if (!w || !logits || !seq || !seq_logp || t < 0) return SET_ERR_ARG;
    DcnetWs W;
    SET_TRY(prep(d, ws, ws_bytes, &W));
    if (t > d->maxT) return SET_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (t == 0) SET_TRY(set_tokens(W.it, 0, W.unfinished, W.alive, d->maxT + 2, d->B, st));
    Slabs lg{logits, 0, ld_logits, 1};
    return greedy_pick(lg, nullptr, d->V, t, max_len, end_idx, (long long*)seq, seq_logp, W.it, W.unfinished, W.alive,
                       w->embed, W.emb, d->E, d->B, st);
}

// free-running decode (dcnet_rl.py:286-346): sample == 0 greedy, 1 multinomial
static int dcnet_rollout(const SetDcnetWeights* w, const SetDcnetDims* d, const int64_t* prev, const int64_t* prevlen,
                         int64_t start_idx, int64_t end_idx, int max_len, int sample, uint64_t seed, uint64_t offset,
                         int64_t* seq, float* seq_logp, void* ws, size_t ws_bytes, void* stream) {
    if (!w || !prev || !prevlen || !seq || !seq_logp || max_len <= 0) return SET_ERR_ARG;
    DcnetWs W;
    SET_TRY(prep(d, ws, ws_bytes, &W));
    if (max_len > d->maxT || start_idx < 0 || start_idx >= d->V) return SET_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int B = d->B;
    SET_TRY(begin_impl(w, d, prev, prevlen, W, st));
    SET_HIP_TRY(hipMemsetAsync(seq, 0, sizeof(int64_t) * B * max_len, st));
    SET_HIP_TRY(hipMemsetAsync(seq_logp, 0, sizeof(float) * B * max_len, st));
    SET_TRY(set_tokens(W.it, start_idx, W.unfinished, W.alive, d->maxT + 2, B, st));
    const bool emb_needed = !table_active(w, d);         // with the token table no consumer reads relu(E[it])
    if (emb_needed) SET_TRY(embed_relu(w->embed, (const int64_t*)W.it, 1, W.emb, d->E, B, d->E, d->V, st));
    static const int fa_merge = env_int("SET_FA_MERGE", 1);
    const bool merge = fa_merge && !emb_needed;
    static const int pick_tail = env_int("SET_PICK_TAIL", 1);    // the pick finishes the next attention-LSTM cell (LstmTail)
    GemmProb a_cur[2], a_nxt[2];
    bool have_a = false, a_done = false;
    // row gate of the loop (set_common.h RowGate; same policy as the EditNet loop, editnet.hip rollout)
    static const int loop_gate = env_int("SET_LOOP_GATE", 1);
    // small batches, greedy: the whole loop as ONE persistent launch (decode_persistent.hip).  The context half of
    // language_lstm's input product is linear in the attention weights: Pc = enc W_ih[:, D:]^T is computed here once
    if (!sample && !emb_needed && !g_row_limit && dcnet_persistent_ok(d, max_len)) {
        const int rc = dcnet_persistent_greedy(w, d, W.pre1, W.att1_c, W.mask, W.pd_pc, W.pd_x, W.it, W.unfinished, W.alive,
                                               start_idx, end_idx, max_len, (long long*)seq, seq_logp, st);
        if (rc != SET_ERR_UNSUPPORTED) return rc;
    }
    for (int t = 0; t <= max_len; ++t) {                                 // dcnet_rl.py:305,315-316
        Slabs lg;
        bool biased = false;
        const bool next_a = merge && t < max_len;
        RowGate gate;
        if (loop_gate && t > 0) gate.alive_prev = W.alive + (t - 1);
        RowGateScope gate_scope(gate);
        SET_TRY(step_impl(w, d, B, W, nullptr, 0, &lg, st, W.it, 1, have_a ? a_cur : nullptr, next_a ? a_nxt : nullptr,
                          &biased, -1, a_done));
        have_a = next_a;
        if (next_a) { a_cur[0] = a_nxt[0]; a_cur[1] = a_nxt[1]; }
        if (t == max_len) break;
        const float* pick_bias = biased ? nullptr : w->fc_b;
        LstmTail tail;
        a_done = next_a && pick_tail;
        if (a_done) {
            tail.g0 = slabs_of(a_nxt[0]);
            tail.pre = W.pre1; tail.ldpre = 4LL * d->D;
            tail.tab = w->tok_table; tail.ld_tab = 4LL * d->D + 8LL * d->C; tail.col0 = 0; tail.nrows = d->V;
            tail.c_in = W.c1; tail.c_out = W.c1; tail.h_out = W.h1; tail.D = d->D;
        }
        if (sample)
            SET_TRY(sample_pick(lg, pick_bias, d->V, t, max_len, end_idx, (long long*)seq, seq_logp, W.it, W.unfinished,
                                W.alive, emb_needed ? w->embed : nullptr, W.emb, d->E, B, seed, offset, nullptr, nullptr,
                                nullptr, st, a_done ? &tail : nullptr));
        else
            SET_TRY(greedy_pick(lg, pick_bias, d->V, t, max_len, end_idx, (long long*)seq, seq_logp, W.it, W.unfinished,
                                W.alive, emb_needed ? w->embed : nullptr, W.emb, d->E, B, st, a_done ? &tail : nullptr));
    }
    return SET_OK;
}

int set_dcnet_greedy(const SetDcnetWeights* w, const SetDcnetDims* d, const int64_t* prev, const int64_t* prevlen,
                     int64_t start_idx, int64_t end_idx, int max_len, int64_t* seq, float* seq_logp, void* ws,
                     size_t ws_bytes, void* stream) {
    return dcnet_rollout(w, d, prev, prevlen, start_idx, end_idx, max_len, 0, 0, 0, seq, seq_logp, ws, ws_bytes, stream);
}

int set_dcnet_sample(const SetDcnetWeights* w, const SetDcnetDims* d, const int64_t* prev, const int64_t* prevlen,
                     int64_t start_idx, int64_t end_idx, int max_len, uint64_t seed, uint64_t offset, int64_t* seq,
                     float* seq_logp, void* ws, size_t ws_bytes, void* stream) {
    return dcnet_rollout(w, d, prev, prevlen, start_idx, end_idx, max_len, 1, seed, offset, seq, seq_logp, ws, ws_bytes,
                         stream);
}

int set_dcnet_xe_forward(const SetDcnetWeights* w, const SetDcnetDims* d, const int64_t* caps, int64_t caps_stride,
                         const int* host_decode_lengths, const int64_t* prev, const int64_t* prevlen,
                         float* predictions, void* ws, size_t ws_bytes, void* stream) {
    if (!w || !caps || !host_decode_lengths || !prev || !prevlen || !predictions) return SET_ERR_ARG;
    DcnetWs W;
    SET_TRY(prep(d, ws, ws_bytes, &W));
    hipStream_t st = (hipStream_t)stream;
    const int B = d->B, V = d->V, maxT = d->maxT;
    for (int b = 0; b < B; ++b) {
        if (host_decode_lengths[b] < 0 || host_decode_lengths[b] > maxT) return SET_ERR_ARG;
        if (b && host_decode_lengths[b] > host_decode_lengths[b - 1]) return SET_ERR_ARG;
    }
    if (caps_stride < maxT) return SET_ERR_ARG;
    SET_TRY(begin_impl(w, d, prev, prevlen, W, st));
    SET_HIP_TRY(hipMemsetAsync(predictions, 0, sizeof(float) * (size_t)B * maxT * V, st));
    const bool emb_needed = !table_active(w, d);
    // small batches: the teacher-forced loop as ONE persistent launch too (decode_persistent.hip, words from caps, scores out)
    if (!emb_needed && dcnet_persistent_ok(d, maxT)) {
        const PDecTeacher teach{caps, caps_stride, predictions, host_decode_lengths};
        const int rc = dcnet_persistent_greedy(w, d, W.pre1, W.att1_c, W.mask, W.pd_pc, W.pd_x, W.it, W.unfinished, W.alive, 0, -1,
                                               maxT, nullptr, nullptr, st, &teach);
        if (rc != SET_ERR_UNSUPPORTED) return rc;
    }
    static const int fa_merge = env_int("SET_FA_MERGE", 1);
    const bool merge = fa_merge && !emb_needed;
    auto rows_at = [&](int t) { int n = 0; while (n < B && host_decode_lengths[n] > t) ++n; return n; };   // dcnet.py:334
    GemmProb a_cur[2], a_nxt[2];
    bool have_a = false;
    for (int t = 0; t < maxT; ++t) {
        const int bt = rows_at(t);
        if (bt == 0) break;
        const int btn = t + 1 < maxT ? rows_at(t + 1) : 0;
        if (emb_needed) SET_TRY(embed_relu(w->embed, caps + t, caps_stride, W.emb, d->E, bt, d->E, V, st));
        const bool next_a = merge && btn > 0;
        SET_TRY(step_impl(w, d, bt, W, predictions + (size_t)t * V, (long long)maxT * V, nullptr, st,
                          (const long long*)(caps + t), caps_stride, have_a ? a_cur : nullptr, next_a ? a_nxt : nullptr, nullptr,
                          btn));
        have_a = next_a;
        if (next_a) { a_cur[0] = a_nxt[0]; a_cur[1] = a_nxt[1]; }
    }
    return SET_OK;
}

size_t set_dcnet_token_table_bytes(const SetDcnetDims* d) {
    if (check_dims(d) != SET_OK) return 0;
    return sizeof(float) * (size_t)d->V * (4 * (size_t)d->D + 8 * (size_t)d->C);
}

size_t set_dcnet_token_table_workspace_bytes(const SetDcnetDims* d) {
    if (check_dims(d) != SET_OK) return 0;
    return round_up(sizeof(float) * (size_t)d->V * d->E, 256) + round_up(sizeof(long long) * (size_t)d->V, 256) + 256;
}

// table[v] = [ W_ih^att[:, :E] relu(E[v]) | W_ih^enc,fwd relu(E[v]) + b_ih^fwd | W_ih^enc,bwd relu(E[v]) + b_ih^bwd ]
int set_dcnet_build_token_table(const SetDcnetWeights* w, const SetDcnetDims* d, float* table, void* ws, size_t ws_bytes,
                                void* stream) {
    if (!w || !table || !ws) return SET_ERR_ARG;
    SET_TRY(check_dims(d));
    if (!aligned16(ws) || !aligned16(table)) return SET_ERR_ARG;
    if (ws_bytes < set_dcnet_token_table_workspace_bytes(d) - 256) return SET_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int V = d->V, D = d->D, C = d->C, E = d->E;
    const long long ldt = 4LL * D + 8LL * C;
    Carver cv(ws);
    float* remb = cv.take<float>((size_t)V * E);
    long long* ids = cv.take<long long>((size_t)V);
    SET_TRY(iota_i64(ids, V, st));
    SET_TRY(embed_relu(w->embed, (const int64_t*)ids, 1, remb, E, V, E, V, st));
    GemmProb p[3];
    p[0] = direct_prob(table, ldt, V, 4 * D, nullptr, SET_ACT_NONE);
    p[0].add(remb, E, w->al_wih, 3LL * E, E);
    p[1] = direct_prob(table + 4 * D, ldt, V, 4 * C, w->enc_bih_f, SET_ACT_NONE);
    p[1].add(remb, E, w->enc_wih_f, E, E);
    p[2] = direct_prob(table + 4 * D + 4 * C, ldt, V, 4 * C, w->enc_bih_b, SET_ACT_NONE);
    p[2].add(remb, E, w->enc_wih_b, E, E);
    return gemm_group(p, 3, st, "gemm:token table");
}

void* set_dcnet_ws_tensor(const SetDcnetDims* d, void* ws, const char* name) {
    if (check_dims(d) != SET_OK || !ws || !name) return nullptr;
    DcnetWs W = carve(d, ws);
    struct { const char* n; void* p; } tab[] = {
        {"enc", W.enc}, {"final_hidden", W.final_hidden}, {"mask", W.mask}, {"att1_c", W.att1_c}, {"pre1", W.pre1},
        {"h1", W.h1}, {"c1", W.c1}, {"h2", W.h2}, {"c2", W.c2}, {"emb", W.emb}, {"attend_cap", W.attend_cap},
        {"alpha_c", W.alpha_c}, {"logits", W.logits}, {"it", W.it}, {"unfinished", W.unfinished}, {"alive", W.alive}};
    for (auto& e : tab)
        if (!strcmp(e.n, name)) return e.p;
    return nullptr;
}

}  // extern "C"
