// Host-side sequencing of the EditNet decode path over the HIP kernels + the C ABI for it.
// Everything is enqueued on the caller's stream; there is no host synchronisation anywhere
// (the reference synchronises every timestep at editnet_rl.py:546 and :506).
//
// Per timestep (eval mode) the work is a chain of six grouped-GEMM launches separated by five
// small pointwise / attention launches (dependencies of editnet.py:527-545):
//   A  gates1 = [emb|h2|h1] x [Wih_emb|Wih_h2|Whh]      + copy-LSTM h2h(h2)          (needs prev state)
//      -> LSTM pointwise (h1, c1)              [final_hidden / image_mean columns + biases hoisted: pre1]
//   B  att2_c, att2, tc_affine, context_gate[word,h1], copy-LSTM x2h[:, h1]            (needs h1)
//      -> caption attention + select, visual attention
//   C  context_gate[ctx], sc_affine(ctx), gate_cmem(sel)                                (needs ctx, sel)
//      -> context gating pointwise (attend_cap)
//   D  copy-LSTM x2h[:, attend_cap | attend_img]                                        (needs both contexts)
//      -> LSTM pointwise (c_new, o-gate)
//   E  gate_cnew(c_new)  -> copy-gate pointwise (h2, c2)
//   F  fc(h2)            -> greedy pick / predictions store
#include <cstdlib>
#include <cstring>
#include "set_common.h"

namespace set {

int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return v && *v ? atoi(v) : dflt;
}
thread_local RowGate g_row_gate;
thread_local const int* g_row_limit = nullptr;
int gemm_target_wgs() {
    static int v = env_int("SET_GEMM_TARGET_WGS", 512);
    return v;
}

struct EditNetWs {
    // per-sequence invariants (prologue)
    float *H, *Mem, *final_hidden, *mask, *att1, *att1_c, *image_mean, *pre1, *rmask;
    float *cap_proj, *mem_proj;       // hoisted [context_gate.W[:,2D:] H | sc_affine.W H] (B,T,2D) and gate_cmem.W Mem (B,T,D)
    // recurrent state + per-step activations
    float *h1, *c1, *h2, *c2, *emb, *ctx_cap, *attend_cap, *attend_img, *sel, *cmem_pre, *c_new, *ogate, *alpha_c, *alpha;
    float* logits;
    long long* it;
    int *unfinished, *alive;
    // split-K slabs, one region per GEMM of the step
    float *sA0, *sA1, *sB0, *sB1, *sB2, *sB3, *sB4, *sC0, *sC1, *sC2, *sD0, *sE0, *sF0;
    // prologue scratch
    float *enc_h, *enc_c, *xg, *emb_seq, *fe, *s_enc, *s_aff, *s_pre;
    int* enc_order;                   // [perm (B) | nactive (T)] of the length-ordered encoder
    char* enc_bar;                    // barrier words of the persistent encoder
    float* pd_pv;                     // persistent small-batch decode (decode_persistent_wide.hip): X x2h[:, 2D:]^T (B, R, 4D) ...
    char* pd_x;                       // ... and its exchange region
    size_t bytes;
};

static int check_dims(const SetEditNetDims* d) {
    if (!d) return SET_ERR_ARG;
    if (d->B <= 0 || d->T <= 0 || d->R <= 0 || d->F <= 0 || d->D <= 0 || d->A <= 0 || d->V <= 0 || d->maxT <= 0)
        return SET_ERR_ARG;
    if ((d->D % GEMM_BK) || (d->A % GEMM_BK) || (d->F % GEMM_BK)) return SET_ERR_UNSUPPORTED;
    if (d->A > 512 || d->T > 256 || d->R > 256) return SET_ERR_UNSUPPORTED;
    return SET_OK;
}

static EditNetWs carve(const SetEditNetDims* d, void* base) {
    EditNetWs w;
    Carver c(base);
    const size_t B = d->B, T = d->T, R = d->R, F = d->F, D = d->D, A = d->A, V = d->V;
    const size_t KS = GEMM_MAX_KSPLIT;
    const size_t Vp = round_up(V, 64);
    w.H = c.take<float>(B * T * D);
    w.Mem = c.take<float>(B * T * D);
    w.final_hidden = c.take<float>(B * D);
    w.mask = c.take<float>(B * T);
    w.att1 = c.take<float>(B * R * A);
    w.att1_c = c.take<float>(B * T * A);
    w.image_mean = c.take<float>(B * F);
    w.pre1 = c.take<float>(B * 4 * D);
    w.rmask = c.take<float>(B * R);
    w.cap_proj = c.take<float>(B * T * 2 * D);
    w.mem_proj = c.take<float>(B * T * D);
    w.h1 = c.take<float>(B * D);
    w.c1 = c.take<float>(B * D);
    w.h2 = c.take<float>(B * D);
    w.c2 = c.take<float>(B * D);
    w.emb = c.take<float>(B * D);
    w.ctx_cap = c.take<float>(B * D);
    w.attend_cap = c.take<float>(B * D);
    w.attend_img = c.take<float>(B * F);
    w.sel = c.take<float>(B * D);
    w.cmem_pre = c.take<float>(B * D);
    w.c_new = c.take<float>(B * D);
    w.ogate = c.take<float>(B * D);
    w.alpha_c = c.take<float>(B * T);
    w.alpha = c.take<float>(B * R);
    w.logits = c.take<float>(B * Vp);
    w.it = c.take<long long>(B);
    w.unfinished = c.take<int>(B);
    w.alive = c.take<int>(d->maxT + 2);
    w.sA0 = c.take<float>(KS * B * 4 * D);
    w.sA1 = c.take<float>(KS * B * 4 * D);
    w.sB0 = c.take<float>(KS * B * A);
    w.sB1 = c.take<float>(KS * B * A);
    w.sB2 = c.take<float>(KS * B * D);
    w.sB3 = c.take<float>(KS * B * D);
    w.sB4 = c.take<float>(KS * B * 4 * D);
    w.sC0 = c.take<float>(KS * B * D);
    w.sC1 = c.take<float>(KS * B * D);
    w.sC2 = c.take<float>(KS * B * D);
    w.sD0 = c.take<float>(KS * B * 4 * D);
    w.sE0 = c.take<float>(KS * B * D);
    w.sF0 = c.take<float>(KS * B * Vp);
    w.enc_h = c.take<float>(B * D);
    w.enc_c = c.take<float>(B * D);
    w.xg = c.take<float>(B * T * 4 * D);
    const size_t seq_rows = B * (T > (size_t)d->maxT ? T : (size_t)d->maxT);
    w.emb_seq = c.take<float>(seq_rows * D);
    w.fe = c.take<float>(B * R * D);
    w.s_enc = c.take<float>(KS * B * 4 * D);
    w.s_aff = c.take<float>(KS * B * D);
    w.s_pre = c.take<float>(KS * B * 4 * D);
    w.enc_order = c.take<int>(B + T);
    w.enc_bar = c.take<char>(persistent_encoder_bar_bytes());
    {
        const size_t pb = B <= (size_t)PDW_MAXB ? B : 0;             // only small batches take the persistent decode
        w.pd_pv = c.take<float>(pb * R * 4 * D);
        w.pd_x = c.take<char>(editnet_persistent_xbytes((int)B, (int)D, (int)A));
    }
    w.bytes = c.off;
    return w;
}

// a GEMM problem whose output goes to a slab region sized for GEMM_MAX_KSPLIT slabs of (B,N)
GemmProb slab_prob(float* slab, int M, int N, int Bmax) {
    GemmProb p;
    p.C = slab;
    p.ldc = N;
    p.slab_stride = (long long)Bmax * N;
    p.M = M;
    p.N = N;
    return p;
}
// a GEMM problem that writes its (M,N) result in place with a fused bias/activation (no K split)
GemmProb direct_prob(float* out, long long ldo, int M, int N, const float* bias, int act) {
    GemmProb p;
    p.C = out;
    p.ldc = ldo;
    p.slab_stride = 0;
    p.M = M;
    p.N = N;
    p.bias = bias;
    p.act = act;
    p.max_ksplit = 1;
    p.ksplit = 1;
    return p;
}

// zero several fp32 buffers with as few fill launches as their placement allows (buffers carved back to back from one
// workspace need ONE fill; a fill is a ~5 us launch on the chain and the decode prologue had ten of them)
int zero_runs(float* const* p, const size_t* n, int cnt, hipStream_t st) {
    int i = 0;
    while (i < cnt) {
        float* base = p[i];
        size_t len = n[i];
        int j = i + 1;
        while (j < cnt && p[j] == base + len) { len += n[j]; ++j; }
        SET_TRY(zero_f32(base, len, st));
        i = j;
    }
    return SET_OK;
}

// CaptionEncoderC.forward (editnet.py:319-348) without the length sort: rows advance while
// t < len[b]; H / Mem rows beyond a caption's length stay zero; mask = (Mem.sum(2) != 0).
int editnet_encoder(const SetEditNetWeights* w, const int64_t* seq, const int64_t* lens, float* H, float* Mem,
                    float* final_hidden, float* mask, int B, int T, int D, int V, float* emb_seq, float* xg,
                    float* enc_h, float* enc_c, float* s_enc, float* s_aff, hipStream_t st, int* order, void* enc_bar) {
    const int tgt = gemm_target_wgs();
    const bool fused = (D % 128 == 0) && env_int("SET_NO_FUSED", 0) == 0;
    // with the token table the hoisted input projection x W_xh^T + b_xh is a row gather done by the step kernel
    const bool tab = fused && w->tok_table != nullptr;
    if (!tab) {
        SET_TRY(embed_relu(w->embed, seq, 1, emb_seq, D, B * T, D, V, st));
        GemmProb p = direct_prob(xg, 4 * D, B * T, 4 * D, w->enc_x2h_b, SET_ACT_NONE);
        p.add(emb_seq, D, w->enc_x2h_w, D, D);
        SET_TRY(gemm_group(&p, 1, st, "gemm:enc x2h"));
    }
    {
        float* zp[4] = {H, Mem, enc_h, enc_c};
        const size_t zn[4] = {(size_t)B * T * D, (size_t)B * T * D, (size_t)B * D, (size_t)B * D};
        SET_TRY(zero_runs(zp, zn, 4, st));
    }
    float* h_cur = enc_h;
    float* h_nxt = s_enc;                         // (B,D) ping-pong partner (the slab region is free in the fused path)
    // visit the rows longest first (device-side ranking, no host sync) so that tiles of finished rows are skipped:
    // order = [perm (B) | nactive (T)]
    static const int skip = env_int("SET_ENC_SKIP", 1);
    int *perm = nullptr, *nactive = nullptr;
    if (fused && order && skip && B <= 4096) {
        perm = order; nactive = order + B;
        SET_TRY(encoder_order(lens, B, T, perm, nactive, st));
    }
    // the whole recurrence as ONE weights-stationary launch (encoder_persistent.hip) when the shape allows it
    bool persistent = fused && perm && enc_bar && persistent_encoder_ok(B, D, T);
    if (persistent) {
        const int rc = tab ? persistent_encoder(w->enc_h2h_w, w->tok_table + 6 * D, 10LL * D, 0, w->enc_h2h_b, lens, seq, T, V,
                                                h_cur, h_nxt, H, Mem, (long long)T * D, D, perm, nactive, enc_bar, B, D, T, st)
                           : persistent_encoder(w->enc_h2h_w, xg, (long long)T * 4 * D, 4 * D, w->enc_h2h_b, lens, nullptr, 0, 0,
                                                h_cur, h_nxt, H, Mem, (long long)T * D, D, perm, nactive, enc_bar, B, D, T, st);
        if (rc == SET_OK) { if (T & 1) { float* tmp = h_cur; h_cur = h_nxt; h_nxt = tmp; } }
        else if (rc == SET_ERR_UNSUPPORTED) persistent = false;
        else return rc;
    }
    for (int t = 0; t < T && !persistent; ++t) {
        if (fused) {
            if (tab)
                SET_TRY(fused_encoder_step(h_cur, h_nxt, enc_c, w->enc_h2h_w, w->tok_table + 6 * D, 10LL * D, 0,
                                           w->enc_h2h_b, lens, t, 0, H, Mem, (long long)T * D, D, 0, B, D, st, seq, T, V, perm, nactive));
            else
                SET_TRY(fused_encoder_step(h_cur, h_nxt, enc_c, w->enc_h2h_w, xg, (long long)T * 4 * D, 4 * D,
                                           w->enc_h2h_b, lens, t, 0, H, Mem, (long long)T * D, D, 0, B, D, st, nullptr, 0, 0,
                                           perm, nactive));
            float* tmp = h_cur; h_cur = h_nxt; h_nxt = tmp;
            continue;
        }
        GemmProb p = slab_prob(s_enc, B, 4 * D, B);
        p.add(enc_h, D, w->enc_h2h_w, D, D);
        plan_ksplit(&p, 1, tgt);
        if (t > 0) SET_TRY(gemm_group(&p, 1, st, "gemm:enc h2h"));   // h == 0 at t == 0: product is exactly zero
        Slabs hh = slabs_of(p);
        if (t == 0) hh.n = 0;
        SET_TRY(encoder_pointwise(hh, xg, (long long)T * 4 * D, 4 * D, t, lens, 0, enc_h, enc_c, H, Mem,
                                  (long long)T * D, D, 0, B, D, w->enc_h2h_b, st));
    }
    const float* h_last = h_cur;
    {
        GemmProb p = slab_prob(s_aff, B, D, B);
        p.add(h_last, D, w->enc_aff_w, D, D);
        plan_ksplit(&p, 1, tgt);
        SET_TRY(gemm_group(&p, 1, st));
        SET_TRY(reduce_bias_act(slabs_of(p), w->enc_aff_b, nullptr, final_hidden, D, B, D, SET_ACT_TANH, st));
    }
    SET_TRY(rowsum_mask(Mem, D, B * T, D, mask, st));
    return SET_OK;
}

static int begin_impl(const SetEditNetWeights* w, const SetEditNetDims* d, const float* X, const float* image_mean,
                      const int64_t* prev, const int64_t* prevlen, EditNetWs& ws, hipStream_t st) {
    const int B = d->B, T = d->T, R = d->R, F = d->F, D = d->D, A = d->A;
    const int tgt = gemm_target_wgs();
    // ---- caption encoder (editnet.py:319-348)
    SET_TRY(editnet_encoder(w, prev, prevlen, ws.H, ws.Mem, ws.final_hidden, ws.mask, B, T, D, d->V, ws.emb_seq, ws.xg,
                            ws.enc_h, ws.enc_c, ws.s_enc, ws.s_aff, st, ws.enc_order, ws.enc_bar));
    // ---- hoisted, loop-invariant projections (eval mode)
    {
        // one grouped launch over the encoder outputs: att1_c (editnet.py:370) and the contractions that are linear in
        // the attention context / the selected memory row, hoisted out of the timestep (see CapAttArgs in attention.hip):
        //   cap_proj[b,t] = [context_gate.W[:, 2D:3D] H_t | sc_affine.W H_t]   (editnet.py:378-379, no bias here)
        //   mem_proj[b,t] = gate_cmem.W Mem_t                                   (editnet.py:281)
        GemmProb p[4];
        p[0] = direct_prob(ws.att1_c, A, B * T, A, w->ca_feat_b, SET_ACT_NONE);
        p[0].add(ws.H, D, w->ca_feat_w, D, D);
        p[1] = direct_prob(ws.cap_proj, 2LL * D, B * T, D, nullptr, SET_ACT_NONE);
        p[1].add(ws.H, D, w->ca_gate_w + 2 * D, 3LL * D, D);
        p[2] = direct_prob(ws.cap_proj + D, 2LL * D, B * T, D, nullptr, SET_ACT_NONE);
        p[2].add(ws.H, D, w->ca_sc_w, D, D);
        p[3] = direct_prob(ws.mem_proj, D, B * T, D, nullptr, SET_ACT_NONE);
        p[3].add(ws.Mem, D, w->cl_cmem_w, D, D);
        SET_TRY(gemm_group(p, 4, st, "gemm:pro cap projections"));
    }
    {
        // small batches (persistent decode, decode_persistent_wide.hip): the region half of copy_lstm.x2h is linear in the
        // visual attention weights — Pv = X x2h[:, 2D:]^T (B, R, 4D) rides the att_embed launch (same operand X)
        GemmProb p[2];
        p[0] = direct_prob(ws.fe, D, B * R, D, w->va_emb_b, SET_ACT_RELU);                // editnet.py:441
        p[0].add(X, F, w->va_emb_w, F, F);
        const bool pv = editnet_persistent_ok(d, 1);
        if (pv) {
            p[1] = direct_prob(ws.pd_pv, 4LL * D, B * R, 4 * D, nullptr, SET_ACT_NONE);
            p[1].add(X, F, w->cl_x2h_w + 2 * D, 2LL * D + F, F);
        }
        SET_TRY(gemm_group(p, pv ? 2 : 1, st, "gemm:pro att_embed"));
        GemmProb q = direct_prob(ws.att1, A, B * R, A, w->va_feat_b, SET_ACT_NONE);       // editnet.py:442
        q.add(ws.fe, D, w->va_feat_w, D, D);
        SET_TRY(gemm_group(&q, 1, st, "gemm:pro features_att"));
        if (d->adaptive) SET_TRY(region_masks(X, ws.fe, ws.rmask, B, R, F, D, st));
    }
    if (image_mean)
        SET_HIP_TRY(hipMemcpyAsync(ws.image_mean, image_mean, sizeof(float) * B * F, hipMemcpyDeviceToDevice, st));
    else
        SET_TRY(mean_regions(X, ws.image_mean, B, R, F, st));                              // editnet.py:503
    {
        // attention_lstm input columns [D,2D) (final_hidden) and [3D,3D+F) (image_mean) + both biases
        const long long ldw = 3LL * D + F;
        GemmProb p = slab_prob(ws.s_pre, B, 4 * D, B);
        p.add(ws.final_hidden, D, w->al_wih + D, ldw, D);
        p.add(ws.image_mean, F, w->al_wih + 3 * D, ldw, F);
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

// One timestep for rows [0,bt).  ws.emb must already hold relu(E[token]).  The vocabulary
// projection is left as split-K slabs (`*logits_out`, bias NOT yet added) unless `dst` is given,
// in which case (bt,V) logits with bias are written to dst (leading stride ld_dst).
// The phase-A problems of a timestep (unplanned): attention-LSTM gate product over [emb|h2|h1] (emb omitted with the
// token table) and the copy-LSTM's h2h(h2).
static void build_phase_a(const SetEditNetWeights* w, const SetEditNetDims* d, EditNetWs& ws, int bt, bool tab,
                          GemmProb a[2]) {
    const int B = d->B, D = d->D;
    const long long ld_ih = 3LL * D + d->F;
    a[0] = slab_prob(ws.sA0, bt, 4 * D, B);
    if (!tab) a[0].add(ws.emb, D, w->al_wih, ld_ih, D);
    a[0].add(ws.h2, D, w->al_wih + 2 * D, ld_ih, D);
    a[0].add(ws.h1, D, w->al_whh, D, D);
    a[1] = slab_prob(ws.sA1, bt, 4 * D, B);
    a[1].add(ws.h2, D, w->cl_h2h_w, D, D);
}

// F/A merge: with the token table NOTHING in phase A of timestep t+1 depends on the token timestep t is about to pick
// (the token only selects a table row that the LSTM pointwise adds), and h1 / h2 of timestep t are final before its
// vocabulary projection starts.  The free-running loop therefore launches fc(h2_t) and the phase-A products of t+1 as ONE
// grouped GEMM (5.8 GFLOP, 32 k-tiles per workgroup, fc unsplit with its bias fused) instead of two launches with
// 16-k-tile workgroups: one launch ramp less per timestep, half the logit traffic into the pick.
//   a_pre  != NULL: phase A of THIS timestep was launched by the previous one; these are its planned problems
//   a_next != NULL: launch the NEXT timestep's phase A together with this timestep's fc and return its problems here
//   *logits_biased: set when the returned logits already include fc.bias (unsplit fc)
//   a_done: the attention-LSTM cell of this timestep was finished by the previous pick (LstmTail): h1 / c1 are current
static int step_impl(const SetEditNetWeights* w, const SetEditNetDims* d, const float* X, int bt, EditNetWs& ws,
                     const long long* tok_ids, long long tok_stride, float* dst,
                     long long ld_dst, Slabs* logits_out, hipStream_t st, const GemmProb* a_pre = nullptr,
                     GemmProb* a_next = nullptr, bool* logits_biased = nullptr, int bt_next = -1, bool a_done = false) {
    const int B = d->B, T = d->T, R = d->R, F = d->F, D = d->D, A = d->A, V = d->V;
    const int tgt = gemm_target_wgs();
    const long long ld_x2h = 2LL * D + F;
    // ---- A
    // token-only contractions folded into the (V,6D) table (inference): their K-segments disappear and the
    // consuming pointwise kernels add the gathered table row instead
    const bool fusedk = (D % 64 == 0) && env_int("SET_NO_FUSED", 0) == 0;
    const bool tab = w->tok_table != nullptr && tok_ids != nullptr && fusedk;
    RowGather g_gates, g_tc, g_cg;
    if (tab) {
        g_gates = RowGather{w->tok_table, tok_ids, tok_stride, 10LL * D, 0, V};
        g_tc = RowGather{w->tok_table, tok_ids, tok_stride, 10LL * D, 4 * D, V};
        g_cg = RowGather{w->tok_table, tok_ids, tok_stride, 10LL * D, 5 * D, V};
    }
    GemmProb a[2];
    if (a_pre) {
        a[0] = a_pre[0]; a[1] = a_pre[1];
    } else {
        build_phase_a(w, d, ws, bt, tab, a);
        plan_ksplit(a, 2, tgt);
        SET_TRY(gemm_group(a, 2, st, "gemm:A gates1+h2h"));
    }
    const Slabs none{nullptr, 0, 0, 0};
    if (!(a_done && a_pre && tab))
        SET_TRY(lstm_pointwise(slabs_of(a[0]), none, none, ws.pre1, 4 * D, nullptr, nullptr, ws.c1, ws.c1, ws.h1, nullptr,
                               bt, D, st, g_gates));
    // ---- B
    GemmProb b[5];
    b[0] = slab_prob(ws.sB0, bt, A, B);
    b[0].add(ws.h1, D, w->ca_dec_w, D, D);
    b[1] = slab_prob(ws.sB1, bt, A, B);
    b[1].add(ws.h1, D, w->va_dec_w, D, D);
    b[2] = slab_prob(ws.sB2, bt, D, B);
    if (!tab) b[2].add(ws.emb, D, w->ca_tc_w, 2 * D, D);
    b[2].add(ws.h1, D, w->ca_tc_w + D, 2 * D, D);
    b[3] = slab_prob(ws.sB3, bt, D, B);
    if (!tab) b[3].add(ws.emb, D, w->ca_gate_w, 3 * D, D);
    b[3].add(ws.h1, D, w->ca_gate_w + D, 3 * D, D);
    b[4] = slab_prob(ws.sB4, bt, 4 * D, B);
    b[4].add(ws.h1, D, w->cl_x2h_w, ld_x2h, D);
    static const int b_bm = env_int("SET_GEMM_B_BM", 0);
    if (bt > 64) b[0].bm_hint = b_bm;
    plan_ksplit(b, 5, tgt);
    SET_TRY(gemm_group(b, 5, st, "gemm:B att2,tc,cg,x2h_h1"));
    // the caption role also applies the context gate: its ctx-side contractions are hoisted (ws.cap_proj), the
    // [word,h1]-side ones arrive as slabs b[3] / b[2] (+ token-table rows); gate_cmem(sel) becomes a row of ws.mem_proj
    CapHoist hoist;
    hoist.P = ws.cap_proj; hoist.Q = ws.mem_proj; hoist.cmem_out = ws.cmem_pre; hoist.gated_out = ws.attend_cap;
    hoist.cg_ab = slabs_of(b[3]); hoist.tc = slabs_of(b[2]); hoist.gz = g_cg; hoist.gtc = g_tc;
    hoist.b_gate = w->ca_gate_b; hoist.b_sc = w->ca_sc_b; hoist.b_tc = w->ca_tc_b;
    SET_TRY(step_attention(ws.att1, slabs_of(b[1]), w->va_dec_b, w->va_full_w, w->va_full_b, X,
                           d->adaptive ? ws.rmask : nullptr, ws.attend_img, ws.alpha, R, F, ws.att1_c, slabs_of(b[0]),
                           w->ca_dec_b, w->ca_full_w, w->ca_full_b, ws.mask, ws.H, ws.Mem, ws.ctx_cap, ws.sel,
                           ws.alpha_c, T, D, A, bt, st, &hoist));
    // from 17 rows the fused E kernel (32-row tiles, the copy gate as its epilogue: 12 us whatever the row count — its
    // serial k-loop bounds it).  Until round 6 it started at 65 rows ("<= 64 workgroups"): re-measured on the 8-wave kernel
    // of round 5, grouped GEMM + pointwise is 17 + 6.5 us at 32 rows: B = 24 / 32 / 48 / 64 greedy decode 2.19 / 2.28 / 2.75 /
    // 2.81 -> 2.16 / 2.25 / 2.72 / 2.79 ms.  Up to 16 rows the gemv class + pointwise stays (SET_FUSED_MIN_ROWS).
    static const int fused_min_rows = env_int("SET_FUSED_MIN_ROWS", 17);
    const bool fused = fusedk && bt >= fused_min_rows;
    // ---- D
    GemmProb dd = slab_prob(ws.sD0, bt, 4 * D, B);
    dd.add(ws.attend_cap, D, w->cl_x2h_w + D, ld_x2h, D);
    dd.add(ws.attend_img, F, w->cl_x2h_w + 2 * D, ld_x2h, F);
    plan_ksplit(&dd, 1, tgt);
    SET_TRY(gemm_group(&dd, 1, st, "gemm:D x2h_ctx"));
    SET_TRY(lstm_pointwise(slabs_of(a[1]), slabs_of(b[4]), slabs_of(dd), nullptr, 0, w->cl_x2h_b, w->cl_h2h_b, ws.c2,
                           ws.c_new, nullptr, ws.ogate, bt, D, st));
    // ---- E (+ copy gate): gate_cnew(c_new) is the only contraction left here
    if (fused) {
        SET_TRY(fused_copy_gate_pre(ws.c_new, ws.sel, ws.cmem_pre, ws.ogate, w->cl_cnew_w, w->cl_cnew_b, w->cl_cmem_b,
                                    ws.c2, ws.h2, bt, D, st));
    } else {
        GemmProb e = slab_prob(ws.sE0, bt, D, B);
        e.add(ws.c_new, D, w->cl_cnew_w, D, D);
        plan_ksplit(&e, 1, tgt);
        SET_TRY(gemm_group(&e, 1, st, "gemm:E cnew"));
        SET_TRY(copy_gate_pointwise(slabs_of(e), w->cl_cnew_b, Slabs{ws.cmem_pre, 0, D, 1}, w->cl_cmem_b, ws.c_new,
                                    ws.sel, ws.ogate, ws.c2, ws.h2, bt, D, st));
    }
    // ---- F
    const long long Vp = (long long)round_up((size_t)V, 64);
    GemmProb f = slab_prob(ws.sF0, bt, V, B);
    f.ldc = Vp;
    f.slab_stride = (long long)B * Vp;
    f.add(ws.h2, D, w->fc_w, D, D);
    static const int f_bm = env_int("SET_GEMM_F_BM", 0);
    if (bt > 64) f.bm_hint = f_bm;
    if (logits_biased) *logits_biased = false;
    if (a_next && tab) {
        // bt_next: rows of the NEXT timestep (teacher-forced loop: the sorted batch shrinks); default = this step's rows
        GemmProb fa[3];
        fa[0] = f;
        build_phase_a(w, d, ws, bt_next > 0 ? bt_next : bt, true, fa + 1);
        plan_ksplit(fa, 3, tgt);
        if (fa[0].ksplit == 1) {             // unsplit fc: write the scores once, bias fused (straight into dst when given)
            fa[0].C = dst ? dst : ws.logits; fa[0].slab_stride = 0; fa[0].bias = w->fc_b;
            if (dst) fa[0].ldc = ld_dst;
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

size_t set_editnet_workspace_bytes(const SetEditNetDims* d) {
    if (check_dims(d) != SET_OK) return 0;
    return carve(d, nullptr).bytes + 256;
}

static int prep(const SetEditNetDims* d, void* ws, size_t ws_bytes, EditNetWs* out) {
    SET_TRY(check_dims(d));
    if (!ws || !aligned16(ws)) return SET_ERR_ARG;
    *out = carve(d, ws);
    if (out->bytes > ws_bytes) return SET_ERR_WORKSPACE;
    return SET_OK;
}

int set_editnet_begin(const SetEditNetWeights* w, const SetEditNetDims* d, const float* X, const float* image_mean,
                      const int64_t* prev, const int64_t* prevlen, void* ws, size_t ws_bytes, void* stream) {
    if (!w || !X || !prev || !prevlen) return SET_ERR_ARG;
    EditNetWs W;
    SET_TRY(prep(d, ws, ws_bytes, &W));
    return begin_impl(w, d, X, image_mean, prev, prevlen, W, (hipStream_t)stream);
}


int set_editnet_step(const SetEditNetWeights* w, const SetEditNetDims* d, const float* X, const int64_t* tokens,
                     int64_t tokens_stride, int bt, float* logits, int64_t ld_logits, void* ws, size_t ws_bytes,
                     void* stream) {
    if (!w || !X || !logits) return SET_ERR_ARG;
    EditNetWs W;
    SET_TRY(prep(d, ws, ws_bytes, &W));
    if (bt <= 0 || bt > d->B || ld_logits < d->V) return SET_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (tokens) SET_TRY(embed_relu(w->embed, tokens, tokens_stride, W.emb, d->D, bt, d->D, d->V, st));
    return step_impl(w, d, X, bt, W, tokens ? (const long long*)tokens : W.it, tokens ? tokens_stride : 1, logits,
                     ld_logits, nullptr, st);
}

int set_editnet_greedy_pick(const SetEditNetWeights* w, const SetEditNetDims* d, const float* logits,
                            int64_t ld_logits, int t, int64_t end_idx, int64_t* seq, float* seq_logp, int max_len,
                            void* ws, size_t ws_bytes, void* stream) {
    if (!w || !logits || !seq || !seq_logp || t < 0) return SET_ERR_ARG;
    EditNetWs W;
    SET_TRY(prep(d, ws, ws_bytes, &W));
    if (t > d->maxT) return SET_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (t == 0) SET_TRY(set_tokens(W.it, 0, W.unfinished, W.alive, d->maxT + 2, d->B, st));
    Slabs lg{logits, 0, ld_logits, 1};
    return greedy_pick(lg, nullptr, d->V, t, max_len, end_idx, (long long*)seq, seq_logp, W.it, W.unfinished, W.alive,
                       w->embed, W.emb, d->D, d->B, st);
}

// free-running decode (editnet_rl.py:485-549): sample == 0 greedy (sample_max), 1 multinomial (sample_rl)
// `begun`: the workspace already holds a completed set_editnet_begin for these inputs (the per-sequence prologue ran
// earlier, possibly on another stream that this one has been made to wait for): only the timestep loop runs here
static int rollout(const SetEditNetWeights* w, const SetEditNetDims* d, const float* X, const float* image_mean,
                   const int64_t* prev, const int64_t* prevlen, int64_t start_idx, int64_t end_idx, int max_len,
                   int sample, uint64_t seed, uint64_t offset, int64_t* seq, float* seq_logp, void* ws, size_t ws_bytes,
                   void* stream, bool begun = false) {
    if (!w || !X || (!begun && (!prev || !prevlen)) || !seq || !seq_logp || max_len <= 0) return SET_ERR_ARG;
    EditNetWs W;
    SET_TRY(prep(d, ws, ws_bytes, &W));
    if (max_len + 1 > d->maxT + 1 || start_idx < 0 || start_idx >= d->V) return SET_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int B = d->B;
    if (!begun) SET_TRY(begin_impl(w, d, X, image_mean, prev, prevlen, W, st));
    SET_HIP_TRY(hipMemsetAsync(seq, 0, sizeof(int64_t) * B * max_len, st));
    SET_HIP_TRY(hipMemsetAsync(seq_logp, 0, sizeof(float) * B * max_len, st));
    SET_TRY(set_tokens(W.it, start_idx, W.unfinished, W.alive, d->maxT + 2, B, st));
    // with the token table the step never reads relu(E[it]) (all its consumers gather the folded products),
    // so neither the loop's first timestep nor the epilogue gathers the embedding
    const bool emb_needed = !(w->tok_table && (d->D % 64 == 0) && env_int("SET_NO_FUSED", 0) == 0);
    if (emb_needed) SET_TRY(embed_relu(w->embed, (const int64_t*)W.it, 1, W.emb, d->D, B, d->D, d->V, st));
    // the reference runs max_len + 1 timesteps and discards the last one (editnet_rl.py:503,517-518)
    static const int fa_merge = env_int("SET_FA_MERGE", 1);
    const bool merge = fa_merge && !emb_needed;      // the token table is active: phase A does not see the token
    // the pick finishes the next timestep's attention-LSTM cell (LstmTail): its gate products ride this timestep's fc
    // launch and only the token-table row waits for the word (SET_PICK_TAIL=0: separate lstm_pointwise launch)
    static const int pick_tail = env_int("SET_PICK_TAIL", 1);
    GemmProb a_cur[2], a_nxt[2];
    bool have_a = false, a_done = false;
    // Row gate (set_common.h): every kernel of timestep t returns at once when alive[t - 1] == 0 — the reference's `break`
    // (editnet_rl.py:546) in TIME, not only in the outputs (SET_LOOP_GATE=0: as before, outputs only).
    static const int loop_gate = env_int("SET_LOOP_GATE", 1);
    // small batches, greedy: the whole loop as ONE persistent launch (decode_persistent_wide.hip).  copy_lstm.x2h's region
    // columns are linear in the visual attention weights: Pv = X x2h[:, 2D:]^T is computed here once per decode
    if (!sample && !emb_needed && !g_row_limit && editnet_persistent_ok(d, max_len)) {
        if (begun) {     // the prologue ran in an earlier call (begin_ahead), possibly under other switches: Pv is not taken on trust
            const int R = d->R, F = d->F, D = d->D;
            GemmProb p = direct_prob(W.pd_pv, 4LL * D, B * R, 4 * D, nullptr, SET_ACT_NONE);
            p.add(X, F, w->cl_x2h_w + 2 * D, 2LL * D + F, F);
            SET_TRY(gemm_group(&p, 1, st, "gemm:pro x2h_img hoist"));
        }
        const int rc = editnet_persistent_greedy(w, d, W.pre1, W.att1, W.att1_c, W.mask, W.cap_proj, W.mem_proj, W.Mem, W.pd_pv,
                                                 W.pd_x, W.it, W.unfinished, W.alive, start_idx, end_idx, max_len, (long long*)seq,
                                                 seq_logp, st);
        if (rc != SET_ERR_UNSUPPORTED) return rc;
    }
    for (int t = 0; t <= max_len; ++t) {
        Slabs lg;
        bool biased = false;
        const bool next_a = merge && t < max_len;     // there is a next timestep to pre-launch phase A for
        RowGate gate;
        if (loop_gate && t > 0) gate.alive_prev = W.alive + (t - 1);
        RowGateScope gate_scope(gate);
        SET_TRY(step_impl(w, d, X, B, W, W.it, 1, nullptr, 0, &lg, st, have_a ? a_cur : nullptr, next_a ? a_nxt : nullptr,
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
            tail.tab = w->tok_table; tail.ld_tab = 10LL * d->D; tail.col0 = 0; tail.nrows = d->V;
            tail.c_in = W.c1; tail.c_out = W.c1; tail.h_out = W.h1; tail.D = d->D;
        }
        if (sample)
            SET_TRY(sample_pick(lg, pick_bias, d->V, t, max_len, end_idx, (long long*)seq, seq_logp, W.it, W.unfinished,
                                W.alive, emb_needed ? w->embed : nullptr, W.emb, d->D, B, seed, offset, nullptr, nullptr,
                                nullptr, st, a_done ? &tail : nullptr));
        else
            SET_TRY(greedy_pick(lg, pick_bias, d->V, t, max_len, end_idx, (long long*)seq, seq_logp, W.it, W.unfinished,
                                W.alive, emb_needed ? w->embed : nullptr, W.emb, d->D, B, st, a_done ? &tail : nullptr));
    }
    return SET_OK;
}

int set_decode_row_limits(const int* row_limit_dev) {
    g_row_limit = row_limit_dev;
    return SET_OK;
}

int set_editnet_greedy(const SetEditNetWeights* w, const SetEditNetDims* d, const float* X, const float* image_mean,
                       const int64_t* prev, const int64_t* prevlen, int64_t start_idx, int64_t end_idx, int max_len,
                       int64_t* seq, float* seq_logp, void* ws, size_t ws_bytes, void* stream) {
    return rollout(w, d, X, image_mean, prev, prevlen, start_idx, end_idx, max_len, 0, 0, 0, seq, seq_logp, ws, ws_bytes,
                   stream);
}

int set_editnet_greedy_begun(const SetEditNetWeights* w, const SetEditNetDims* d, const float* X, int64_t start_idx,
                             int64_t end_idx, int max_len, int64_t* seq, float* seq_logp, void* ws, size_t ws_bytes,
                             void* stream) {
    return rollout(w, d, X, nullptr, nullptr, nullptr, start_idx, end_idx, max_len, 0, 0, 0, seq, seq_logp, ws, ws_bytes,
                   stream, true);
}

int set_editnet_beam_persistent(const SetEditNetWeights* w, const SetEditNetDims* d, const float* X, const float* image_mean,
                                const int64_t* prev, const int64_t* prevlen, int64_t start_idx, int64_t end_idx, int max_picks,
                                int32_t* hist_parent, int64_t* hist_word, float* best_score, int64_t* best_word,
                                int32_t* result, void* ws, size_t ws_bytes, void* stream) {
    if (!w || !d || !X || !prev || !prevlen || !hist_parent || !hist_word || !best_score || !best_word || !result || max_picks < 1)
        return SET_ERR_ARG;
    if (start_idx < 0 || start_idx >= d->V) return SET_ERR_ARG;
    // (nothing is touched before the checks that can answer SET_ERR_UNSUPPORTED)
    if (!(w->tok_table && (d->D % 64 == 0) && env_int("SET_NO_FUSED", 0) == 0)) return SET_ERR_UNSUPPORTED;
    if (d->B > 4 || d->adaptive || !env_int("SET_DEC_PERSISTENT", 1) || persistent_disabled() ||
        !editnet_persistent_wide_ok(d->B, d->D, d->A, d->T, d->R, d->V) || !editnet_persistent_ok(d, 1))
        return SET_ERR_UNSUPPORTED;
    EditNetWs W;
    SET_TRY(prep(d, ws, ws_bytes, &W));
    hipStream_t st = (hipStream_t)stream;
    SET_TRY(begin_impl(w, d, X, image_mean, prev, prevlen, W, st));      // (includes Pv = X x2h[:, 2D:]^T: editnet_persistent_ok)
    const PDecBeam beam{hist_parent, hist_word, best_score, best_word, result};
    return editnet_persistent_greedy(w, d, W.pre1, W.att1, W.att1_c, W.mask, W.cap_proj, W.mem_proj, W.Mem, W.pd_pv, W.pd_x, W.it,
                                     W.unfinished, W.alive, start_idx, end_idx, max_picks, nullptr, nullptr, st, nullptr, &beam);
}

int set_editnet_sample(const SetEditNetWeights* w, const SetEditNetDims* d, const float* X, const float* image_mean,
                       const int64_t* prev, const int64_t* prevlen, int64_t start_idx, int64_t end_idx, int max_len,
                       uint64_t seed, uint64_t offset, int64_t* seq, float* seq_logp, void* ws, size_t ws_bytes,
                       void* stream) {
    return rollout(w, d, X, image_mean, prev, prevlen, start_idx, end_idx, max_len, 1, seed, offset, seq, seq_logp, ws,
                   ws_bytes, stream);
}

int set_editnet_xe_forward(const SetEditNetWeights* w, const SetEditNetDims* d, const float* X,
                           const float* image_mean, const int64_t* caps, int64_t caps_stride,
                           const int* host_decode_lengths, const int64_t* prev, const int64_t* prevlen,
                           float* predictions, void* ws, size_t ws_bytes, void* stream) {
    if (!w || !X || !caps || !host_decode_lengths || !prev || !prevlen || !predictions) return SET_ERR_ARG;
    EditNetWs W;
    SET_TRY(prep(d, ws, ws_bytes, &W));
    hipStream_t st = (hipStream_t)stream;
    const int B = d->B, V = d->V, maxT = d->maxT;
    for (int b = 0; b < B; ++b) {
        if (host_decode_lengths[b] < 0 || host_decode_lengths[b] > maxT) return SET_ERR_ARG;
        if (b && host_decode_lengths[b] > host_decode_lengths[b - 1]) return SET_ERR_ARG;   // sorted desc
    }
    if (caps_stride < maxT) return SET_ERR_ARG;
    SET_TRY(begin_impl(w, d, X, image_mean, prev, prevlen, W, st));
    SET_HIP_TRY(hipMemsetAsync(predictions, 0, sizeof(float) * (size_t)B * maxT * V, st));
    // with the token table the step never reads relu(E[it]) and phase A of timestep t+1 does not depend on its token:
    // fc(h2_t) and the phase-A products of t+1 (over the rows still in the batch then) ride one launch, as in the
    // free-running loop
    const bool emb_needed = !(w->tok_table && (d->D % 64 == 0) && env_int("SET_NO_FUSED", 0) == 0);
    // small batches: the teacher-forced loop as ONE persistent launch too (decode_persistent_wide.hip)
    if (!emb_needed && editnet_persistent_ok(d, maxT)) {
        const PDecTeacher teach{caps, caps_stride, predictions, host_decode_lengths};
        const int rc = editnet_persistent_greedy(w, d, W.pre1, W.att1, W.att1_c, W.mask, W.cap_proj, W.mem_proj, W.Mem, W.pd_pv,
                                                 W.pd_x, W.it, W.unfinished, W.alive, 0, -1, maxT, nullptr, nullptr, st, &teach);
        if (rc != SET_ERR_UNSUPPORTED) return rc;
    }
    static const int fa_merge = env_int("SET_FA_MERGE", 1);
    const bool merge = fa_merge && !emb_needed;
    auto rows_at = [&](int t) { int n = 0; while (n < B && host_decode_lengths[n] > t) ++n; return n; };   // editnet.py:506
    GemmProb a_cur[2], a_nxt[2];
    bool have_a = false;
    for (int t = 0; t < maxT; ++t) {
        const int bt = rows_at(t);
        if (bt == 0) break;
        const int btn = t + 1 < maxT ? rows_at(t + 1) : 0;
        if (emb_needed) SET_TRY(embed_relu(w->embed, caps + t, caps_stride, W.emb, d->D, bt, d->D, V, st));
        const bool next_a = merge && btn > 0;
        SET_TRY(step_impl(w, d, X, bt, W, (const long long*)(caps + t), caps_stride, predictions + (size_t)t * V,
                          (long long)maxT * V, nullptr, st, have_a ? a_cur : nullptr, next_a ? a_nxt : nullptr, nullptr, btn));
        have_a = next_a;
        if (next_a) { a_cur[0] = a_nxt[0]; a_cur[1] = a_nxt[1]; }
    }
    return SET_OK;
}

size_t set_editnet_token_table_bytes(const SetEditNetDims* d) {
    if (check_dims(d) != SET_OK) return 0;
    return sizeof(float) * (size_t)d->V * 10 * d->D;
}

size_t set_editnet_token_table_workspace_bytes(const SetEditNetDims* d) {
    if (check_dims(d) != SET_OK) return 0;
    return round_up(sizeof(float) * (size_t)d->V * d->D, 256) + round_up(sizeof(long long) * (size_t)d->V, 256) + 256;
}

int set_editnet_build_token_table(const SetEditNetWeights* w, const SetEditNetDims* d, float* table, void* ws,
                                  size_t ws_bytes, void* stream) {
    if (!w || !table || !ws) return SET_ERR_ARG;
    SET_TRY(check_dims(d));
    if (!aligned16(ws) || !aligned16(table)) return SET_ERR_ARG;
    if (ws_bytes < set_editnet_token_table_workspace_bytes(d) - 256) return SET_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int V = d->V, D = d->D;
    Carver cv(ws);
    float* remb = cv.take<float>((size_t)V * D);
    long long* ids = cv.take<long long>((size_t)V);
    SET_TRY(iota_i64(ids, V, st));
    SET_TRY(embed_relu(w->embed, (const int64_t*)ids, 1, remb, D, V, D, V, st));           // relu(E), all rows
    GemmProb p[4];
    p[0] = direct_prob(table, 10LL * D, V, 4 * D, nullptr, SET_ACT_NONE);
    p[0].add(remb, D, w->al_wih, 3LL * D + d->F, D);
    p[1] = direct_prob(table + 4 * D, 10LL * D, V, D, nullptr, SET_ACT_NONE);
    p[1].add(remb, D, w->ca_tc_w, 2 * D, D);
    p[2] = direct_prob(table + 5 * D, 10LL * D, V, D, nullptr, SET_ACT_NONE);
    p[2].add(remb, D, w->ca_gate_w, 3 * D, D);
    p[3] = direct_prob(table + 6 * D, 10LL * D, V, 4 * D, w->enc_x2h_b, SET_ACT_NONE);
    p[3].add(remb, D, w->enc_x2h_w, D, D);
    return gemm_group(p, 4, st, "gemm:token table");
}

void* set_editnet_ws_tensor(const SetEditNetDims* d, void* ws, const char* name) {
    if (check_dims(d) != SET_OK || !ws || !name) return nullptr;
    EditNetWs W = carve(d, ws);
    struct { const char* n; void* p; } tab[] = {
        {"H", W.H}, {"M", W.Mem}, {"final_hidden", W.final_hidden}, {"mask", W.mask}, {"att1", W.att1},
        {"att1_c", W.att1_c}, {"image_mean", W.image_mean}, {"pre1", W.pre1}, {"rmask", W.rmask}, {"h1", W.h1},
        {"c1", W.c1}, {"h2", W.h2}, {"c2", W.c2}, {"emb", W.emb}, {"ctx_cap", W.ctx_cap},
        {"attend_cap", W.attend_cap}, {"attend_img", W.attend_img}, {"sel", W.sel}, {"c_new", W.c_new},
        {"alpha_c", W.alpha_c}, {"alpha", W.alpha}, {"logits", W.logits}, {"it", W.it},
        {"cap_proj", W.cap_proj}, {"mem_proj", W.mem_proj}, {"enc_bar", W.enc_bar},
        {"unfinished", W.unfinished}, {"alive", W.alive}};
    for (auto& e : tab)
        if (!strcmp(e.n, name)) return e.p;
    return nullptr;
}

}  // extern "C"
