// Host side of EditNet's persistent greedy / teacher-forced / beam decode: the argument block of one launch and the dispatch
// to csrc/decode_persistent_wide.hip, which since round 5 serves every batch of 1 .. 16 rows (it measured faster than the
// <= 8-row kernel of round 4 that lived here, at every row count — B = 4: 1.24 vs 1.28 ms, B = 8: 1.41 vs 1.67 ms — so
// that kernel was removed).
#include "decode_persistent.h"

namespace set {

// exchange region of one decode (laid out by decode_persistent_wide.hip)
size_t editnet_persistent_xbytes(int B, int D, int A) { return B > PDW_MAXB ? 0 : editnet_persistent_wide_xbytes(B, D, A); }

bool editnet_persistent_ok(const SetEditNetDims* d, int max_len) {
    const int on = env_int("SET_DEC_PERSISTENT", 1);                 // (read per call: tests and A/B runs flip it inside one process)
    const int maxb = env_int("SET_DEC_PERSISTENT_MAXB", PDW_MAXB);
    if (!on || d->B > maxb || max_len < 1 || d->adaptive) return false;
    return editnet_persistent_wide_ok(d->B, d->D, d->A, d->T, d->R, d->V) && !persistent_disabled();
}

// the greedy loop after the prologue (set_editnet_begin).  pv = X x2h[:, 2D:]^T (B, R, 4D), xbuf = exchange region.
// SET_ERR_UNSUPPORTED: nothing was touched, the caller runs the per-step loop.
int editnet_persistent_greedy(const SetEditNetWeights* w, const SetEditNetDims* d, const float* pre1, const float* att1,
                              const float* att1_c, const float* mask, const float* capP, const float* memQ, const float* Mem,
                              const float* pv, void* xbuf, long long* it, int* unfinished, int* alive, long long start_idx,
                              long long end_idx, int max_len, long long* seq, float* seq_logp, hipStream_t s,
                              const PDecTeacher* teach, const PDecBeam* beam) {
    if (beam) {         // beam mode lives in the wide variant, whatever the row count
        if (!env_int("SET_DEC_PERSISTENT", 1) || max_len < 1 || d->adaptive || persistent_disabled() ||
            !editnet_persistent_wide_ok(d->B, d->D, d->A, d->T, d->R, d->V)) return SET_ERR_UNSUPPORTED;
    } else if (!editnet_persistent_ok(d, max_len)) return SET_ERR_UNSUPPORTED;
    const int B = d->B, D = d->D, A = d->A, F = d->F, G = D / 4;
    PDecEditArgs P{};
    P.al_wih = w->al_wih; P.ld_ih = 3LL * D + F; P.al_whh = w->al_whh; P.cl_h2h_w = w->cl_h2h_w;
    P.cl_x2h_w = w->cl_x2h_w; P.ld_x2h = 2LL * D + F; P.cl_x2h_b = w->cl_x2h_b; P.cl_h2h_b = w->cl_h2h_b;
    P.ca_gate_w = w->ca_gate_w; P.ca_gate_b = w->ca_gate_b; P.ca_tc_w = w->ca_tc_w; P.ca_tc_b = w->ca_tc_b; P.ca_sc_b = w->ca_sc_b;
    P.ca_dec_w = w->ca_dec_w; P.ca_dec_b = w->ca_dec_b; P.ca_full_w = w->ca_full_w; P.ca_full_b = w->ca_full_b;
    P.va_dec_w = w->va_dec_w; P.va_dec_b = w->va_dec_b; P.va_full_w = w->va_full_w; P.va_full_b = w->va_full_b;
    P.cl_cnew_w = w->cl_cnew_w; P.cl_cnew_b = w->cl_cnew_b; P.cl_cmem_b = w->cl_cmem_b;
    P.fc_w = w->fc_w; P.fc_b = w->fc_b; P.tok_table = w->tok_table; P.ld_tab = 10LL * D;
    P.pre1 = pre1; P.att1 = att1; P.att1_c = att1_c; P.mask = mask; P.capP = capP; P.memQ = memQ; P.Mem = Mem; P.pv = pv;
    P.it = it; P.unfinished = unfinished; P.alive = alive; P.seq = seq; P.seq_logp = seq_logp;
    P.B = B; P.D = D; P.T = d->T; P.R = d->R; P.A = A; P.V = d->V; P.max_len = max_len; P.rpw = (d->V + G - 1) / G;
    P.start_idx = start_idx; P.end_idx = end_idx;
    if (teach) {
        P.caps = (const long long*)teach->caps; P.caps_stride = teach->caps_stride;
        P.predictions = teach->predictions; P.ld_pred_b = (long long)max_len * d->V;
        for (int b = 0; b < B; ++b) P.dlen[b] = teach->host_decode_lengths[b];
    }
    PersistentGuard guard;
    if (guard.rc != SET_OK) return guard.rc;
    P.spin_limit = guard.spin_limit();
    P.test_stall = guard.test_stall(); P.fault = guard.fault;
    if (beam) {
        P.bm_hist_par = beam->hist_par; P.bm_hist_word = (long long*)beam->hist_word; P.bm_best_score = beam->best_score;
        P.bm_best_word = (long long*)beam->best_word; P.bm_result = beam->result;
    }
    bool unsupported = true;
    const int rc = editnet_persistent_wide_launch(P, xbuf, guard, s, &unsupported, beam != nullptr);      // (lays out the exchange region)
    return rc != SET_OK ? rc : (unsupported ? SET_ERR_UNSUPPORTED : SET_OK);
}

}  // namespace set
