"""The DCNet decode loops (reference `dcnet.py:333-348` teacher-forced, `dcnet_rl.py:305-344` sampled) as ONE autograd
node — the text-only twin of `xe_sequence.py` (see there for the design: per-sequence logs, Philox dropout kernels,
every "+=" of back-propagation through time as the accumulate flag of a GEMM problem or of the attention backward,
one contraction per parameter gradient over all T x B rows)."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from . import autograd_ops as A
from . import rng
from ._lib import EditNetWeights, check
from .xe_sequence import SeqConfig, _Ops, _dvalues, _e, _rows, _z  # noqa: F401

PARAM_NAMES = ("E", "al_wih", "al_whh", "al_bih", "al_bhh", "ca_dec_w", "ca_dec_b", "ca_full_w", "ca_full_b",
               "ll_wih", "ll_whh", "ll_bih", "ll_bhh", "fc_w", "fc_b")


def dae_params(dae):
    al, ll, ca = dae.attention_lstm, dae.language_lstm, dae.caption_attention
    return (dae.embed.embedding.weight, al.weight_ih, al.weight_hh, al.bias_ih, al.bias_hh,
            ca.cap_decoder_att.weight, ca.cap_decoder_att.bias, ca.cap_full_att.weight, ca.cap_full_att.bias,
            ll.weight_ih, ll.weight_hh, ll.bias_ih, ll.bias_hh, dae.fc.weight, dae.fc.bias)


class _DcnetSequence(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cfg, enc, final_hidden, mask, att1_c, caps, *params):
        P = dict(zip(PARAM_NAMES, params))
        dev = enc.device
        ops = _Ops(dev)
        lib, st = ops.lib, ops.st
        ro = cfg.rollout
        B = enc.shape[0]
        lens = cfg.decode_lengths if ro is None else [int(ro["max_len"])] * B
        T = max(lens)
        bts = [sum(1 for l in lens if l > t) for t in range(T)]
        Tc, Dh = enc.shape[1], enc.shape[2]                  # Dh = 2C
        E, D = P["E"].shape[1], P["al_whh"].shape[1]
        Adim, V = P["ca_dec_w"].shape[0], P["fc_w"].shape[0]
        K1, K2 = E + Dh + D, D + Dh
        train = cfg.train
        enc, final_hidden, mask, att1_c, caps = (t.contiguous() for t in (enc, final_hidden, mask, att1_c, caps))
        uniform = min(lens) == T
        _zl = _e if uniform else _z
        L = {"FH": final_hidden, "EMB": _zl(T, B, E, dev=dev), "X2": _zl(T, B, K2, dev=dev),
             "G1": _zl(T, B, 4 * D, dev=dev), "G2": _zl(T, B, 4 * D, dev=dev), "ALPHAC": _zl(T, B, Tc, dev=dev),
             "ATT2": _e(T, B, Adim, dev=dev)}
        for k in ("H1", "C1", "H2", "C2"):
            L[k] = _zl(T + 1, B, D, dev=dev)
            if uniform:
                L[k][0].zero_()
        if train and cfg.p_out > 0:
            L["H2D"] = _zl(T, B, D, dev=dev)
        cx = _e(B, Dh, dev=dev)
        # dcnet.py:336 feeds [emb | final_hidden | h2] to the attention LSTM: the final_hidden columns (+ both biases) are
        # contracted once per sequence, every step contracts only the emb / h2 column blocks of weight_ih
        al_wih = P["al_wih"]
        pre1 = _e(B, 4 * D, dev=dev)
        ops.linear(final_hidden, al_wih[:, E:E + Dh], P["al_bih"] + P["al_bhh"], pre1, B)
        w = EditNetWeights()          # the attention kernel reads its four caption-attention pointers from this struct
        w.ca_dec_w, w.ca_dec_b = P["ca_dec_w"].data_ptr(), P["ca_dec_b"].data_ptr()
        w.ca_full_w, w.ca_full_b = P["ca_full_w"].data_ptr(), P["ca_full_b"].data_ptr()
        wref = C.byref(w)
        ws_l = ops.ws("lstm", lib.set_lstm_cell_workspace_bytes(B, D, max(K1, K2)))
        ws_c = ops.ws("cap", lib.set_caption_attention_workspace_bytes(B, Tc, max(Dh, D), Adim))
        Etab = P["E"]
        cap_stride = caps.stride(0) if ro is None else 1
        off = rng.offset
        state = None
        if ro is not None:
            state = A.SampleState(B, T, ro["start_idx"], ro["end_idx"], dev, seed=ro.get("seed"), offset=ro.get("offset", 0))
            L["LOGITS"] = _e(T, B, V, dev=dev)
            L["RAW"] = torch.empty(T, B, dtype=torch.long, device=dev)
            L["LSE"], L["LOGP"] = _e(T, B, dev=dev), _e(T, B, dev=dev)
        for t in range(T):
            bt = bts[t]
            emb = L["EMB"][t]
            tok = caps[:, t] if ro is None else state.tokens[t]
            if train and cfg.p_embed > 0:
                check(lib.set_embed_relu_dropout_f32(Etab.data_ptr(), tok.data_ptr(), cap_stride, emb.data_ptr(), E, bt, E,
                                                     Etab.shape[0], cfg.p_embed, cfg.seed, off(1, t), st),
                      "set_embed_relu_dropout_f32")
            else:
                check(lib.set_embed_relu_f32(Etab.data_ptr(), tok.data_ptr(), cap_stride, emb.data_ptr(), E, bt, E, Etab.shape[0], st),
                      "set_embed_relu_f32")
            h1 = L["H1"][t + 1]
            check(lib.set_lstm_cell_pre_train_f32(emb.data_ptr(), E, al_wih.data_ptr(), K1, E, L["H2"][t].data_ptr(), D,
                                                  al_wih[:, E + Dh:].data_ptr(), K1, D, L["H1"][t].data_ptr(),
                                                  P["al_whh"].data_ptr(), pre1.data_ptr(), 4 * D, L["C1"][t].data_ptr(),
                                                  h1.data_ptr(), L["C1"][t + 1].data_ptr(), L["G1"][t].data_ptr(), bt, D,
                                                  ws_l.data_ptr(), ws_l.numel(), st), "set_lstm_cell_pre_train_f32")
            check(lib.set_caption_attention_att2_f32(wref, enc.data_ptr(), att1_c.data_ptr(), h1.data_ptr(), mask.data_ptr(),
                                                     cx.data_ptr(), L["ALPHAC"][t].data_ptr(), L["ATT2"][t].data_ptr(), bt, Tc,
                                                     Dh, D, Adim, ws_c.data_ptr(), ws_c.numel(), st),
                  "set_caption_attention_att2_f32")
            x2 = L["X2"][t]
            ops.pack(x2, bt, [h1, cx])
            check(lib.set_lstm_cell_train_f32(x2.data_ptr(), K2, K2, L["H2"][t].data_ptr(), L["C2"][t].data_ptr(),
                                              P["ll_wih"].data_ptr(), K2, P["ll_whh"].data_ptr(), P["ll_bih"].data_ptr(),
                                              P["ll_bhh"].data_ptr(), L["H2"][t + 1].data_ptr(), L["C2"][t + 1].data_ptr(),
                                              L["G2"][t].data_ptr(), bt, D, ws_l.data_ptr(), ws_l.numel(), st),
                  "set_lstm_cell_train_f32")
            if train and cfg.p_out > 0:
                ops.dropout(L["H2"][t + 1], L["H2D"][t], bt, D, cfg.p_out, cfg.seed, off(3, t))
            if ro is not None:
                hz = L["H2D"][t] if (train and cfg.p_out > 0) else L["H2"][t + 1]
                ops.linear(hz, P["fc_w"], P["fc_b"], L["LOGITS"][t], B)
                check(lib.set_sample_pick_f32(L["LOGITS"][t].data_ptr(), V, B, V, t, T, state.end_idx, state.seed, state.offset,
                                              state.seq.data_ptr(), state.tokens[t + 1].data_ptr(), state.unfinished.data_ptr(),
                                              state.alive.data_ptr(), L["RAW"][t].data_ptr(), L["LSE"][t].data_ptr(),
                                              L["LOGP"][t].data_ptr(), st), "set_sample_pick_f32")
        hout = L["H2D"] if (train and cfg.p_out > 0) else L["H2"][1:]
        ctx.cfg, ctx.L, ctx.bts, ctx.uniform, ctx.hout = cfg, L, bts, uniform, hout
        ctx.dims = (T, B, Tc, Dh, E, D, Adim, V)
        ctx.save_for_backward(enc, mask, att1_c, caps, *params)
        if ro is not None:
            ctx.tokens = state.tokens
            ctx.mark_non_differentiable(state.seq)
            return state.seq, L["LOGP"].t()
        if uniform:
            pred_tb = _e(T, B, V, dev=dev)
            ops.linear(hout.reshape(T * B, D), P["fc_w"], P["fc_b"], pred_tb.view(T * B, V), T * B)
            return pred_tb.transpose(0, 1)
        out = _z(B, T, V, dev=dev)
        for t in range(T):
            ops.linear(hout[t], P["fc_w"], P["fc_b"], out[:, t], bts[t])
        return out

    @staticmethod
    def backward(ctx, dpred, dlogp=None):
        enc, mask, att1_c, caps = ctx.saved_tensors[:4]
        params = ctx.saved_tensors[4:]
        P = dict(zip(PARAM_NAMES, params))
        cfg, L, bts = ctx.cfg, ctx.L, ctx.bts
        T, B, Tc, Dh, E, D, Adim, V = ctx.dims
        dev = enc.device
        ops = _Ops(dev)
        lib, st = ops.lib, ops.st
        train = cfg.train
        pidx = {n: i for i, n in enumerate(PARAM_NAMES)}
        g = [None] * len(PARAM_NAMES)
        if cfg.rollout is not None:
            dl = dlogp.t().contiguous()
            dp = A.zero_padded_rows(T, B, V, dev)          # rows padded to 16 bytes: the fc contractions read them in place
            for t in range(T):
                check(lib.set_sample_logp_bwd_f32(L["LOGITS"][t].data_ptr(), V, L["LSE"][t].data_ptr(), L["RAW"][t].data_ptr(),
                                                  dl[t].data_ptr(), dp[t].data_ptr(), dp.stride(1), B, V, st),
                      "set_sample_logp_bwd_f32")
            L["LOGITS"] = None
            dp2 = dp.as_strided((T * B, V), (dp.stride(1), 1), dp.storage_offset())
        else:
            dp2 = A.score_grad_rows(dpred, bts, ctx.uniform)
        dH2D = A._dgrad(dp2, P["fc_w"]).view(T, B, D)
        need_p = ctx.needs_input_grad[6:]
        if need_p[pidx["fc_b"]]:
            g[pidx["fc_b"]] = A._bgrad(params[pidx["fc_b"]], dp2)
        if need_p[pidx["fc_w"]]:          # final now: contracted eagerly so that its all-reduce runs underneath the BPTT loop
            g[pidx["fc_w"]] = A._wgrad(params[pidx["fc_w"]], dp2, ctx.hout.reshape(T * B, D), eager=True)

        _zl = _e if ctx.uniform else _z
        DG1, DG2 = _zl(T, B, 4 * D, dev=dev), _zl(T, B, 4 * D, dev=dev)
        DATT2, DWF, DE = _zl(T, B, Adim, dev=dev), _zl(T, B, Adim, dev=dev), _zl(T, B, Tc, dev=dev)
        DCTX, DEMBRAW = _zl(T, B, Dh, dev=dev), _zl(T, B, E, dev=dev)
        datt1c = torch.zeros_like(att1_c)
        DH1, DH2 = _z(B, D, dev=dev), _z(B, D, dev=dev)
        DC1 = [_z(B, D, dev=dev), _z(B, D, dev=dev)]
        DC2 = [_z(B, D, dev=dev), _z(B, D, dev=dev)]
        demb = _e(B, E, dev=dev)
        al_wih, ll_wih = P["al_wih"], P["ll_wih"]
        sc_out = 1.0 / (1.0 - cfg.p_out) if (train and cfg.p_out > 0) else 1.0
        sc_emb = 1.0 / (1.0 - cfg.p_embed) if (train and cfg.p_embed > 0) else 1.0
        ca_full = P["ca_full_w"].reshape(-1)

        def gg(items):
            A.gemm_group([(dy, wv, dy.shape[0], wv.shape[1], dy.shape[1], out, acc) for dy, wv, out, acc in items], False, True)

        for t in range(T - 1, -1, -1):
            bt = bts[t]
            r = lambda x: _rows(x, bt)
            h1 = L["H1"][t + 1]
            if train and cfg.p_out > 0:    # h2 does not follow a ReLU: the mask is regenerated, not read off the zero pattern
                check(lib.set_dropout_bwd_philox_f32(dH2D[t].data_ptr(), D, DH2.data_ptr(), D, bt, D, cfg.p_out, cfg.seed,
                                                     rng.offset(rng.SITE_OUT, t), 1, st), "set_dropout_bwd_philox_f32")
            else:
                ops.pack(DH2, bt, [dH2D[t]], accumulate=True)
            dc2_in, dc2_out = DC2[t & 1], DC2[(t & 1) ^ 1]
            check(lib.set_lstm_cell_bwd_f32(DH2.data_ptr(), dc2_in.data_ptr(), L["G2"][t].data_ptr(), L["C2"][t].data_ptr(),
                                            L["C2"][t + 1].data_ptr(), DG2[t].data_ptr(), dc2_out.data_ptr(), bt, D, st),
                  "set_lstm_cell_bwd_f32")
            dg2, dctx = r(DG2[t]), DCTX[t]
            gg([(dg2, ll_wih[:, :D], r(DH1), True), (dg2, ll_wih[:, D:], r(dctx), False), (dg2, P["ll_whh"], r(DH2), False)])
            check(lib.set_attention_bwd_acc_f32(dctx.data_ptr(), None, L["ALPHAC"][t].data_ptr(), enc.data_ptr(),
                                                att1_c.data_ptr(), L["ATT2"][t].data_ptr(), ca_full.data_ptr(), datt1c.data_ptr(),
                                                DATT2[t].data_ptr(), DWF[t].data_ptr(), None, DE[t].data_ptr(), bt, Tc, Dh, Adim,
                                                1, 1, 0, Adim, st), "set_attention_bwd_acc_f32")
            gg([(r(DATT2[t]), P["ca_dec_w"], r(DH1), True)])
            dc1_in, dc1_out = DC1[t & 1], DC1[(t & 1) ^ 1]
            check(lib.set_lstm_cell_bwd_f32(DH1.data_ptr(), dc1_in.data_ptr(), L["G1"][t].data_ptr(), L["C1"][t].data_ptr(),
                                            L["C1"][t + 1].data_ptr(), DG1[t].data_ptr(), dc1_out.data_ptr(), bt, D, st),
                  "set_lstm_cell_bwd_f32")
            dg1 = r(DG1[t])
            # (the final_hidden columns are loop-invariant: their gradients come from sum_t dgates after the loop)
            gg([(dg1, al_wih[:, :E], r(demb), False), (dg1, al_wih[:, E + Dh:], r(DH2), True), (dg1, P["al_whh"], r(DH1), False)])
            ops.dropout_bwd(demb, L["EMB"][t], DEMBRAW[t], bt, E, sc_emb, False)

        denc = _dvalues(L["ALPHAC"], DCTX, ops)
        TB = T * B
        sdg1 = DG1.sum(0)                  # loop-invariant input: d final_hidden = (sum_t dgates) . W_ih[:, E:E+Dh]
        dFH = A.gemm(sdg1, False, al_wih[:, E:E + Dh], True, B, Dh, 4 * D)

        need = ctx.needs_input_grad[6:]                       # frozen parameters (requires_grad False) get no gradient

        def W(name, dy, x):
            if need[pidx[name]]:
                g[pidx[name]] = A._wgrad(params[pidx[name]], dy, x)

        def Bg(name, dy):
            if need[pidx[name]]:
                g[pidx[name]] = A._bgrad(params[pidx[name]], dy)

        ids = caps[:, :T].t().reshape(-1) if cfg.rollout is None else ctx.tokens[:T].reshape(-1)
        if need[pidx["E"]]:
            dE = torch.zeros_like(P["E"])
            dE.index_add_(0, ids, DEMBRAW.view(TB, E))
            g[pidx["E"]] = dE
        dg1 = DG1.view(TB, 4 * D)
        if need[pidx["al_wih"]]:           # column blocks [emb | final_hidden | h2]; the invariant one from sum_t dgates
            g[pidx["al_wih"]] = A._wgrad_blocks(params[pidx["al_wih"]], [
                (dg1, L["EMB"].view(TB, E), 0), (sdg1, L["FH"], E), (dg1, L["H2"][:T].reshape(TB, D), E + Dh)])
        W("al_whh", dg1, L["H1"][:T].reshape(TB, D))
        Bg("al_bih", dg1); Bg("al_bhh", dg1)
        dg2 = DG2.view(TB, 4 * D)
        W("ll_wih", dg2, L["X2"].view(TB, -1)); W("ll_whh", dg2, L["H2"][:T].reshape(TB, D))
        Bg("ll_bih", dg2); Bg("ll_bhh", dg2)
        W("ca_dec_w", DATT2.view(TB, Adim), L["H1"][1:].reshape(TB, D)); Bg("ca_dec_b", DATT2.view(TB, Adim))
        if need[pidx["ca_full_w"]]:
            g[pidx["ca_full_w"]] = A._colsum(DWF.view(TB, Adim)).view(1, Adim)
        if need[pidx["ca_full_b"]]:
            g[pidx["ca_full_b"]] = DE.sum().reshape(1)
        ctx.L = None
        # inputs: cfg, enc, final_hidden, mask, att1_c, caps
        return (None, denc, dFH, None, datt1c, None) + tuple(g)


def dcnet_sequence(cfg, enc, final_hidden, mask, att1_c, caps, params):
    return _DcnetSequence.apply(cfg, enc, final_hidden, mask, att1_c, caps, *params)
