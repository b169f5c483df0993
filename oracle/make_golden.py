"""ORACLE tooling (authoring container only): capture golden vectors from the reference itself.

    python -m oracle.make_golden            # rewrites tests/golden/*.npz

Loads the reference's model classes out of /root/reference (oracle/ref_slice.py), loads the
deterministic synthetic weights of each case in oracle/cases.py, runs the reference's own
`forward`s on CPU in eval mode, and stores ONLY input-free data: the reference's outputs (full
tensors for the reduced-dimension cases; token ids, log-probs, top-8 logits, a 64-column logit
slice, log-sum-exp, attention rows and state slices for the full-size cases).  Inputs and weights
are regenerated from seeds by the tests, so nothing of the reference travels except numbers it
computed.  The reference cannot run on the GPU box; these files are what pins parity there.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from oracle import cases, ref_slice  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
T_ = torch.from_numpy


def _np(x):
    return x.detach().cpu().numpy()


def _summ_logits(logits, V):
    """Compact, order-independent summary of a (N,V) logit block."""
    lg = logits.astype(np.float64)
    idx = np.argsort(-lg, axis=1, kind="stable")[:, :8]
    top = np.take_along_axis(logits, idx, 1)
    m = lg.max(1, keepdims=True)
    lse = (m[:, 0] + np.log(np.exp(lg - m).sum(1))).astype(np.float32)
    return dict(top_idx=idx.astype(np.int32), top_val=top.astype(np.float32), lse=lse,
                cols=logits[:, cases.logit_slice_cols(V)].astype(np.float32))


def _trace_greedy_editnet(dec, wm, prev, plen, X, image_mean=None):
    """Re-run the reference's sub-modules step by step exactly as editnet_rl.py:497-547 does, to
    capture per-step intermediates (the forward itself does not expose them)."""
    B = X.shape[0]
    it = torch.full((B,), wm["<start>"], dtype=torch.long)
    h1, c1 = dec.init_hidden_state(B)
    h2, c2 = dec.init_hidden_state(B)
    H, M, fh, mask = dec.caption_encoder(prev, plen)
    mean = X.mean(1) if image_mean is None else image_mean
    steps = []
    unfinished = None
    for t in range(19):
        emb = dec.embed(it)
        h1, c1 = dec.attention_lstm(torch.cat([emb, fh, h2, mean], 1), (h1, c1))
        ac, alpha_c = dec.caption_attention(H, h1, emb, mask)
        ai = dec.visual_attention(X, h1)
        sel = dec.select(M, alpha_c)
        h2, c2 = dec.copy_lstm(torch.cat([h1, ac, ai], 1), (h2, c2), sel)
        pt = dec.fc(dec.dropout(h2))
        steps.append(dict(h1=_np(h1), c1=_np(c1), h2=_np(h2), c2=_np(c2), attend_cap=_np(ac), alpha_c=_np(alpha_c),
                          attend_img=_np(ai), sel=_np(sel), logits=_np(pt)))
        if t == 18:
            break
        lp = torch.log_softmax(pt, 1)
        _, it = lp.max(1)
        it = it.clone()
        it[it == wm["<end>"]] = 0
        unfinished = (it > 0) if t == 0 else unfinished * (it > 0)
        it = it * unfinished.type_as(it)
        if unfinished.sum() == 0:
            break
    return steps


def make_editnet(name, adaptive=False):
    d = cases.build_editnet(name)
    c, wm = d["case"], d["wm"]
    small = c["D"] < 1024
    big = c["B"] > 8          # big-batch cases keep summaries only (fixture size)
    xe_cls = (ref_slice.editnet_adaptive() if adaptive else ref_slice.editnet_xe())["DecoderC"]
    rl_cls = ref_slice.editnet_rl()
    if adaptive:
        # the RL loop is the same for adaptive features except for the visual attention / image_mean;
        # reuse the editnet_rl loop text but with the adaptive VisualAttentionC class.
        rl_cls = dict(rl_cls)
    args = (wm, c["D"], c["D"], c["D"], c["A"], c["F"])
    dec_xe = ref_slice.load_state(xe_cls(*args), d["sd"]).eval()
    dec_rl = ref_slice.load_state(rl_cls["DecoderC"](*args), d["sd"]).eval()
    if adaptive:
        dec_rl.visual_attention = dec_xe.visual_attention
    X, prev, plen, caps, clen = (T_(d[k]) for k in ("X", "prev", "plen", "caps", "clen"))
    out = {}
    with torch.no_grad():
        # ---- prologue (a11)
        H, M, fh, mask = dec_xe.caption_encoder(prev, plen)
        if big:
            out.update(enc_H_sum=_np(H).astype(np.float64).sum(2), enc_M_sum=_np(M).astype(np.float64).sum(2),
                       enc_final=_np(fh)[:, :32], enc_mask=_np(mask))
        else:
            out.update(enc_H=_np(H), enc_M=_np(M), enc_final=_np(fh), enc_mask=_np(mask))
        # ---- per-operator probes (a1..a7)
        p = {k: T_(v) for k, v in d["probe"].items()}
        emb = dec_xe.embed(p["ids"])
        mean = T_(d["image_mean"]) if adaptive else X.mean(1)
        h1n, c1n = dec_xe.attention_lstm(torch.cat([emb, fh, p["h2"], mean], 1), (p["h1"], p["c1"]))
        gated, alpha_c = dec_xe.caption_attention(H, p["h1"], p["word"], mask)
        vis = dec_xe.visual_attention(X, p["h1"])
        sel = dec_xe.select(M, alpha_c)
        h2n, c2n = dec_xe.copy_lstm(torch.cat([p["h1"], gated, vis], 1), (p["h2"], p["c2"]), sel)
        logits = dec_xe.fc(h2n)
        ops = dict(op_embed=emb, op_h1=h1n, op_c1=c1n, op_gated=gated, op_alpha_c=alpha_c, op_vis=vis, op_sel=sel,
                   op_h2=h2n, op_c2=c2n)
        if not big:
            out.update({k: _np(v) for k, v in ops.items()})
        if big:
            pass
        elif small:
            out["op_logits"] = _np(logits)
        else:
            out.update({"op_logits_" + k: v for k, v in _summ_logits(_np(logits), c["V"]).items()})
        # ---- XE forward (a9), eval mode, no scheduled sampling
        if adaptive:
            pred, caps_s, dl, sort_ind, gd_fh, last_h = dec_xe(X, T_(d["image_mean"]), caps, clen, prev, plen, False, 0.0)
            out.update(xe_gd_final=_np(gd_fh), xe_last_hidden=_np(last_h))
        else:
            pred, caps_s, dl, sort_ind = dec_xe(X, caps, clen, prev, plen, False, 0.0)
        out.update(xe_sort_ind=_np(sort_ind), xe_decode_lengths=np.asarray(dl, np.int64), xe_caps_sorted=_np(caps_s))
        pred = _np(pred)
        if small:
            out["xe_pred"] = pred
        else:
            flat = pred.reshape(-1, c["V"])
            out.update({"xe_pred_" + k: v.reshape(pred.shape[0], pred.shape[1], -1).squeeze(-1)
                        if v.ndim == 1 else v.reshape(pred.shape[0], pred.shape[1], -1)
                        for k, v in _summ_logits(flat, c["V"]).items() if not (big and k == "cols")})
        # XE loss through the reference's own criterion path (editnet.py:571-577)
        from torch.nn.utils.rnn import pack_padded_sequence
        sc = pack_padded_sequence(T_(pred), dl, batch_first=True).data
        tg = pack_padded_sequence(caps_s[:, 1:], dl, batch_first=True).data
        out["xe_loss"] = np.float64(torch.nn.CrossEntropyLoss()(sc.double(), tg).item())
    # ---- a13: gradients of the XE loss (eval-mode dropout so they are deterministic), reference autograd
    # (big-batch cases keep norms + 64-element slices only: config 1's backward at B=128)
    if not adaptive or small:
        from torch.nn.utils.rnn import pack_padded_sequence as _pps
        dec_xe.zero_grad()
        if adaptive:     # the adaptive train() adds MSE(decoder_last_hidden, gd_final_hidden) (editnet_adaptive.py:594-596)
            predg, caps_g, dlg, _, gdg, lastg = dec_xe(X, T_(d["image_mean"]), caps, clen, prev, plen, False, 0.0)
        else:
            predg, caps_g, dlg, _ = dec_xe(X, caps, clen, prev, plen, False, 0.0)
        lossg = torch.nn.CrossEntropyLoss()(_pps(predg, dlg, batch_first=True).data,
                                            _pps(caps_g[:, 1:], dlg, batch_first=True).data)
        if adaptive:
            lossg = lossg + torch.nn.MSELoss()(lastg, gdg)
        lossg.backward()
        out["grad_loss"] = np.float64(lossg.item())
        for k_, p_ in dec_xe.named_parameters():
            g_ = _np(p_.grad)
            out["gradnorm." + k_] = np.float64(np.sqrt((g_.astype(np.float64) ** 2).sum()))
            if small:
                out["grad." + k_] = g_
            else:
                out["gradslice." + k_] = g_.reshape(-1)[:: max(1, g_.size // 64)][:64].copy()
        dec_xe.zero_grad()
    with torch.no_grad():
        # ---- greedy decode (a10)
        if adaptive:
            # free-running loop with the supplied image_mean: restate editnet_rl.py:497-547 over the
            # adaptive sub-modules (the reference has no RL script for adaptive features)
            seq = logp = None
        else:
            seq, logp = dec_rl(wm, prev, plen, X, True, False)
            out.update(greedy_seq=_np(seq), greedy_logp=_np(logp))
            rc = rl_cls["RewardCriterion"]()
            reward = T_(np.repeat(cases.synth.uniform(c["iseed"], "reward", (c["B"], 1), -1, 1), 18, 1))
            out["reward_loss"] = np.float64(rc(logp.double(), seq, reward.double()).item())
        steps = _trace_greedy_editnet(dec_rl, wm, prev, plen, X, T_(d["image_mean"]) if adaptive else None)
        out["greedy_nsteps"] = np.int64(len(steps))
        lg = np.stack([s["logits"] for s in steps])                  # (S,B,V)
        srt = np.sort(lg, axis=2)
        out["greedy_margin"] = (srt[:, :, -1] - srt[:, :, -2]).astype(np.float32)
        if small:
            for k in steps[0]:
                out["greedy_" + k] = np.stack([s[k] for s in steps])
        else:
            summ = _summ_logits(lg.reshape(-1, c["V"]), c["V"])
            for k, v in summ.items():
                if big and k == "cols":
                    continue
                out["greedy_logits_" + k] = v.reshape(lg.shape[0], lg.shape[1], *v.shape[1:])
            keepB = min(c["B"], 8)
            for k in ("alpha_c",):
                out["greedy_" + k] = np.stack([s[k] for s in steps])[:, :keepB]
            for k in ("h1", "c1", "h2", "c2", "attend_cap", "attend_img", "sel"):
                full = np.stack([s[k] for s in steps])
                out["greedy_" + k + "_slice"] = full[:, :keepB, :32].astype(np.float32)
                out["greedy_" + k + "_sum"] = full.astype(np.float64).sum(2)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    sz = os.path.getsize(os.path.join(OUT, name + ".npz"))
    print("%-28s %8.1f KiB   greedy margin min %.2e  steps %d" % (name, sz / 1024, out["greedy_margin"].min(), len(steps)))


def _trace_greedy_dcnet(dae, wm, prev, plen):
    B = prev.shape[0]
    it = torch.full((B,), wm["<start>"], dtype=torch.long)
    h1, c1 = dae.init_hidden_state(B)
    h2, c2 = dae.init_hidden_state(B)
    enc, fh, mask = dae.caption_encoder(prev, plen)
    steps, unfinished = [], None
    for t in range(19):
        emb = dae.embed(it)
        h1, c1 = dae.attention_lstm(torch.cat([emb, fh, h2], 1), (h1, c1))
        ac = dae.caption_attention(enc, h1, mask)
        h2, c2 = dae.language_lstm(torch.cat([h1, ac], 1), (h2, c2))
        pt = dae.fc(dae.dropout(h2))
        steps.append(dict(h1=_np(h1), c1=_np(c1), h2=_np(h2), c2=_np(c2), attend_cap=_np(ac), logits=_np(pt)))
        if t == 18:
            break
        _, it = torch.log_softmax(pt, 1).max(1)
        it = it.clone()
        it[it == wm["<end>"]] = 0
        unfinished = (it > 0) if t == 0 else unfinished * (it > 0)
        it = it * unfinished.type_as(it)
        if unfinished.sum() == 0:
            break
    return steps


def make_dcnet(name):
    d = cases.build_dcnet(name)
    c, wm = d["case"], d["wm"]
    small = c["D"] < 1024
    args = (wm, None, c["D"], c["A"], c["C"], c["E"])
    dae_xe = ref_slice.load_state(ref_slice.dcnet_xe()["DAE"](*args), d["sd"]).eval()
    dae_rl = ref_slice.load_state(ref_slice.dcnet_rl()["DAE"](*args), d["sd"]).eval()
    prev, plen, caps, clen = (T_(d[k]) for k in ("prev", "plen", "caps", "clen"))
    out = {}
    with torch.no_grad():
        enc, fh, mask = dae_xe.caption_encoder(prev, plen)
        out.update(enc_out=_np(enc), enc_final=_np(fh), enc_mask=_np(mask))
        out["op_ctx"] = _np(dae_xe.caption_attention(enc, T_(d["probe"]["h1"]), mask))
        pred, caps_s, dl, sort_ind = dae_xe(caps, clen, prev, plen)
        out.update(xe_sort_ind=_np(sort_ind), xe_decode_lengths=np.asarray(dl, np.int64), xe_caps_sorted=_np(caps_s))
        pred = _np(pred)
        if small:
            out["xe_pred"] = pred
        else:
            flat = pred.reshape(-1, c["V"])
            out.update({"xe_pred_" + k: v.reshape(pred.shape[0], pred.shape[1], -1).squeeze(-1)
                        if v.ndim == 1 else v.reshape(pred.shape[0], pred.shape[1], -1)
                        for k, v in _summ_logits(flat, c["V"]).items()})
    # ---- a13 for DCNet: gradients of the XE loss (dcnet.py:353-402: CrossEntropyLoss over the packed rows),
    # eval-mode dropout so they are deterministic, reference autograd
    from torch.nn.utils.rnn import pack_padded_sequence as _pps
    dae_xe.zero_grad()
    predg, caps_g, dlg, _ = dae_xe(caps, clen, prev, plen)
    lossg = torch.nn.CrossEntropyLoss()(_pps(predg, dlg, batch_first=True).data,
                                        _pps(caps_g[:, 1:], dlg, batch_first=True).data)
    lossg.backward()
    out["grad_loss"] = np.float64(lossg.item())
    for k_, p_ in dae_xe.named_parameters():
        g_ = _np(p_.grad)
        out["gradnorm." + k_] = np.float64(np.sqrt((g_.astype(np.float64) ** 2).sum()))
        if small:
            out["grad." + k_] = g_
        else:
            out["gradslice." + k_] = g_.reshape(-1)[:: max(1, g_.size // 64)][:64].copy()
    dae_xe.zero_grad()
    with torch.no_grad():
        seq, logp = dae_rl(wm, prev, plen, True, False)
        out.update(greedy_seq=_np(seq), greedy_logp=_np(logp))
        steps = _trace_greedy_dcnet(dae_rl, wm, prev, plen)
        out["greedy_nsteps"] = np.int64(len(steps))
        lg = np.stack([s["logits"] for s in steps])
        srt = np.sort(lg, axis=2)
        out["greedy_margin"] = (srt[:, :, -1] - srt[:, :, -2]).astype(np.float32)
        if small:
            for k in steps[0]:
                out["greedy_" + k] = np.stack([s[k] for s in steps])
        else:
            summ = _summ_logits(lg.reshape(-1, c["V"]), c["V"])
            for k, v in summ.items():
                out["greedy_logits_" + k] = v.reshape(lg.shape[0], lg.shape[1], *v.shape[1:])
            for k in ("h1", "c1", "h2", "c2", "attend_cap"):
                full = np.stack([s[k] for s in steps])
                out["greedy_" + k + "_slice"] = full[:, :, :32].astype(np.float32)
                out["greedy_" + k + "_sum"] = full.astype(np.float64).sum(2)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    sz = os.path.getsize(os.path.join(OUT, name + ".npz"))
    print("%-28s %8.1f KiB   greedy margin min %.2e  steps %d" % (name, sz / 1024, out["greedy_margin"].min(), len(steps)))


def main(argv):
    assert ref_slice.have_reference(), "needs /root/reference (authoring container only)"
    torch.manual_seed(0)
    os.makedirs(OUT, exist_ok=True)
    want = set(argv[1:])
    for name in cases.EDITNET_CASES:
        if not want or name in want:
            make_editnet(name)
    for name in cases.ADAPTIVE_CASES:
        if not want or name in want:
            make_editnet(name, adaptive=True)
    for name in cases.DCNET_CASES:
        if not want or name in want:
            make_dcnet(name)


if __name__ == "__main__":
    main(sys.argv)
