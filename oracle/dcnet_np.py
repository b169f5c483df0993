"""ORACLE (test infrastructure, not product): numpy fp32 restatement of the DCNet (DAE) decode path.

Follows `/root/reference/dcnet.py` (XE loop) and `dcnet_rl.py` (free-running greedy loop).
DCNet is text-only: it takes no image features (`dcnet.py:303`, `dcnet.py:336-338`).
Pinned against golden vectors captured from the reference classes (tests/golden/dcnet_*.npz).
Same import rules as oracle/editnet_np.py: tests/, smoke() and bench.py's cpu_baseline only.
"""
from __future__ import annotations

import numpy as np

from .editnet_np import _linear, _log_softmax, _sigmoid, _softmax, cast_params, lstm_cell  # noqa: F401


def embed(P, ids):
    """Embedding.forward, non-GloVe branch (dcnet.py:199-206), eval mode."""
    return np.maximum(P["embed.embedding.weight"][ids], 0)


def _lstm_dir(P, sfx, x, lens, reverse):
    """One direction of nn.LSTM over a packed batch (dcnet.py:217,233): per row, run over its own
    valid positions only (forward 0..len-1, reverse len-1..0); padded outputs stay zero."""
    p = "caption_encoder.lstm_encoder."
    W_ih, W_hh = P[p + "weight_ih_l0" + sfx], P[p + "weight_hh_l0" + sfx]
    b = P[p + "bias_ih_l0" + sfx] + P[p + "bias_hh_l0" + sfx]
    B, T, _ = x.shape
    C = W_hh.shape[1]
    h = np.zeros((B, C), x.dtype)
    c = np.zeros((B, C), x.dtype)
    out = np.zeros((B, T, C), x.dtype)
    steps = range(T - 1, -1, -1) if reverse else range(T)
    for t in steps:
        act = lens > t
        if not act.any():
            continue
        gates = x[act, t] @ W_ih.T + h[act] @ W_hh.T + b
        i, f, g, o = np.split(gates, 4, axis=1)
        cn = _sigmoid(f) * c[act] + _sigmoid(i) * np.tanh(g)
        hn = _sigmoid(o) * np.tanh(cn)
        h[act], c[act] = hn, cn
        out[act, t] = hn
    return out, h


def caption_encoder(P, src, src_len):
    """CaptionEncoder.forward (dcnet.py:220-243): BiLSTM on packed sequence, outputs padded to
    Tmax = max(len); mask = outputs.sum(2) != 0; final_hidden = tanh(concat([h_fwd, h_bwd]))."""
    lens = np.asarray(src_len).reshape(-1).astype(np.int64)
    Tmax = int(lens.max())
    x = embed(P, np.asarray(src)[:, :Tmax])
    of, hf = _lstm_dir(P, "", x, lens, False)
    ob, hb = _lstm_dir(P, "_reverse", x, lens, True)
    outputs = np.concatenate([of, ob], 2)
    mask = (outputs.sum(2) != 0).astype(x.dtype)
    final_hidden = np.tanh(_linear(np.concatenate([hf, hb], 1), P, "caption_encoder.concat"))
    return outputs, final_hidden, mask


def caption_attention(P, feats, h1, mask, att1_c=None, return_alpha=False):
    """CaptionAttention.forward (dcnet.py:254-270): additive attention, no gating."""
    p = "caption_attention."
    if att1_c is None:
        att1_c = _linear(feats, P, p + "cap_features_att")
    att2_c = _linear(h1, P, p + "cap_decoder_att")
    e = np.tanh(att1_c + att2_c[:, None, :]) @ P[p + "cap_full_att.weight"][0] + P[p + "cap_full_att.bias"][0]
    e = np.where(mask == 0, np.asarray(-1e10, e.dtype), e)
    alpha = _softmax(e, 1)
    ctx = (feats * alpha[:, :, None]).sum(1)
    return (ctx, alpha) if return_alpha else ctx


class SeqState:
    def __init__(self, P, prev, prevlen):
        self.P = P
        self.enc, self.final_hidden, self.mask = caption_encoder(P, prev, prevlen)
        self.att1_c = _linear(self.enc, P, "caption_attention.cap_features_att")
        B, D = self.enc.shape[0], P["attention_lstm.weight_hh"].shape[1]
        z = lambda: np.zeros((B, D), self.enc.dtype)
        self.h1, self.c1, self.h2, self.c2 = z(), z(), z(), z()


def step(S: SeqState, it, bt=None, trace=None):
    """dcnet.py:336-347 / dcnet_rl.py:306-312, eval mode."""
    P = S.P
    bt = S.enc.shape[0] if bt is None else bt
    emb = embed(P, it)
    x1 = np.concatenate([emb, S.final_hidden[:bt], S.h2[:bt]], 1)
    h1, c1 = lstm_cell(P, "attention_lstm", x1, S.h1[:bt], S.c1[:bt])
    attend_cap, alpha = caption_attention(P, S.enc[:bt], h1, S.mask[:bt], S.att1_c[:bt], True)
    x2 = np.concatenate([h1, attend_cap], 1)
    h2, c2 = lstm_cell(P, "language_lstm", x2, S.h2[:bt], S.c2[:bt])
    logits = _linear(h2, P, "fc")
    S.h1, S.c1, S.h2, S.c2 = h1, c1, h2, c2
    if trace is not None:
        trace.append(dict(h1=h1, c1=c1, attend_cap=attend_cap, alpha_c=alpha, h2=h2, c2=c2, logits=logits))
    return logits


def xe_forward(P, caps, caplens, prev, prevlen, trace=None):
    """DAE.forward (dcnet.py:303-350), eval mode."""
    caplens = np.asarray(caplens).reshape(-1)
    sort_ind = np.argsort(-caplens, kind="stable")
    caplens = caplens[sort_ind]
    caps, prev, prevlen = caps[sort_ind], prev[sort_ind], np.asarray(prevlen)[sort_ind]
    decode_lengths = (caplens - 1).tolist()
    S = SeqState(P, prev, prevlen)
    V = P["fc.weight"].shape[0]
    pred = np.zeros((caps.shape[0], max(decode_lengths), V), S.enc.dtype)
    for t in range(max(decode_lengths)):
        bt = sum(l > t for l in decode_lengths)
        pred[:bt, t] = step(S, caps[:bt, t], bt, trace)
    return pred, caps, decode_lengths, sort_ind


def greedy_decode(P, start_idx, end_idx, prev, prevlen, max_len=18, trace=None):
    """DAE.forward, RL variant with sample_max=True (dcnet_rl.py:286-346)."""
    B = prev.shape[0]
    S = SeqState(P, prev, prevlen)
    seq = np.zeros((B, max_len), np.int64)
    seq_logp = np.zeros((B, max_len), S.enc.dtype)
    it = np.full((B,), start_idx, np.int64)
    unfinished = None
    for t in range(max_len + 1):
        logp = _log_softmax(step(S, it, None, trace), 1)
        if t == max_len:
            break
        it = logp.argmax(1)
        sample_logp = logp[np.arange(B), it]
        it = it.copy()
        it[it == end_idx] = 0
        unfinished = (it > 0) if t == 0 else (unfinished & (it > 0))
        it = it * unfinished
        seq[:, t] = it
        seq_logp[:, t] = sample_logp
        if unfinished.sum() == 0:
            break
    return seq, seq_logp
