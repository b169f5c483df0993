"""ORACLE (test infrastructure, not product): numpy fp32 restatement of the EditNet decode path.

This file restates, in PyTorch-free numpy, the algorithm of the reference's EditNet model
(`/root/reference/editnet.py`, `editnet_rl.py`, `adaptive_features/editnet_adaptive.py`) for
the hot path named in BASELINE.json: the per-timestep decode step (visual attention + caption
attention + copy-gate selection + two LSTM cell updates + vocabulary projection), its
per-sequence prologue and the XE / greedy loops around it.  Each function cites the reference
lines it follows.  It is pinned against golden vectors captured from the reference classes
themselves (tests/golden/*.npz, made by oracle/make_golden.py) in tests/test_oracle_golden.py.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product (show_edit_tell_amd) never does; it fails loudly when the HIP library is missing.

Weights are passed as a dict keyed by the reference's state_dict names, all
`(out, in)` row-major fp32.  `dtype` may be set to np.float64 to obtain a high-precision
statement of the same algorithm (used to decide which of two fp32 answers is closer).
"""
from __future__ import annotations

import numpy as np


def _sigmoid(x):
    # numerically-stable logistic, same value class as torch.sigmoid
    out = np.empty_like(x)
    pos = x >= 0
    out[pos] = 1.0 / (1.0 + np.exp(-x[pos]))
    ex = np.exp(x[~pos])
    out[~pos] = ex / (1.0 + ex)
    return out


def _softmax(x, axis=-1):
    m = x.max(axis=axis, keepdims=True)
    e = np.exp(x - m)
    return e / e.sum(axis=axis, keepdims=True)


def _log_softmax(x, axis=-1):
    m = x.max(axis=axis, keepdims=True)
    s = x - m
    return s - np.log(np.exp(s).sum(axis=axis, keepdims=True))


def _linear(x, P, name):
    y = x @ P[name + ".weight"].T
    b = P.get(name + ".bias")
    return y + b if b is not None else y


def cast_params(sd, dtype=np.float32):
    return {k: np.asarray(v).astype(dtype) for k, v in sd.items()}


# ---------------------------------------------------------------------------------------------
# a1  EmbeddingC.forward  (editnet.py:300-304) — eval mode: relu(W_emb[ids]); dropout is identity
# ---------------------------------------------------------------------------------------------
def embed(P, ids):
    return np.maximum(P["embed.embedding.weight"][ids], 0)


# ---------------------------------------------------------------------------------------------
# LSTMCellC.forward (editnet.py:226-244); gate order i, f, g, o
# ---------------------------------------------------------------------------------------------
def lstm_cell_c(P, prefix, x, h, c):
    gates = _linear(x, P, prefix + ".x2h") + _linear(h, P, prefix + ".h2h")
    i, f, g, o = np.split(gates, 4, axis=1)
    c_new = _sigmoid(f) * c + _sigmoid(i) * np.tanh(g)
    h_new = _sigmoid(o) * np.tanh(c_new)
    return h_new, c_new


# ---------------------------------------------------------------------------------------------
# a2  nn.LSTMCell (attention_lstm, editnet.py:468,532): x W_ih^T + b_ih + h W_hh^T + b_hh
# ---------------------------------------------------------------------------------------------
def lstm_cell(P, prefix, x, h, c):
    gates = (x @ P[prefix + ".weight_ih"].T + P[prefix + ".bias_ih"]
             + h @ P[prefix + ".weight_hh"].T + P[prefix + ".bias_hh"])
    i, f, g, o = np.split(gates, 4, axis=1)
    c_new = _sigmoid(f) * c + _sigmoid(i) * np.tanh(g)
    h_new = _sigmoid(o) * np.tanh(c_new)
    return h_new, c_new


# ---------------------------------------------------------------------------------------------
# a11 CaptionEncoderC.forward (editnet.py:319-348)
#   The reference sorts by length and runs a prefix-shrinking batch; per row that is simply
#   "update while t < len, otherwise leave H/M rows zero and keep the last h".  Output length
#   is Tmax = max(len).  mask is data-derived: (sum_d M[b,t,d]) != 0  (editnet.py:340).
# ---------------------------------------------------------------------------------------------
def caption_encoder(P, seq, seq_len):
    seq = np.asarray(seq)
    lens = np.asarray(seq_len).reshape(-1).astype(np.int64)
    B = seq.shape[0]
    dt = P["caption_encoder.affine_hn.weight"].dtype
    D = P["caption_encoder.affine_hn.weight"].shape[0]
    Tmax = int(lens.max())
    H = np.zeros((B, Tmax, D), dt)
    M = np.zeros((B, Tmax, D), dt)
    h = np.zeros((B, D), dt)
    c = np.zeros((B, D), dt)
    final = np.zeros((B, D), dt)
    emb = embed(P, seq[:, :Tmax])
    for t in range(Tmax):
        act = lens > t
        hn, cn = lstm_cell_c(P, "caption_encoder.lstm_encoder_cell", emb[act, t], h[act], c[act])
        h[act], c[act] = hn, cn
        H[act, t], M[act, t] = hn, cn
        final[act] = hn
    mask = (M.sum(2) != 0).astype(dt)
    final_hidden = np.tanh(_linear(final, P, "caption_encoder.affine_hn"))
    return H, M, final_hidden, mask


# ---------------------------------------------------------------------------------------------
# a3  CaptionAttentionC.forward (editnet.py:364-381)
# ---------------------------------------------------------------------------------------------
def caption_attention(P, H, h1, word, mask, att1_c=None):
    p = "caption_attention."
    if att1_c is None:
        att1_c = _linear(H, P, p + "cap_features_att")            # (B,T,A) loop invariant
    att2_c = _linear(h1, P, p + "cap_decoder_att")                 # (B,A)
    e = np.tanh(att1_c + att2_c[:, None, :]) @ P[p + "cap_full_att.weight"][0] + P[p + "cap_full_att.bias"][0]
    e = np.where(mask == 0, np.asarray(-1e10, e.dtype), e)
    alpha_c = _softmax(e, 1)
    context = (H * alpha_c[:, :, None]).sum(1)
    zt = _sigmoid(_linear(np.concatenate([word, h1, context], 1), P, p + "context_gate"))
    tc_in = np.concatenate([word, h1], 1)
    gated = zt * np.tanh(_linear(context, P, p + "sc_affine")) + (1 - zt) * np.tanh(_linear(tc_in, P, p + "tc_affine"))
    return gated, alpha_c


# ---------------------------------------------------------------------------------------------
# a5  SelectC.forward, hard mode (editnet.py:403-421)
#   w = alpha*onehot(j*) + onehot(j*)*(1-alpha[j*]); sel = sum_t w_t M_t ; first index on ties
# ---------------------------------------------------------------------------------------------
def select(M, alpha_c):
    B = M.shape[0]
    j = alpha_c.argmax(1)
    a = alpha_c[np.arange(B), j]
    w = a * 1 + (1 - a)              # rounds exactly as the reference's fp32 expression
    return M[np.arange(B), j] * w[:, None]


# ---------------------------------------------------------------------------------------------
# a4  VisualAttentionC.forward (editnet.py:439-447) — ReLU (not tanh), no mask, context over raw X
# ---------------------------------------------------------------------------------------------
def visual_att1(P, X):
    p = "visual_attention."
    fe = np.maximum(_linear(X, P, p + "att_embed.0"), 0)           # eval: dropout identity
    return _linear(fe, P, p + "features_att")                      # (B,R,A) loop invariant in eval


def visual_attention(P, X, h1, att1=None, return_alpha=False):
    p = "visual_attention."
    if att1 is None:
        att1 = visual_att1(P, X)
    att2 = _linear(h1, P, p + "decoder_att")
    e = np.maximum(att1 + att2[:, None, :], 0) @ P[p + "full_att.weight"][0] + P[p + "full_att.bias"][0]
    alpha = _softmax(e, 1)
    ctx = (X * alpha[:, :, None]).sum(1)
    return (ctx, alpha) if return_alpha else ctx


# ---------------------------------------------------------------------------------------------
# a4' adaptive VisualAttentionC.forward (adaptive_features/editnet_adaptive.py:438-457)
#   valid-region mask from X.sum(2)!=0; att_embed only on valid rows (zero elsewhere);
#   att_masks = att_embed.sum(2)!=0; masked_fill(-1e10); context over first Lmax regions.
# ---------------------------------------------------------------------------------------------
def visual_attention_adaptive(P, X, h1, return_alpha=False):
    p = "visual_attention."
    tmp_mask = X.sum(2) != 0
    att_len = tmp_mask.sum(1)
    Lmax = int(att_len.max())
    B = X.shape[0]
    D = P[p + "att_embed.0.weight"].shape[0]
    # pack_padded_sequence takes the FIRST att_len[b] rows of each sample (not the masked rows)
    fe = np.zeros((B, Lmax, D), X.dtype)
    for b in range(B):
        n = int(att_len[b])
        fe[b, :n] = np.maximum(_linear(X[b, :n], P, p + "att_embed.0"), 0)
    att_masks = fe.sum(2) != 0
    att1 = _linear(fe, P, p + "features_att")
    att2 = _linear(h1, P, p + "decoder_att")
    e = np.maximum(att1 + att2[:, None, :], 0) @ P[p + "full_att.weight"][0] + P[p + "full_att.bias"][0]
    e = np.where(att_masks == 0, np.asarray(-1e10, e.dtype), e)
    alpha = _softmax(e, 1)
    L2 = int(att_masks.sum(1).max())
    alpha = alpha[:, :L2]
    ctx = (X[:, :L2] * alpha[:, :, None]).sum(1)
    return (ctx, alpha) if return_alpha else ctx


# ---------------------------------------------------------------------------------------------
# a6  CopyLSTMCellC.forward (editnet.py:265-285)
# ---------------------------------------------------------------------------------------------
def copy_lstm(P, x, h2, c2, c_memory):
    p = "copy_lstm."
    gates = _linear(x, P, p + "x2h") + _linear(h2, P, p + "h2h")
    i, f, g, o = np.split(gates, 4, axis=1)
    c_new = _sigmoid(f) * c2 + _sigmoid(i) * np.tanh(g)
    copy_gate = _sigmoid(_linear(c_new, P, p + "gate_cnew") + _linear(c_memory, P, p + "gate_cmem"))
    adaptive = copy_gate * c_memory + (1 - copy_gate) * c_new
    h_new = _sigmoid(o) * np.tanh(adaptive)
    return h_new, adaptive


# ---------------------------------------------------------------------------------------------
# one decode step a1..a7 (editnet.py:527-545 / editnet_rl.py:505-513), eval mode
# ---------------------------------------------------------------------------------------------
class SeqState:
    """Per-sequence invariants + recurrent state (rows = the active batch prefix)."""

    def __init__(self, P, X, prev, prevlen, image_mean=None, adaptive=False):
        self.P, self.X, self.adaptive = P, X, adaptive
        self.H, self.M, self.final_hidden, self.mask = caption_encoder(P, prev, prevlen)
        self.image_mean = X.mean(1, dtype=X.dtype) if image_mean is None else image_mean
        self.att1 = None if adaptive else visual_att1(P, X)
        self.att1_c = _linear(self.H, P, "caption_attention.cap_features_att")
        B, D = X.shape[0], self.H.shape[2]
        z = lambda: np.zeros((B, D), X.dtype)
        self.h1, self.c1, self.h2, self.c2 = z(), z(), z(), z()


def step(S: SeqState, it, bt=None, trace=None):
    """Advance rows [:bt] by one timestep with input tokens `it` (bt,); returns logits (bt,V)."""
    P = S.P
    bt = S.X.shape[0] if bt is None else bt
    emb = embed(P, it)
    x1 = np.concatenate([emb, S.final_hidden[:bt], S.h2[:bt], S.image_mean[:bt]], 1)
    h1, c1 = lstm_cell(P, "attention_lstm", x1, S.h1[:bt], S.c1[:bt])
    attend_cap, alpha_c = caption_attention(P, S.H[:bt], h1, emb, S.mask[:bt], S.att1_c[:bt])
    if S.adaptive:
        attend_img, alpha = visual_attention_adaptive(P, S.X[:bt], h1, return_alpha=True)
    else:
        attend_img, alpha = visual_attention(P, S.X[:bt], h1, S.att1[:bt], return_alpha=True)
    x2 = np.concatenate([h1, attend_cap, attend_img], 1)
    sel = select(S.M[:bt], alpha_c)
    h2, c2 = copy_lstm(P, x2, S.h2[:bt], S.c2[:bt], sel)
    logits = _linear(h2, P, "fc")                                   # eval: dropout identity
    S.h1, S.c1, S.h2, S.c2 = h1, c1, h2, c2
    if trace is not None:
        trace.append(dict(emb=emb, h1=h1, c1=c1, attend_cap=attend_cap, alpha_c=alpha_c,
                          attend_img=attend_img, alpha=alpha, sel=sel, h2=h2, c2=c2, logits=logits))
    return logits


# ---------------------------------------------------------------------------------------------
# a9  DecoderC.forward, XE teacher-forced loop (editnet.py:479-548), eval mode, use_ss=False
# ---------------------------------------------------------------------------------------------
def xe_forward(P, X, caps, caplens, prev, prevlen, image_mean=None, adaptive=False, trace=None):
    caplens = np.asarray(caplens).reshape(-1)
    # torch.sort(descending=True) is not stable; callers that need the exact permutation pass
    # distinct lengths or compare through sort_ind.  We use a stable descending order.
    sort_ind = np.argsort(-caplens, kind="stable")
    caplens = caplens[sort_ind]
    X, caps, prev, prevlen = X[sort_ind], caps[sort_ind], prev[sort_ind], np.asarray(prevlen)[sort_ind]
    if image_mean is not None:
        image_mean = image_mean[sort_ind]
    decode_lengths = (caplens - 1).tolist()
    S = SeqState(P, X, prev, prevlen, image_mean, adaptive)
    V = P["fc.weight"].shape[0]
    B = X.shape[0]
    pred = np.zeros((B, max(decode_lengths), V), X.dtype)
    for t in range(max(decode_lengths)):
        bt = sum(l > t for l in decode_lengths)
        pred[:bt, t] = step(S, caps[:bt, t], bt, trace)
    return pred, caps, decode_lengths, sort_ind


# ---------------------------------------------------------------------------------------------
# a10 DecoderC.forward, free-running greedy loop (editnet_rl.py:485-549), sample_max=True
# ---------------------------------------------------------------------------------------------
def greedy_decode(P, start_idx, end_idx, prev, prevlen, X, max_len=18, image_mean=None,
                  adaptive=False, trace=None):
    B = X.shape[0]
    seq = np.zeros((B, max_len), np.int64)
    seq_logp = np.zeros((B, max_len), X.dtype)
    it = np.full((B,), start_idx, np.int64)
    S = SeqState(P, X, prev, prevlen, image_mean, adaptive)
    unfinished = None
    for t in range(max_len + 1):
        logits = step(S, it, None, trace)
        logp = _log_softmax(logits, 1)
        if t == max_len:
            break
        it = logp.argmax(1)                       # first index on ties, as torch.max on CPU
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


# ---------------------------------------------------------------------------------------------
# a12 losses on the path's output
# ---------------------------------------------------------------------------------------------
def xe_loss(pred, caps_sorted, decode_lengths):
    """CrossEntropyLoss(mean) over pack_padded rows (editnet.py:571-577)."""
    tot, n = 0.0, 0
    for b, L in enumerate(decode_lengths):
        lp = _log_softmax(pred[b, :L].astype(np.float64), 1)
        tot -= lp[np.arange(L), caps_sorted[b, 1:L + 1]].sum()
        n += L
    return tot / n


def reward_criterion(seq_logp, seq, reward):
    """RewardCriterion.forward (editnet_rl.py:557-573)."""
    mask = (seq > 0).astype(seq_logp.dtype)
    mask = np.concatenate([np.ones((mask.shape[0], 1), mask.dtype), mask[:, :-1]], 1)
    return (-(seq_logp * reward * mask).sum() / mask.sum())
