"""ORACLE / TEST INFRASTRUCTURE — never imported by the product package.

torch-CPU restatement of the reference's free-running EditNet decode *as written*
(`/root/reference/editnet_rl.py:485-549` and the sub-module forwards `:231-447`): nothing is
hoisted out of the timestep loop — `att_embed` + `features_att` over all 36 regions and
`cap_features_att` over the previous caption are recomputed every timestep, the `(B,R,F)·alpha`
product is materialised, the encoder runs on the length-sorted, prefix-shrinking batch — and every
contraction is an `aten::addmm` / `aten::linear` on the CPU BLAS torch was built with.  This is
the op stream the reference PyTorch CPU path dispatches, so timing it (bench.py `cpu_baseline`)
is the faithful "reference on the host cores" figure; `oracle/editnet_np.py` (numpy/OpenBLAS,
loop invariants hoisted) is the algorithmically cheaper port kept beside it.

Written functionally over a plain {state_dict key: tensor} mapping (no nn.Module, no reference
source): each function cites the lines it restates.  Pinned by tests/test_oracle_golden.py
against the golden vectors captured from the reference's own classes.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def params_from_numpy(sd_np):
    """fp32 CPU tensors from the synthetic / checkpoint numpy state dict."""
    return {k: torch.from_numpy(v.copy()).float() for k, v in sd_np.items()}


def _lin(P, name, x):
    return F.linear(x, P[name + ".weight"], P[name + ".bias"])


def embed(P, ids):
    """EmbeddingC.forward, eval mode (editnet_rl.py:305-309): dropout is the identity"""
    return torch.relu(F.embedding(ids, P["embed.embedding.weight"]))


def lstm_cell_c(P, pre, x, h, c):
    """LSTMCellC.forward (editnet_rl.py:231-248): gates = x2h(x) + h2h(h), chunk order i, f, g, o"""
    i, f, g, o = (_lin(P, pre + ".x2h", x) + _lin(P, pre + ".h2h", h)).chunk(4, 1)
    c_new = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
    return torch.sigmoid(o) * torch.tanh(c_new), c_new


def caption_encoder(P, seq, seq_len):
    """CaptionEncoderC.forward (editnet_rl.py:325-353): sort by length, shrink the batch prefix, scatter h / c
    per position, final_hidden = tanh(affine(last h)), mask from the data, undo the sort"""
    B = seq.shape[0]
    D = P["caption_encoder.affine_hn.weight"].shape[0]
    lens, order = seq_len.squeeze(1).sort(dim=0, descending=True)
    inv = order.clone()
    inv[order] = torch.arange(B)
    lens = lens.tolist()
    emb = embed(P, seq[order])
    tmax = max(lens)
    H = torch.zeros(B, tmax, D)
    M = torch.zeros(B, tmax, D)
    last = torch.zeros(B, D)
    h, c = torch.zeros(B, D), torch.zeros(B, D)
    for t in range(tmax):
        bt = sum(l > t for l in lens)
        h, c = lstm_cell_c(P, "caption_encoder.lstm_encoder_cell", emb[:bt, t], h[:bt], c[:bt])
        H[:bt, t] = h
        M[:bt, t] = c
        last[:bt] = h
    mask = (M.sum(2) != 0).float()
    final_hidden = torch.tanh(_lin(P, "caption_encoder.affine_hn", last))
    return H[inv], M[inv], final_hidden[inv], mask[inv]


def attention_lstm(P, x, h, c):
    """nn.LSTMCell (editnet_rl.py:467,504): x W_ih^T + b_ih + h W_hh^T + b_hh, order i, f, g, o"""
    g = F.linear(x, P["attention_lstm.weight_ih"], P["attention_lstm.bias_ih"]) + \
        F.linear(h, P["attention_lstm.weight_hh"], P["attention_lstm.bias_hh"])
    i, f, gg, o = g.chunk(4, 1)
    c_new = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
    return torch.sigmoid(o) * torch.tanh(c_new), c_new


def caption_attention(P, H, h1, word, mask):
    """CaptionAttentionC.forward (editnet_rl.py:369-388), att1_c recomputed every call as written"""
    att1 = _lin(P, "caption_attention.cap_features_att", H)
    att2 = _lin(P, "caption_attention.cap_decoder_att", h1)
    e = _lin(P, "caption_attention.cap_full_att", torch.tanh(att1 + att2.unsqueeze(1))).squeeze(2)
    e = e.masked_fill(mask == 0, -1e10)
    alpha = F.softmax(e, dim=1)
    ctx = (H * alpha.unsqueeze(2)).sum(1)
    zt = torch.sigmoid(_lin(P, "caption_attention.context_gate", torch.cat([word, h1, ctx], 1)))
    tc = _lin(P, "caption_attention.tc_affine", torch.cat([word, h1], 1))
    return zt * torch.tanh(_lin(P, "caption_attention.sc_affine", ctx)) + (1 - zt) * torch.tanh(tc), alpha


def select_hard(M, alpha):
    """SelectC.forward, soft=False (editnet_rl.py:410-423): weight alpha*1 + (1 - alpha) on the arg-max row"""
    a = alpha.detach()
    val, idx = a.max(1)
    onehot = torch.zeros_like(a).scatter_(1, idx.unsqueeze(1), 1.0)
    diff = onehot.clone()
    diff[diff == 1] = 1 - val
    w = alpha * onehot + diff
    return (w.unsqueeze(2) * M).sum(1)


def visual_attention(P, X, h1):
    """VisualAttentionC.forward (editnet_rl.py:440-447): region embedding and its projection are recomputed
    per call; ReLU scores; context over the raw features with the product materialised"""
    fe = torch.relu(_lin(P, "visual_attention.att_embed.0", X))
    att1 = _lin(P, "visual_attention.features_att", fe)
    att2 = _lin(P, "visual_attention.decoder_att", h1)
    e = _lin(P, "visual_attention.full_att", torch.relu(att1 + att2.unsqueeze(1))).squeeze(2)
    alpha = F.softmax(e, dim=1)
    return (X * alpha.unsqueeze(2)).sum(1)


def copy_lstm(P, x, h, c, sel):
    """CopyLSTMCellC.forward (editnet_rl.py:269-287)"""
    i, f, g, o = (_lin(P, "copy_lstm.x2h", x) + _lin(P, "copy_lstm.h2h", h)).chunk(4, 1)
    c_new = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
    cg = torch.sigmoid(_lin(P, "copy_lstm.gate_cnew", c_new) + _lin(P, "copy_lstm.gate_cmem", sel))
    mem = cg * sel + (1 - cg) * c_new
    return torch.sigmoid(o) * torch.tanh(mem), mem


@torch.no_grad()
def greedy_decode(P, start_idx, end_idx, prev, plen, X, max_len=18):
    """DecoderC.forward, sample_max=True (editnet_rl.py:485-549).  prev (B,T) int64, plen (B,1) int64,
    X (B,R,F) fp32 -> (seq (B,max_len) int64, seqLogprobs (B,max_len) fp32)"""
    B = X.shape[0]
    D = P["fc.weight"].shape[1]
    seq = torch.zeros(B, max_len, dtype=torch.long)
    logps = torch.zeros(B, max_len)
    it = torch.full((B,), int(start_idx), dtype=torch.long)
    h1, c1, h2, c2 = (torch.zeros(B, D) for _ in range(4))
    H, M, final_hidden, mask = caption_encoder(P, prev, plen)
    mean = X.mean(1)
    unfinished = None
    for t in range(max_len + 1):
        emb = embed(P, it)
        h1, c1 = attention_lstm(P, torch.cat([emb, final_hidden, h2, mean], 1), h1, c1)
        attend_cap, alpha_c = caption_attention(P, H, h1, emb, mask)
        attend_img = visual_attention(P, X, h1)
        sel = select_hard(M, alpha_c)
        h2, c2 = copy_lstm(P, torch.cat([h1, attend_cap, attend_img], 1), h2, c2, sel)
        logp = F.log_softmax(_lin(P, "fc", h2), dim=1)
        if t == max_len:
            break
        best, it = logp.max(1)
        it = it.clone()
        it[it == int(end_idx)] = 0
        unfinished = (it > 0) if t == 0 else unfinished * (it > 0)
        it = it * unfinished.type_as(it)
        seq[:, t] = it
        logps[:, t] = best
        if unfinished.sum() == 0:
            break
    return seq, logps
