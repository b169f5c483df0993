"""Autograd wrappers of the operator-level HIP entry points.

Forward of every op runs the gfx950 kernels through the C ABI (include/set_hip.h).  Backward is
PyTorch autograd over a differentiable restatement of the same operator, re-evaluated from the
saved inputs (recompute-in-backward, like activation checkpointing): BASELINE.json's north star
keeps autograd on the PyTorch-ROCm host side; hand-written backward kernels are the next round
(DESIGN.md §0 row a13).  Gradient parity is tested against the reference's own autograd
(tests/golden/*: `grad.*`), not against these formulas.

Each formula cites the reference lines it restates; none of them is used in any forward pass.
"""
from __future__ import annotations

import ctypes as C

import torch
import torch.nn.functional as F

from . import _lib
from ._lib import EditNetWeights, check, ptr, stream_of


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


class _HipFn(torch.autograd.Function):
    """forward: `hip(*inputs)` (no grad); backward: autograd over `formula(*inputs)`."""

    @staticmethod
    def forward(ctx, hip, formula, nout, *inputs):
        with torch.no_grad():
            outs = hip(*inputs)
        if not isinstance(outs, tuple):
            outs = (outs,)
        ctx.formula = formula
        ctx.is_tensor = [isinstance(t, torch.Tensor) for t in inputs]
        ctx.consts = [None if isinstance(t, torch.Tensor) else t for t in inputs]
        ctx.save_for_backward(*[t for t in inputs if isinstance(t, torch.Tensor)])
        ctx.nout = nout
        return outs if nout > 1 else outs[0]

    @staticmethod
    def backward(ctx, *grads):
        saved = list(ctx.saved_tensors)
        inputs, k = [], 0
        for is_t, cst in zip(ctx.is_tensor, ctx.consts):
            if is_t:
                t = saved[k]
                k += 1
                inputs.append(t.detach().requires_grad_(t.is_floating_point()))
            else:
                inputs.append(cst)
        need = [i for i, (is_t, t) in enumerate(zip(ctx.is_tensor, inputs))
                if is_t and t.requires_grad and ctx.needs_input_grad[3 + i]]
        with torch.enable_grad():
            outs = ctx.formula(*inputs)
        if not isinstance(outs, tuple):
            outs = (outs,)
        pairs = [(o, g) for o, g in zip(outs, grads) if g is not None and o.requires_grad]
        gin = [None] * len(inputs)
        if pairs and need:
            res = torch.autograd.grad([o for o, _ in pairs], [inputs[i] for i in need], [g for _, g in pairs],
                                      allow_unused=True)
            for i, r in zip(need, res):
                gin[i] = r
        return (None, None, None) + tuple(gin)


def _apply(hip, formula, nout, *inputs):
    return _HipFn.apply(hip, formula, nout, *inputs)


# ------------------------------------------------------------------------------------------------
# nn.Linear (+ activation)
# ------------------------------------------------------------------------------------------------
def linear(x, weight, bias, act=_lib.ACT_NONE):
    def hip(x, w, b):
        lib = _lib.load()
        x2 = _c(x.reshape(-1, x.shape[-1]))
        M, K, N = x2.shape[0], x2.shape[1], w.shape[0]
        y = torch.empty(M, N, dtype=torch.float32, device=x.device)
        ws = torch.empty(max(16, lib.set_linear_workspace_bytes(M, N, K)), dtype=torch.uint8, device=x.device)
        check(lib.set_linear_f32(ptr(x2), K, ptr(w), K, ptr(b), ptr(y), N, M, N, K, act, ptr(ws), ws.numel(),
                                 stream_of(x.device)), "set_linear_f32")
        return y.reshape(*x.shape[:-1], N)

    def formula(x, w, b):
        y = F.linear(x, w, b)
        if act == _lib.ACT_RELU:
            y = torch.relu(y)
        elif act == _lib.ACT_TANH:
            y = torch.tanh(y)
        elif act == _lib.ACT_SIGMOID:
            y = torch.sigmoid(y)
        return y

    return _apply(hip, formula, 1, x, weight, bias)


# ------------------------------------------------------------------------------------------------
# EmbeddingC.forward without the dropout (editnet.py:301-302)
# ------------------------------------------------------------------------------------------------
def embed_relu(ids, table):
    def hip(ids, table):
        lib = _lib.load()
        ids_c = _c(ids)
        n, D = ids_c.numel(), table.shape[1]
        out = torch.empty(tuple(ids_c.shape) + (D,), dtype=torch.float32, device=ids.device)
        check(lib.set_embed_relu_f32(ptr(table), ptr(ids_c), 1, ptr(out), D, n, D, table.shape[0],
                                     stream_of(ids.device)), "set_embed_relu_f32")
        return out

    return _apply(hip, lambda ids, table: torch.relu(F.embedding(ids, table)), 1, ids, table)


# ------------------------------------------------------------------------------------------------
# nn.LSTMCell / LSTMCellC (editnet.py:226-244)
# ------------------------------------------------------------------------------------------------
def _lstm_formula(x, h, c, w_ih, w_hh, b_ih, b_hh):
    gates = F.linear(x, w_ih, b_ih) + F.linear(h, w_hh, b_hh)
    i, f, g, o = gates.chunk(4, 1)
    c_new = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
    return torch.sigmoid(o) * torch.tanh(c_new), c_new


def lstm_cell(x, h, c, w_ih, w_hh, b_ih, b_hh):
    def hip(x, h, c, w_ih, w_hh, b_ih, b_hh):
        lib = _lib.load()
        x, h, c = _c(x), _c(h), _c(c)
        M, K, D = x.shape[0], x.shape[1], h.shape[1]
        h_new, c_new = torch.empty_like(h), torch.empty_like(c)
        ws = torch.empty(lib.set_lstm_cell_workspace_bytes(M, D, K), dtype=torch.uint8, device=x.device)
        check(lib.set_lstm_cell_f32(ptr(x), K, K, ptr(h), ptr(c), ptr(w_ih), K, ptr(w_hh), ptr(b_ih), ptr(b_hh),
                                    ptr(h_new), ptr(c_new), M, D, ptr(ws), ws.numel(), stream_of(x.device)),
              "set_lstm_cell_f32")
        return h_new, c_new

    return _apply(hip, _lstm_formula, 2, x, h, c, w_ih, w_hh, b_ih, b_hh)


# ------------------------------------------------------------------------------------------------
# CaptionAttentionC.forward (editnet.py:364-381)
# ------------------------------------------------------------------------------------------------
def caption_attention(H, h1, word, mask, feat_w, feat_b, dec_w, dec_b, full_w, full_b, gate_w, gate_b, sc_w, sc_b,
                      tc_w, tc_b):
    def hip(H, h1, word, mask, feat_w, feat_b, dec_w, dec_b, full_w, full_b, gate_w, gate_b, sc_w, sc_b, tc_w, tc_b):
        lib = _lib.load()
        H, h1, word, mask = _c(H), _c(h1), _c(word), _c(mask)
        M, T, D = H.shape
        A = dec_w.shape[0]
        w = EditNetWeights()
        w.ca_feat_w, w.ca_feat_b, w.ca_dec_w, w.ca_dec_b = feat_w.data_ptr(), feat_b.data_ptr(), dec_w.data_ptr(), dec_b.data_ptr()
        w.ca_full_w, w.ca_full_b, w.ca_gate_w, w.ca_gate_b = full_w.data_ptr(), full_b.data_ptr(), gate_w.data_ptr(), gate_b.data_ptr()
        w.ca_sc_w, w.ca_sc_b, w.ca_tc_w, w.ca_tc_b = sc_w.data_ptr(), sc_b.data_ptr(), tc_w.data_ptr(), tc_b.data_ptr()
        gated = torch.empty(M, D, dtype=torch.float32, device=H.device)
        alpha = torch.empty(M, T, dtype=torch.float32, device=H.device)
        ws = torch.empty(lib.set_caption_attention_workspace_bytes(M, T, D, A), dtype=torch.uint8, device=H.device)
        check(lib.set_caption_attention_f32(C.byref(w), ptr(H), None, ptr(h1), ptr(word), ptr(mask), ptr(gated),
                                            ptr(alpha), M, T, D, D, A, ptr(ws), ws.numel(), stream_of(H.device)),
              "set_caption_attention_f32")
        return gated, alpha

    def formula(H, h1, word, mask, feat_w, feat_b, dec_w, dec_b, full_w, full_b, gate_w, gate_b, sc_w, sc_b, tc_w, tc_b):
        att1 = F.linear(H, feat_w, feat_b)
        att2 = F.linear(h1, dec_w, dec_b)
        e = F.linear(torch.tanh(att1 + att2.unsqueeze(1)), full_w, full_b).squeeze(2)
        e = e.masked_fill(mask == 0, -1e10)
        alpha = F.softmax(e, dim=1)
        ctx = (H * alpha.unsqueeze(2)).sum(1)
        zt = torch.sigmoid(F.linear(torch.cat([word, h1, ctx], 1), gate_w, gate_b))
        out = zt * torch.tanh(F.linear(ctx, sc_w, sc_b)) + (1 - zt) * torch.tanh(F.linear(torch.cat([word, h1], 1), tc_w, tc_b))
        return out, alpha

    return _apply(hip, formula, 2, H, h1, word, mask, feat_w, feat_b, dec_w, dec_b, full_w, full_b, gate_w, gate_b,
                  sc_w, sc_b, tc_w, tc_b)


# ------------------------------------------------------------------------------------------------
# VisualAttentionC.forward given att1 = features_att(att_embed(X)) (editnet.py:443-446)
# ------------------------------------------------------------------------------------------------
def visual_attention_from_att1(X, att1, h1, dec_w, dec_b, full_w, full_b):
    def hip(X, att1, h1, dec_w, dec_b, full_w, full_b):
        lib = _lib.load()
        X, att1, h1 = _c(X), _c(att1), _c(h1)
        M, R, Fd = X.shape
        D, A = dec_w.shape[1], dec_w.shape[0]
        w = EditNetWeights()
        w.va_dec_w, w.va_dec_b, w.va_full_w, w.va_full_b = dec_w.data_ptr(), dec_b.data_ptr(), full_w.data_ptr(), full_b.data_ptr()
        ctx = torch.empty(M, Fd, dtype=torch.float32, device=X.device)
        ws = torch.empty(lib.set_visual_attention_workspace_bytes(M, R, Fd, D, A), dtype=torch.uint8, device=X.device)
        check(lib.set_visual_attention_f32(C.byref(w), ptr(X), ptr(att1), ptr(h1), ptr(ctx), None, M, R, Fd, D, A, 0,
                                           ptr(ws), ws.numel(), stream_of(X.device)), "set_visual_attention_f32")
        return ctx

    def formula(X, att1, h1, dec_w, dec_b, full_w, full_b):
        att2 = F.linear(h1, dec_w, dec_b)
        e = F.linear(torch.relu(att1 + att2.unsqueeze(1)), full_w, full_b).squeeze(2)
        alpha = F.softmax(e, dim=1)
        return (X * alpha.unsqueeze(2)).sum(1)

    return _apply(hip, formula, 1, X, att1, h1, dec_w, dec_b, full_w, full_b)


# ------------------------------------------------------------------------------------------------
# SelectC.forward, hard mode (editnet.py:409-420): straight-through weight on the arg-max row
# ------------------------------------------------------------------------------------------------
def select(Mem, alpha):
    def hip(Mem, alpha):
        lib = _lib.load()
        Mem, alpha = _c(Mem), _c(alpha)
        B, T, D = Mem.shape
        sel = torch.empty(B, D, dtype=torch.float32, device=Mem.device)
        check(lib.set_select_f32(ptr(Mem), ptr(alpha), ptr(sel), B, T, D, stream_of(Mem.device)), "set_select_f32")
        return sel

    def formula(Mem, alpha):
        a_d = alpha.detach()
        val, idx = a_d.max(1)
        onehot = torch.zeros_like(a_d).scatter_(1, idx.unsqueeze(1), 1.0)
        w = alpha * onehot + onehot * (1 - val).unsqueeze(1)
        return (w.unsqueeze(2) * Mem).sum(1)

    return _apply(hip, formula, 1, Mem, alpha)


# ------------------------------------------------------------------------------------------------
# CopyLSTMCellC.forward (editnet.py:265-285)
# ------------------------------------------------------------------------------------------------
def copy_lstm(x, h2, c2, cmem, x2h_w, x2h_b, h2h_w, h2h_b, cnew_w, cnew_b, cmem_w, cmem_b):
    def hip(x, h2, c2, cmem, x2h_w, x2h_b, h2h_w, h2h_b, cnew_w, cnew_b, cmem_w, cmem_b):
        lib = _lib.load()
        x, h2, c2, cmem = _c(x), _c(h2), _c(c2), _c(cmem)
        M, K, D = x.shape[0], x.shape[1], h2.shape[1]
        w = EditNetWeights()
        w.cl_x2h_w, w.cl_x2h_b, w.cl_h2h_w, w.cl_h2h_b = x2h_w.data_ptr(), x2h_b.data_ptr(), h2h_w.data_ptr(), h2h_b.data_ptr()
        w.cl_cnew_w, w.cl_cnew_b, w.cl_cmem_w, w.cl_cmem_b = cnew_w.data_ptr(), cnew_b.data_ptr(), cmem_w.data_ptr(), cmem_b.data_ptr()
        h_new, c_new = torch.empty_like(h2), torch.empty_like(c2)
        ws = torch.empty(lib.set_copy_lstm_workspace_bytes(M, D, K), dtype=torch.uint8, device=x.device)
        check(lib.set_copy_lstm_f32(C.byref(w), ptr(x), K, K, ptr(h2), ptr(c2), ptr(cmem), ptr(h_new), ptr(c_new), M, D,
                                    ptr(ws), ws.numel(), stream_of(x.device)), "set_copy_lstm_f32")
        return h_new, c_new

    def formula(x, h2, c2, cmem, x2h_w, x2h_b, h2h_w, h2h_b, cnew_w, cnew_b, cmem_w, cmem_b):
        gates = F.linear(x, x2h_w, x2h_b) + F.linear(h2, h2h_w, h2h_b)
        i, f, g, o = gates.chunk(4, 1)
        c_new = torch.sigmoid(f) * c2 + torch.sigmoid(i) * torch.tanh(g)
        copy = torch.sigmoid(F.linear(c_new, cnew_w, cnew_b) + F.linear(cmem, cmem_w, cmem_b))
        adaptive = copy * cmem + (1 - copy) * c_new
        return torch.sigmoid(o) * torch.tanh(adaptive), adaptive

    return _apply(hip, formula, 2, x, h2, c2, cmem, x2h_w, x2h_b, h2h_w, h2h_b, cnew_w, cnew_b, cmem_w, cmem_b)


# ------------------------------------------------------------------------------------------------
# DCNet CaptionAttention.forward (dcnet.py:254-270): additive attention without gating
# ------------------------------------------------------------------------------------------------
def dcnet_caption_attention(feats, h1, mask, feat_w, feat_b, dec_w, dec_b, full_w, full_b):
    def hip(feats, h1, mask, feat_w, feat_b, dec_w, dec_b, full_w, full_b):
        lib = _lib.load()
        feats, h1, mask = _c(feats), _c(h1), _c(mask)
        M, T, Dh = feats.shape
        D, A = h1.shape[1], dec_w.shape[0]
        w = EditNetWeights()
        w.ca_feat_w, w.ca_feat_b, w.ca_dec_w, w.ca_dec_b = feat_w.data_ptr(), feat_b.data_ptr(), dec_w.data_ptr(), dec_b.data_ptr()
        w.ca_full_w, w.ca_full_b = full_w.data_ptr(), full_b.data_ptr()
        ctx = torch.empty(M, Dh, dtype=torch.float32, device=feats.device)
        ws = torch.empty(lib.set_caption_attention_workspace_bytes(M, T, max(Dh, D), A), dtype=torch.uint8,
                         device=feats.device)
        check(lib.set_caption_attention_f32(C.byref(w), ptr(feats), None, ptr(h1), None, ptr(mask), ptr(ctx), None, M,
                                            T, Dh, D, A, ptr(ws), ws.numel(), stream_of(feats.device)),
              "set_caption_attention_f32")
        return ctx

    def formula(feats, h1, mask, feat_w, feat_b, dec_w, dec_b, full_w, full_b):
        att1 = F.linear(feats, feat_w, feat_b)
        att2 = F.linear(h1, dec_w, dec_b)
        e = F.linear(torch.tanh(att1 + att2.unsqueeze(1)), full_w, full_b).squeeze(2)
        e = e.masked_fill(mask == 0, -1e10)
        alpha = F.softmax(e, dim=1)
        return (feats * alpha.unsqueeze(2)).sum(1)

    return _apply(hip, formula, 1, feats, h1, mask, feat_w, feat_b, dec_w, dec_b, full_w, full_b)
