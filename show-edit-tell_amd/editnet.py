"""EditNet on MI355X: the reference's `DecoderC` module surface over the HIP decode path.

Mirrors `/root/reference/editnet.py:210-548`: same class names, constructor signatures,
attribute names and `state_dict` keys, so checkpoints and the reference's train / evaluate
loops (`editnet.py:551-740`) drop in.  The modules hold parameters in ordinary torch containers
(nn.Linear / nn.LSTMCell / nn.Embedding); every `forward` runs hand-written gfx950 kernels
through the C ABI in include/set_hip.h (csrc/libset_hip.so).  There is no PyTorch fallback:
CPU tensors, non-fp32 parameters or a missing library raise.

Two execution paths:
  * no-grad (inference / evaluation / the greedy baseline of SCST): the whole loop runs inside the
    C library (`set_editnet_xe_forward`, `set_editnet_greedy`), loop invariants hoisted.
  * grad-enabled (training, or eval-mode gradient checks): `_forward_autograd` follows the
    reference loop (`editnet.py:479-548`) step by step over autograd-wrapped HIP operators
    (`autograd_ops.py`: HIP forward, PyTorch-autograd backward), with the reference's three dropout
    sites and scheduled sampling.
"""
from __future__ import annotations

import ctypes as C
import math
import numpy as np

import torch
import torch.nn as nn

from . import _lib
from ._lib import EditNetDims, EditNetWeights, EDITNET_WEIGHT_FIELDS, check, ptr, stream_of


import os as _os

_XE_SEQUENCE = _os.environ.get("SET_XE_SEQUENCE", "1") != "0"     # 0: keep the per-operator autograd loop everywhere


_arange = {}


def _arange_cached(n, dev):
    """torch.arange(n) on `dev` (the identity sort order handed back to the caller), made once per (n, device)"""
    key = (n, dev)
    if key not in _arange:
        _arange[key] = torch.arange(n, device=dev)
    return _arange[key]


def _f32c(t):
    return t if (t.dtype == torch.float32 and t.is_contiguous()) else t.float().contiguous()


def _i64c(t):
    return t if (t.dtype == torch.int64 and t.is_contiguous()) else t.long().contiguous()


def _require_cuda(t, what):
    if not t.is_cuda:
        raise _lib.SetError("%s must live on the GPU: the decode path has no CPU fallback" % what)


def _wants_grad(mod, *tensors):
    """True when a direct sub-module call has to be differentiable (the reference's sub-modules always are:
    evaluate()-style callers may use them under autograd) or runs in train mode (dropout sites): the call then
    goes through the autograd-wrapped HIP operators of autograd_ops.py instead of the plain forward kernels."""
    if not torch.is_grad_enabled():
        return False
    return (any(p.requires_grad for p in mod.parameters())
            or any(torch.is_tensor(t) and t.requires_grad for t in tensors))


class LSTMCellC(nn.Module):
    """reference editnet.py:210-244"""

    def __init__(self, input_size, hidden_size):
        super().__init__()
        self.hidden_size = hidden_size
        self.input_size = input_size
        self.x2h = nn.Linear(input_size, 4 * hidden_size)
        self.h2h = nn.Linear(hidden_size, 4 * hidden_size)
        self.tanh = nn.Tanh()
        self.init_parameters()

    def init_parameters(self):
        std = 1.0 / math.sqrt(self.hidden_size)
        for p in self.parameters():
            p.data.uniform_(-std, std)

    def forward(self, x, states):
        ht, ct = states
        _require_cuda(x, "LSTMCellC input")
        if _wants_grad(self, x, ht, ct):
            from . import autograd_ops as A
            return A.lstm_cell(_f32c(x), _f32c(ht), _f32c(ct), self.x2h.weight, self.h2h.weight, self.x2h.bias,
                               self.h2h.bias)
        lib = _lib.load()
        x, ht, ct = _f32c(x), _f32c(ht), _f32c(ct)
        M, K, D = x.shape[0], x.shape[1], self.hidden_size
        h_new, c_new = torch.empty_like(ht), torch.empty_like(ct)
        ws = torch.empty(lib.set_lstm_cell_workspace_bytes(M, D, K), dtype=torch.uint8, device=x.device)
        check(lib.set_lstm_cell_f32(ptr(x), K, K, ptr(ht), ptr(ct), ptr(self.x2h.weight), K, ptr(self.h2h.weight),
                                    ptr(self.x2h.bias), ptr(self.h2h.bias), ptr(h_new), ptr(c_new), M, D, ptr(ws),
                                    ws.numel(), stream_of(x.device)), "set_lstm_cell_f32")
        return h_new, c_new


class CopyLSTMCellC(nn.Module):
    """reference editnet.py:247-285"""

    def __init__(self, input_size, hidden_size):
        super().__init__()
        self.hidden_size = hidden_size
        self.input_size = input_size
        self.x2h = nn.Linear(input_size, 4 * hidden_size)
        self.h2h = nn.Linear(hidden_size, 4 * hidden_size)
        self.gate_cnew = nn.Linear(hidden_size, hidden_size)
        self.gate_cmem = nn.Linear(hidden_size, hidden_size)
        self.tanh = nn.Tanh()
        self.init_parameters()

    def init_parameters(self):
        std = 1.0 / math.sqrt(self.hidden_size)
        for p in self.parameters():
            p.data.uniform_(-std, std)

    def forward(self, x, states, c_memory):
        ht, ct = states
        _require_cuda(x, "CopyLSTMCellC input")
        if _wants_grad(self, x, ht, ct, c_memory):
            from . import autograd_ops as A
            return A.copy_lstm(_f32c(x), _f32c(ht), _f32c(ct), _f32c(c_memory), self.x2h.weight, self.x2h.bias,
                               self.h2h.weight, self.h2h.bias, self.gate_cnew.weight, self.gate_cnew.bias,
                               self.gate_cmem.weight, self.gate_cmem.bias)
        lib = _lib.load()
        x, ht, ct, cm = _f32c(x), _f32c(ht), _f32c(ct), _f32c(c_memory)
        M, K, D = x.shape[0], x.shape[1], self.hidden_size
        w = EditNetWeights()
        w.cl_x2h_w, w.cl_x2h_b = self.x2h.weight.data_ptr(), self.x2h.bias.data_ptr()
        w.cl_h2h_w, w.cl_h2h_b = self.h2h.weight.data_ptr(), self.h2h.bias.data_ptr()
        w.cl_cnew_w, w.cl_cnew_b = self.gate_cnew.weight.data_ptr(), self.gate_cnew.bias.data_ptr()
        w.cl_cmem_w, w.cl_cmem_b = self.gate_cmem.weight.data_ptr(), self.gate_cmem.bias.data_ptr()
        h_new, c_new = torch.empty_like(ht), torch.empty_like(ct)
        ws = torch.empty(lib.set_copy_lstm_workspace_bytes(M, D, K), dtype=torch.uint8, device=x.device)
        check(lib.set_copy_lstm_f32(C.byref(w), ptr(x), K, K, ptr(ht), ptr(ct), ptr(cm), ptr(h_new), ptr(c_new), M, D,
                                    ptr(ws), ws.numel(), stream_of(x.device)), "set_copy_lstm_f32")
        return h_new, c_new


class EmbeddingC(nn.Module):
    """reference editnet.py:288-304 (ReLU + Dropout(0.5) after the lookup; no padding_idx)"""

    def __init__(self, word_map, emb_dim):
        super().__init__()
        self.emb_dim = emb_dim
        self.word_map = word_map
        self.embedding = nn.Embedding(len(word_map), self.emb_dim)
        self.relu = nn.ReLU()
        self.dropout = nn.Dropout(0.5)

    def forward(self, x):
        _require_cuda(x, "token ids")
        if self.training or _wants_grad(self):
            from . import autograd_ops as A
            from . import rng
            # editnet.py:301-303; a direct call draws its own seed (DecoderC.forward addresses the site itself)
            return A.philox_dropout(A.embed_relu(_i64c(x), self.embedding.weight), self.dropout.p, rng.next_seed(),
                                    rng.offset(rng.SITE_EMBED), self.training)
        lib = _lib.load()
        ids = _i64c(x)
        n, D = ids.numel(), self.emb_dim
        out = torch.empty(tuple(ids.shape) + (D,), dtype=torch.float32, device=ids.device)
        check(lib.set_embed_relu_f32(ptr(self.embedding.weight), ptr(ids), 1, ptr(out), D, n, D,
                                     self.embedding.num_embeddings, stream_of(ids.device)), "set_embed_relu_f32")
        return out


class CaptionEncoderC(nn.Module):
    """reference editnet.py:307-348"""

    def __init__(self, vocab_size, emb_dim, enc_hid_dim, embed):
        super().__init__()
        self.vocab_size = vocab_size
        self.emb_dim = emb_dim
        self.enc_hid_dim = enc_hid_dim
        self.embed = embed
        self.lstm_encoder_cell = LSTMCellC(emb_dim, enc_hid_dim)
        self.affine_hn = nn.Linear(enc_hid_dim, enc_hid_dim)
        self.tanh = nn.Tanh()

    def forward(self, seq, seq_len):
        _require_cuda(seq, "previous captions")
        if self.training or _wants_grad(self):
            return _caption_encoder_autograd(self, seq, seq_len)
        lib = _lib.load()
        seq, lens = _i64c(seq), _i64c(seq_len.reshape(-1))
        B, T, D = seq.shape[0], seq.shape[1], self.enc_hid_dim
        dev = seq.device
        w = EditNetWeights()
        cell = self.lstm_encoder_cell
        w.embed = self.embed.embedding.weight.data_ptr()
        w.enc_x2h_w, w.enc_x2h_b = cell.x2h.weight.data_ptr(), cell.x2h.bias.data_ptr()
        w.enc_h2h_w, w.enc_h2h_b = cell.h2h.weight.data_ptr(), cell.h2h.bias.data_ptr()
        w.enc_aff_w, w.enc_aff_b = self.affine_hn.weight.data_ptr(), self.affine_hn.bias.data_ptr()
        H = torch.empty(B, T, D, dtype=torch.float32, device=dev)
        M = torch.empty(B, T, D, dtype=torch.float32, device=dev)
        fh = torch.empty(B, D, dtype=torch.float32, device=dev)
        mask = torch.empty(B, T, dtype=torch.float32, device=dev)
        ws = torch.empty(lib.set_caption_encoder_workspace_bytes(B, T, D), dtype=torch.uint8, device=dev)
        check(lib.set_caption_encoder_f32(C.byref(w), ptr(seq), ptr(lens), ptr(H), ptr(M), ptr(fh), ptr(mask), B, T, D,
                                          self.vocab_size, ptr(ws), ws.numel(), stream_of(dev)),
              "set_caption_encoder_f32")
        tmax = int(lens.max().item())            # the reference pads to max(len) (editnet.py:327)
        return H[:, :tmax], M[:, :tmax], fh, mask[:, :tmax]


def _caption_encoder_autograd(enc, seq, seq_len, seed=None, site=None):
    """CaptionEncoderC.forward (editnet.py:319-348), grad-enabled: embedding (+ its dropout) for all positions, then the
    whole recurrence as one autograd node (autograd_ops.encoder_lstm: rows advance while t < len, padded outputs zero).
    The embedding's dropout mask (editnet.py:329 calls the shared EmbeddingC) is the Philox stream (seed, site): element
    (b, l, :) is row b * Tmax + l of the (B * Tmax, E) operand, rows in the caller's order (rng.py)."""
    from . import autograd_ops as A
    from . import rng
    cell = enc.lstm_encoder_cell
    lens = seq_len.reshape(-1)
    tmax = int(lens.max().item())
    seed = rng.next_seed() if seed is None else seed
    emb = A.philox_dropout(A.embed_relu(seq[:, :tmax], enc.embed.embedding.weight), enc.embed.dropout.p, seed,
                           rng.offset(rng.SITE_ENC_EMBED if site is None else site), enc.embed.training)
    H, M, h_last = A.encoder_lstm(emb, lens, cell.x2h.weight, cell.x2h.bias, cell.h2h.weight, cell.h2h.bias)
    mask = (M.detach().sum(2) != 0).float()
    final_hidden = A.linear(h_last, enc.affine_hn.weight, enc.affine_hn.bias, _lib.ACT_TANH)
    return H, M, final_hidden, mask


class CaptionAttentionC(nn.Module):
    """reference editnet.py:351-381"""

    def __init__(self, caption_features_dim, decoder_dim, attention_dim):
        super().__init__()
        self.cap_features_att = nn.Linear(caption_features_dim, attention_dim)
        self.cap_decoder_att = nn.Linear(decoder_dim, attention_dim)
        self.cap_full_att = nn.Linear(attention_dim, 1)
        self.context_gate = nn.Linear((caption_features_dim * 2) + decoder_dim, caption_features_dim)
        self.sc_affine = nn.Linear(caption_features_dim, caption_features_dim)
        self.tc_affine = nn.Linear(decoder_dim * 2, caption_features_dim)
        self.tanh = nn.Tanh()

    def _weights(self):
        w = EditNetWeights()
        for f, m in (("ca_feat", self.cap_features_att), ("ca_dec", self.cap_decoder_att),
                     ("ca_full", self.cap_full_att), ("ca_gate", self.context_gate), ("ca_sc", self.sc_affine),
                     ("ca_tc", self.tc_affine)):
            setattr(w, f + "_w", m.weight.data_ptr())
            setattr(w, f + "_b", m.bias.data_ptr())
        return w

    def forward(self, caption_features, decoder_hidden, word, prev_caption_mask):
        _require_cuda(caption_features, "caption features")
        if _wants_grad(self, caption_features, decoder_hidden, word):
            from . import autograd_ops as A
            return A.caption_attention(
                _f32c(caption_features), _f32c(decoder_hidden), _f32c(word), _f32c(prev_caption_mask),
                self.cap_features_att.weight, self.cap_features_att.bias, self.cap_decoder_att.weight,
                self.cap_decoder_att.bias, self.cap_full_att.weight, self.cap_full_att.bias, self.context_gate.weight,
                self.context_gate.bias, self.sc_affine.weight, self.sc_affine.bias, self.tc_affine.weight,
                self.tc_affine.bias)
        lib = _lib.load()
        H, h1, word, mask = _f32c(caption_features), _f32c(decoder_hidden), _f32c(word), _f32c(prev_caption_mask)
        M, T, D = H.shape
        A = self.cap_decoder_att.out_features
        w = self._weights()
        gated = torch.empty(M, D, dtype=torch.float32, device=H.device)
        alpha = torch.empty(M, T, dtype=torch.float32, device=H.device)
        ws = torch.empty(lib.set_caption_attention_workspace_bytes(M, T, D, A), dtype=torch.uint8, device=H.device)
        check(lib.set_caption_attention_f32(C.byref(w), ptr(H), None, ptr(h1), ptr(word), ptr(mask), ptr(gated),
                                            ptr(alpha), M, T, D, D, A, ptr(ws), ws.numel(), stream_of(H.device)),
              "set_caption_attention_f32")
        return gated, alpha


class SelectC(nn.Module):
    """SCMA selection, reference editnet.py:383-421: hard (soft=False, the only mode the reference's loops use) keeps the
    arg-max row with the straight-through weight; soft=True is the plain weighted sum over the memory rows (:419-420)."""

    def __init__(self, prev_caption_dim, decoder_dim):
        super().__init__()

    def forward(self, previous_encoded_m, sim_weights, soft=False):
        _require_cuda(previous_encoded_m, "encoder memory")
        if soft:
            from . import autograd_ops as A
            Mem, alpha = _f32c(previous_encoded_m), _f32c(sim_weights)
            if _wants_grad(self, previous_encoded_m, sim_weights):
                return A.select_soft(Mem, alpha)
            return A.select_soft_nograd(Mem, alpha)
        if _wants_grad(self, previous_encoded_m, sim_weights):
            from . import autograd_ops as A
            return A.select(_f32c(previous_encoded_m), _f32c(sim_weights))
        lib = _lib.load()
        Mem, alpha = _f32c(previous_encoded_m), _f32c(sim_weights)
        B, T, D = Mem.shape
        sel = torch.empty(B, D, dtype=torch.float32, device=Mem.device)
        check(lib.set_select_f32(ptr(Mem), ptr(alpha), ptr(sel), B, T, D, stream_of(Mem.device)), "set_select_f32")
        return sel


class VisualAttentionC(nn.Module):
    """reference editnet.py:424-447"""

    adaptive = 0

    def __init__(self, image_features_dim, decoder_dim, attention_dim):
        super().__init__()
        self.att_embed = nn.Sequential(nn.Linear(image_features_dim, decoder_dim), nn.ReLU(), nn.Dropout(0.5))
        self.features_att = nn.Linear(decoder_dim, attention_dim)
        self.decoder_att = nn.Linear(decoder_dim, attention_dim)
        self.full_att = nn.Linear(attention_dim, 1)
        self.softmax = nn.Softmax(dim=1)

    def _weights(self):
        w = EditNetWeights()
        for f, m in (("va_emb", self.att_embed[0]), ("va_feat", self.features_att), ("va_dec", self.decoder_att),
                     ("va_full", self.full_att)):
            setattr(w, f + "_w", m.weight.data_ptr())
            setattr(w, f + "_b", m.bias.data_ptr())
        return w

    def forward(self, image_features, decoder_hidden):
        _require_cuda(image_features, "image features")
        if self.training or _wants_grad(self, decoder_hidden):
            # editnet.py:441-446 as written: region embedding (+ its Dropout(0.5) in train mode) recomputed per call
            from . import autograd_ops as A
            from . import rng
            X = _f32c(image_features)
            fe = A.philox_dropout(A.linear(X, self.att_embed[0].weight, self.att_embed[0].bias, _lib.ACT_RELU),
                                  self.att_embed[2].p, rng.next_seed(), rng.offset(rng.SITE_REGION), self.training)
            rmask = None
            if self.adaptive:
                # editnet_adaptive.py:440-449: att_embed runs over the packed valid regions only — the first att_len[b] =
                # #(nonzero feature rows) regions of a sample; padded rows of the embedding stay zero — and the attention
                # mask is re-derived from the (dropped-out) embedding's row sums
                valid = (X.sum(2) != 0)
                keep = torch.arange(X.shape[1], device=X.device)[None, :] < valid.sum(1, keepdim=True)
                fe = fe * keep[:, :, None].to(fe.dtype)
                rmask = (fe.detach().sum(2) != 0).float()
            att1 = A.linear(fe, self.features_att.weight, self.features_att.bias)
            return A.visual_attention_from_att1(X, att1, _f32c(decoder_hidden), self.decoder_att.weight,
                                                self.decoder_att.bias, self.full_att.weight, self.full_att.bias, rmask)
        lib = _lib.load()
        X, h1 = _f32c(image_features), _f32c(decoder_hidden)
        M, R, F = X.shape
        D, A = self.decoder_att.in_features, self.decoder_att.out_features
        w = self._weights()
        ctx = torch.empty(M, F, dtype=torch.float32, device=X.device)
        ws = torch.empty(lib.set_visual_attention_workspace_bytes(M, R, F, D, A), dtype=torch.uint8, device=X.device)
        check(lib.set_visual_attention_f32(C.byref(w), ptr(X), None, ptr(h1), ptr(ctx), None, M, R, F, D, A,
                                           self.adaptive, ptr(ws), ws.numel(), stream_of(X.device)),
              "set_visual_attention_f32")
        return ctx


class _HipLSTMCell(nn.LSTMCell):
    """nn.LSTMCell parameters (weight_ih, weight_hh, bias_ih, bias_hh) with a HIP forward."""

    def forward(self, x, states=None):
        _require_cuda(x, "LSTMCell input")
        lib = _lib.load()
        x = _f32c(x)
        M, K, D = x.shape[0], x.shape[1], self.hidden_size
        if states is None:
            z = torch.zeros(M, D, dtype=torch.float32, device=x.device)
            states = (z, z)
        ht, ct = _f32c(states[0]), _f32c(states[1])
        if _wants_grad(self, x, ht, ct):
            from . import autograd_ops as A
            return A.lstm_cell(x, ht, ct, self.weight_ih, self.weight_hh, self.bias_ih, self.bias_hh)
        h_new, c_new = torch.empty_like(ht), torch.empty_like(ct)
        ws = torch.empty(lib.set_lstm_cell_workspace_bytes(M, D, K), dtype=torch.uint8, device=x.device)
        check(lib.set_lstm_cell_f32(ptr(x), K, K, ptr(ht), ptr(ct), ptr(self.weight_ih), K, ptr(self.weight_hh),
                                    ptr(self.bias_ih), ptr(self.bias_hh), ptr(h_new), ptr(c_new), M, D, ptr(ws),
                                    ws.numel(), stream_of(x.device)), "set_lstm_cell_f32")
        return h_new, c_new


class _HipLinear(nn.Linear):
    """nn.Linear parameters with a HIP forward (fp32 MFMA GEMM)."""

    def forward(self, x):
        _require_cuda(x, "Linear input")
        if _wants_grad(self, x):
            from . import autograd_ops as A
            return A.linear(_f32c(x), self.weight, self.bias)
        lib = _lib.load()
        x = _f32c(x)
        lead = x.shape[:-1]
        x2 = x.reshape(-1, x.shape[-1])
        M, K, N = x2.shape[0], x2.shape[1], self.out_features
        y = torch.empty(M, N, dtype=torch.float32, device=x.device)
        ws = torch.empty(max(16, lib.set_linear_workspace_bytes(M, N, K)), dtype=torch.uint8, device=x.device)
        check(lib.set_linear_f32(ptr(x2), K, ptr(self.weight), K, ptr(self.bias), ptr(y), N, M, N, K, 0, ptr(ws),
                                 ws.numel(), stream_of(x.device)), "set_linear_f32")
        return y.reshape(*lead, N)


class DecoderC(nn.Module):
    """reference editnet.py:449-548 — XE (teacher-forced) forward."""

    _visual_attention_cls = VisualAttentionC
    _adaptive = 0

    def __init__(self, word_map, decoder_dim=1024, caption_features_dim=1024, emb_dim=1024, attention_dim=512,
                 image_features_dim=2048):
        super().__init__()
        self.vocab_size = len(word_map)
        self.dropout = nn.Dropout(0.5)
        self.decoder_dim = decoder_dim
        self.embed = EmbeddingC(word_map, emb_dim)
        self.caption_encoder = CaptionEncoderC(len(word_map), emb_dim, caption_features_dim, self.embed)
        self.caption_attention = CaptionAttentionC(caption_features_dim, decoder_dim, attention_dim)
        self.visual_attention = self._visual_attention_cls(image_features_dim, decoder_dim, attention_dim)
        self.select = SelectC(caption_features_dim, decoder_dim)
        self.attention_lstm = _HipLSTMCell((emb_dim * 3) + image_features_dim, decoder_dim)
        self.copy_lstm = CopyLSTMCellC((emb_dim * 2) + image_features_dim, decoder_dim)
        self.tanh = nn.Tanh()
        self.fc = _HipLinear(decoder_dim, self.vocab_size)
        if not (decoder_dim == caption_features_dim == emb_dim):
            raise ValueError("the reference's cat shapes force decoder_dim == caption_features_dim == emb_dim")
        self._attention_dim = attention_dim
        self._image_features_dim = image_features_dim
        self._ws = None
        self._ws_key = None

    # ---- runtime state is NOT part of the module's persistent state --------------------------------------
    # The reference checkpoints pickle the whole module (editnet.py:168-175, `'decoder': decoder`) and callers may
    # copy.deepcopy a decoder: GPU workspaces, the derived token table and the last autograd graph must not travel.
    _RUNTIME_ATTRS = ("_ws", "_ws_key", "_ws_cache", "_tok_state", "_last_hidden", "_fwd_seed", "_fed_tokens", "_grad_buckets",
                      "_ahead", "_ahead_free", "_ahead_hits", "_ahead_busy")

    def __getstate__(self):
        state = dict(self.__dict__)
        for k in self._RUNTIME_ATTRS:
            state.pop(k, None)
        state["_ws"] = state["_ws_key"] = None
        return state

    def invalidate_token_table(self):
        """Drop the derived inference-time token table (see _token_table).  It is rebuilt automatically after two
        further no-grad calls.  Called on every train() <-> eval() switch, load_state_dict() and device / dtype move;
        call it yourself after writing weights in a way autograd cannot see (`p.data.add_()`, `dist.broadcast(p.data)`,
        raw-pointer updates): such writes do not bump `tensor._version`, which is all the cache can observe
        without a device->host synchronisation (SET_TOKEN_TABLE_VERIFY=1 adds that check for debugging)."""
        self.__dict__.pop("_tok_state", None)

    def train(self, mode=True):
        if bool(mode) != self.training:          # an actual train <-> eval switch (eval() on an eval module keeps the table)
            self.invalidate_token_table()
        return super().train(mode)

    def load_state_dict(self, *args, **kwargs):
        self.invalidate_token_table()
        return super().load_state_dict(*args, **kwargs)

    def _apply(self, fn, *args, **kwargs):
        self.invalidate_token_table()
        self.__dict__.pop("_ws_cache", None)
        self.__dict__.pop("_ahead", None)
        self.__dict__.pop("_ahead_free", None)
        self.__dict__.pop("_grad_buckets", None)
        self._ws = self._ws_key = None
        return super()._apply(fn, *args, **kwargs)

    # ---- reference API ---------------------------------------------------------------------
    def init_hidden_state(self, batch_size):
        dev = self.fc.weight.device          # the parameters' device (the reference uses a module global)
        h = torch.zeros(batch_size, self.decoder_dim, device=dev)
        c = torch.zeros(batch_size, self.decoder_dim, device=dev)
        return h, c

    # ---- runtime plumbing ------------------------------------------------------------------
    def _weights(self, dims=None):
        """Pack the parameter pointers; with `dims`, also attach the inference-time token table when it
        is valid (see _token_table)."""
        dev = self.fc.weight.device
        params = dict(self.named_parameters())
        w = _lib.pack_weights(EditNetWeights, EDITNET_WEIGHT_FIELDS, params, dev)
        if dims is not None:
            tab = self._token_table(dims)
            if tab is not None:
                w.tok_table = tab.data_ptr()
        return w

    # The three contractions of the step whose only input is the current token are folded into a
    # (V,10D) table (include/set_hip.h: tok_table).  The table is derived from six parameter tensors
    # and is rebuilt whenever any of them changes (tensor._version / data_ptr), on every train()/eval()
    # switch, load_state_dict() and device move (invalidate_token_table); it is only built once the same
    # weights have been seen on two consecutive no-grad calls, so SCST training (weights change every
    # iteration, and the loop toggles eval()/train()) never pays for it.  In-place writes through `.data`
    # are invisible to `_version`: call invalidate_token_table() after them.
    # SET_TOKEN_TABLE=0 disables, =1 forces, SET_TOKEN_TABLE_VERIFY=1 re-checks a checksum of the six
    # source tensors on every use (one device->host sync per call; debugging aid).
    def _token_table(self, dims):
        import os
        mode = os.environ.get("SET_TOKEN_TABLE", "auto")
        if mode == "0" or dims.D % 64:
            return None
        cell = self.caption_encoder.lstm_encoder_cell
        src = (self.embed.embedding.weight, self.attention_lstm.weight_ih, self.caption_attention.tc_affine.weight,
               self.caption_attention.context_gate.weight, cell.x2h.weight, cell.x2h.bias)
        from . import optim as _optim
        sig = tuple((t.data_ptr(), t._version) for t in src) + (_optim.weights_epoch(),)
        st = self.__dict__.setdefault("_tok_state", {"sig": None, "seen": 0, "table": None})
        if st["sig"] != sig:
            st.update(sig=sig, seen=1, table=None)
        else:
            st["seen"] += 1
        if st["table"] is None and (mode == "1" or st["seen"] >= 2):
            lib = _lib.load()
            dev = self.fc.weight.device
            table = torch.empty(lib.set_editnet_token_table_bytes(C.byref(dims)) // 4, dtype=torch.float32, device=dev)
            ws = torch.empty(lib.set_editnet_token_table_workspace_bytes(C.byref(dims)), dtype=torch.uint8, device=dev)
            w = _lib.pack_weights(EditNetWeights, EDITNET_WEIGHT_FIELDS, dict(self.named_parameters()), dev)
            check(lib.set_editnet_build_token_table(C.byref(w), C.byref(dims), ptr(table), ptr(ws), ws.numel(),
                                                    stream_of(dev)), "set_editnet_build_token_table")
            torch.cuda.current_stream(dev).synchronize()        # other streams may use the table next
            st["table"] = table
            st["check"] = torch.stack([t.detach().double().sum() for t in src]).cpu()
        if st["table"] is not None and os.environ.get("SET_TOKEN_TABLE_VERIFY") == "1":
            now = torch.stack([t.detach().double().sum() for t in src]).cpu()
            if not torch.equal(now, st["check"]):
                raise _lib.SetError("token table is stale: a source weight changed without bumping tensor._version "
                                    "(in-place .data write?); call decoder.invalidate_token_table()")
        return st["table"]

    def _dims(self, B, T, R, maxT):
        return EditNetDims(B=B, T=T, R=R, F=self._image_features_dim, D=self.decoder_dim, A=self._attention_dim,
                           V=self.vocab_size, maxT=maxT, adaptive=self._adaptive)

    def _workspace(self, dims):
        """One workspace per (dims, device, stream): concurrent decodes on different streams must not share
        recurrent state or split-K slabs."""
        lib = _lib.load()
        dev = self.fc.weight.device
        key = tuple(getattr(dims, f) for f, _ in EditNetDims._fields_) + (str(dev), torch.cuda.current_stream(dev).cuda_stream)
        cache = self.__dict__.setdefault("_ws_cache", {})
        ws = cache.get(key)
        if ws is None:
            n = lib.set_editnet_workspace_bytes(C.byref(dims))
            if n == 0:
                raise _lib.SetError("unsupported EditNet dims %r (contraction dims must be multiples of 32)" % (key,))
            if len(cache) >= 24:
                cache.clear()
            ws = torch.empty(n, dtype=torch.uint8, device=dev)
            cache[key] = ws
        self._ws, self._ws_key = ws, key
        return ws

    def ws_tensor(self, dims, name, shape, dtype=torch.float32):
        """View of a named workspace tensor (debug / tests)."""
        lib = _lib.load()
        p = lib.set_editnet_ws_tensor(C.byref(dims), ptr(self._ws), name.encode())
        if not p:
            raise KeyError(name)
        off = p - self._ws.data_ptr()
        n = int(torch.tensor(shape).prod().item()) * torch.empty((), dtype=dtype).element_size()
        return self._ws[off:off + n].view(dtype).view(*shape)

    def forward(self, image_features, encoded_captions, caption_lengths, encoded_previous_captions,
                previous_cap_length, use_ss=False, ss_prob=0.0, image_mean=None):
        """Teacher-forced XE forward, reference editnet.py:479-548.  Returns
        (predictions (B,max(decode_lengths),V), encoded_captions sorted, decode_lengths, sort_ind)."""
        _require_cuda(image_features, "image features")
        # the fused C path is the eval-mode, no-grad, teacher-forced loop; everything else (train mode =
        # dropout, scheduled sampling, gradients) follows the reference loop over the HIP operators
        if (self.training or (use_ss and ss_prob > 0.0)
                or (torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()))):
            return self._forward_autograd(image_features, encoded_captions, caption_lengths,
                                          encoded_previous_captions, previous_cap_length, use_ss, ss_prob, image_mean)
        lib = _lib.load()
        dev = image_features.device
        batch_size = encoded_captions.size(0)
        caption_lengths, sort_ind = caption_lengths.squeeze(1).sort(dim=0, descending=True, stable=True)
        X = _f32c(image_features[sort_ind])
        encoded_captions = _i64c(encoded_captions[sort_ind])
        prev = _i64c(encoded_previous_captions[sort_ind])
        plen = _i64c(previous_cap_length[sort_ind].reshape(-1))
        mean = None if image_mean is None else _f32c(image_mean[sort_ind])
        decode_lengths = (caption_lengths - 1).tolist()
        maxT = max(decode_lengths)
        dims = self._dims(batch_size, prev.shape[1], X.shape[1], maxT)
        ws = self._workspace(dims)
        w = self._weights(dims)
        predictions = torch.empty(batch_size, maxT, self.vocab_size, dtype=torch.float32, device=dev)
        dl = (C.c_int * batch_size)(*decode_lengths)
        check(lib.set_editnet_xe_forward(C.byref(w), C.byref(dims), ptr(X), ptr(mean), ptr(encoded_captions),
                                         encoded_captions.shape[1], dl, ptr(prev), ptr(plen), ptr(predictions),
                                         ptr(ws), ws.numel(), stream_of(dev)), "set_editnet_xe_forward")
        return predictions, encoded_captions, decode_lengths, sort_ind


    # ---- grad-enabled path -------------------------------------------------------------------
    def with_host_lengths(self, caption_lengths_host):
        """the NEXT grad-enabled forward takes its sort order / decode lengths from this host copy of `caption_lengths`
        (one call; see _forward_autograd)"""
        self.__dict__["_caplens_host"] = caption_lengths_host
        return self

    def _encoder_autograd(self, seq, seq_len, seed=None, site=None):
        return _caption_encoder_autograd(self.caption_encoder, seq, seq_len, seed, site)

    def _forward_autograd(self, image_features, encoded_captions, caption_lengths, encoded_previous_captions,
                          previous_cap_length, use_ss, ss_prob, image_mean=None):
        """The reference loop (editnet.py:479-548), one autograd-wrapped HIP operator per module call."""
        from . import autograd_ops as A
        from . import rng
        dev = image_features.device
        batch_size = encoded_captions.size(0)
        # (stable: rows of equal length keep their input order — any order is a valid outcome of the reference's unstable
        # sort, this one makes the row <-> dropout-stream assignment a function of the inputs alone)
        # `caption_lengths_host` (set by the training steps from the data loader's host copy of the lengths, editnet.py:560-563):
        # the sort order and the decode lengths are computed on the host — no device sort, no `.tolist()` round trip in the
        # middle of the step — and a batch that already is in order (uniform lengths, a sorting loader) is not gathered at all
        host = self.__dict__.pop("_caplens_host", None)
        order = None
        if host is not None:
            hl = np.asarray(host.cpu() if torch.is_tensor(host) else host, dtype=np.int64).reshape(-1)
            if hl.shape[0] == batch_size:
                order = np.argsort(-hl, kind="stable")           # the same permutation as the stable device sort below
        if order is not None:
            decode_lengths = (hl[order] - 1).tolist()
            if bool((order == np.arange(batch_size)).all()):
                sort_ind = _arange_cached(batch_size, dev)
                X, prev, plen = _f32c(image_features), encoded_previous_captions, previous_cap_length
            else:
                sort_ind = torch.from_numpy(order).to(dev, non_blocking=True)
                X = _f32c(image_features[sort_ind])
                encoded_captions = encoded_captions[sort_ind]
                prev = encoded_previous_captions[sort_ind]
                plen = previous_cap_length[sort_ind]
        else:
            caption_lengths, sort_ind = caption_lengths.squeeze(1).sort(dim=0, descending=True, stable=True)
            X = _f32c(image_features[sort_ind])
            encoded_captions = encoded_captions[sort_ind]
            prev = encoded_previous_captions[sort_ind]
            plen = previous_cap_length[sort_ind]
            decode_lengths = (caption_lengths - 1).tolist()
        if not _XE_SEQUENCE:                                      # (the sequence node starts from its own zero state)
            h1, c1 = self.init_hidden_state(batch_size)
            h2, c2 = self.init_hidden_state(batch_size)
        preds_t = []
        # ONE seed per forward call; every dropout / sampling site is its own Philox offset (rng.py)
        seed = self.__dict__["_fwd_seed"] = rng.next_seed()
        p_emb, p_reg, p_out = self.embed.dropout.p, self.visual_attention.att_embed[2].p, self.dropout.p
        training = self.training
        H, M, final_hidden, mask = self._encoder_autograd(prev, plen, seed)
        mean = X.mean(1) if image_mean is None else image_mean[sort_ind]
        ca, va, cl, al = self.caption_attention, self.visual_attention, self.copy_lstm, self.attention_lstm
        E = self.embed.embedding.weight
        att1_c_all = A.linear(H, ca.cap_features_att.weight, ca.cap_features_att.bias)   # loop invariant (editnet.py:370)
        # adaptive features (editnet_adaptive.py:438-453): att_embed only sees the valid (non-zero) regions, the
        # padded ones stay exactly zero; the score mask is re-derived from the embedded rows
        valid = (X.sum(2) != 0).float().unsqueeze(2) if self._adaptive else None

        # relu(att_embed.0(X)) (editnet.py:441, 19.3 GFLOP at B=128) does not depend on the timestep: only the Dropout(0.5)
        # mask drawn on top of it does.  It is contracted ONCE per sequence — and so is its weight gradient, from the sum
        # over the timesteps of the masked upstream gradients that autograd accumulates — instead of once per timestep
        # as the reference writes it; values and gradients are those of the reference for the same masks.
        Y = A.linear(X, va.att_embed[0].weight, va.att_embed[0].bias, _lib.ACT_RELU)

        def embed_regions(Yb, vb, t=0):
            fe = A.philox_dropout(Yb, p_reg, seed, rng.offset(rng.SITE_REGION, t), training)   # fresh mask per timestep
            if vb is None:
                return fe, None
            fe = fe * vb
            return fe, (fe.detach().sum(2) != 0).float()

        if _XE_SEQUENCE:
            # the whole loop as ONE autograd node (xe_sequence.py): logs instead of cat / add / per-step bias sums
            from . import xe_sequence as S
            Yv = Y if valid is None else Y * valid      # adaptive: padded regions stay exactly zero through the dropout
            Yin = Yv if self.training else A.linear(Yv, va.features_att.weight, va.features_att.bias)
            cfg = S.SeqConfig(decode_lengths, self.training, p_emb, p_reg, p_out, seed)
            if use_ss and ss_prob > 0.0:
                cfg.ss_prob = float(ss_prob)
            if self._adaptive:
                cfg.adaptive = True
                cfg.rmask = None if self.training else (Yv.detach().sum(2) != 0).float()
            out = S.xe_sequence(cfg, X, mean, H, M, final_hidden, mask, att1_c_all, Yin, encoded_captions,
                                S.decoder_params(self))
            if self._adaptive:
                out, self._last_hidden = out
            if cfg.ss_prob > 0.0:
                self.__dict__["_fed_tokens"] = cfg.fed_tokens       # (T, B) words the steps consumed (diagnostics / tests)
            return out, encoded_captions, decode_lengths, sort_ind

        att1_eval = rmask_eval = None
        if not self.training:            # dropout inactive: features_att(att_embed(X)) is loop invariant
            fe, rmask_eval = embed_regions(Y, valid)
            att1_eval = A.linear(fe, va.features_att.weight, va.features_att.bias)
        prev_scores = ss_coin = None
        if use_ss and ss_prob > 0.0:
            fed = self.__dict__["_fed_tokens"] = encoded_captions[:, :max(decode_lengths)].t().clone()
        last_parts = []                  # rows leaving the batch after this step, with their final h2 (adaptive :560)

        def head(x, n):                  # x[:n] without a SliceBackward (zero-fill + copy of the whole tensor) when n is all rows
            return x if x.shape[0] == n else x[:n]

        # fc over all timesteps in ONE contraction (the weight is read once, not 19 times) when no step needs
        # the previous step's scores and no sequence finishes early
        batch_fc = not use_ss and min(decode_lengths) == max(decode_lengths)
        h2_t = []
        for t in range(max(decode_lengths)):
            bt = sum([l > t for l in decode_lengths])
            it = encoded_captions[:bt, t]
            if use_ss and t >= 1 and ss_prob > 0.0:                                   # editnet.py:508-520
                # the coin and the draw come from the same Philox streams as on the whole-sequence node (rng.py)
                if ss_coin is None:
                    ss_coin = (rng.uniforms(max(decode_lengths) * batch_size, seed, rng.offset(rng.SITE_SS_COIN), dev)
                               < ss_prob).view(-1, batch_size)
                drawn = A.philox_categorical(prev_scores[:bt].detach(), seed, rng.offset(rng.SITE_SS_DRAW, t))
                it = torch.where(ss_coin[t, :bt], drawn, it)
                fed[t, :bt] = it
            emb = A.philox_dropout(A.embed_relu(it, E), p_emb, seed, rng.offset(rng.SITE_EMBED, t), training)
            x1 = torch.cat([emb, head(final_hidden, bt), head(h2, bt), head(mean, bt)], 1)
            h1, c1 = A.lstm_cell(x1, head(h1, bt), head(c1, bt), al.weight_ih, al.weight_hh, al.bias_ih, al.bias_hh)
            attend_cap, alpha_c = A.caption_attention(
                head(H, bt), h1, emb, head(mask, bt), ca.cap_features_att.weight, ca.cap_features_att.bias,
                ca.cap_decoder_att.weight, ca.cap_decoder_att.bias, ca.cap_full_att.weight, ca.cap_full_att.bias,
                ca.context_gate.weight, ca.context_gate.bias, ca.sc_affine.weight, ca.sc_affine.bias,
                ca.tc_affine.weight, ca.tc_affine.bias, att1_c=head(att1_c_all, bt))
            if att1_eval is not None:
                att1, rmask = head(att1_eval, bt), (None if rmask_eval is None else head(rmask_eval, bt))
            else:                                                                     # fresh dropout mask per step
                fe, rmask = embed_regions(head(Y, bt), None if valid is None else head(valid, bt), t)
                att1 = A.linear(fe, va.features_att.weight, va.features_att.bias)
            attend_img = A.visual_attention_from_att1(head(X, bt), att1, h1, va.decoder_att.weight, va.decoder_att.bias,
                                                      va.full_att.weight, va.full_att.bias, rmask)
            sel = A.select(head(M, bt), alpha_c)
            h2, c2 = A.copy_lstm(torch.cat([h1, attend_cap, attend_img], 1), head(h2, bt), head(c2, bt), sel,
                                 cl.x2h.weight, cl.x2h.bias, cl.h2h.weight, cl.h2h.bias, cl.gate_cnew.weight,
                                 cl.gate_cnew.bias, cl.gate_cmem.weight, cl.gate_cmem.bias)
            bt_next = sum([l > t + 1 for l in decode_lengths])
            if bt_next < bt:
                last_parts.append(h2[bt_next:bt])
            h2d = A.philox_dropout(h2, p_out, seed, rng.offset(rng.SITE_OUT, t), training)
            if batch_fc:
                h2_t.append(h2d)
                continue
            preds = A.linear(h2d, self.fc.weight, self.fc.bias)
            prev_scores = preds
            if bt < batch_size:
                preds = torch.cat([preds, preds.new_zeros(batch_size - bt, preds.shape[1])], 0)
            preds_t.append(preds)
        if self._adaptive:           # only the adaptive forward returns it (editnet_adaptive.py:560-562)
            self._last_hidden = torch.cat(last_parts[::-1], 0)     # row order: the longest captions leave last
        if batch_fc:                     # (T, B, V) computed at once; returned as its (B, T, V) view
            predictions = A.linear(torch.stack(h2_t, 0), self.fc.weight, self.fc.bias).transpose(0, 1)
            return predictions, encoded_captions, decode_lengths, sort_ind
        predictions = torch.stack(preds_t, 1)
        return predictions, encoded_captions, decode_lengths, sort_ind
