"""DCNet (DAE) on MI355X: the reference's `DAE` module surface over the HIP decode path.

Mirrors `/root/reference/dcnet.py:147-350`: `Embedding`, `CaptionEncoder`, `CaptionAttention`,
`DAE` with the same constructor signatures, attribute names and `state_dict` keys.  DCNet is
text-only (no image features, `dcnet.py:303`).
"""
from __future__ import annotations

import ctypes as C

import torch
import torch.nn as nn

from . import _lib
from ._lib import DcnetDims, DcnetWeights, DCNET_WEIGHT_FIELDS, EditNetWeights, check, ptr, stream_of
from .editnet import _HipLinear, _HipLSTMCell, _f32c, _i64c, _require_cuda, _wants_grad


class Embedding(nn.Module):
    """reference dcnet.py:147-206 (the GloVe branch is dead code there: load_glove_embedding=False)"""

    def __init__(self, word_map, emb_file, emb_dim, load_glove_embedding=False):
        super().__init__()
        if load_glove_embedding:
            raise NotImplementedError("the reference hard-wires load_glove_embedding=False (dcnet.py:288)")
        self.emb_dim = emb_dim
        self.load_glove_embedding = load_glove_embedding
        self.emb_file = emb_file
        self.word_map = word_map
        self.embedding = nn.Embedding(len(word_map), self.emb_dim)
        self.relu = nn.ReLU()
        self.dropout = nn.Dropout(0.5)

    def forward(self, x):
        _require_cuda(x, "token ids")
        if self.training or _wants_grad(self):
            from . import autograd_ops as A
            from . import rng
            # dcnet.py:203-205; a direct call draws its own seed (DAE.forward addresses the site itself)
            return A.philox_dropout(A.embed_relu(_i64c(x), self.embedding.weight), self.dropout.p, rng.next_seed(),
                                    rng.offset(rng.SITE_EMBED), self.training)
        lib = _lib.load()
        ids = _i64c(x)
        n, D = ids.numel(), self.emb_dim
        out = torch.empty(tuple(ids.shape) + (D,), dtype=torch.float32, device=ids.device)
        check(lib.set_embed_relu_f32(ptr(self.embedding.weight), ptr(ids), 1, ptr(out), D, n, D,
                                     self.embedding.num_embeddings, stream_of(ids.device)), "set_embed_relu_f32")
        return out


class CaptionEncoder(nn.Module):
    """reference dcnet.py:209-243 — parameters live in an nn.LSTM container (same state_dict keys);
    the forward runs through DAE's workspace (set_dcnet_begin)."""

    def __init__(self, vocab_size, emb_dim, enc_hid_dim, concat_output_dim, embed):
        super().__init__()
        self.vocab_size = vocab_size
        self.emb_dim = emb_dim
        self.enc_hid_dim = enc_hid_dim
        self.embed = embed
        self.lstm_encoder = nn.LSTM(emb_dim, enc_hid_dim, batch_first=True, bidirectional=True)
        self.concat = nn.Linear(enc_hid_dim * 2, concat_output_dim)
        self._owner = None      # set by DAE (plain attribute, not a sub-module)

    def __getstate__(self):          # the owner link is a weakref (not picklable); DAE.__setstate__ restores it
        state = dict(self.__dict__)
        state["_owner"] = None
        return state

    def forward(self, src, src_len):
        if self._owner is None:
            raise _lib.SetError("CaptionEncoder must be owned by a DAE (it runs through the DAE workspace)")
        if self.training or _wants_grad(self):
            return self._owner()._encoder_autograd(src, src_len)
        return self._owner()._encode(src, src_len)


class CaptionAttention(nn.Module):
    """reference dcnet.py:245-270"""

    def __init__(self, caption_features_dim, decoder_dim, attention_dim):
        super().__init__()
        self.cap_features_att = nn.Linear(caption_features_dim * 2, attention_dim)
        self.cap_decoder_att = nn.Linear(decoder_dim, attention_dim)
        self.cap_full_att = nn.Linear(attention_dim, 1)

    def forward(self, caption_features, decoder_hidden, prev_caption_mask):
        _require_cuda(caption_features, "caption features")
        if _wants_grad(self, caption_features, decoder_hidden):
            from . import autograd_ops as A
            return A.dcnet_caption_attention(_f32c(caption_features), _f32c(decoder_hidden), _f32c(prev_caption_mask),
                                             self.cap_features_att.weight, self.cap_features_att.bias,
                                             self.cap_decoder_att.weight, self.cap_decoder_att.bias,
                                             self.cap_full_att.weight, self.cap_full_att.bias)
        lib = _lib.load()
        H, h1, mask = _f32c(caption_features), _f32c(decoder_hidden), _f32c(prev_caption_mask)
        M, T, Dh = H.shape
        D, A = h1.shape[1], self.cap_decoder_att.out_features
        w = EditNetWeights()          # gate weights stay NULL -> plain context (include/set_hip.h)
        w.ca_feat_w, w.ca_feat_b = self.cap_features_att.weight.data_ptr(), self.cap_features_att.bias.data_ptr()
        w.ca_dec_w, w.ca_dec_b = self.cap_decoder_att.weight.data_ptr(), self.cap_decoder_att.bias.data_ptr()
        w.ca_full_w, w.ca_full_b = self.cap_full_att.weight.data_ptr(), self.cap_full_att.bias.data_ptr()
        ctx = torch.empty(M, Dh, dtype=torch.float32, device=H.device)
        ws = torch.empty(lib.set_caption_attention_workspace_bytes(M, T, max(Dh, D), A), dtype=torch.uint8,
                         device=H.device)
        check(lib.set_caption_attention_f32(C.byref(w), ptr(H), None, ptr(h1), None, ptr(mask), ptr(ctx), None, M, T,
                                            Dh, D, A, ptr(ws), ws.numel(), stream_of(H.device)),
              "set_caption_attention_f32")
        return ctx


class DAE(nn.Module):
    """reference dcnet.py:273-350 — XE (teacher-forced) forward."""

    def __init__(self, word_map, emb_file, decoder_dim=1024, attention_dim=512, caption_features_dim=512, emb_dim=1024):
        super().__init__()
        import weakref
        self.vocab_size = len(word_map)
        self.attention_lstm = _HipLSTMCell(emb_dim * 3, decoder_dim)
        self.language_lstm = _HipLSTMCell(emb_dim * 2, decoder_dim)
        self.embed = Embedding(word_map, emb_file, emb_dim, load_glove_embedding=False)
        self.caption_encoder = CaptionEncoder(len(word_map), emb_dim, caption_features_dim, caption_features_dim * 2,
                                              self.embed)
        self.caption_encoder._owner = weakref.ref(self)
        self.caption_attention = CaptionAttention(caption_features_dim, decoder_dim, attention_dim)
        self.fc = _HipLinear(decoder_dim, len(word_map))
        self.tanh = nn.Tanh()
        self.decoder_dim = decoder_dim
        self.dropout = nn.Dropout(0.5)
        self._dims_cfg = (decoder_dim, attention_dim, caption_features_dim, emb_dim)
        self._ws = None
        self._ws_key = None

    # the reference checkpoints pickle the whole module (dcnet.py:131-138): GPU workspaces and the derived token table
    # must not travel
    def __getstate__(self):
        state = dict(self.__dict__)
        state["_ws"] = state["_ws_key"] = None
        for k in ("_tok_state", "_ws_cache", "_fwd_seed", "_grad_buckets"):
            state.pop(k, None)
        return state

    def invalidate_token_table(self):
        """Drop the derived inference-time token table (see _token_table); same contract as editnet.DecoderC's."""
        self.__dict__.pop("_tok_state", None)

    def train(self, mode=True):
        if bool(mode) != self.training:
            self.invalidate_token_table()
        return super().train(mode)

    def load_state_dict(self, *args, **kwargs):
        self.invalidate_token_table()
        return super().load_state_dict(*args, **kwargs)

    def __setstate__(self, state):
        import weakref
        super().__setstate__(state)
        self.caption_encoder._owner = weakref.ref(self)

    def _apply(self, fn, *args, **kwargs):
        self._ws = self._ws_key = None
        self.__dict__.pop("_ws_cache", None)
        self.__dict__.pop("_grad_buckets", None)
        self.invalidate_token_table()
        return super()._apply(fn, *args, **kwargs)

    def init_hidden_state(self, batch_size):
        dev = self.fc.weight.device
        return (torch.zeros(batch_size, self.decoder_dim, device=dev),
                torch.zeros(batch_size, self.decoder_dim, device=dev))

    # ---- runtime plumbing
    def _weights(self, dims=None):
        """Pack the parameter pointers; with `dims` (no-grad decode paths) also attach the token table when valid."""
        w = _lib.pack_weights(DcnetWeights, DCNET_WEIGHT_FIELDS, dict(self.named_parameters()), self.fc.weight.device)
        if dims is not None:
            tab = self._token_table(dims)
            if tab is not None:
                w.tok_table = tab.data_ptr()
        return w

    # The contractions whose only input is a token (attention_lstm.W_ih[:, :E] relu(E[v]) and the BiLSTM encoder's two
    # input projections) are folded into a (V, 4D + 8C) table (include/set_hip.h: SetDcnetWeights.tok_table).  Same
    # life cycle as editnet.DecoderC._token_table: built once the same source weights have been seen on two consecutive
    # no-grad calls, dropped when any of them changes (tensor._version / data_ptr), on train()/eval() switches,
    # load_state_dict() and device moves; SET_TOKEN_TABLE=0 disables, =1 forces, SET_TOKEN_TABLE_VERIFY=1 re-checks.
    def _token_table(self, dims):
        import os
        mode = os.environ.get("SET_TOKEN_TABLE", "auto")
        if mode == "0" or dims.D % 64 or dims.C % 128:
            return None
        enc = self.caption_encoder.lstm_encoder
        src = (self.embed.embedding.weight, self.attention_lstm.weight_ih, enc.weight_ih_l0, enc.bias_ih_l0,
               enc.weight_ih_l0_reverse, enc.bias_ih_l0_reverse)
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
            table = torch.empty(lib.set_dcnet_token_table_bytes(C.byref(dims)) // 4, dtype=torch.float32, device=dev)
            ws = torch.empty(lib.set_dcnet_token_table_workspace_bytes(C.byref(dims)), dtype=torch.uint8, device=dev)
            w = _lib.pack_weights(DcnetWeights, DCNET_WEIGHT_FIELDS, dict(self.named_parameters()), dev)
            check(lib.set_dcnet_build_token_table(C.byref(w), C.byref(dims), ptr(table), ptr(ws), ws.numel(), stream_of(dev)),
                  "set_dcnet_build_token_table")
            torch.cuda.current_stream(dev).synchronize()
            st["table"] = table
            st["check"] = torch.stack([t.detach().double().sum() for t in src]).cpu()
        if st["table"] is not None and os.environ.get("SET_TOKEN_TABLE_VERIFY") == "1":
            now = torch.stack([t.detach().double().sum() for t in src]).cpu()
            if not torch.equal(now, st["check"]):
                raise _lib.SetError("token table is stale: a source weight changed without bumping tensor._version "
                                    "(in-place .data write?); call dae.invalidate_token_table()")
        return st["table"]

    def _dims(self, B, T, maxT):
        D, A, Cc, E = self._dims_cfg
        return DcnetDims(B=B, T=T, D=D, A=A, C=Cc, E=E, V=self.vocab_size, maxT=maxT)

    def _workspace(self, dims):
        """One workspace per (dims, device, stream), as DecoderC._workspace: concurrent decodes on different streams (the
        self-critical step runs the greedy baseline on a side stream underneath the sampled rollout) must not share
        recurrent state or split-K slabs."""
        lib = _lib.load()
        dev = self.fc.weight.device
        key = tuple(getattr(dims, f) for f, _ in DcnetDims._fields_) + (str(dev), torch.cuda.current_stream(dev).cuda_stream)
        cache = self.__dict__.setdefault("_ws_cache", {})
        ws = cache.get(key)
        if ws is None:
            n = lib.set_dcnet_workspace_bytes(C.byref(dims))
            if n == 0:
                raise _lib.SetError("unsupported DCNet dims %r" % (key,))
            if len(cache) >= 24:
                cache.clear()
            ws = cache[key] = torch.empty(n, dtype=torch.uint8, device=dev)
        self._ws, self._ws_key = ws, key
        return ws

    def ws_tensor(self, dims, name, shape, dtype=torch.float32):
        lib = _lib.load()
        p = lib.set_dcnet_ws_tensor(C.byref(dims), ptr(self._ws), name.encode())
        if not p:
            raise KeyError(name)
        off = p - self._ws.data_ptr()
        n = int(torch.tensor(shape).prod().item()) * torch.empty((), dtype=dtype).element_size()
        return self._ws[off:off + n].view(dtype).view(*shape)

    def _encode(self, src, src_len):
        """caption_encoder(src, src_len) -> (outputs (B,Tmax,2C), final_hidden (B,2C), mask (B,Tmax))"""
        _require_cuda(src, "previous captions")
        lib = _lib.load()
        src, lens = _i64c(src), _i64c(src_len.reshape(-1))
        B, T = src.shape
        dims = self._dims(B, T, 19)
        ws = self._workspace(dims)
        w = self._weights(dims)
        check(lib.set_dcnet_begin(C.byref(w), C.byref(dims), ptr(src), ptr(lens), ptr(ws), ws.numel(),
                                  stream_of(src.device)), "set_dcnet_begin")
        Cc = self._dims_cfg[2]
        tmax = int(lens.max().item())
        enc = self.ws_tensor(dims, "enc", (B, T, 2 * Cc)).clone()
        fh = self.ws_tensor(dims, "final_hidden", (B, 2 * Cc)).clone()
        mask = self.ws_tensor(dims, "mask", (B, T)).clone()
        return enc[:, :tmax], fh, mask[:, :tmax]

    def forward(self, encoded_captions, caption_lengths, encoded_previous_captions, previous_cap_length):
        """reference dcnet.py:303-350; returns (predictions, encoded_captions sorted, decode_lengths, sort_ind)."""
        _require_cuda(encoded_captions, "captions")
        if self.training or (torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())):
            return self._forward_autograd(encoded_captions, caption_lengths, encoded_previous_captions,
                                          previous_cap_length)
        lib = _lib.load()
        dev = encoded_captions.device
        batch_size = encoded_captions.size(0)
        caption_lengths, sort_ind = caption_lengths.squeeze(1).sort(dim=0, descending=True, stable=True)
        encoded_captions = _i64c(encoded_captions[sort_ind])
        prev = _i64c(encoded_previous_captions[sort_ind])
        plen = _i64c(previous_cap_length[sort_ind].reshape(-1))
        decode_lengths = (caption_lengths - 1).tolist()
        maxT = max(decode_lengths)
        dims = self._dims(batch_size, prev.shape[1], maxT)
        ws = self._workspace(dims)
        w = self._weights(dims)
        predictions = torch.empty(batch_size, maxT, self.vocab_size, dtype=torch.float32, device=dev)
        dl = (C.c_int * batch_size)(*decode_lengths)
        check(lib.set_dcnet_xe_forward(C.byref(w), C.byref(dims), ptr(encoded_captions), encoded_captions.shape[1], dl,
                                       ptr(prev), ptr(plen), ptr(predictions), ptr(ws), ws.numel(), stream_of(dev)),
              "set_dcnet_xe_forward")
        return predictions, encoded_captions, decode_lengths, sort_ind


    # ---- grad-enabled path (HIP forward operators, autograd backward) ------------------------
    def _encoder_autograd(self, src, src_len, seed=None):
        """CaptionEncoder.forward (dcnet.py:220-243): packed BiLSTM == per-row masked recurrences.  The embedding's
        dropout (dcnet.py:224) is the Philox stream (seed, SITE_ENC_EMBED): row b * Tmax + l, see rng.py."""
        from . import autograd_ops as A
        from . import rng
        enc = self.caption_encoder
        lstm = enc.lstm_encoder
        lens = src_len.reshape(-1)
        tmax = int(lens.max().item())
        B, Cc = src.shape[0], enc.enc_hid_dim
        emb = A.philox_dropout(A.embed_relu(src[:, :tmax], self.embed.embedding.weight), self.embed.dropout.p,
                               rng.next_seed() if seed is None else seed, rng.offset(rng.SITE_ENC_EMBED), self.embed.training)
        outs, finals = [], []
        for sfx, reverse in (("", False), ("_reverse", True)):      # each direction = one autograd node
            w_ih, w_hh = getattr(lstm, "weight_ih_l0" + sfx), getattr(lstm, "weight_hh_l0" + sfx)
            b_ih, b_hh = getattr(lstm, "bias_ih_l0" + sfx), getattr(lstm, "bias_hh_l0" + sfx)
            Hd, h_last = A.encoder_lstm(emb, lens, w_ih, b_ih, w_hh, b_hh, reverse=reverse, want_mem=False)
            outs.append(Hd)
            finals.append(h_last)
        outputs = torch.cat(outs, 2)
        mask = (outputs.detach().sum(2) != 0).float()
        final_hidden = A.linear(torch.cat(finals, 1), enc.concat.weight, enc.concat.bias, _lib.ACT_TANH)
        return outputs, final_hidden, mask

    def _step_autograd(self, emb, final_hidden, enc, mask, h1, c1, h2, c2, att1_c=None):
        from . import autograd_ops as A
        al, ll, ca = self.attention_lstm, self.language_lstm, self.caption_attention
        h1, c1 = A.lstm_cell(torch.cat([emb, final_hidden, h2], 1), h1, c1, al.weight_ih, al.weight_hh, al.bias_ih,
                             al.bias_hh)
        attend_cap = A.dcnet_caption_attention(enc, h1, mask, ca.cap_features_att.weight, ca.cap_features_att.bias,
                                               ca.cap_decoder_att.weight, ca.cap_decoder_att.bias,
                                               ca.cap_full_att.weight, ca.cap_full_att.bias, att1_c=att1_c)
        h2, c2 = A.lstm_cell(torch.cat([h1, attend_cap], 1), h2, c2, ll.weight_ih, ll.weight_hh, ll.bias_ih, ll.bias_hh)
        return h1, c1, h2, c2

    def _forward_autograd(self, encoded_captions, caption_lengths, encoded_previous_captions, previous_cap_length):
        """dcnet.py:303-350 over autograd ops."""
        from . import autograd_ops as A
        from . import rng
        batch_size = encoded_captions.size(0)
        caption_lengths, sort_ind = caption_lengths.squeeze(1).sort(dim=0, descending=True, stable=True)
        encoded_captions = encoded_captions[sort_ind]
        prev = encoded_previous_captions[sort_ind]
        plen = previous_cap_length[sort_ind]
        h1, c1 = self.init_hidden_state(batch_size)
        h2, c2 = self.init_hidden_state(batch_size)
        decode_lengths = (caption_lengths - 1).tolist()
        seed = self.__dict__["_fwd_seed"] = rng.next_seed()       # one seed per forward call, one Philox offset per site (rng.py)
        training, p_emb, p_out = self.training, self.embed.dropout.p, self.dropout.p
        enc, final_hidden, mask = self._encoder_autograd(prev, plen, seed)
        ca = self.caption_attention
        att1_c = A.linear(enc, ca.cap_features_att.weight, ca.cap_features_att.bias)       # loop invariant (dcnet.py:261)
        from . import editnet as _editnet
        if _editnet._XE_SEQUENCE:         # the whole loop as ONE autograd node (dcnet_sequence.py)
            from . import dcnet_sequence as S
            cfg = S.SeqConfig(decode_lengths, training, p_emb, 0.0, p_out, seed)
            preds = S.dcnet_sequence(cfg, enc, final_hidden, mask, att1_c, encoded_captions, S.dae_params(self))
            return preds, encoded_captions, decode_lengths, sort_ind
        # dcnet.py:325 embeds (and drops out) all positions at once; position t is consumed by timestep t only, so its
        # mask is the (SITE_EMBED, t) stream over the rows still in the batch — as on the whole-sequence node
        raw = A.embed_relu(encoded_captions, self.embed.embedding.weight)
        preds_t = []
        for t in range(max(decode_lengths)):
            bt = sum([l > t for l in decode_lengths])
            emb = A.philox_dropout(raw[:bt, t], p_emb, seed, rng.offset(rng.SITE_EMBED, t), training)
            h1, c1, h2, c2 = self._step_autograd(emb, final_hidden[:bt], enc[:bt], mask[:bt], h1[:bt],
                                                 c1[:bt], h2[:bt], c2[:bt], att1_c[:bt])
            preds = A.linear(A.philox_dropout(h2, p_out, seed, rng.offset(rng.SITE_OUT, t), training),
                             self.fc.weight, self.fc.bias)
            if bt < batch_size:
                preds = torch.cat([preds, preds.new_zeros(batch_size - bt, preds.shape[1])], 0)
            preds_t.append(preds)
        return torch.stack(preds_t, 1), encoded_captions, decode_lengths, sort_ind
