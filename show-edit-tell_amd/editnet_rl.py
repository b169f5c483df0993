"""EditNet free-running decode (greedy / sampling) on MI355X.

Mirrors `/root/reference/editnet_rl.py:455-549`: same `DecoderC` constructor and attributes as
`editnet.DecoderC`; `forward(word_map, encoded_previous_captions, previous_cap_length,
image_features, sample_max, sample_rl)` returns `(seq (B,18) int64, seqLogprobs (B,18))`.
The whole 19-timestep loop runs on the device without host synchronisation (the reference
synchronises every step at editnet_rl.py:546).
"""
from __future__ import annotations

import ctypes as C

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib
from ._lib import check, ptr, stream_of
from .editnet import (CaptionAttentionC, CaptionEncoderC, CopyLSTMCellC, EmbeddingC, LSTMCellC,  # noqa: F401
                      SelectC, VisualAttentionC, _f32c, _i64c, _require_cuda)
from .editnet import DecoderC as _DecoderXE


class DecoderC(_DecoderXE):
    """reference editnet_rl.py:455-549"""

    max_len = 18
    # Optional per-row cap on the caption length of the no-grad greedy loop (include/set_hip.h set_decode_row_limits): an
    # int32 device tensor of B entries, or None.  Row b is ended by the loop after at most row_limits[b] words.
    row_limits = None

    def forward(self, word_map, encoded_previous_captions, previous_cap_length, image_features, sample_max=True,
                sample_rl=False, image_mean=None, repeat_images=1):
        """repeat_images = n (an extension for BASELINE.json configs[4], n sampled rollouts per image): `image_features`
        holds B images while the captions hold n * B rows, sample-major (row s * B + b belongs to image b)."""
        _require_cuda(image_features, "image features")
        if (self.training or (torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()))):
            return self._rollout_autograd(word_map, encoded_previous_captions, previous_cap_length, image_features,
                                          sample_max, sample_rl, image_mean, repeat_images)
        if repeat_images > 1:
            image_features = image_features.repeat(repeat_images, 1, 1)
            image_mean = None if image_mean is None else image_mean.repeat(repeat_images, 1)
        lib = _lib.load()
        limits = self.row_limits
        if limits is not None:
            if limits.dtype != torch.int32 or not limits.is_cuda or limits.numel() != image_features.shape[0]:
                raise _lib.SetError("row_limits must be an int32 device tensor with one entry per row")
            limits = limits.contiguous()
        # the limit pointer is thread-local state of the library that every greedy pick of this host thread reads: it is set
        # for the duration of THIS enqueue only (it used to stay set — a later caller with another B, or after the
        # tensor was freed, would have read it)
        lib.set_decode_row_limits(_lib.ptr(limits) if limits is not None else None)
        try:
            return self._decode_nograd(lib, word_map, encoded_previous_captions, previous_cap_length, image_features,
                                       sample_rl, image_mean)
        finally:
            if limits is not None:
                lib.set_decode_row_limits(None)

    def _decode_nograd(self, lib, word_map, encoded_previous_captions, previous_cap_length, image_features, sample_rl,
                       image_mean):
        dev = image_features.device
        X = _f32c(image_features)
        prev = _i64c(encoded_previous_captions)
        plen = _i64c(previous_cap_length.reshape(-1))
        mean = None if image_mean is None else _f32c(image_mean)
        B = X.shape[0]
        max_len = self.max_len
        dims = self._dims(B, prev.shape[1], X.shape[1], max_len + 1)
        seq = torch.empty(B, max_len, dtype=torch.long, device=dev)
        seq_logp = torch.empty(B, max_len, dtype=torch.float32, device=dev)
        ticket = None if sample_rl else self._take_ahead(X, prev, plen, mean)
        if ticket is not None and ticket.get("out") is not None:
            # decode_ahead: the whole decode of this batch already ran (or is running) on another stream
            cur = torch.cuda.current_stream(dev)
            cur.wait_event(ticket["event"])
            out = ticket.pop("out")
            for t in out:
                t.record_stream(cur)
            ticket["inputs"] = None
            self.__dict__["_ahead_hits"] = self.__dict__.get("_ahead_hits", 0) + 1
            return out
        if ticket is not None:
            # the prologue of this batch already ran (begin_ahead) on another stream into its own workspace: order this
            # stream after it and run the timestep loop alone
            cur = torch.cuda.current_stream(dev)
            cur.wait_event(ticket["event"])
            w = ticket["weights"]
            check(lib.set_editnet_greedy_begun(C.byref(w), C.byref(dims), ptr(X), int(word_map['<start>']),
                                               int(word_map['<end>']), max_len, ptr(seq), ptr(seq_logp), ptr(ticket["ws"]),
                                               ticket["ws"].numel(), stream_of(dev)), "set_editnet_greedy_begun")
            ticket["done"].record(cur)                     # the workspace may be begun again once this loop has finished
            ticket["inputs"] = None
            self.__dict__.setdefault("_ahead_free", []).append(ticket)
            self.__dict__["_ahead_hits"] = self.__dict__.get("_ahead_hits", 0) + 1
            return seq, seq_logp
        ws = self._workspace(dims)
        w = self._weights(dims)
        if sample_rl:        # multinomial sampling, eval mode, no gradients: fused device loop with the Philox epilogue
            from . import rng
            seed = rng.next_seed()                                      # torch.manual_seed() makes it reproducible
            check(lib.set_editnet_sample(C.byref(w), C.byref(dims), ptr(X), ptr(mean), ptr(prev), ptr(plen),
                                         int(word_map['<start>']), int(word_map['<end>']), max_len, seed,
                                         rng.offset(rng.SITE_ROLLOUT), ptr(seq),
                                         ptr(seq_logp), ptr(ws), ws.numel(), stream_of(dev)), "set_editnet_sample")
            return seq, seq_logp
        check(lib.set_editnet_greedy(C.byref(w), C.byref(dims), ptr(X), ptr(mean), ptr(prev), ptr(plen),
                                     int(word_map['<start>']), int(word_map['<end>']), max_len, ptr(seq),
                                     ptr(seq_logp), ptr(ws), ws.numel(), stream_of(dev)), "set_editnet_greedy")
        return seq, seq_logp

    # ---- prologue-ahead (the reference's callers issue one decode after the other: train(), evaluate()) -------------
    def begin_ahead(self, encoded_previous_captions, previous_cap_length, image_features, image_mean=None):
        """Run the per-sequence prologue (caption encoder + hoisted projections, editnet_rl.py:499-501) of a batch that
        will be decoded LATER, on the CURRENT stream, into a workspace of its own.  The next greedy `forward` that is given
        these very tensors finds the prologue done and runs only its timestep loop (bit-identical results).  Call it from a
        side stream while the previous batch decodes — `pipeline.DevicePrefetcher(..., begin_ahead=...)` does that.
        Eval mode / no-grad only; anything else is ignored (the forward then runs its own prologue)."""
        if self.training or (torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())):
            return False
        _require_cuda(image_features, "image features")
        lib = _lib.load()
        dev = image_features.device
        X = _f32c(image_features)
        prev = _i64c(encoded_previous_captions)
        plen = _i64c(previous_cap_length.reshape(-1))
        mean = None if image_mean is None else _f32c(image_mean)
        dims = self._dims(X.shape[0], prev.shape[1], X.shape[1], self.max_len + 1)
        cur = torch.cuda.current_stream(dev)
        free = self.__dict__.setdefault("_ahead_free", [])
        key = tuple(getattr(dims, f) for f, _ in _lib.EditNetDims._fields_) + (str(dev),)
        ticket = None
        for i, t in enumerate(free):
            if t["dims_key"] == key:
                ticket = free.pop(i)
                cur.wait_event(ticket["done"])          # its previous decode must have finished with the workspace
                break
        if ticket is None:
            n = lib.set_editnet_workspace_bytes(C.byref(dims))
            if n == 0:
                raise _lib.SetError("unsupported EditNet dims %r" % (key,))
            ticket = {"dims_key": key, "ws": torch.empty(n, dtype=torch.uint8, device=dev),
                      "event": torch.cuda.Event(), "done": torch.cuda.Event()}
        w = self._weights(dims)
        check(lib.set_editnet_begin(C.byref(w), C.byref(dims), ptr(X), ptr(mean), ptr(prev), ptr(plen), ptr(ticket["ws"]),
                                    ticket["ws"].numel(), stream_of(dev)), "set_editnet_begin")
        ticket["event"].record(cur)
        # the decode must see the SAME weights view (token table attached or not) the prologue was built with
        ticket.update(weights=w, inputs=(X, prev, plen, mean), sig=self._ahead_sig(X, prev, plen, mean),
                      tab=self.__dict__.get("_tok_state", {}).get("table"), wsig=self._weights_sig())
        pending = self.__dict__.setdefault("_ahead", [])
        pending.append(ticket)
        while len(pending) > 4:                          # prologues nobody came back for: recycle the oldest
            old = pending.pop(0)
            old["done"].record(cur)
            old["inputs"] = None
            free.append(old)
        return True

    def _weights_sig(self):
        from . import optim as _optim
        return tuple(p._version for p in self.parameters()) + (_optim.weights_epoch(),)

    def decode_ahead(self, word_map, encoded_previous_captions, previous_cap_length, image_features, image_mean=None):
        """The whole greedy decode of a batch that the caller will ask for LATER, issued now on the CURRENT stream (a side
        stream of pipeline.DevicePrefetcher): the `forward` call that is given these very tensors — with unchanged
        weights — returns this result after ordering the caller's stream behind it.  For inference loops (the reference's
        evaluate(): fixed weights, one batch after the other) this keeps several decodes in flight without the caller
        managing streams; a weight update in between discards the result and the forward decodes again.  Same kernels,
        same arithmetic: bit-identical to the plain call."""
        if self.training or (torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())):
            return False
        pending = self.__dict__.setdefault("_ahead", [])
        X = _f32c(image_features)
        prev = _i64c(encoded_previous_captions)
        plen = _i64c(previous_cap_length.reshape(-1))
        mean = None if image_mean is None else _f32c(image_mean)
        self.__dict__["_ahead_busy"] = True              # (the forward below must not look for a ticket itself)
        try:
            out = self.forward(word_map, prev, plen, X, True, False, image_mean=mean)
        finally:
            self.__dict__["_ahead_busy"] = False
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(X.device))
        pending.append({"out": out, "event": ev, "inputs": (X, prev, plen, mean), "sig": self._ahead_sig(X, prev, plen, mean),
                        "tab": self.__dict__.get("_tok_state", {}).get("table"), "wsig": self._weights_sig()})
        while len(pending) > 8:
            pending.pop(0)
        return True

    @staticmethod
    def _ahead_sig(X, prev, plen, mean):
        return tuple((t.data_ptr(), tuple(t.shape), t._version) for t in (X, prev, plen)) + \
            ((mean.data_ptr(), mean._version) if mean is not None else None,)

    def _take_ahead(self, X, prev, plen, mean):
        pending = self.__dict__.get("_ahead")
        if not pending or self.__dict__.get("_ahead_busy"):
            return None
        sig = self._ahead_sig(X, prev, plen, mean)
        tab = self.__dict__.get("_tok_state", {}).get("table")
        for i, t in enumerate(pending):
            if t["sig"] == sig:
                pending.pop(i)
                if t["tab"] is not tab or t["wsig"] != self._weights_sig():
                    # a weight changed (or the token table appeared / was dropped) in between: the prologue is stale
                    if "ws" in t:
                        self.__dict__.setdefault("_ahead_free", []).append(t)
                        t["done"].record(torch.cuda.current_stream(X.device))
                    return None
                return t
        return None

    def _rollout_autograd(self, word_map, encoded_previous_captions, previous_cap_length, image_features, sample_max,
                          sample_rl, image_mean=None, repeat_images=1):
        """The reference loop editnet_rl.py:485-549 over autograd-wrapped HIP operators: used for the
        sampled SCST rollout (train mode, dropout active, gradients flow through seqLogprobs) and for
        the grad-enabled greedy decode.  Sampling runs in the HIP epilogue `set_sample_pick_f32` (Philox draw,
        log-prob gather, <end> / unfinished / break bookkeeping on the device): the sampled loop never synchronises
        with the host (the reference does every step, editnet_rl.py:546)."""
        from . import autograd_ops as A
        from . import rng
        if self._adaptive:
            raise NotImplementedError("rollout with adaptive features is not built yet")
        # ONE seed per forward call: dropout sites and the multinomial draws are separate Philox offsets of it (rng.py)
        seed = self.__dict__["_fwd_seed"] = rng.next_seed()
        training = self.training
        p_emb, p_reg, p_out = self.embed.dropout.p, self.visual_attention.att_embed[2].p, self.dropout.p
        dev = image_features.device
        X1 = _f32c(image_features)                       # one row per IMAGE
        rep = (lambda t: t) if repeat_images <= 1 else (lambda t: t.repeat(repeat_images, *([1] * (t.dim() - 1))))
        X = rep(X1)                                      # one row per ROLLOUT (the region stream of the visual attention)
        B, max_len = X.shape[0], self.max_len
        seq = torch.zeros(B, max_len, dtype=torch.long, device=dev)
        logps = []
        it = torch.full((B,), int(word_map['<start>']), dtype=torch.long, device=dev)
        h1, c1 = self.init_hidden_state(B)
        h2, c2 = self.init_hidden_state(B)
        H, M, final_hidden, mask = self._encoder_autograd(encoded_previous_captions, previous_cap_length, seed)
        mean = rep(X1.mean(1) if image_mean is None else image_mean)
        ca, va, cl, al = self.caption_attention, self.visual_attention, self.copy_lstm, self.attention_lstm
        E = self.embed.embedding.weight
        att1_c_all = A.linear(H, ca.cap_features_att.weight, ca.cap_features_att.bias)
        # relu(att_embed.0(X)) is loop invariant (only its dropout mask is per step): contracted once, see editnet.py — and
        # once per IMAGE: the rows of the n rollouts of an image are copies (autograd sums their gradients back)
        Y = rep(A.linear(X1, va.att_embed[0].weight, va.att_embed[0].bias, _lib.ACT_RELU))
        att1_eval = None
        if not self.training:
            att1_eval = A.linear(Y, va.features_att.weight, va.features_att.bias)
        from . import editnet as _editnet
        if sample_rl and _editnet._XE_SEQUENCE:
            # the sampled rollout as ONE autograd node (xe_sequence.py, rollout mode): the same kernels, logs instead of
            # per-step autograd nodes; the final, unused step of the reference loop (t == max_len) is not run
            from . import xe_sequence as S
            cfg = S.SeqConfig([], training, p_emb, p_reg, p_out, seed,
                              rollout=dict(max_len=max_len, start_idx=int(word_map['<start>']), end_idx=int(word_map['<end>']),
                                           seed=seed, offset=rng.offset(rng.SITE_ROLLOUT)))
            return S.xe_sequence(cfg, X, mean, H, M, final_hidden, mask, att1_c_all, Y if self.training else att1_eval,
                                 torch.zeros(1, 1, dtype=torch.long, device=dev), S.decoder_params(self))
        unfinished = None
        state = (A.SampleState(B, max_len, word_map['<start>'], word_map['<end>'], dev, seed=seed,
                               offset=rng.offset(rng.SITE_ROLLOUT)) if sample_rl else None)
        for t in range(max_len + 1):
            if sample_rl:
                it = state.tokens[t]
            emb = A.philox_dropout(A.embed_relu(it, E), p_emb, seed, rng.offset(rng.SITE_EMBED, t), training)
            h1, c1 = A.lstm_cell(torch.cat([emb, final_hidden, h2, mean], 1), h1, c1, al.weight_ih, al.weight_hh,
                                 al.bias_ih, al.bias_hh)
            attend_cap, alpha_c = A.caption_attention(
                H, h1, emb, mask, ca.cap_features_att.weight, ca.cap_features_att.bias, ca.cap_decoder_att.weight,
                ca.cap_decoder_att.bias, ca.cap_full_att.weight, ca.cap_full_att.bias, ca.context_gate.weight,
                ca.context_gate.bias, ca.sc_affine.weight, ca.sc_affine.bias, ca.tc_affine.weight, ca.tc_affine.bias,
                att1_c=att1_c_all)
            if att1_eval is not None:
                att1 = att1_eval
            else:
                att1 = A.linear(A.philox_dropout(Y, p_reg, seed, rng.offset(rng.SITE_REGION, t), training),
                                va.features_att.weight, va.features_att.bias)
            attend_img = A.visual_attention_from_att1(X, att1, h1, va.decoder_att.weight, va.decoder_att.bias,
                                                      va.full_att.weight, va.full_att.bias)
            sel = A.select(M, alpha_c)
            h2, c2 = A.copy_lstm(torch.cat([h1, attend_cap, attend_img], 1), h2, c2, sel, cl.x2h.weight, cl.x2h.bias,
                                 cl.h2h.weight, cl.h2h.bias, cl.gate_cnew.weight, cl.gate_cnew.bias,
                                 cl.gate_cmem.weight, cl.gate_cmem.bias)
            logits = A.linear(A.philox_dropout(h2, p_out, seed, rng.offset(rng.SITE_OUT, t), training),
                              self.fc.weight, self.fc.bias)
            if t == max_len:
                break
            if sample_rl:                # editnet_rl.py:521-543 on the device, no host sync
                logps.append(A.sample_pick(logits, state, t))
                continue
            logprobs = F.log_softmax(logits, dim=1)
            sample_logp, it = torch.max(logprobs, 1)
            it = it.clone()
            it[it == int(word_map['<end>'])] = 0
            unfinished = (it > 0) if t == 0 else unfinished * (it > 0)
            it = it * unfinished.type_as(it)
            seq[:, t] = it
            logps.append(sample_logp.view(-1))
            if unfinished.sum() == 0:
                break
        if sample_rl:
            seq = state.seq
        seq_logp = torch.stack(logps, 1)
        if seq_logp.shape[1] < max_len:
            seq_logp = torch.cat([seq_logp, seq_logp.new_zeros(B, max_len - seq_logp.shape[1])], 1)
        return seq, seq_logp


class RewardCriterion(nn.Module):
    """reference editnet_rl.py:553-573 (a loss on the path's output; plain tensor ops)."""

    def forward(self, sample_logprobs, seq, reward):
        sample_logprobs = sample_logprobs.reshape(-1)
        reward = reward.reshape(-1)
        mask = (seq > 0).float()
        mask = torch.cat([mask.new_ones(mask.size(0), 1), mask[:, :-1]], 1).reshape(-1)
        return torch.sum(-sample_logprobs * reward * mask) / torch.sum(mask)
