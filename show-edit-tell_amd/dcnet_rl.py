"""DCNet free-running decode on MI355X (reference `dcnet_rl.py:256-361`)."""
from __future__ import annotations

import ctypes as C

import torch
import torch.nn as nn

from . import _lib
from ._lib import check, ptr, stream_of
from .dcnet import DAE as _DAE_XE
from .dcnet import CaptionAttention, CaptionEncoder, Embedding  # noqa: F401
from .editnet import _i64c, _require_cuda


class DAE(_DAE_XE):
    """reference dcnet_rl.py:256-346: forward(word_map, prev, prevlen, sample_max, sample_rl)"""

    max_len = 18

    def forward(self, word_map, encoded_previous_captions, previous_cap_length, sample_max=True, sample_rl=False):
        _require_cuda(encoded_previous_captions, "previous captions")
        if (self.training or (torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()))):
            return self._rollout_autograd(word_map, encoded_previous_captions, previous_cap_length, sample_max, sample_rl)
        lib = _lib.load()
        dev = encoded_previous_captions.device
        prev = _i64c(encoded_previous_captions)
        plen = _i64c(previous_cap_length.reshape(-1))
        B, max_len = prev.shape[0], self.max_len
        dims = self._dims(B, prev.shape[1], max_len + 1)
        ws = self._workspace(dims)
        w = self._weights(dims)
        limits = getattr(self, "row_limits", None)             # see editnet_rl.DecoderC.row_limits
        if limits is not None:
            if limits.dtype != torch.int32 or not limits.is_cuda or limits.numel() != B:
                raise _lib.SetError("row_limits must be an int32 device tensor with one entry per row")
            limits = limits.contiguous()
        seq = torch.empty(B, max_len, dtype=torch.long, device=dev)
        seq_logp = torch.empty(B, max_len, dtype=torch.float32, device=dev)
        # (thread-local library state, set for the duration of this enqueue only — see editnet_rl.DecoderC.forward)
        lib.set_decode_row_limits(ptr(limits) if limits is not None else None)
        try:
            if sample_rl:        # multinomial sampling, eval mode, no gradients: fused device loop, Philox epilogue
                from . import rng
                seed = rng.next_seed()
                check(lib.set_dcnet_sample(C.byref(w), C.byref(dims), ptr(prev), ptr(plen), int(word_map['<start>']),
                                           int(word_map['<end>']), max_len, seed, rng.offset(rng.SITE_ROLLOUT), ptr(seq),
                                           ptr(seq_logp), ptr(ws),
                                           ws.numel(), stream_of(dev)), "set_dcnet_sample")
                return seq, seq_logp
            check(lib.set_dcnet_greedy(C.byref(w), C.byref(dims), ptr(prev), ptr(plen), int(word_map['<start>']),
                                       int(word_map['<end>']), max_len, ptr(seq), ptr(seq_logp), ptr(ws), ws.numel(),
                                       stream_of(dev)), "set_dcnet_greedy")
            return seq, seq_logp
        finally:
            if limits is not None:
                lib.set_decode_row_limits(None)


def _dae_rollout(self, word_map, encoded_previous_captions, previous_cap_length, sample_max, sample_rl):
    """dcnet_rl.py:286-346 over autograd-wrapped HIP operators (sampled SCST rollout / multinomial sampling)."""
    import torch.nn.functional as F
    from . import autograd_ops as A
    from . import rng
    dev = encoded_previous_captions.device
    B, max_len = encoded_previous_captions.shape[0], self.max_len
    seed = self.__dict__["_fwd_seed"] = rng.next_seed()       # one seed per forward call (rng.py)
    training, p_emb, p_out = self.training, self.embed.dropout.p, self.dropout.p
    seq = torch.zeros(B, max_len, dtype=torch.long, device=dev)
    logps = []
    it = torch.full((B,), int(word_map['<start>']), dtype=torch.long, device=dev)
    h1, c1 = self.init_hidden_state(B)
    h2, c2 = self.init_hidden_state(B)
    enc, final_hidden, mask = self._encoder_autograd(encoded_previous_captions, previous_cap_length, seed)
    ca = self.caption_attention
    att1_c = A.linear(enc, ca.cap_features_att.weight, ca.cap_features_att.bias)
    from . import editnet as _editnet
    if sample_rl and _editnet._XE_SEQUENCE:       # the sampled rollout as ONE autograd node (dcnet_sequence.py, rollout mode)
        from . import dcnet_sequence as S
        cfg = S.SeqConfig([], training, p_emb, 0.0, p_out, seed,
                          rollout=dict(max_len=max_len, start_idx=int(word_map['<start>']), end_idx=int(word_map['<end>']),
                                       seed=seed, offset=rng.offset(rng.SITE_ROLLOUT)))
        return S.dcnet_sequence(cfg, enc, final_hidden, mask, att1_c, torch.zeros(1, 1, dtype=torch.long, device=dev),
                                S.dae_params(self))
    unfinished = None
    state = (A.SampleState(B, max_len, word_map['<start>'], word_map['<end>'], dev, seed=seed,
                           offset=rng.offset(rng.SITE_ROLLOUT)) if sample_rl else None)
    for t in range(max_len + 1):
        if sample_rl:
            it = state.tokens[t]
        emb = A.philox_dropout(A.embed_relu(it, self.embed.embedding.weight), p_emb, seed, rng.offset(rng.SITE_EMBED, t), training)
        h1, c1, h2, c2 = self._step_autograd(emb, final_hidden, enc, mask, h1, c1, h2, c2, att1_c)
        logits = A.linear(A.philox_dropout(h2, p_out, seed, rng.offset(rng.SITE_OUT, t), training), self.fc.weight, self.fc.bias)
        if t == max_len:
            break
        if sample_rl:                    # dcnet_rl.py:320-340 on the device (Philox draw), no host sync
            logps.append(A.sample_pick(logits, state, t))
            continue
        sample_logp, it = torch.max(F.log_softmax(logits, dim=1), 1)
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


DAE._rollout_autograd = _dae_rollout


class DAEWithAR(nn.Module):
    """reference dcnet_rl.py:348-361: wraps a trained DAE (+ an `affine_hidden` layer for the MSE
    variant).  The reference loads 'BEST_checkpoint_3_dae.pth.tar' inside __init__; here the DAE is
    passed in (or loaded from `checkpoint` if given) so the class is usable without that file."""

    def __init__(self, dae=None, checkpoint=None):
        super().__init__()
        if dae is None:
            if checkpoint is None:
                checkpoint = 'BEST_checkpoint_3_dae.pth.tar'
            dae = torch.load(checkpoint, weights_only=False)['dae']
        self.dae = dae
        decoder_dim = self.dae.decoder_dim
        self.affine_hidden = nn.Linear(decoder_dim, decoder_dim)

    def forward(self, *args, **kwargs):
        return self.dae(*args, **kwargs)
