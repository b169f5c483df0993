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
        if sample_rl:
            raise NotImplementedError("multinomial sampling rollout (dcnet_rl.py:322-326) is not built yet")
        if self.training:
            raise NotImplementedError("train-mode rollout (dropout active) is not built yet; call .eval()")
        _require_cuda(encoded_previous_captions, "previous captions")
        lib = _lib.load()
        dev = encoded_previous_captions.device
        prev = _i64c(encoded_previous_captions)
        plen = _i64c(previous_cap_length.reshape(-1))
        B, max_len = prev.shape[0], self.max_len
        dims = self._dims(B, prev.shape[1], max_len + 1)
        ws = self._workspace(dims)
        w = self._weights()
        seq = torch.empty(B, max_len, dtype=torch.long, device=dev)
        seq_logp = torch.empty(B, max_len, dtype=torch.float32, device=dev)
        check(lib.set_dcnet_greedy(C.byref(w), C.byref(dims), ptr(prev), ptr(plen), int(word_map['<start>']),
                                   int(word_map['<end>']), max_len, ptr(seq), ptr(seq_logp), ptr(ws), ws.numel(),
                                   stream_of(dev)), "set_dcnet_greedy")
        return seq, seq_logp


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
