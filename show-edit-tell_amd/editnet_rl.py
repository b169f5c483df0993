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

from . import _lib
from ._lib import check, ptr, stream_of
from .editnet import (CaptionAttentionC, CaptionEncoderC, CopyLSTMCellC, EmbeddingC, LSTMCellC,  # noqa: F401
                      SelectC, VisualAttentionC, _f32c, _i64c, _require_cuda)
from .editnet import DecoderC as _DecoderXE


class DecoderC(_DecoderXE):
    """reference editnet_rl.py:455-549"""

    max_len = 18

    def forward(self, word_map, encoded_previous_captions, previous_cap_length, image_features, sample_max=True,
                sample_rl=False, image_mean=None):
        if sample_rl:
            raise NotImplementedError("multinomial sampling rollout (editnet_rl.py:524-528) is not built yet")
        if self.training:
            raise NotImplementedError("train-mode rollout (dropout active) is not built yet; call .eval()")
        _require_cuda(image_features, "image features")
        lib = _lib.load()
        dev = image_features.device
        X = _f32c(image_features)
        prev = _i64c(encoded_previous_captions)
        plen = _i64c(previous_cap_length.reshape(-1))
        mean = None if image_mean is None else _f32c(image_mean)
        B = X.shape[0]
        max_len = self.max_len
        dims = self._dims(B, prev.shape[1], X.shape[1], max_len + 1)
        ws = self._workspace(dims)
        w = self._weights()
        seq = torch.empty(B, max_len, dtype=torch.long, device=dev)
        seq_logp = torch.empty(B, max_len, dtype=torch.float32, device=dev)
        check(lib.set_editnet_greedy(C.byref(w), C.byref(dims), ptr(X), ptr(mean), ptr(prev), ptr(plen),
                                     int(word_map['<start>']), int(word_map['<end>']), max_len, ptr(seq),
                                     ptr(seq_logp), ptr(ws), ws.numel(), stream_of(dev)), "set_editnet_greedy")
        return seq, seq_logp
