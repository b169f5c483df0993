"""EditNet with adaptive bottom-up features (10-100 regions, zero padded) on MI355X.

Mirrors `/root/reference/adaptive_features/editnet_adaptive.py:423-562`: `VisualAttentionC`
masks padded regions (`:438-457`); `DecoderC.forward` takes a pre-computed `image_mean` and also
returns `gd_final_hidden` (caption encoder run on the ground-truth captions) and
`decoder_last_hidden` for the optional MSE loss (`:489-562`).
"""
from __future__ import annotations

from .editnet import (CaptionAttentionC, CaptionEncoderC, CopyLSTMCellC, EmbeddingC, LSTMCellC,  # noqa: F401
                      SelectC)
from .editnet import DecoderC as _DecoderXE
from .editnet import VisualAttentionC as _VisualAttentionC


class VisualAttentionC(_VisualAttentionC):
    """reference adaptive_features/editnet_adaptive.py:423-457"""
    adaptive = 1


class DecoderC(_DecoderXE):
    """reference adaptive_features/editnet_adaptive.py:459-562"""

    _visual_attention_cls = VisualAttentionC
    _adaptive = 1

    def forward(self, image_features, image_mean, encoded_captions, caption_lengths, encoded_previous_captions,
                previous_cap_length, use_ss=False, ss_prob=0.0):
        pred, caps_sorted, decode_lengths, sort_ind = super().forward(
            image_features, encoded_captions, caption_lengths, encoded_previous_captions, previous_cap_length,
            use_ss, ss_prob, image_mean=image_mean)
        B = caps_sorted.shape[0]
        sorted_lengths = caption_lengths.squeeze(1)[sort_ind].unsqueeze(1)
        if pred.requires_grad:           # grad-enabled path: both extra outputs stay in the autograd graph (MSE loss :594-596)
            decoder_last_hidden = self.__dict__.pop("_last_hidden")        # do not keep the graph alive on the module
            from . import rng
            # editnet_adaptive.py:516: the second encoder pass shares the forward call's seed at its own site (rng.py)
            _, _, gd_final_hidden, _ = self._encoder_autograd(caps_sorted, sorted_lengths, self.__dict__.pop("_fwd_seed", None),
                                                              rng.SITE_ENC2_EMBED)
            return pred, caps_sorted, decode_lengths, sort_ind, gd_final_hidden, decoder_last_hidden
        dims = self._dims(B, encoded_previous_captions.shape[1], image_features.shape[1], max(decode_lengths))
        # decoder_last_hidden[:bt] = h2 at every step (:560) == the h2 state left in the workspace
        decoder_last_hidden = self.ws_tensor(dims, "h2", (B, self.decoder_dim)).clone()
        _, _, gd_final_hidden, _ = self.caption_encoder(caps_sorted, sorted_lengths)          # :516
        return pred, caps_sorted, decode_lengths, sort_ind, gd_final_hidden, decoder_last_hidden
