"""Token cross-entropy of the training loops (editnet.py:571-577, dcnet.py:391-397) on the library's kernels
(csrc/loss.hip): pack_padded_sequence + CrossEntropyLoss as one forward launch over the (B, T, V) scores where they lie
and one backward launch that writes the score gradient in the row-padded (T, B, V4) layout the fc contractions of the
sequence nodes read in place (no packing copies, no log-softmax temporaries, no transposes)."""
from __future__ import annotations

import ctypes as C
import os

import torch

from . import _lib
from . import autograd_ops as A
from ._lib import check, stream_of

_FUSED = os.environ.get("SET_FUSED_XE_LOSS", "1") != "0"
MAX_T = 64


class _XELossSum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, scores, targets, live):
        lib = _lib.load()
        B, T, V = scores.shape
        dev = scores.device
        rowloss = torch.empty(T * B, dtype=torch.float32, device=dev)
        lse = torch.empty(T * B, dtype=torch.float32, device=dev)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        live_c = (C.c_int * T)(*live)
        check(lib.set_xe_loss_f32(scores.data_ptr(), scores.stride(0), scores.stride(1), targets.data_ptr(), targets.stride(0),
                                  targets.stride(1), live_c, B, T, V, rowloss.data_ptr(), lse.data_ptr(), loss.data_ptr(),
                                  stream_of(dev)), "set_xe_loss_f32")
        ctx.save_for_backward(scores, targets, lse)
        ctx.live = live
        return loss

    @staticmethod
    def backward(ctx, dloss):
        scores, targets, lse = ctx.saved_tensors
        lib = _lib.load()
        B, T, V = scores.shape
        dev = scores.device
        n4 = (V + 3) & ~3
        grad = torch.empty(T, B, n4, dtype=torch.float32, device=dev)
        A.register_zero_padded(grad)
        d = dloss.detach().to(torch.float32).contiguous()
        live_c = (C.c_int * T)(*ctx.live)
        check(lib.set_xe_loss_bwd_f32(scores.data_ptr(), scores.stride(0), scores.stride(1), targets.data_ptr(), targets.stride(0),
                                      targets.stride(1), live_c, B, T, V, lse.data_ptr(), d.data_ptr(), grad.data_ptr(), n4,
                                      stream_of(dev)), "set_xe_loss_bwd_f32")
        return grad.transpose(0, 1)[:, :, :V], None, None


def fusable(scores, targets, decode_lengths):
    if not (_FUSED and scores.is_cuda and scores.dtype == torch.float32 and scores.dim() == 3 and scores.stride(2) == 1):
        return False
    if targets.dtype != torch.long or targets.device != scores.device or targets.dim() != 2:
        return False
    lens = list(decode_lengths)
    B, T = scores.shape[0], scores.shape[1]
    return (len(lens) == B and T <= MAX_T and targets.shape[0] == B and targets.shape[1] >= T and 0 < max(lens) <= T and
            min(lens) >= 0 and all(lens[i] >= lens[i + 1] for i in range(B - 1)))


def xe_loss_sum(scores, targets, decode_lengths):
    """sum over the packed rows (b, t < decode_lengths[b]) of CrossEntropy(scores[b, t], targets[b, t]), 0-dim tensor"""
    lens = [int(l) for l in decode_lengths]
    T = scores.shape[1]
    live = [sum(1 for l in lens if l > t) for t in range(T)]
    return _XELossSum.apply(scores, targets, live)
