"""clip_grad_norm_ + Adam step of the reference's training loops (editnet.py:580-581, dcnet.py:399-400,
editnet_rl.py:684-686) on the library's two-launch kernel pair (csrc/optim.hip).

The training steps of `train.py` keep the reference's calling convention — the user owns a `torch.optim.Adam` — and hand
it to `clip_grad_norm_and_step`: when the optimizer is a plain Adam over dense fp32 device tensors the update runs
through `set_clip_adam_f32`, reading and writing the optimizer's own state tensors (`step`, `exp_avg`, `exp_avg_sq`, laid
out exactly as torch.optim.Adam creates them, so `state_dict()` / `load_state_dict()` / `adjust_learning_rate`
(utils.py of the reference) keep working and a checkpoint moves freely between the two).  Anything else (another
optimizer class, amsgrad, sparse gradients, ...) takes the two torch calls of the reference unchanged.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

from . import _lib
from ._lib import check, stream_of

_FUSED = os.environ.get("SET_FUSED_ADAM", "1") != "0"
_ws = {}
# Process-wide count of weight updates done behind autograd's back (raw-pointer kernels, collectives on `.data`).  Part of
# the token tables' cache signature next to (data_ptr, _version), so ANY such writer invalidates every derived table even
# when it cannot name the tensors it touched.
_weights_epoch = 0


def weights_epoch():
    return _weights_epoch


def bump_weights_epoch():
    global _weights_epoch
    _weights_epoch += 1


def _plain_adam(opt):
    if type(opt) is not torch.optim.Adam:
        return False
    for g in opt.param_groups:
        if g.get("amsgrad") or g.get("maximize") or g.get("capturable") or g.get("differentiable") or g.get("fused"):
            return False
        if g.get("decoupled_weight_decay"):
            return False
        if isinstance(g["lr"], torch.Tensor) or any(isinstance(b, torch.Tensor) for b in g["betas"]):
            return False
    return True


def _dense_f32(p):
    g = p.grad
    return (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and g is not None and not g.is_sparse and
            g.dtype == torch.float32 and g.is_contiguous() and g.device == p.device and p.data_ptr() % 16 == 0 and
            g.data_ptr() % 16 == 0)


def clip_grad_norm_and_step(parameters, optimizer, max_norm, scale_grads=False):
    """torch.nn.utils.clip_grad_norm_(parameters, max_norm); optimizer.step() — returns the total gradient norm (0-dim
    tensor).  scale_grads=False leaves `.grad` unclipped (the reference zeroes it before it is read again); True also
    writes the clipped gradients back, as clip_grad_norm_ does."""
    params = [p for p in parameters if p.grad is not None]
    todo = []
    if _FUSED and params and _plain_adam(optimizer):
        for g in optimizer.param_groups:
            todo += [(p, g) for p in g["params"] if p.grad is not None]
    same = len(todo) == len(params) and {id(p) for p, _ in todo} == {id(p) for p in params}
    if not same or not all(_dense_f32(p) for p in params) or len({p.device for p in params}) != 1:
        norm = torch.nn.utils.clip_grad_norm_(params, max_norm)
        optimizer.step()
        return norm
    lib = _lib.load()
    dev = params[0].device
    n = len(todo)
    P, G, M, V = ((C.c_void_p * n)() for _ in range(4))
    numel, step = (C.c_int64 * n)(), (C.c_int64 * n)()
    lr, b1, b2, eps, wd = ((C.c_double * n)() for _ in range(5))
    for i, (p, g) in enumerate(todo):
        st = optimizer.state[p]
        if len(st) == 0:               # as torch.optim.Adam._init_group creates it
            st["step"] = torch.tensor(0.0, dtype=torch.float32)
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        m, v = st["exp_avg"], st["exp_avg_sq"]
        if not (m.is_contiguous() and v.is_contiguous() and m.data_ptr() % 16 == 0 and v.data_ptr() % 16 == 0 and
                m.device == dev and v.device == dev and st["step"].device.type == "cpu"):
            raise _lib.SetError("Adam state of a parameter is not a dense fp32 device tensor with a host step counter")
        P[i], G[i], M[i], V[i] = p.data_ptr(), p.grad.data_ptr(), m.data_ptr(), v.data_ptr()
        # the step counters only advance once the launch has been accepted (below): an error half-way through the
        # marshalling must not skew the bias correction of the parameters already visited
        numel[i], step[i] = p.numel(), int(st["step"].item()) + 1
        lr[i], b1[i], b2[i], eps[i], wd[i] = g["lr"], g["betas"][0], g["betas"][1], g["eps"], g["weight_decay"]
    nbytes = lib.set_clip_adam_workspace_bytes(n, numel)
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
    ws = _ws.get(key)
    if ws is None or ws.numel() < nbytes:
        if len(_ws) >= 32:                    # keyed by (device, stream): bounded
            _ws.clear()
        ws = _ws[key] = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    norm = torch.empty((), dtype=torch.float32, device=dev)
    check(lib.set_clip_adam_f32(n, P, G, M, V, numel, step, lr, b1, b2, eps, wd, float(max_norm), int(bool(scale_grads)),
                                norm.data_ptr(), ws.data_ptr(), ws.numel(), stream_of(dev)), "set_clip_adam_f32")
    touched = []
    for p, _ in todo:
        st = optimizer.state[p]
        st["step"] += 1
        touched += [p, st["exp_avg"], st["exp_avg_sq"]]
    # The kernel wrote through raw pointers.  torch.optim.Adam's in-place ops bump tensor._version, and the modules'
    # derived caches (the inference-time token tables, editnet.py / dcnet.py `_token_table`) key on it: bump it here
    # too, so that a no-grad decode after this step never reads a table built from the old weights.
    if scale_grads:
        touched += [p.grad for p, _ in todo]
    torch.autograd.graph.increment_version(touched)
    bump_weights_epoch()
    return norm
