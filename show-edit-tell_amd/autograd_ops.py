"""Autograd wrappers of the operator-level HIP entry points.

Forward of every operator runs the gfx950 kernels through the C ABI (`*_train_f32` variants keep
what the backward needs).  Backward: the pointwise / attention part of each operator is a
hand-written HIP kernel (`csrc/backward.hip`: `set_lstm_cell_bwd_f32`, `set_copy_gate_bwd_f32`,
`set_lstm_gates_bwd_f32`, `set_context_gate_bwd_f32`, `set_attention_bwd_f32`,
`set_select_bwd_f32`); the plain contractions between them (dX = dY W, dW = dY^T X, bias column
sums) run on the package's general-layout fp32 MFMA kernel (`csrc/gemm_gen.hip`, `set_gemm_f32`:
dX = dY.W reads W in place as a k-minor operand, dW += dY^T.X accumulates straight into `.grad`).
Inside `deferred_param_grads()` the parameter gradients of all timesteps are contracted in ONE GEMM per
parameter.  Gradient parity is tested against the REFERENCE's own
autograd (tests/golden `grad.*`, tests/test_hip_train.py).
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from . import rng as _rng
from ._lib import EditNetWeights, check, ptr, stream_of


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


def _ws(lib_fn, *dims, device):
    return torch.empty(max(16, lib_fn(*dims)), dtype=torch.uint8, device=device)


import contextlib
import os as _os

_FUSED_WGRAD = _os.environ.get("SET_FUSED_WGRAD", "1") != "0"
# every Linear-backward contraction runs on this package's fp32 MFMA kernel (set_gemm_f32): there is no vendor-BLAS
# route on the path
_GEMM_WS_BYTES = 64 << 20
_gemm_ws = {}


def _scratch(device):
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    ws = _gemm_ws.get(key)
    if ws is None:
        if len(_gemm_ws) >= 16:               # 64 MB each, keyed by (device, stream): bounded
            _gemm_ws.clear()
        # zeroed once: with SET_GEN_COMBINE=1 the library keeps its arrival counters in the tail of this buffer
        ws = _gemm_ws[key] = torch.zeros(_GEMM_WS_BYTES, dtype=torch.uint8, device=device)
    return ws


def _mat(t):
    """(tensor, leading dimension) of a 2-D fp32 operand the GEMM can read in place (unit inner stride,
    16-byte aligned rows); anything else is made contiguous first."""
    if t.stride(1) != 1 or t.stride(0) % 4 or t.data_ptr() % 16 or t.stride(0) < t.shape[1]:
        t = t.contiguous()
    return t, t.stride(0)


def gemm(a, a_kminor, b, b_kminor, M, N, K, out=None, accumulate=False):
    """out[m,n] (+)= sum_k a(m,k) b(n,k) through set_gemm_f32 (see include/set_hip.h for the layouts)."""
    lib = _lib.load()
    a, lda = _mat(a)
    b, ldb = _mat(b)
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=a.device)
        accumulate = False
    ws = _scratch(a.device)
    check(lib.set_gemm_f32(ptr(a), lda, int(a_kminor), ptr(b), ldb, int(b_kminor), ptr(out), out.stride(0), M, N, K,
                           int(accumulate), ptr(ws), ws.numel(), stream_of(a.device)), "set_gemm_f32")
    return out


def gemm_group(items, a_kminor, b_kminor):
    """items: [(a, b, M, N, K, out or None, accumulate)], all of one operand layout -> list of outputs, ONE launch
    (set_gemm_group_f32; at most 6 problems, same row-tile class)."""
    lib = _lib.load()
    n = len(items)
    descs = (_lib.GemmDesc * n)()
    keep, outs = [], []
    dev = items[0][0].device
    for i, (a, b, M, N, K, out, acc) in enumerate(items):
        a, lda = _mat(a)
        b, ldb = _mat(b)
        if out is None:
            out, acc = torch.empty(M, N, dtype=torch.float32, device=dev), False
        keep += [a, b]
        outs.append(out)
        descs[i] = _lib.GemmDesc(ptr(a), lda, ptr(b), ldb, ptr(out), out.stride(0), M, N, K, int(acc))
    ws = _scratch(dev)
    check(lib.set_gemm_group_f32(descs, n, int(a_kminor), int(b_kminor), ptr(ws), ws.numel(), stream_of(dev)),
          "set_gemm_group_f32")
    return outs


def _dgrad_group(pairs):
    """[(dy, w, out or None)] -> [dX (+)= dy . w], grouped into one launch when the kernel can take them all"""
    ok = all(_native_ok(dy, w) and not (dy.shape[1] & 3) and not (w.shape[1] & 3) and w.shape[1] >= 4 and
             (out is None or out.is_contiguous()) for dy, w, out in pairs)
    same_class = len({dy.shape[0] <= 128 for dy, _, _ in pairs}) == 1
    if not ok or not same_class or len(pairs) > 6:
        return [_dgrad(dy, w, out) for dy, w, out in pairs]
    return gemm_group([(dy, w, dy.shape[0], w.shape[1], dy.shape[1], out, out is not None) for dy, w, out in pairs],
                      False, True)


def _linear_nograd(x, w, b):
    """y = x w^T + b through set_linear_f32 (no autograd node): recomputation of small forward products inside a
    backward (the decoder-side attention projection att2 = decoder_att(h1), editnet.py:371,443)."""
    lib = _lib.load()
    x = _c(x)
    M, K, N = x.shape[0], x.shape[1], w.shape[0]
    y = torch.empty(M, N, dtype=torch.float32, device=x.device)
    ws = _ws(lib.set_linear_workspace_bytes, M, N, K, device=x.device)
    check(lib.set_linear_f32(ptr(x), K, ptr(w), K, ptr(b), ptr(y), N, M, N, K, _lib.ACT_NONE, ptr(ws), ws.numel(),
                             stream_of(x.device)), "set_linear_f32")
    return y


def _native_ok(*ts):
    return all(t.is_cuda and t.dtype == torch.float32 for t in ts)


# Buffers whose rows are ALREADY padded with zeros to a multiple of 4 floats (the score gradient written by
# set_xe_loss_bwd_f32): storage data pointer -> weak reference of the owning tensor.  A 2-D view (M, N) of such a buffer
# with row stride N4 is read by the contractions in place, as its (M, N4) form.
import weakref as _weakref

_zero_padded = {}


def register_zero_padded(buf):
    for k in [k for k, r in _zero_padded.items() if r() is None]:
        del _zero_padded[k]
    _zero_padded[buf.untyped_storage().data_ptr()] = _weakref.ref(buf)


def _padded_view(t):
    """the (M, N4) form of a 2-D view (M, N) of a registered zero-padded buffer, or None"""
    if t.dim() != 2 or not (t.shape[1] & 3):
        return None
    ref = _zero_padded.get(t.untyped_storage().data_ptr())
    if ref is None or ref() is None:
        return None
    n4 = (t.shape[1] + 3) & ~3
    if t.stride(1) != 1 or t.stride(0) != n4 or t.data_ptr() % 16 or t.dtype != torch.float32:
        return None
    return t.as_strided((t.shape[0], n4), (n4, 1), t.storage_offset())


def zero_padded_rows(T, B, V, device):
    """a (T, B, V) fp32 view whose rows are padded with zeros to a multiple of 4 floats (registered: the contractions
    read its 2-D forms in place); the V columns are NOT initialised"""
    n4 = (V + 3) & ~3
    buf = torch.empty(T, B, n4, dtype=torch.float32, device=device)
    if n4 != V:
        buf[:, :, V:].zero_()
    register_zero_padded(buf)
    return buf[:, :, :V]


def score_grad_rows(dpred, bts, uniform):
    """the (B, T, V) gradient of a node's scores -> its (T*B, V) rows in (t, b) order with the rows of finished sequences
    zeroed.  A gradient that already lives in a zero-padded (T, B, V4) buffer (set_xe_loss_bwd_f32) is used where it lies;
    anything else is brought into (T, B, V) order with one copy."""
    B, T, V = dpred.shape
    dp = dpred.transpose(0, 1)
    n4 = (V + 3) & ~3
    ref = _zero_padded.get(dp.untyped_storage().data_ptr())
    if (V & 3) and ref is not None and ref() is not None and dp.stride() == (B * n4, n4, 1) and dp.data_ptr() % 16 == 0:
        rows = dp.as_strided((T * B, V), (n4, 1), dp.storage_offset())
    else:
        dp = dp if (dp.is_contiguous() and uniform) else dp.contiguous()
        rows = dp.view(T * B, V)
    if not uniform:
        for t in range(T):
            if bts[t] < B:
                rows[t * B + bts[t]:(t + 1) * B].zero_()
    return rows


def _pad_cols4(t):
    """(M, N) -> (M, N4) zero-padded form with N4 = N rounded up to 4 (only for feature counts that are not a multiple
    of 4, e.g. a vocabulary of 9490 words): the kernel reads 16-byte chunks along its unit-stride dimension.  A copy,
    unless `t` already lives in a zero-padded buffer."""
    pad = (-t.shape[1]) % 4
    if pad == 0:
        return t
    v = _padded_view(t)
    return v if v is not None else torch.nn.functional.pad(t, (0, pad))


def _dgrad(dy, w, out=None):
    """dX (+)= dy . w   (dy: (M, N), w: (N, K) possibly a column slice of a wider weight) -> (M, K);
    accumulates into `out` (a tensor this backward owns) when given.  Always on the package's fp32 MFMA kernel."""
    M, N = dy.shape
    K = w.shape[1]
    if (K & 3) or K < 4 or (out is not None and not out.is_contiguous()):
        raise _lib.SetError("dX = dY.W needs an input feature count that is a multiple of 4 (got %d)" % K)
    if N & 3:             # contraction over a ragged feature count: zero-pad dy's columns; W rows >= N are never read
        dy = _pad_cols4(dy)
    return gemm(dy, False, w, True, M, K, N, out=out, accumulate=out is not None)


def _wgrad_mm(dy, x, out=None):
    """dW (+)= dy^T . x   (dy: (M, N), x: (M, K)) -> (N, K); accumulates into `out` when given"""
    M, N = dy.shape
    K = x.shape[1]
    if (K & 3) or K < 4:
        raise _lib.SetError("dW = dY^T.X needs an input feature count that is a multiple of 4 (got %d)" % K)
    if (N & 3) and N >= 4 and (out is None or out.is_contiguous()):
        dyp = _padded_view(dy)
        if dyp is not None:    # rows readable up to N4: the kernel reads them in place and never stores rows >= N
            return gemm(dyp, True, x, True, N, K, M, out=out, accumulate=out is not None)
    if (N & 3) or N < 4 or (out is not None and not out.is_contiguous()):
        # ragged output-feature count (e.g. V = 9490 rows of fc.weight): contract into a padded buffer, keep N rows
        dyp = _pad_cols4(dy)
        full = gemm(dyp, True, x, True, dyp.shape[1], K, M)
        if out is None:
            return full[:N].contiguous()
        return out.add_(full[:N])
    return gemm(dy, True, x, True, N, K, M, out=out, accumulate=out is not None)


# ---- deferred (time-batched) parameter gradients -----------------------------------------------------
# Inside `deferred_param_grads()` the per-timestep weight/bias gradient contributions are only recorded;
# on exit every parameter gets ONE contraction over all timesteps (dW += cat(dy)^T . cat(x): contraction
# length 19 x B instead of 19 GEMMs of length B that each re-read and re-write the whole .grad).
import threading as _threading

_tls = _threading.local()          # the pending table is per thread (autograd runs a .backward() on the calling thread's
                                   # device thread; two models training on two threads must not share it)


def _pending():
    return getattr(_tls, "deferred", None)


@contextlib.contextmanager
def deferred_param_grads(on_ready=None):
    """`on_ready(param)` is called as soon as a parameter's gradient is final (used by the data-parallel step to
    start the all-reduce of a bucket while the remaining weight-gradient contractions still run).
    Only `loss.backward()` inside the context is supported: parameter gradients are written straight into `.grad`
    (`torch.autograd.grad(loss, params)` and parameter hooks do not see them)."""
    if _pending() is not None or not _FUSED_WGRAD:
        yield
        return
    _tls.deferred = {}
    _active[_threading.get_ident()] = _tls.deferred
    _active_cb[id(_tls.deferred)] = on_ready
    try:
        yield
        pending, cat_cache = _tls.deferred, {}

        def cat(ts):
            if len(ts) == 1:
                return ts[0]
            key = tuple(id(t) for t in ts)
            if key not in cat_cache:
                cat_cache[key] = torch.cat(ts, 0)
            return cat_cache[key]

        # bias gradients first, all of them as one grouped column-sum launch; two biases fed by the SAME output gradients
        # (nn.LSTMCell's bias_ih / bias_hh, the x2h / h2h pair of the copy cell, gate_cnew / gate_cmem) share one problem
        problems = {}
        for param, (dys, xs) in pending.values():
            if xs is None:
                problems.setdefault(tuple(id(t) for t in dys), (dys, []))[1].append(param)
        if problems:
            _colsum_group([(cat(dys), plist) for dys, plist in problems.values()])
            if on_ready is not None:
                for _, plist in problems.values():
                    for q in plist:
                        on_ready(q)
        for param, (dys, xs) in pending.values():
            if xs is None:
                continue
            dy, x = cat(dys), cat(xs)
            if param.grad is None:
                param.grad = _wgrad_mm(dy, x)
            else:
                _wgrad_mm(dy, x, out=param.grad)
            if on_ready is not None:
                on_ready(param)
    finally:
        _active.pop(_threading.get_ident(), None)
        _active_cb.pop(id(_tls.deferred), None)
        _tls.deferred = None


_active_cb = {}     # id(table) -> the context's on_ready callback (eager gradients are announced through it at once)
_active = {}        # thread id of the thread that opened a deferred context -> its table (autograd's backward may run on
                    # a worker thread of the engine: it looks the table up through the single open context)


def _deferred_table():
    t = _pending()
    if t is not None:
        return t
    # backward kernels are enqueued from the autograd engine's device thread, not from the thread that called
    # loss.backward(): fall back to the one context that is open process-wide (None if there is none or several)
    if len(_active) == 1:
        return next(iter(_active.values()))
    return None


def _is_leaf_param(param):
    return _FUSED_WGRAD and isinstance(param, torch.nn.Parameter) and param.is_leaf


def _announce(param, eager=True):
    """finished gradient inside a deferred context: tell the context's on_ready (the data-parallel reducer) right away"""
    tab = _deferred_table()
    cb = _active_cb.get(id(tab)) if tab is not None else None
    if cb is not None:
        try:
            cb(param, eager=eager)
        except TypeError:
            cb(param)


def _wgrad_blocks(param, blocks):
    """Weight gradient assembled from column blocks: dW[:, c0:c0+Kb] (+)= dy^T x for every (dy, x, c0) — the blocks cover
    all columns of `param` and may contract over different row counts (a per-timestep input block over all T*B rows, a
    loop-invariant input block over B rows against sum_t dy).  This is the parameter's ONLY contribution of the backward:
    written straight into param.grad (see `_wgrad`) and announced; non-leaf tensors get the gradient returned."""
    leaf = _is_leaf_param(param)
    acc = leaf and param.grad is not None
    out = param.grad if acc else torch.empty_like(param)
    N = param.shape[0]
    for dy, x, c0 in blocks:
        Kb = x.shape[1]
        if (Kb & 3) or (N & 3) or (c0 & 3):
            raise _lib.SetError("column-block weight gradient needs multiples of 4 (got N=%d, K=%d at column %d)" % (N, Kb, c0))
        gemm(dy, True, x, True, N, Kb, dy.shape[0], out=out[:, c0:c0 + Kb], accumulate=acc)
    if not leaf:
        return out
    param.grad = out
    _announce(param, eager=False)
    return None


def _wgrad(param, dy, x, eager=False):
    """Weight gradient dW = dy^T x, accumulated IN PLACE into param.grad (fused weight-gradient
    accumulation): returning dW to autograd would make it allocate a weight-sized temporary and run a
    separate weight-sized add per timestep (19 x 355 MB per training step).  Returns None so autograd
    skips its own accumulation.  Leaf parameters only; anything else gets the gradient returned.
    eager: contract now even inside `deferred_param_grads()` (the caller guarantees this is the parameter's ONLY
    contribution of the backward, e.g. fc in the sequence nodes) and announce it, so that its all-reduce can run
    underneath the rest of the backward."""
    if not _is_leaf_param(param):
        return _wgrad_mm(dy, x)
    if eager and _deferred_table() is not None and id(param) not in _deferred_table():
        if param.grad is None:
            param.grad = _wgrad_mm(dy, x)
        else:
            _wgrad_mm(dy, x, out=param.grad)
        _announce(param)
        return None
    if _deferred_table() is not None:
        ent = _deferred_table().setdefault(id(param), (param, ([], [])))
        ent[1][0].append(dy)
        ent[1][1].append(x)
    elif param.grad is None:
        param.grad = _wgrad_mm(dy, x)
    else:
        _wgrad_mm(dy, x, out=param.grad)
    return None


def _colsum(dy, out=None):
    """sum over the rows of a 2-D fp32 matrix on the library's two-pass kernel (set_colsum_f32); out (+)= when given"""
    if dy.is_cuda and dy.dim() == 2 and dy.shape[1] % 4 and dy.shape[0] >= 64 and _padded_view(dy) is not None:
        g = _colsum(_padded_view(dy))[:dy.shape[1]]          # zero-padded rows: sum the padded form, drop the padding
        return g.clone() if out is None else out.add_(g.reshape(out.shape))
    if not (dy.is_cuda and dy.dtype == torch.float32 and dy.dim() == 2 and dy.shape[1] % 4 == 0 and dy.shape[0] >= 64):
        g = dy.sum(0)
        return g if out is None else out.add_(g.reshape(out.shape))
    lib = _lib.load()
    dy, ld = _mat(dy)
    rows, cols = dy.shape
    acc = out is not None
    if out is None:
        out = torch.empty(cols, dtype=torch.float32, device=dy.device)
    if not out.is_contiguous() or out.data_ptr() % 16:
        return out.add_(dy.sum(0).reshape(out.shape)) if acc else dy.sum(0)
    ws = _scratch(dy.device)
    check(lib.set_colsum_f32(ptr(dy), ld, rows, cols, ptr(out), int(acc), ptr(ws), ws.numel(), stream_of(dy.device)),
          "set_colsum_f32")
    return out


_colsum_ws = {}


def _colsum_group(problems):
    """problems = [(dy, [param, ...])]: every listed parameter's .grad (+)= the column sums of dy, all problems in ONE pair of
    launches (set_colsum_group_f32) — a training step has ~16 bias gradients, each of which was two launches plus an add.
    Problems the grouped kernel cannot take (non-fp32, ragged columns, unaligned .grad views) go through _colsum."""
    lib = _lib.load()
    descs, keep, rest = [], [], []
    for dy, plist in problems:
        ok = dy.is_cuda and dy.dtype == torch.float32 and dy.dim() == 2 and dy.shape[1] % 4 == 0 and dy.shape[0] >= 64
        if ok:
            for q in plist:
                g = q.grad
                if q.numel() != dy.shape[1] or q.dtype != torch.float32:
                    ok = False                  # (not a bias of this gradient: let the plain path raise / broadcast as torch would)
                elif g is not None and (not g.is_contiguous() or g.data_ptr() % 16 or g.dtype != torch.float32 or g.numel() != dy.shape[1]):
                    ok = False
        if not ok or len(descs) + (len(plist) + 1) // 2 > 24:
            rest.append((dy, plist))
            continue
        dy, ld = _mat(dy)
        keep.append(dy)
        outs = []
        for q in plist:
            acc = q.grad is not None
            if not acc:
                q.grad = torch.empty(q.shape, dtype=torch.float32, device=dy.device)
            outs.append((q.grad, acc))
        for i in range(0, len(outs), 2):
            o2 = outs[i + 1] if i + 1 < len(outs) else (None, False)
            descs.append((dy.data_ptr(), ld, dy.shape[0], dy.shape[1], outs[i][0].data_ptr(), int(outs[i][1]),
                          o2[0].data_ptr() if o2[0] is not None else None, int(o2[1])))
    if descs:
        arr = (_lib.ColsumDesc * len(descs))(*[_lib.ColsumDesc(*d) for d in descs])
        dev = keep[0].device
        need = lib.set_colsum_group_workspace_bytes(arr, len(descs))
        # keyed by (device, stream) like _scratch: the two launches of the grouped column sum share the partials buffer, and two
        # streams flushing deferred bias gradients at the same time must not share it
        key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
        ws = _colsum_ws.get(key)
        if ws is None or ws.numel() * 4 < need:
            if len(_colsum_ws) >= 16:
                _colsum_ws.clear()
            ws = _colsum_ws[key] = torch.empty((need + 3) // 4 + 1024, dtype=torch.float32, device=dev)
        check(lib.set_colsum_group_f32(arr, len(descs), ptr(ws), ws.numel() * 4, stream_of(dev)), "set_colsum_group_f32")
    for dy, plist in rest:
        g = _colsum(dy)
        for q in plist:
            if q.grad is None:
                q.grad = g.reshape(q.shape).clone()
            else:
                q.grad.add_(g.reshape(q.shape))


def _bgrad(param, dy, eager=False):
    if not _is_leaf_param(param):
        return _colsum(dy).reshape(param.shape)
    if eager and _deferred_table() is not None and id(param) not in _deferred_table():
        if param.grad is None:
            param.grad = _colsum(dy).reshape(param.shape)
        else:
            _colsum(dy, out=param.grad)
        _announce(param)
        return None
    if _deferred_table() is not None:
        ent = _deferred_table().setdefault(id(param), (param, ([], None)))
        ent[1][0].append(dy)
        return None
    if param.grad is None:
        param.grad = _colsum(dy).reshape(param.shape)
    else:
        _colsum(dy, out=param.grad)
    return None


# ------------------------------------------------------------------------------------------------
# nn.Linear (+ activation)
# ------------------------------------------------------------------------------------------------
class _Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, act):
        lib = _lib.load()
        x2 = _c(x.reshape(-1, x.shape[-1]))
        M, K, N = x2.shape[0], x2.shape[1], w.shape[0]
        y = torch.empty(M, N, dtype=torch.float32, device=x.device)
        ws = _ws(lib.set_linear_workspace_bytes, M, N, K, device=x.device)
        check(lib.set_linear_f32(ptr(x2), K, ptr(w), K, ptr(b), ptr(y), N, M, N, K, act, ptr(ws), ws.numel(),
                                 stream_of(x.device)), "set_linear_f32")
        ctx.act, ctx.xshape = act, x.shape
        ctx.params = (w, b)
        ctx.save_for_backward(x2, w, y if act != _lib.ACT_NONE else None)
        return y.reshape(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        x2, w, y = ctx.saved_tensors
        dy = _c(dy).reshape(-1, dy.shape[-1])
        if ctx.act == _lib.ACT_RELU:
            dy = _relu_bwd(dy, y.reshape(dy.shape))
        elif ctx.act == _lib.ACT_TANH:
            dy = dy * (1 - y * y)
        elif ctx.act == _lib.ACT_SIGMOID:
            dy = dy * y * (1 - y)
        pw, pb = ctx.params
        dx = _dgrad(dy, w).reshape(ctx.xshape) if ctx.needs_input_grad[0] else None
        dw = _wgrad(pw, dy, x2) if ctx.needs_input_grad[1] else None
        db = _bgrad(pb, dy) if ctx.needs_input_grad[2] else None
        return dx, dw, db, None


def _relu_bwd(dy, y):
    """dy * (y > 0) for a ReLU output y (>= 0, so y > 0 <=> y != 0): one launch of the library's dropout-backward kernel with
    scale 1 instead of a compare and a multiply; anything the kernel does not take goes through torch"""
    if (dy.is_cuda and dy.dtype == torch.float32 and y.dtype == torch.float32 and dy.dim() == 2 and dy.shape == y.shape and
            dy.shape[1] % 4 == 0 and dy.is_contiguous() and y.is_contiguous() and dy.data_ptr() % 16 == 0 and y.data_ptr() % 16 == 0):
        out = torch.empty_like(dy)
        rows, cols = dy.shape
        check(_lib.load().set_dropout_bwd_f32(ptr(dy), cols, ptr(y), cols, ptr(out), cols, rows, cols, 1.0, 0, stream_of(dy.device)),
              "set_dropout_bwd_f32")
        return out
    return dy * (y > 0)


def linear(x, weight, bias, act=_lib.ACT_NONE):
    return _Linear.apply(x, weight, bias, act)


# ------------------------------------------------------------------------------------------------
# EmbeddingC.forward without the dropout (editnet.py:301-302)
# ------------------------------------------------------------------------------------------------
class _EmbedRelu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ids, table):
        lib = _lib.load()
        ids_c = _c(ids)
        n, D = ids_c.numel(), table.shape[1]
        out = torch.empty(tuple(ids_c.shape) + (D,), dtype=torch.float32, device=ids.device)
        check(lib.set_embed_relu_f32(ptr(table), ptr(ids_c), 1, ptr(out), D, n, D, table.shape[0],
                                     stream_of(ids.device)), "set_embed_relu_f32")
        ctx.save_for_backward(ids_c, out)
        ctx.V = table.shape[0]
        return out

    @staticmethod
    def backward(ctx, dout):
        ids, out = ctx.saved_tensors
        D = out.shape[-1]
        g = _relu_bwd(_c(dout).reshape(-1, D), out.reshape(-1, D))
        dt = torch.zeros(ctx.V, D, dtype=torch.float32, device=out.device)
        dt.index_add_(0, ids.reshape(-1), g)
        return None, dt


def embed_relu(ids, table):
    return _EmbedRelu.apply(ids, table)


# ------------------------------------------------------------------------------------------------
# nn.LSTMCell / LSTMCellC (editnet.py:226-244)
# ------------------------------------------------------------------------------------------------
class _LstmCell(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, h, c, w_ih, w_hh, b_ih, b_hh):
        lib = _lib.load()
        x, h, c = _c(x), _c(h), _c(c)
        M, K, D = x.shape[0], x.shape[1], h.shape[1]
        h_new, c_new = torch.empty_like(h), torch.empty_like(c)
        gates = torch.empty(M, 4 * D, dtype=torch.float32, device=x.device)
        ws = _ws(lib.set_lstm_cell_workspace_bytes, M, D, K, device=x.device)
        check(lib.set_lstm_cell_train_f32(ptr(x), K, K, ptr(h), ptr(c), ptr(w_ih), K, ptr(w_hh), ptr(b_ih), ptr(b_hh),
                                          ptr(h_new), ptr(c_new), ptr(gates), M, D, ptr(ws), ws.numel(),
                                          stream_of(x.device)), "set_lstm_cell_train_f32")
        ctx.params = (w_ih, w_hh, b_ih, b_hh)
        ctx.set_materialize_grads(False)          # a missing dh / dc reaches the kernel as NULL, not as a zero tensor
        ctx.save_for_backward(x, h, c, w_ih, w_hh, gates, c_new)
        return h_new, c_new

    @staticmethod
    def backward(ctx, dh, dc):
        x, h, c, w_ih, w_hh, gates, c_new = ctx.saved_tensors
        lib = _lib.load()
        M, D = h.shape
        dg = torch.empty_like(gates)
        dcp = torch.empty_like(c)
        check(lib.set_lstm_cell_bwd_f32(ptr(None if dh is None else _c(dh)), ptr(None if dc is None else _c(dc)),
                                        ptr(gates), ptr(c), ptr(c_new), ptr(dg), ptr(dcp), M, D,
                                        stream_of(h.device)), "set_lstm_cell_bwd_f32")
        p_ih, p_hh, pb_ih, pb_hh = ctx.params
        dx, dhp = _dgrad_group([(dg, w_ih, None), (dg, w_hh, None)])
        return (dx, dhp, dcp, _wgrad(p_ih, dg, x), _wgrad(p_hh, dg, h), _bgrad(pb_ih, dg), _bgrad(pb_hh, dg))


def lstm_cell(x, h, c, w_ih, w_hh, b_ih, b_hh):
    return _LstmCell.apply(x, h, c, w_ih, w_hh, b_ih, b_hh)


# ------------------------------------------------------------------------------------------------
# The caption encoders' recurrence as ONE autograd node (CaptionEncoderC editnet.py:319-348; each direction of the
# packed BiLSTM of dcnet.py:220-243).  The reference runs a length-sorted, prefix-shrinking batch; per row that is
# "advance while t < len, then hold the state, outputs beyond the length stay zero", which the cell kernels implement
# directly (set_encoder_cell_train_f32 / _bwd_f32).  Compared with a per-step chain of lstm_cell + masked torch
# arithmetic this removes ~25 tiny kernels per step and direction in forward + backward, and the input projection
# x W_x^T (and its two gradient contractions) runs ONCE over all (b, t) instead of once per step.
# ------------------------------------------------------------------------------------------------
class _EncoderLSTM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, emb, lens, w_ih, b_ih, w_hh, b_hh, reverse, want_mem):
        lib = _lib.load()
        emb = _c(emb)
        lens = _c(lens.reshape(-1).long())
        B, T, E = emb.shape
        D = w_hh.shape[1]
        dev = emb.device
        st = stream_of(dev)
        xg = _linear_nograd(emb.reshape(B * T, E), w_ih, b_ih)                 # (B*T, 4D): hoisted x W_x^T + b_x
        H = torch.empty(B, T, D, dtype=torch.float32, device=dev)
        Mem = torch.empty(B, T, D, dtype=torch.float32, device=dev) if want_mem else None
        Hprev = torch.empty(B, T, D, dtype=torch.float32, device=dev)
        hs = torch.empty(T + 1, B, D, dtype=torch.float32, device=dev)
        cs = torch.empty(T + 1, B, D, dtype=torch.float32, device=dev)
        hs[0].zero_()
        cs[0].zero_()
        gates = torch.empty(T, B, 4 * D, dtype=torch.float32, device=dev)
        ws = _ws(lib.set_encoder_cell_workspace_bytes, B, D, device=dev)
        order = list(range(T - 1, -1, -1)) if reverse else list(range(T))
        for k, t in enumerate(order):
            check(lib.set_encoder_cell_train_f32(ptr(xg), T * 4 * D, 4 * D, ptr(hs[k]), ptr(cs[k]), ptr(w_hh), ptr(b_hh),
                                                 ptr(lens), t, ptr(hs[k + 1]), ptr(cs[k + 1]), ptr(H), ptr(Mem), ptr(Hprev),
                                                 T * D, D, 0, ptr(gates[k]), B, D, ptr(ws), ws.numel(), st),
                  "set_encoder_cell_train_f32")
        ctx.order = order
        ctx.params = (w_ih, b_ih, w_hh, b_hh)
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(emb, lens, w_ih, w_hh, cs, gates, Hprev)
        h_last = hs[T]
        if want_mem:
            return H, Mem, h_last
        return H, h_last

    @staticmethod
    def backward(ctx, dH, *rest):
        emb, lens, w_ih, w_hh, cs, gates, Hprev = ctx.saved_tensors
        if len(rest) == 2:
            dM, dh_last = rest
        else:
            dM, dh_last = None, rest[0]
        lib = _lib.load()
        B, T, E = emb.shape
        D = w_hh.shape[1]
        dev = emb.device
        st = stream_of(dev)
        dH = None if dH is None else _c(dH)
        dM = None if dM is None else _c(dM)
        dG = torch.empty(B, T, 4 * D, dtype=torch.float32, device=dev)       # gradient of xg = pre-activation gate gradients
        dh = None if dh_last is None else _c(dh_last)
        dc = None
        for k in range(T - 1, -1, -1):
            t = ctx.order[k]
            dc_prev = torch.empty(B, D, dtype=torch.float32, device=dev)
            dh_pass = torch.empty(B, D, dtype=torch.float32, device=dev)
            dg_t = dG[:, t]                                                     # (B, 4D) rows with stride T*4D
            check(lib.set_encoder_cell_bwd_f32(ptr(dh), ptr(dc), ptr(dH), ptr(dM), T * D, D, 0, ptr(lens), t, ptr(gates[k]),
                                               ptr(cs[k]), ptr(cs[k + 1]), ptr(dg_t), T * 4 * D, ptr(dc_prev), ptr(dh_pass),
                                               B, D, st), "set_encoder_cell_bwd_f32")
            dh = gemm(dg_t, False, w_hh, True, B, D, 4 * D, out=dh_pass, accumulate=True)    # dh_{k-1} = dh_pass + dgates W_hh
            dc = dc_prev
        p_ih, pb_ih, p_hh, pb_hh = ctx.params
        dG2 = dG.reshape(B * T, 4 * D)
        d_emb = _dgrad(dG2, w_ih).reshape(B, T, E) if ctx.needs_input_grad[0] else None
        return (d_emb, None, _wgrad(p_ih, dG2, emb.reshape(B * T, E)), _bgrad(pb_ih, dG2),
                _wgrad(p_hh, dG2, Hprev.reshape(B * T, D)), _bgrad(pb_hh, dG2), None, None)


def encoder_lstm(emb, lens, w_ih, b_ih, w_hh, b_hh, reverse=False, want_mem=True):
    """emb (B,T,E) [dropout already applied], lens (B,) -> (H (B,T,D), Mem (B,T,D), h_last (B,D)) — or (H, h_last) when
    want_mem is False; rows are advanced while t < lens[b] (reverse: positions T-1 .. 0), padded outputs are zero."""
    return _EncoderLSTM.apply(emb, lens, w_ih, b_ih, w_hh, b_hh, reverse, want_mem)


# ------------------------------------------------------------------------------------------------
# nn.Dropout(p) of the train-mode path on the library's Philox stream (rng.py: one seed per forward call, one offset per
# (site, timestep)): the same masks on the whole-sequence nodes and on the per-operator route, reproducible in numpy
# (oracle/philox_np.dropout_mask) — which is what pins train mode to the reference's autograd
# ------------------------------------------------------------------------------------------------
class _PhiloxDropout(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p, seed, offset):
        lib = _lib.load()
        x = _c(x)
        cols = x.shape[-1]
        rows = x.numel() // cols
        y = torch.empty_like(x)
        check(lib.set_dropout_f32(ptr(x), cols, ptr(y), cols, rows, cols, float(p), int(seed), int(offset),
                                  stream_of(x.device)), "set_dropout_f32")
        ctx.args = (rows, cols, float(p), int(seed), int(offset))
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        rows, cols, p, seed, offset = ctx.args
        dy = _c(dy)
        dx = torch.empty_like(dy)
        check(lib.set_dropout_bwd_philox_f32(ptr(dy), cols, ptr(dx), cols, rows, cols, p, seed, offset, 0,
                                             stream_of(dy.device)), "set_dropout_bwd_philox_f32")
        return dx, None, None, None


def philox_dropout(x, p, seed, offset, training=True):
    """nn.Dropout(p)(x) with the mask of (seed, offset): element (r, c) of x viewed as (rows, last dim) is kept iff the
    24-bit uniform of counter (r, c // 4, offset), word c % 4 is >= p.  Identity when not training or p == 0."""
    if not training or p <= 0.0 or x.numel() == 0:
        return x
    if x.shape[-1] & 3:
        raise _lib.SetError("Philox dropout needs a feature count that is a multiple of 4 (got %d)" % x.shape[-1])
    return _PhiloxDropout.apply(x, p, seed, offset)


# ------------------------------------------------------------------------------------------------
# additive attention backward shared by the caption (tanh) and visual (relu) attentions
# ------------------------------------------------------------------------------------------------
def _attention_bwd(dctx, dalpha_ext, alpha, values, att1, att2, w_full, use_tanh, want_dvalues):
    lib = _lib.load()
    M, L, Dv = values.shape
    A = att1.shape[2]
    dev = values.device
    datt1 = torch.empty(M, L, A, dtype=torch.float32, device=dev)
    datt2 = torch.empty(M, A, dtype=torch.float32, device=dev)
    dwf = torch.empty(M, A, dtype=torch.float32, device=dev)
    de = torch.empty(M, L, dtype=torch.float32, device=dev)
    dval = torch.empty(M, L, Dv, dtype=torch.float32, device=dev) if want_dvalues else None
    check(lib.set_attention_bwd_f32(ptr(_c(dctx)), ptr(None if dalpha_ext is None else _c(dalpha_ext)), ptr(alpha),
                                    ptr(values), ptr(att1), ptr(att2), ptr(_c(w_full.reshape(-1))), ptr(datt1),
                                    ptr(datt2), ptr(dwf), ptr(dval), ptr(de), M, L, Dv, A, 1 if use_tanh else 0,
                                    stream_of(dev)), "set_attention_bwd_f32")
    return datt1, datt2, dwf.sum(0, keepdim=True), dval, de.sum().reshape(1)


# ------------------------------------------------------------------------------------------------
# CaptionAttentionC.forward (editnet.py:364-381) with the loop-invariant att1_c = cap_features_att(H)
# passed in (its gradient accumulates over the timesteps and flows once through `linear`)
# ------------------------------------------------------------------------------------------------
class _CaptionAttention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, H, att1_c, h1, word, mask, dec_w, dec_b, full_w, full_b, gate_w, gate_b, sc_w, sc_b, tc_w, tc_b):
        lib = _lib.load()
        H, att1_c, h1, word, mask = _c(H), _c(att1_c), _c(h1), _c(word), _c(mask)
        M, T, D = H.shape
        A = dec_w.shape[0]
        dev = H.device
        w = EditNetWeights()
        w.ca_dec_w, w.ca_dec_b = dec_w.data_ptr(), dec_b.data_ptr()
        w.ca_full_w, w.ca_full_b, w.ca_gate_w, w.ca_gate_b = full_w.data_ptr(), full_b.data_ptr(), gate_w.data_ptr(), gate_b.data_ptr()
        w.ca_sc_w, w.ca_sc_b, w.ca_tc_w, w.ca_tc_b = sc_w.data_ptr(), sc_b.data_ptr(), tc_w.data_ptr(), tc_b.data_ptr()
        gated, cx, zt, s, t = (torch.empty(M, D, dtype=torch.float32, device=dev) for _ in range(5))
        alpha = torch.empty(M, T, dtype=torch.float32, device=dev)
        ws = _ws(lib.set_caption_attention_workspace_bytes, M, T, D, A, device=dev)
        check(lib.set_caption_attention_train_f32(C.byref(w), ptr(H), ptr(att1_c), ptr(h1), ptr(word), ptr(mask),
                                                  ptr(gated), ptr(alpha), ptr(cx), ptr(zt), ptr(s), ptr(t), M, T, D, D, A,
                                                  ptr(ws), ws.numel(), stream_of(dev)), "set_caption_attention_train_f32")
        ctx.params = (dec_w, dec_b, gate_w, gate_b, sc_w, sc_b, tc_w, tc_b)
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(H, att1_c, h1, word, dec_w, dec_b, full_w, gate_w, sc_w, tc_w, alpha, cx, zt, s, t)
        return gated, alpha

    @staticmethod
    def backward(ctx, dgated, dalpha):
        H, att1_c, h1, word, dec_w, dec_b, full_w, gate_w, sc_w, tc_w, alpha, cx, zt, s, t = ctx.saved_tensors
        lib = _lib.load()
        M, D = cx.shape
        dev = H.device
        if dgated is None:
            dgated = torch.zeros_like(cx)
        dz, ds, dt = (torch.empty(M, D, dtype=torch.float32, device=dev) for _ in range(3))
        check(lib.set_context_gate_bwd_f32(ptr(_c(dgated)), ptr(zt), ptr(s), ptr(t), ptr(dz), ptr(ds), ptr(dt), M, D,
                                           stream_of(dev)), "set_context_gate_bwd_f32")
        # six products in two launches: the three column blocks of the context gate, then the += terms
        dctx, dword, dh1 = _dgrad_group([(dz, gate_w[:, 2 * D:], None), (dz, gate_w[:, :D], None),
                                         (dz, gate_w[:, D:2 * D], None)])
        _dgrad_group([(ds, sc_w, dctx), (dt, tc_w[:, :D], dword), (dt, tc_w[:, D:], dh1)])
        p_dec_w, p_dec_b, p_gate_w, p_gate_b, p_sc_w, p_sc_b, p_tc_w, p_tc_b = ctx.params
        wh = torch.cat([word, h1], 1)
        d_gate_w = _wgrad(p_gate_w, dz, torch.cat([wh, cx], 1))
        d_sc_w = _wgrad(p_sc_w, ds, cx)
        d_tc_w = _wgrad(p_tc_w, dt, wh)
        att2 = _linear_nograd(h1, dec_w, dec_b)
        datt1, datt2, dfull_w, dH, dfull_b = _attention_bwd(dctx, dalpha, alpha, H, att1_c, att2, full_w, True, True)
        dh1 = _dgrad(datt2, dec_w, out=dh1)
        return (dH, datt1, dh1, dword, None, _wgrad(p_dec_w, datt2, h1), _bgrad(p_dec_b, datt2), dfull_w, dfull_b,
                d_gate_w, _bgrad(p_gate_b, dz), d_sc_w, _bgrad(p_sc_b, ds), d_tc_w, _bgrad(p_tc_b, dt))


def caption_attention(H, h1, word, mask, feat_w, feat_b, dec_w, dec_b, full_w, full_b, gate_w, gate_b, sc_w, sc_b,
                      tc_w, tc_b, att1_c=None):
    if att1_c is None:
        att1_c = linear(H, feat_w, feat_b)
    return _CaptionAttention.apply(H, att1_c, h1, word, mask, dec_w, dec_b, full_w, full_b, gate_w, gate_b, sc_w, sc_b,
                                   tc_w, tc_b)


# ------------------------------------------------------------------------------------------------
# DCNet CaptionAttention.forward (dcnet.py:254-270): additive attention without gating
# ------------------------------------------------------------------------------------------------
class _DcnetCaptionAttention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feats, att1_c, h1, mask, dec_w, dec_b, full_w, full_b):
        lib = _lib.load()
        feats, att1_c, h1, mask = _c(feats), _c(att1_c), _c(h1), _c(mask)
        M, T, Dh = feats.shape
        D, A = h1.shape[1], dec_w.shape[0]
        dev = feats.device
        w = EditNetWeights()          # gate weights stay NULL -> plain context
        w.ca_dec_w, w.ca_dec_b, w.ca_full_w, w.ca_full_b = dec_w.data_ptr(), dec_b.data_ptr(), full_w.data_ptr(), full_b.data_ptr()
        cx = torch.empty(M, Dh, dtype=torch.float32, device=dev)
        alpha = torch.empty(M, T, dtype=torch.float32, device=dev)
        ws = _ws(lib.set_caption_attention_workspace_bytes, M, T, max(Dh, D), A, device=dev)
        check(lib.set_caption_attention_f32(C.byref(w), ptr(feats), ptr(att1_c), ptr(h1), None, ptr(mask), ptr(cx),
                                            ptr(alpha), M, T, Dh, D, A, ptr(ws), ws.numel(), stream_of(dev)),
              "set_caption_attention_f32")
        ctx.params = (dec_w, dec_b)
        ctx.save_for_backward(feats, att1_c, h1, dec_w, dec_b, full_w, alpha)
        return cx

    @staticmethod
    def backward(ctx, dctx):
        feats, att1_c, h1, dec_w, dec_b, full_w, alpha = ctx.saved_tensors
        att2 = _linear_nograd(h1, dec_w, dec_b)
        datt1, datt2, dfull_w, dF, dfull_b = _attention_bwd(dctx, None, alpha, feats, att1_c, att2, full_w, True, True)
        return (dF, datt1, _dgrad(datt2, dec_w), None, _wgrad(ctx.params[0], datt2, h1), _bgrad(ctx.params[1], datt2), dfull_w,
                dfull_b)


def dcnet_caption_attention(feats, h1, mask, feat_w, feat_b, dec_w, dec_b, full_w, full_b, att1_c=None):
    if att1_c is None:
        att1_c = linear(feats, feat_w, feat_b)
    return _DcnetCaptionAttention.apply(feats, att1_c, h1, mask, dec_w, dec_b, full_w, full_b)


# ------------------------------------------------------------------------------------------------
# VisualAttentionC.forward given att1 = features_att(att_embed(X)) (editnet.py:443-446)
# ------------------------------------------------------------------------------------------------
class _VisualAttention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, X, att1, h1, dec_w, dec_b, full_w, full_b, rmask):
        lib = _lib.load()
        X, att1, h1 = _c(X), _c(att1), _c(h1)
        rmask = None if rmask is None else _c(rmask.float())
        M, R, Fd = X.shape
        D, A = dec_w.shape[1], dec_w.shape[0]
        dev = X.device
        w = EditNetWeights()
        w.va_dec_w, w.va_dec_b, w.va_full_w, w.va_full_b = dec_w.data_ptr(), dec_b.data_ptr(), full_w.data_ptr(), full_b.data_ptr()
        cx = torch.empty(M, Fd, dtype=torch.float32, device=dev)
        alpha = torch.empty(M, R, dtype=torch.float32, device=dev)
        ws = _ws(lib.set_visual_attention_workspace_bytes, M, R, Fd, D, A, device=dev)
        check(lib.set_visual_attention_masked_f32(C.byref(w), ptr(X), ptr(att1), ptr(rmask), ptr(h1), ptr(cx), ptr(alpha),
                                                  M, R, Fd, D, A, ptr(ws), ws.numel(), stream_of(dev)),
              "set_visual_attention_masked_f32")
        ctx.params = (dec_w, dec_b)
        ctx.save_for_backward(X, att1, h1, dec_w, dec_b, full_w, alpha)
        return cx

    @staticmethod
    def backward(ctx, dctx):
        X, att1, h1, dec_w, dec_b, full_w, alpha = ctx.saved_tensors
        att2 = _linear_nograd(h1, dec_w, dec_b)
        datt1, datt2, dfull_w, _, dfull_b = _attention_bwd(dctx, None, alpha, X, att1, att2, full_w, False, False)
        return (None, datt1, _dgrad(datt2, dec_w), _wgrad(ctx.params[0], datt2, h1), _bgrad(ctx.params[1], datt2), dfull_w,
                dfull_b, None)


def visual_attention_from_att1(X, att1, h1, dec_w, dec_b, full_w, full_b, rmask=None):
    """rmask (M,R) 0/1: the adaptive model's valid-region mask (editnet_adaptive.py:449-453); masked regions get
    alpha == 0 exactly, so the shared attention backward needs no mask of its own."""
    return _VisualAttention.apply(X, att1, h1, dec_w, dec_b, full_w, full_b, rmask)


# ------------------------------------------------------------------------------------------------
# SelectC.forward, hard mode (editnet.py:409-420): straight-through weight on the arg-max row
# ------------------------------------------------------------------------------------------------
class _Select(torch.autograd.Function):
    @staticmethod
    def forward(ctx, Mem, alpha):
        lib = _lib.load()
        Mem, alpha = _c(Mem), _c(alpha)
        B, T, D = Mem.shape
        sel = torch.empty(B, D, dtype=torch.float32, device=Mem.device)
        check(lib.set_select_f32(ptr(Mem), ptr(alpha), ptr(sel), B, T, D, stream_of(Mem.device)), "set_select_f32")
        ctx.save_for_backward(Mem, alpha)
        return sel

    @staticmethod
    def backward(ctx, dsel):
        Mem, alpha = ctx.saved_tensors
        lib = _lib.load()
        B, T, D = Mem.shape
        dM = torch.empty_like(Mem)
        dalpha = torch.empty_like(alpha)
        check(lib.set_select_bwd_f32(ptr(_c(dsel)), ptr(Mem), ptr(alpha), ptr(dM), ptr(dalpha), B, T, D,
                                     stream_of(Mem.device)), "set_select_bwd_f32")
        return dM, dalpha


def select(Mem, alpha):
    return _Select.apply(Mem, alpha)


class _SelectSoft(torch.autograd.Function):
    """SelectC.forward with soft = True (editnet.py:419-420): sel = sum_t alpha_t M_t, differentiable in both."""

    @staticmethod
    def forward(ctx, Mem, alpha):
        Mem, alpha = _c(Mem), _c(alpha)
        ctx.save_for_backward(Mem, alpha)
        return select_soft_nograd(Mem, alpha)

    @staticmethod
    def backward(ctx, dsel):
        Mem, alpha = ctx.saved_tensors
        lib = _lib.load()
        B, T, D = Mem.shape
        dM = torch.empty_like(Mem)
        dalpha = torch.empty_like(alpha)
        check(lib.set_select_soft_bwd_f32(ptr(_c(dsel)), ptr(Mem), ptr(alpha), ptr(dM), ptr(dalpha), B, T, D,
                                          stream_of(Mem.device)), "set_select_soft_bwd_f32")
        return dM, dalpha


def select_soft_nograd(Mem, alpha):
    lib = _lib.load()
    B, T, D = Mem.shape
    sel = torch.empty(B, D, dtype=torch.float32, device=Mem.device)
    check(lib.set_select_soft_f32(ptr(Mem), ptr(alpha), ptr(sel), B, T, D, stream_of(Mem.device)), "set_select_soft_f32")
    return sel


def select_soft(Mem, alpha):
    return _SelectSoft.apply(Mem, alpha)


# ------------------------------------------------------------------------------------------------
# CopyLSTMCellC.forward (editnet.py:265-285)
# ------------------------------------------------------------------------------------------------
class _CopyLstm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, h2, c2, cmem, x2h_w, x2h_b, h2h_w, h2h_b, cnew_w, cnew_b, cmem_w, cmem_b):
        lib = _lib.load()
        x, h2, c2, cmem = _c(x), _c(h2), _c(c2), _c(cmem)
        M, K, D = x.shape[0], x.shape[1], h2.shape[1]
        dev = x.device
        w = EditNetWeights()
        w.cl_x2h_w, w.cl_x2h_b, w.cl_h2h_w, w.cl_h2h_b = x2h_w.data_ptr(), x2h_b.data_ptr(), h2h_w.data_ptr(), h2h_b.data_ptr()
        w.cl_cnew_w, w.cl_cnew_b, w.cl_cmem_w, w.cl_cmem_b = cnew_w.data_ptr(), cnew_b.data_ptr(), cmem_w.data_ptr(), cmem_b.data_ptr()
        h_new, adp, c_new, cg = (torch.empty(M, D, dtype=torch.float32, device=dev) for _ in range(4))
        gates = torch.empty(M, 4 * D, dtype=torch.float32, device=dev)
        ws = _ws(lib.set_copy_lstm_workspace_bytes, M, D, K, device=dev)
        check(lib.set_copy_lstm_train_f32(C.byref(w), ptr(x), K, K, ptr(h2), ptr(c2), ptr(cmem), ptr(h_new), ptr(adp),
                                          ptr(gates), ptr(c_new), ptr(cg), M, D, ptr(ws), ws.numel(), stream_of(dev)),
              "set_copy_lstm_train_f32")
        ctx.params = (x2h_w, x2h_b, h2h_w, h2h_b, cnew_w, cnew_b, cmem_w, cmem_b)
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(x, h2, c2, cmem, x2h_w, h2h_w, cnew_w, cmem_w, gates, c_new, cg, adp)
        return h_new, adp

    @staticmethod
    def backward(ctx, dh, dadp):
        x, h2, c2, cmem, x2h_w, h2h_w, cnew_w, cmem_w, gates, c_new, cg, adp = ctx.saved_tensors
        lib = _lib.load()
        M, D = h2.shape
        dev = x.device
        st = stream_of(dev)
        du, dcm, dcn, dop = (torch.empty(M, D, dtype=torch.float32, device=dev) for _ in range(4))
        ogate = gates[:, 3 * D:].contiguous()
        check(lib.set_copy_gate_bwd_f32(ptr(None if dh is None else _c(dh)), ptr(None if dadp is None else _c(dadp)),
                                        ptr(ogate), ptr(adp), ptr(cg), ptr(cmem), ptr(c_new), ptr(du), ptr(dcm), ptr(dcn),
                                        ptr(dop), M, D, st), "set_copy_gate_bwd_f32")
        _dgrad_group([(du, cnew_w, dcn), (du, cmem_w, dcm)])
        dgw = torch.empty_like(gates)
        dc2 = torch.empty_like(c2)
        check(lib.set_lstm_gates_bwd_f32(ptr(dcn), ptr(dop), ptr(gates), ptr(c2), ptr(dgw), ptr(dc2), M, D, st),
              "set_lstm_gates_bwd_f32")
        p = ctx.params
        dx, dh2 = _dgrad_group([(dgw, x2h_w, None), (dgw, h2h_w, None)])
        return (dx, dh2, dc2, dcm, _wgrad(p[0], dgw, x), _bgrad(p[1], dgw), _wgrad(p[2], dgw, h2),
                _bgrad(p[3], dgw), _wgrad(p[4], du, c_new), _bgrad(p[5], du), _wgrad(p[6], du, cmem), _bgrad(p[7], du))


def copy_lstm(x, h2, c2, cmem, x2h_w, x2h_b, h2h_w, h2h_b, cnew_w, cnew_b, cmem_w, cmem_b):
    return _CopyLstm.apply(x, h2, c2, cmem, x2h_w, x2h_b, h2h_w, h2h_b, cnew_w, cnew_b, cmem_w, cmem_b)


# ------------------------------------------------------------------------------------------------
# Multinomial sampling epilogue of a free-running step (editnet_rl.py:521-543, dcnet_rl.py:320-340)
# ------------------------------------------------------------------------------------------------
class SampleState:
    """Device-side bookkeeping of one sampled rollout: seq (B,max_len), the token fed to every step (row t of
    `tokens`), the `unfinished` latch and the per-step alive counters that emulate the reference's early `break`
    without a host synchronisation.  Seed / offset select the Philox stream (seed drawn from torch's CPU generator,
    so torch.manual_seed makes rollouts reproducible)."""

    def __init__(self, B, max_len, start_idx, end_idx, device, seed=None, offset=0):
        self.B, self.max_len, self.end_idx = B, max_len, int(end_idx)
        self.seq = torch.zeros(B, max_len, dtype=torch.long, device=device)
        self.tokens = torch.zeros(max_len + 2, B, dtype=torch.long, device=device)
        self.tokens[0].fill_(int(start_idx))
        self.unfinished = torch.empty(B, dtype=torch.int32, device=device)
        self.alive = torch.empty(max_len + 2, dtype=torch.int32, device=device)
        self.seed = _rng.next_seed() if seed is None else int(seed)
        self.offset = int(offset)


class _SamplePick(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, state, t):
        lib = _lib.load()
        logits = _c(logits)
        B, V = logits.shape
        dev = logits.device
        raw = torch.empty(B, dtype=torch.long, device=dev)
        lse = torch.empty(B, dtype=torch.float32, device=dev)
        logp = torch.empty(B, dtype=torch.float32, device=dev)
        check(lib.set_sample_pick_f32(ptr(logits), logits.stride(0), B, V, t, state.max_len, state.end_idx, state.seed,
                                      state.offset, ptr(state.seq), ptr(state.tokens[t + 1]), ptr(state.unfinished),
                                      ptr(state.alive), ptr(raw), ptr(lse), ptr(logp), stream_of(dev)),
              "set_sample_pick_f32")
        ctx.save_for_backward(logits, lse, raw)
        return logp

    @staticmethod
    def backward(ctx, g):
        logits, lse, raw = ctx.saved_tensors
        lib = _lib.load()
        B, V = logits.shape
        d = torch.empty_like(logits)
        check(lib.set_sample_logp_bwd_f32(ptr(logits), logits.stride(0), ptr(lse), ptr(raw), ptr(_c(g)), ptr(d),
                                          d.stride(0), B, V, stream_of(logits.device)), "set_sample_logp_bwd_f32")
        return d, None, None


def philox_categorical(logits, seed, offset):
    """it ~ softmax(logits) per row (no gradient, no bookkeeping): the scheduled-sampling draw of editnet.py:515-517
    (`torch.multinomial(torch.exp(scores))` normalises, i.e. samples softmax(scores)) on the device sampling epilogue:
    row b uses the uniform of counter (b, 0, offset), inverse CDF over the epilogue's fixed word enumeration."""
    lib = _lib.load()
    logits = _c(logits)
    B, V = logits.shape
    dev = logits.device
    i64 = torch.zeros(3, B, dtype=torch.long, device=dev)
    i32 = torch.zeros(B + 8, dtype=torch.int32, device=dev)
    f32 = torch.empty(2, B, dtype=torch.float32, device=dev)
    check(lib.set_sample_pick_f32(ptr(logits), logits.stride(0), B, V, 0, 1, -1, int(seed), int(offset), ptr(i64[0]),
                                  ptr(i64[1]), ptr(i32), ptr(i32[B:]), ptr(i64[2]), ptr(f32[0]), ptr(f32[1]),
                                  stream_of(dev)), "set_sample_pick_f32")
    return i64[2]


def sample_pick(logits, state, t):
    """One sampled step: draws it ~ softmax(logits) on the device (Philox), applies the <end> / unfinished / break
    bookkeeping into `state` (seq[:, t], tokens[t + 1]) and returns log_softmax(logits)[it] (B), differentiable
    w.r.t. logits.  No host synchronisation."""
    return _SamplePick.apply(logits, state, t)
