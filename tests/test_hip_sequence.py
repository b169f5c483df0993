"""GPU: the whole-sequence XE training node (show_edit_tell_amd/xe_sequence.py) and its helper kernels.
Gradient parity against the REFERENCE's autograd runs in tests/test_hip_train.py (both routes); here: the node against
the per-operator route on ragged / uniform batches, the Philox dropout kernels, the row packer, the accumulating
backward kernels."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device("cuda:0")


def _build(V, D, A, F, seed=5):
    from show_edit_tell_amd import editnet, synth
    wm = synth.word_map(V)
    sd = synth.editnet_state(seed, V, D, A, F)
    sd["caption_encoder.embed.embedding.weight"] = sd["embed.embedding.weight"]
    m = editnet.DecoderC(wm, D, D, D, A, F)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return m.to(_dev())


def _inputs(B, R, F, T, V, min_len):
    from show_edit_tell_amd import synth
    dev = _dev()
    X = torch.from_numpy(synth.features(3, B, R, F)).to(dev)
    prev, plen = (torch.from_numpy(x).to(dev) for x in synth.prev_captions(3, B, T, V, 5))
    caps, clen = (torch.from_numpy(x).to(dev) for x in synth.captions(3, B, V, T, min_len))
    return X, caps, clen, prev, plen


def _loss_and_grads(m, inputs, seq, deferred, monkeypatch):
    import contextlib
    from show_edit_tell_amd import editnet
    from show_edit_tell_amd.autograd_ops import deferred_param_grads
    from show_edit_tell_amd.train import xe_loss_sum
    monkeypatch.setattr(editnet, "_XE_SEQUENCE", seq)
    m.zero_grad(set_to_none=True)
    X, caps, clen, prev, plen = inputs
    pred, caps_s, dl, _ = m(X, caps, clen, prev, plen, False, 0.0)
    ls, n, _, _ = xe_loss_sum(pred, caps_s, dl)
    with (deferred_param_grads() if deferred else contextlib.nullcontext()):
        (ls / n).backward()
    return pred.detach().clone(), {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}


@pytest.mark.parametrize("B,V,D,A,F,min_len,deferred", [(4, 203, 64, 32, 256, 8, False), (6, 203, 64, 32, 256, 20, True),
                                                       (16, 1000, 256, 128, 2048, 6, True)])
def test_sequence_node_equals_per_operator_route(B, V, D, A, F, min_len, deferred, monkeypatch):
    """same kernels, two host schedules: the node contracts the loop-invariant final_hidden / image_mean columns of the
    attention LSTM once per sequence (as the decode path does), so scores agree to fp32 summation order; gradients equal
    up to the order of the sums over timesteps"""
    m = _build(V, D, A, F).eval()
    inputs = _inputs(B, 36, F, 20, V, min_len)
    p0, g0 = _loss_and_grads(m, inputs, False, deferred, monkeypatch)
    p1, g1 = _loss_and_grads(m, inputs, True, deferred, monkeypatch)
    assert float((p0 - p1).abs().max()) <= 2e-5 * max(1.0, float(p0.abs().max()))
    assert set(g0) == set(g1)
    gmax = max(float(g.abs().max()) for g in g0.values())
    for k in g0:
        if k.endswith("full_att.bias"):          # mathematically zero (softmax is shift invariant): rounding noise
            assert float(g1[k].abs().max()) < 1e-4 * gmax
            continue
        err = float((g0[k] - g1[k]).abs().max()) / max(float(g0[k].abs().max()), 1e-6 * gmax)
        assert err < 1e-3, (k, err)


def test_sequence_node_train_mode_learns(monkeypatch):
    """train(): the three dropout sites run on the library's Philox kernels; ragged lengths; the loss falls"""
    from show_edit_tell_amd import editnet
    from show_edit_tell_amd.autograd_ops import deferred_param_grads
    from show_edit_tell_amd.train import xe_loss_sum
    monkeypatch.setattr(editnet, "_XE_SEQUENCE", True)
    torch.manual_seed(0)
    m = _build(203, 64, 32, 256).train()
    X, caps, clen, prev, plen = _inputs(8, 36, 256, 20, 203, 6)
    opt = torch.optim.Adam(m.parameters(), lr=2e-3)
    losses = []
    for _ in range(12):
        opt.zero_grad()
        pred, caps_s, dl, _ = m(X, caps, clen, prev, plen, False, 0.0)
        ls, n, _, _ = xe_loss_sum(pred, caps_s, dl)
        with deferred_param_grads():
            (ls / n).backward()
        opt.step()
        losses.append(float((ls / n).detach()))
    assert all(np.isfinite(losses))
    assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in m.parameters())
    assert np.mean(losses[-3:]) < np.mean(losses[:3]) - 0.2, losses
    # two forwards draw different masks (the seed comes from torch's CPU generator)
    a = m(X, caps, clen, prev, plen, False, 0.0)[0].detach()
    b = m(X, caps, clen, prev, plen, False, 0.0)[0].detach()
    assert not torch.equal(a, b)


@pytest.mark.parametrize("min_len", [6, 20])
def test_training_loop_schedules_agree(min_len, monkeypatch):
    """train mode (all three dropout sites on), ragged and uniform lengths: the C timestep loops (merged backward launches,
    all-timestep embedding / output dropout, one packing launch per timestep), the same with per-timestep output logs (no
    packing launch inside the loop, the copy cell contracting [h1 | gated | attend_img] as segments) and the Python loops
    draw the same masks and give the same scores and gradients up to fp32 summation order"""
    from show_edit_tell_amd import xe_sequence as xs
    m = _build(203, 64, 32, 256).train()
    inputs = _inputs(8, 36, 256, 20, 203, min_len)
    runs = {}
    for name, c_loops, step_logs in (("c", True, False), ("c_logs", True, True), ("py", False, False)):
        monkeypatch.setattr(xs, "_C_LOOPS", c_loops)
        monkeypatch.setattr(xs, "_STEP_LOGS", step_logs)
        torch.manual_seed(11)
        runs[name] = _loss_and_grads(m, inputs, True, True, monkeypatch)
    p0, g0 = runs["py"]
    gmax = max(float(g.abs().max()) for g in g0.values())
    for name in ("c", "c_logs"):
        p1, g1 = runs[name]
        assert float((p0 - p1).abs().max()) <= 2e-5 * max(1.0, float(p0.abs().max())), name
        assert set(g0) == set(g1)
        for k in g0:
            if k.endswith("full_att.bias"):
                continue
            err = float((g0[k] - g1[k]).abs().max()) / max(float(g0[k].abs().max()), 1e-6 * gmax)
            assert err < 1e-3, (name, k, err)


@pytest.mark.parametrize("order", ["shuffled", "sorted"])
def test_host_caption_lengths_give_the_same_forward(order, monkeypatch):
    """`DecoderC.with_host_lengths` (the training steps pass the loader's host copy of the lengths): the sort order and the
    decode lengths come from the host — same permutation as the stable device sort, an already ordered batch is not gathered —
    and scores, returned captions, lengths and sort indices are those of the plain call, bit for bit"""
    from show_edit_tell_amd import editnet
    monkeypatch.setattr(editnet, "_XE_SEQUENCE", True)
    m = _build(203, 64, 32, 256).eval()
    X, caps, clen, prev, plen = _inputs(8, 36, 256, 20, 203, 6)
    if order == "sorted":
        idx = clen.squeeze(1).sort(descending=True, stable=True)[1]
        X, caps, clen, prev, plen = X[idx], caps[idx], clen[idx], prev[idx], plen[idx]
    with torch.enable_grad():
        a = m(X, caps, clen, prev, plen, False, 0.0)
        b = m.with_host_lengths(clen.cpu())(X, caps, clen, prev, plen, False, 0.0)
    assert "_caplens_host" not in m.__dict__                       # consumed by the call
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and a[2] == b[2] and torch.equal(a[3], b[3])
    if order == "sorted":
        assert torch.equal(b[3], torch.arange(8, device=b[3].device))


def test_dropout_kernels():
    from show_edit_tell_amd import _lib
    lib, dev = _lib.load(), _dev()
    st = _lib.stream_of(dev)
    rows, cols, p = 512, 1024, 0.5
    x = torch.rand(rows, cols, device=dev) + 0.5
    y = torch.empty_like(x)
    _lib.check(lib.set_dropout_f32(x.data_ptr(), cols, y.data_ptr(), cols, rows, cols, p, 1234, 7, st), "set_dropout_f32")
    keep = (y != 0)
    frac = float(keep.float().mean())
    assert abs(frac - (1 - p)) < 4 * np.sqrt(p * (1 - p) / (rows * cols))
    assert torch.allclose(y[keep], x[keep] / (1 - p))
    # per-row and per-column keep rates are unbiased too (no stripe pattern from the counter layout)
    assert float((keep.float().mean(0) - (1 - p)).abs().max()) < 0.12
    assert float((keep.float().mean(1) - (1 - p)).abs().max()) < 0.09
    y2 = torch.empty_like(x)
    _lib.check(lib.set_dropout_f32(x.data_ptr(), cols, y2.data_ptr(), cols, rows, cols, p, 1234, 7, st), "set_dropout_f32")
    assert torch.equal(y, y2)                                  # (seed, offset) reproduces the mask
    _lib.check(lib.set_dropout_f32(x.data_ptr(), cols, y2.data_ptr(), cols, rows, cols, p, 1234, 8, st), "set_dropout_f32")
    assert float(((y2 != 0) == keep).float().mean()) < 0.55    # another offset: an independent mask
    # p = 0.2, strided views
    big = torch.zeros(rows, 2 * cols, device=dev)
    _lib.check(lib.set_dropout_f32(x.data_ptr(), cols, big[:, cols:].data_ptr(), 2 * cols, rows, cols, 0.2, 5, 0, st),
               "set_dropout_f32")
    assert float(big[:, :cols].abs().max()) == 0.0
    assert abs(float((big[:, cols:] != 0).float().mean()) - 0.8) < 0.01
    # backward: dx (+)= dy * mask * scale
    dy = torch.randn(rows, cols, device=dev)
    dx = torch.ones(rows, cols, device=dev)
    _lib.check(lib.set_dropout_bwd_f32(dy.data_ptr(), cols, y.data_ptr(), cols, dx.data_ptr(), cols, rows, cols, 2.0, 1, st),
               "set_dropout_bwd_f32")
    assert torch.allclose(dx, 1.0 + dy * keep * 2.0)
    _lib.check(lib.set_dropout_bwd_f32(dy.data_ptr(), cols, y.data_ptr(), cols, dx.data_ptr(), cols, rows, cols, 2.0, 0, st),
               "set_dropout_bwd_f32")
    assert torch.allclose(dx, dy * keep * 2.0)
    assert lib.set_dropout_f32(x.data_ptr(), cols, y.data_ptr(), cols, rows, cols, 1.0, 1, 0, st) == 1      # SET_ERR_ARG


def test_dropout_steps_kernels_equal_per_step_launches():
    """set_dropout_steps_f32 / set_dropout_bwd_steps_f32 (all timesteps of the region dropout in one launch) are bit for bit
    the per-timestep launches they replace: same counters (offset + t), same accumulation order over t."""
    from show_edit_tell_amd import _lib, rng
    lib, dev = _lib.load(), _dev()
    st = _lib.stream_of(dev)
    rows, cols, T, p, seed = 36 * 7, 256, 5, 0.5, 991
    x = torch.relu(torch.randn(rows, cols, device=dev))
    off0 = rng.offset(rng.SITE_REGION, 0)
    ys = torch.empty(T, rows, cols, device=dev)
    _lib.check(lib.set_dropout_steps_f32(x.data_ptr(), cols, ys.data_ptr(), cols, rows * cols, rows, cols, T, p, seed, off0, st),
               "set_dropout_steps_f32")
    ref = torch.empty_like(ys)
    for t in range(T):
        _lib.check(lib.set_dropout_f32(x.data_ptr(), cols, ref[t].data_ptr(), cols, rows, cols, p, seed,
                                       rng.offset(rng.SITE_REGION, t), st), "set_dropout_f32")
    assert torch.equal(ys, ref)
    assert not torch.equal(ys[0], ys[1])
    dy = torch.randn(T, rows, cols, device=dev)
    dx = torch.randn(rows, cols, device=dev)
    dx_ref = dx.clone()
    _lib.check(lib.set_dropout_bwd_steps_f32(dy.data_ptr(), cols, rows * cols, ys.data_ptr(), cols, rows * cols, dx.data_ptr(), cols,
                                             rows, cols, T, 2.0, 1, st), "set_dropout_bwd_steps_f32")
    for t in range(T):
        _lib.check(lib.set_dropout_bwd_f32(dy[t].data_ptr(), cols, ys[t].data_ptr(), cols, dx_ref.data_ptr(), cols, rows, cols,
                                           2.0, 1, st), "set_dropout_bwd_f32")
    assert torch.equal(dx, dx_ref)
    _lib.check(lib.set_dropout_bwd_steps_f32(dy.data_ptr(), cols, rows * cols, ys.data_ptr(), cols, rows * cols, dx.data_ptr(), cols,
                                             rows, cols, T, 2.0, 0, st), "set_dropout_bwd_steps_f32")
    dx_ref.zero_()
    for t in range(T):
        _lib.check(lib.set_dropout_bwd_f32(dy[t].data_ptr(), cols, ys[t].data_ptr(), cols, dx_ref.data_ptr(), cols, rows, cols,
                                           2.0, 1, st), "set_dropout_bwd_f32")
    assert torch.equal(dx, dx_ref)


def test_pack_rows():
    from show_edit_tell_amd import _lib
    lib, dev = _lib.load(), _dev()
    st = _lib.stream_of(dev)
    a, b, c = torch.randn(37, 64, device=dev), torch.randn(40, 256, device=dev)[:, 128:], torch.randn(37, 8, device=dev)
    dst = torch.full((40, 512), 9.0, device=dev)
    srcs = [a, b, c]
    n = len(srcs)
    ps = (C.c_void_p * n)(*[s.data_ptr() for s in srcs])
    ls = (C.c_int64 * n)(*[s.stride(0) for s in srcs])
    cs = (C.c_int * n)(*[s.shape[1] for s in srcs])
    _lib.check(lib.set_pack_f32(dst.data_ptr(), 512, 37, n, ps, ls, cs, 0, st), "set_pack_f32")
    ref = torch.cat([a, b[:37], c], 1)
    assert torch.equal(dst[:37, :200], ref)
    assert float((dst[:37, 200:] - 9.0).abs().max()) == 0.0 and float((dst[37:] - 9.0).abs().max()) == 0.0
    _lib.check(lib.set_pack_f32(dst.data_ptr(), 512, 37, n, ps, ls, cs, 1, st), "set_pack_f32")
    assert torch.allclose(dst[:37, :200], 2 * ref)


def test_accumulating_backward_kernels():
    """set_attention_bwd_acc_f32 / set_select_bwd_acc_f32 with the flags set add exactly what the plain forms write;
    set_attention_dvalues_f32 equals the sum over timesteps of the per-step dvalues"""
    from show_edit_tell_amd import _lib
    lib, dev = _lib.load(), _dev()
    st = _lib.stream_of(dev)
    M, L, Dv, A = 9, 20, 64, 32
    g = torch.Generator(device="cpu").manual_seed(3)
    rnd = lambda *s: torch.randn(*s, generator=g).to(dev)
    dctx, alpha = rnd(M, Dv), torch.softmax(rnd(M, L), 1)
    vals, att1, att2, wf, dal = rnd(M, L, Dv), rnd(M, L, A), rnd(M, A), rnd(A), rnd(M, L)

    def run(acc1, accv, datt1, dval):
        datt2, dwf, de = torch.empty(M, A, device=dev), torch.empty(M, A, device=dev), torch.empty(M, L, device=dev)
        _lib.check(lib.set_attention_bwd_acc_f32(dctx.data_ptr(), dal.data_ptr(), alpha.data_ptr(), vals.data_ptr(),
                                                 att1.data_ptr(), att2.data_ptr(), wf.data_ptr(), datt1.data_ptr(),
                                                 datt2.data_ptr(), dwf.data_ptr(), dval.data_ptr(), de.data_ptr(), M, L, Dv, A, 1,
                                                 acc1, accv, 0, st), "set_attention_bwd_acc_f32")
        return datt2, dwf, de

    d1, dv = torch.empty(M, L, A, device=dev), torch.empty(M, L, Dv, device=dev)
    base = run(0, 0, d1, dv)
    d1b, dvb = d1.clone() * 0 + 1.5, dv.clone() * 0 - 2.0
    again = run(1, 1, d1b, dvb)
    assert torch.allclose(d1b, d1 + 1.5) and torch.allclose(dvb, dv - 2.0)
    for x, y in zip(base, again):
        assert torch.equal(x, y)
    # select
    Mem, dsel = rnd(M, L, Dv), rnd(M, Dv)
    dM, da = torch.empty(M, L, Dv, device=dev), torch.empty(M, L, device=dev)
    _lib.check(lib.set_select_bwd_acc_f32(dsel.data_ptr(), Mem.data_ptr(), alpha.data_ptr(), dM.data_ptr(), da.data_ptr(), M, L,
                                          Dv, 0, st), "set_select_bwd_acc_f32")
    dM2, da2 = torch.full((M, L, Dv), 3.0, device=dev), torch.empty(M, L, device=dev)
    _lib.check(lib.set_select_bwd_acc_f32(dsel.data_ptr(), Mem.data_ptr(), alpha.data_ptr(), dM2.data_ptr(), da2.data_ptr(), M,
                                          L, Dv, 1, st), "set_select_bwd_acc_f32")
    assert torch.allclose(dM2, dM + 3.0) and torch.equal(da, da2)
    # batched dvalues over T steps
    T = 7
    al, dc = torch.softmax(rnd(T, M, L), 2), rnd(T, M, Dv)
    out = torch.empty(M, L, Dv, device=dev)
    _lib.check(lib.set_attention_dvalues_f32(al.data_ptr(), dc.data_ptr(), out.data_ptr(), T, M, L, Dv, 0, st),
               "set_attention_dvalues_f32")
    ref = torch.einsum("tbl,tbd->bld", al.double(), dc.double()).float()
    assert torch.allclose(out, ref, atol=1e-5)


def test_rollout_node_equals_per_operator_rollout(monkeypatch):
    """sampled SCST rollout (editnet_rl.py:485-549, sample_rl) in eval mode (no dropout): the node and the per-operator
    route draw the same Philox stream from the same scores -> identical sequences and log-probs, equal gradients of a
    weighted sum of the log-probs"""
    from show_edit_tell_amd import editnet, editnet_rl, synth
    V, D, A, F, B = 203, 64, 32, 256, 6
    wm = synth.word_map(V)
    sd = synth.editnet_state(9, V, D, A, F, emb_scale=3.0, fc_scale=4.0, gain=2.0)
    sd["caption_encoder.embed.embedding.weight"] = sd["embed.embedding.weight"]
    m = editnet_rl.DecoderC(wm, D, D, D, A, F)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    m = m.to(_dev()).eval()
    X, _, _, prev, plen = _inputs(B, 36, F, 20, V, 20)
    wgt = torch.linspace(0.5, 1.5, B * m.max_len, device=_dev()).view(B, m.max_len)
    out = []
    for seq_node in (False, True):
        monkeypatch.setattr(editnet, "_XE_SEQUENCE", seq_node)
        m.zero_grad(set_to_none=True)
        torch.manual_seed(11)
        seq, logp = m(wm, prev, plen, X, sample_max=False, sample_rl=True)
        assert logp.requires_grad and not seq.requires_grad
        (logp * wgt).sum().backward()
        out.append((seq.clone(), logp.detach().clone(), {k: p.grad.detach().clone() for k, p in m.named_parameters()}))
    (s0, l0, g0), (s1, l1, g1) = out
    assert torch.equal(s0, s1) and int((s0 > 0).sum()) > B
    assert torch.allclose(l0, l1, atol=1e-5)
    gmax = max(float(g.abs().max()) for g in g0.values())
    for k in g0:
        if k.endswith("full_att.bias"):
            continue
        err = float((g0[k] - g1[k]).abs().max()) / max(float(g0[k].abs().max()), 1e-6 * gmax)
        assert err < 1e-3, (k, err)


def test_colsum_kernel():
    from show_edit_tell_amd import autograd_ops as A
    dev = _dev()
    g = torch.Generator(device="cpu").manual_seed(5)
    for rows, cols in ((2432, 4096), (87552, 512), (100, 64), (63, 8)):
        x = torch.randn(rows, cols, generator=g).to(dev)
        ref = x.double().sum(0).float()
        out = A._colsum(x)
        assert torch.allclose(out, ref, rtol=1e-4, atol=1e-3 * float(ref.abs().max()))
        acc = torch.ones(cols, device=dev)
        A._colsum(x, out=acc)
        assert torch.allclose(acc, ref + 1.0, rtol=1e-4, atol=1e-3 * float(ref.abs().max()))
        assert torch.equal(A._colsum(x), out)                 # deterministic
    big = torch.randn(512, 256, generator=g).to(dev)
    assert torch.allclose(A._colsum(big[:, 64:192]), big[:, 64:192].sum(0), atol=1e-4)


def test_grouped_colsum_equals_single_colsums():
    """set_colsum_group_f32 (every bias gradient of a step as one pair of launches): per problem the sums of the single kernel
    (to rounding: the partials are combined in a different order), `.grad` created or accumulated, a shared problem
    writing both of its parameters, and the problems the grouped kernel cannot take (ragged columns, few rows) falling back"""
    from show_edit_tell_amd import autograd_ops as A
    dev = _dev()
    g = torch.Generator(device="cpu").manual_seed(9)
    shapes = [(2432, 4096), (87552, 512), (2432, 1024), (100, 64), (2432, 6), (40, 8)]
    xs = [torch.randn(r, c, generator=g).to(dev) for r, c in shapes]
    strided = torch.randn(2432, 3072, generator=g).to(dev)
    xs.append(strided[:, 1024:2048])                               # a column block of a wider log
    params = [[torch.nn.Parameter(torch.zeros(x.shape[1], device=dev)) for _ in range(1 + (i % 2))] for i, x in enumerate(xs)]
    params[2][0].grad = torch.full((1024,), 2.0, device=dev)        # accumulate
    params[0] = [torch.nn.Parameter(torch.zeros(1, 4096, device=dev))]   # a (1, A) parameter (full_att.weight)
    A._colsum_group([(x, pl) for x, pl in zip(xs, params)])
    for i, (x, pl) in enumerate(zip(xs, params)):
        ref = x.double().sum(0).float()
        tol = 1e-3 * float(ref.abs().max())
        for j, q in enumerate(pl):
            want = ref + (2.0 if (i == 2 and j == 0) else 0.0)
            assert q.grad.shape == q.shape
            assert torch.allclose(q.grad.reshape(-1), want, rtol=1e-4, atol=tol), (i, j)
        if len(pl) == 2:
            assert torch.equal(pl[0].grad, pl[1].grad)
    first = [q.grad.clone() for pl in params for q in pl]
    for pl in params:
        for q in pl:
            q.grad = None
    params[2][0].grad = torch.full((1024,), 2.0, device=dev)
    A._colsum_group([(x, pl) for x, pl in zip(xs, params)])
    assert all(torch.equal(a, q.grad) for a, q in zip(first, [q for pl in params for q in pl]))    # deterministic


def test_grouped_colsum_more_problems_than_one_launch_takes():
    """more than SET_COLSUM_MAX (24) problems: the surplus goes through the single kernel, every parameter still gets its sums"""
    from show_edit_tell_amd import autograd_ops as A
    dev = _dev()
    g = torch.Generator(device="cpu").manual_seed(10)
    xs = [torch.randn(128, 64, generator=g).to(dev) for _ in range(30)]
    ps = [[torch.nn.Parameter(torch.zeros(64, device=dev))] for _ in xs]
    A._colsum_group(list(zip(xs, ps)))
    for x, pl in zip(xs, ps):
        assert torch.allclose(pl[0].grad, x.double().sum(0).float(), rtol=1e-4, atol=1e-4)


def test_dcnet_rollout_node_equals_per_operator_rollout(monkeypatch):
    """DCNet (dcnet_rl.py:286-346, sample_rl) in eval mode: node and per-operator route draw the same words, same
    log-probs, equal gradients; the teacher-forced node equals the per-operator XE route on a ragged batch"""
    from show_edit_tell_amd import dcnet, dcnet_rl, editnet, synth
    from show_edit_tell_amd.train import xe_loss_sum
    V, D, A, Cc, E, B = 203, 64, 32, 32, 64, 6
    wm = synth.word_map(V)
    sd = synth.dcnet_state(4, V, D, A, Cc, E, 3.0, 4.0, 2.0)
    dev = _dev()
    prev, plen = (torch.from_numpy(x).to(dev) for x in synth.prev_captions(3, B, 20, V, 5))
    caps, clen = (torch.from_numpy(x).to(dev) for x in synth.captions(3, B, V, 20, 7))

    def grads(m):
        return {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}

    def close(g0, g1):
        assert set(g0) == set(g1)
        gmax = max(float(g.abs().max()) for g in g0.values())
        for k in g0:
            if k.endswith("full_att.bias"):
                continue
            err = float((g0[k] - g1[k]).abs().max()) / max(float(g0[k].abs().max()), 1e-6 * gmax)
            assert err < 1e-3, (k, err)

    rl = dcnet_rl.DAE(wm, None, D, A, Cc, E)
    rl.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    rl = rl.to(dev).eval()
    wgt = torch.linspace(0.5, 1.5, B * rl.max_len, device=dev).view(B, rl.max_len)
    out = []
    for seq_node in (False, True):
        monkeypatch.setattr(editnet, "_XE_SEQUENCE", seq_node)
        rl.zero_grad(set_to_none=True)
        torch.manual_seed(5)
        seq, logp = rl(wm, prev, plen, sample_max=False, sample_rl=True)
        (logp * wgt).sum().backward()
        out.append((seq.clone(), logp.detach().clone(), grads(rl)))
    assert torch.equal(out[0][0], out[1][0]) and torch.allclose(out[0][1], out[1][1], atol=1e-5)
    close(out[0][2], out[1][2])

    xe = dcnet.DAE(wm, None, D, A, Cc, E)
    xe.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    xe = xe.to(dev).eval()
    res = []
    for seq_node in (False, True):
        monkeypatch.setattr(editnet, "_XE_SEQUENCE", seq_node)
        xe.zero_grad(set_to_none=True)
        pred, caps_s, dl, _ = xe(caps, clen, prev, plen)
        ls, n, _, _ = xe_loss_sum(pred, caps_s, dl)
        (ls / n).backward()
        res.append((pred.detach().clone(), grads(xe)))
    assert float((res[0][0] - res[1][0]).abs().max()) <= 2e-5 * max(1.0, float(res[0][0].abs().max()))
    close(res[0][1], res[1][1])


def test_scheduled_sampling_in_node(monkeypatch):
    """editnet.py:508-520 inside the node (device Philox draw from softmax(previous scores), no host sync): about ss_prob
    of the input words from step 1 on are replaced; feeding the SAME words teacher-forced through the per-operator route
    reproduces the node's scores and gradients (the sampled words carry no gradient, as in the reference); train mode
    runs and stays finite."""
    from show_edit_tell_amd import editnet, xe_sequence
    from show_edit_tell_amd.train import xe_loss_sum
    V, D, A, F, B = 203, 64, 32, 256, 16
    m = _build(V, D, A, F).eval()
    X, caps, clen, prev, plen = _inputs(B, 36, F, 20, V, 9)
    seen = []
    real = xe_sequence.xe_sequence
    monkeypatch.setattr(xe_sequence, "xe_sequence", lambda cfg, *a: (seen.append(cfg), real(cfg, *a))[1])
    monkeypatch.setattr(editnet, "_XE_SEQUENCE", True)
    m.zero_grad(set_to_none=True)
    torch.manual_seed(2)
    pred, caps_s, dl, sort_ind = m(X, caps, clen, prev, plen, True, 0.6)
    ls, n, _, _ = xe_loss_sum(pred, caps_s, dl)
    (ls / n).backward()
    g_ss = {k: p.grad.detach().clone() for k, p in m.named_parameters()}
    tok = seen[-1].fed_tokens                                   # (T, B), sorted row order
    T = tok.shape[0]
    assert torch.equal(tok[0], caps_s[:, 0])
    active = torch.tensor([[l > t for l in dl] for t in range(T)], device=tok.device)
    changed = (tok != caps_s[:, :T].t()) & active
    frac = float(changed[1:].sum()) / float(active[1:].sum())
    assert 0.45 < frac < 0.75, frac                             # ss_prob = 0.6 (a draw may coincide with the ground truth)
    # the same words, teacher-forced, per-operator route
    fed = caps_s.clone()
    fed[:, :T] = torch.where(active.t(), tok.t(), caps_s[:, :T])
    monkeypatch.setattr(editnet, "_XE_SEQUENCE", False)
    m.zero_grad(set_to_none=True)
    lens_sorted = clen[sort_ind]
    pred2, _, dl2, sort2 = m(X[sort_ind], fed, lens_sorted, prev[sort_ind], plen[sort_ind], False, 0.0)
    assert dl2 == dl                                            # (the sort may permute rows of equal length: sort2)
    assert torch.allclose(pred[sort2], pred2, atol=1e-5)
    ls2, n2, _, _ = xe_loss_sum(pred2, caps_s[sort2], dl)       # targets: the ORIGINAL captions in both runs
    (ls2 / n2).backward()
    gmax = max(float(g.abs().max()) for g in g_ss.values())
    for k, p in m.named_parameters():
        if k.endswith("full_att.bias"):
            continue
        err = float((g_ss[k] - p.grad).abs().max()) / max(float(g_ss[k].abs().max()), 1e-6 * gmax)
        assert err < 1e-3, (k, err)
    # train mode
    monkeypatch.setattr(editnet, "_XE_SEQUENCE", True)
    m2 = _build(V, D, A, F).train()
    pred, caps_s, dl, _ = m2(X, caps, clen, prev, plen, True, 0.5)
    ls, n, _, _ = xe_loss_sum(pred, caps_s, dl)
    (ls / n).backward()
    assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in m2.parameters())


def test_frozen_parameters_get_no_gradient(monkeypatch):
    """requires_grad = False on some parameters: the node must not write their .grad (an optimizer would otherwise
    update a frozen weight) and must still produce the others"""
    from show_edit_tell_amd import editnet
    from show_edit_tell_amd.autograd_ops import deferred_param_grads
    from show_edit_tell_amd.train import xe_loss_sum
    monkeypatch.setattr(editnet, "_XE_SEQUENCE", True)
    m = _build(203, 64, 32, 256).train()
    frozen = [m.fc.weight, m.attention_lstm.weight_ih, m.caption_attention.cap_full_att.weight, m.copy_lstm.x2h.bias]
    for p in frozen:
        p.requires_grad_(False)
    X, caps, clen, prev, plen = _inputs(6, 36, 256, 20, 203, 8)
    pred, caps_s, dl, _ = m(X, caps, clen, prev, plen, False, 0.0)
    ls, n, _, _ = xe_loss_sum(pred, caps_s, dl)
    with deferred_param_grads():
        (ls / n).backward()
    assert all(p.grad is None for p in frozen)
    others = [p for p in m.parameters() if p.requires_grad]
    assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in others)


def test_fused_embedding_dropout_equals_two_kernels():
    from show_edit_tell_amd import _lib
    lib, dev = _lib.load(), _dev()
    st = _lib.stream_of(dev)
    V, D, n = 57, 64, 33
    g = torch.Generator(device="cpu").manual_seed(2)
    table = torch.randn(V, D, generator=g).to(dev)
    ids = torch.randint(0, V, (n, 3), generator=g).to(dev)
    a = torch.zeros(n, 2 * D, device=dev)
    b = torch.zeros(n, 2 * D, device=dev)
    _lib.check(lib.set_embed_relu_dropout_f32(table.data_ptr(), ids[:, 1].data_ptr(), 3, a[:, D:].data_ptr(), 2 * D, n, D, V, 0.5, 9,
                                              (1 << 40) | 4, st), "set_embed_relu_dropout_f32")
    _lib.check(lib.set_embed_relu_f32(table.data_ptr(), ids[:, 1].data_ptr(), 3, b[:, D:].data_ptr(), 2 * D, n, D, V, st),
               "set_embed_relu_f32")
    _lib.check(lib.set_dropout_f32(b[:, D:].data_ptr(), 2 * D, b[:, D:].data_ptr(), 2 * D, n, D, 0.5, 9, (1 << 40) | 4, st),
               "set_dropout_f32")
    assert torch.equal(a, b) and float(a[:, :D].abs().max()) == 0.0
    assert 0.15 < float((a[:, D:] != 0).float().mean()) < 0.35          # relu keeps ~half, dropout half of that


@pytest.mark.parametrize("min_len", [5, 20])
def test_all_timestep_region_projection_equals_per_step(min_len, monkeypatch):
    """train mode: att1(t) = features_att(dropout_t(.)) contracted for all timesteps at once before the loop, or per
    timestep over the live rows only — same Philox masks, same scores, same gradients (ragged and uniform batches)"""
    from show_edit_tell_amd import editnet, xe_sequence
    from show_edit_tell_amd.train import xe_loss_sum
    monkeypatch.setattr(editnet, "_XE_SEQUENCE", True)
    m = _build(203, 64, 32, 256).train()
    X, caps, clen, prev, plen = _inputs(8, 36, 256, 20, 203, min_len)
    res = []
    for live in (0.0, 2.0):                          # always hoisted / never hoisted
        monkeypatch.setattr(xe_sequence, "_ATT1_HOIST_LIVE", live)
        m.zero_grad(set_to_none=True)
        torch.manual_seed(17)
        pred, caps_s, dl, _ = m(X, caps, clen, prev, plen, False, 0.0)
        ls, n, _, _ = xe_loss_sum(pred, caps_s, dl)
        (ls / n).backward()
        res.append((pred.detach().clone(), {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}))
    (p0, g0), (p1, g1) = res
    assert float((p0 - p1).abs().max()) <= 2e-5 * max(1.0, float(p0.abs().max()))
    gmax = max(float(g.abs().max()) for g in g0.values())
    for k in g0:
        assert float((g0[k] - g1[k]).abs().max()) <= 1e-4 * max(float(g0[k].abs().max()), 1e-3 * gmax), k


def test_column_block_weight_gradient():
    """dW assembled from column blocks with different contraction lengths (per-timestep input over T*B rows, loop-invariant
    input over B rows against sum_t dY): written into .grad in place (first call creates it, second accumulates), returned
    for non-leaf tensors"""
    from show_edit_tell_amd import autograd_ops as A
    dev = _dev()
    g = torch.Generator().manual_seed(3)
    T, B, N = 5, 8, 64
    dy = torch.randn(T * B, N, generator=g).to(dev)
    sdy = dy.view(T, B, N).sum(0)
    x0, x1, x2 = (torch.randn(r, k, generator=g).to(dev) for r, k in ((T * B, 32), (B, 16), (T * B, 24)))
    ref = torch.cat([dy.t() @ x0, sdy.t() @ x1, dy.t() @ x2], 1)
    blocks = [(dy, x0, 0), (sdy, x1, 32), (dy, x2, 48)]
    w = torch.nn.Parameter(torch.zeros(N, 72, device=dev))
    assert A._wgrad_blocks(w, blocks) is None and torch.allclose(w.grad, ref, rtol=1e-4, atol=1e-5)
    A._wgrad_blocks(w, blocks)
    assert torch.allclose(w.grad, 2 * ref, rtol=1e-4, atol=1e-5)
    out = A._wgrad_blocks(torch.zeros(N, 72, device=dev), blocks)          # not a leaf parameter: returned
    assert torch.allclose(out, ref, rtol=1e-4, atol=1e-5)
