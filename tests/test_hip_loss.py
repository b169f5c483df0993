"""Token cross-entropy on the library's kernels (csrc/loss.hip) against the reference's own formulation
(editnet.py:571-577: pack_padded_sequence of scores and targets + CrossEntropyLoss), and the row-padded gradient
layout its backward hands to the fc contractions."""
import pytest
import torch
from torch.nn.utils.rnn import pack_padded_sequence

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device("cuda:0")


def _reference(scores, caps, lens):
    sc = pack_padded_sequence(scores, lens, batch_first=True).data
    tg = pack_padded_sequence(caps[:, 1:], lens, batch_first=True).data
    return torch.nn.functional.cross_entropy(sc, tg, reduction="sum"), sc.shape[0]


@pytest.mark.parametrize("B,T,V,lens,time_major", [
    (6, 7, 203, [7, 7, 5, 3, 1, 1], True), (4, 5, 1000, [5, 5, 5, 5], True), (9, 19, 9490, [19] * 4 + [12] * 3 + [2] * 2, False),
    (3, 4, 6, [4, 2, 1], False)])
def test_loss_and_gradient_match_pack_padded_cross_entropy(B, T, V, lens, time_major):
    from show_edit_tell_amd.train import xe_loss_sum
    g = torch.Generator().manual_seed(B * T + V)
    base = (torch.randn(T, B, V, generator=g) * 3.0).to(_dev())
    caps = torch.randint(0, V, (B, T + 3), generator=g).to(_dev())
    outs = []
    for fused in (False, True):
        leaf = base.clone().requires_grad_(True)
        scores = leaf.transpose(0, 1) if time_major else leaf.transpose(0, 1).contiguous()
        if fused:
            ls, n, _, _ = xe_loss_sum(scores, caps, lens)
        else:
            ls, n = _reference(scores, caps, lens)
        (ls * 0.37).backward()
        outs.append((float(ls.detach()), n, leaf.grad.clone()))
    (l0, n0, g0), (l1, n1, g1) = outs
    assert n0 == n1 == sum(lens)
    assert abs(l0 - l1) <= 2e-6 * abs(l0) * max(1, n0 ** 0.5)
    assert torch.allclose(g0, g1, rtol=1e-5, atol=1e-7)
    for b, l in enumerate(lens):                       # rows beyond a caption's length carry no gradient
        assert float(g1[l:, b].abs().max()) == 0.0 if l < T else True


def test_padded_gradient_feeds_the_contractions_in_place():
    """the (B, T, V) gradient the loss returns lives in a (T, B, V4) buffer: dX, dW and db of fc read it without a copy
    (a ragged row count M = V of dW = dY^T X is masked inside the kernel) and agree with torch"""
    from show_edit_tell_amd import autograd_ops as A
    from show_edit_tell_amd.train import xe_loss_sum
    B, T, V, D = 8, 10, 203, 64
    g = torch.Generator().manual_seed(5)
    scores = (torch.randn(B, T, V, generator=g)).to(_dev()).requires_grad_(True)
    caps = torch.randint(0, V, (B, T + 1), generator=g).to(_dev())
    ls, n, _, _ = xe_loss_sum(scores, caps, [T] * B)
    dpred, = torch.autograd.grad(ls, scores)
    assert dpred.shape == (B, T, V) and dpred.stride() == (V + 1, B * (V + 1), 1)
    rows = A.score_grad_rows(dpred, [B] * T, True)
    assert rows.data_ptr() == dpred.data_ptr() and rows.stride() == (V + 1, 1)       # no copy
    ref = dpred.transpose(0, 1).reshape(T * B, V)
    w = torch.randn(V, D, generator=g).to(_dev())
    x = torch.randn(T * B, D, generator=g).to(_dev())
    assert torch.allclose(A._dgrad(rows, w), ref @ w, rtol=1e-4, atol=1e-5)
    assert torch.allclose(A._wgrad_mm(rows, x), ref.t() @ x, rtol=1e-4, atol=1e-5)
    acc = torch.ones(V, D, device=_dev())
    A._wgrad_mm(rows, x, out=acc)
    assert torch.allclose(acc, 1.0 + ref.t() @ x, rtol=1e-4, atol=1e-5)
    assert torch.allclose(A._colsum(rows), ref.sum(0), rtol=1e-4, atol=1e-6)


def test_out_of_range_target_poisons_the_loss():
    """a label outside [0, V) must not train silently on a clamped word (F.cross_entropy raises): the fused loss and its
    gradient row become NaN"""
    from show_edit_tell_amd.train import xe_loss_sum
    B, T, V = 4, 5, 203
    g = torch.Generator().manual_seed(1)
    scores = torch.randn(B, T, V, generator=g).to(_dev()).requires_grad_(True)
    caps = torch.randint(0, V, (B, T + 1), generator=g).to(_dev())
    caps[1, 2] = V + 7
    ls, n, _, _ = xe_loss_sum(scores, caps, [5, 4, 3, 2])
    assert torch.isnan(ls)
    ls.backward()
    gr = scores.grad
    assert torch.isnan(gr[1, 1]).all() and torch.isfinite(gr[0]).all() and torch.isfinite(gr[1, 0]).all()
