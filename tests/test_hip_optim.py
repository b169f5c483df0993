"""clip_grad_norm_ + Adam on the library's two-launch kernels (csrc/optim.hip) against torch's own
clip_grad_norm_ + torch.optim.Adam (editnet.py:580-581) on the same parameters and gradients."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device("cuda:0")


def _make(shapes, seed, lr=4e-4, wd=0.0):
    g = torch.Generator().manual_seed(seed)
    ps = [torch.nn.Parameter(torch.randn(*s, generator=g).to(_dev())) for s in shapes]
    groups = [{"params": ps[:2], "lr": lr}, {"params": ps[2:], "lr": 2.5 * lr, "betas": (0.8, 0.99), "weight_decay": wd}]
    return ps, torch.optim.Adam(groups, lr=lr)


@pytest.mark.parametrize("scale_grads", [False, True])
@pytest.mark.parametrize("max_norm", [0.25, 1e9])
def test_clip_adam_matches_torch(max_norm, scale_grads):
    from show_edit_tell_amd.optim import clip_grad_norm_and_step
    shapes = [(1000, 512), (4096,), (3,), (1,), (37, 53), (16384 * 3 + 5,), (2048, 1024)]
    pa, oa = _make(shapes, 1, wd=0.01)
    pb, ob = _make(shapes, 1, wd=0.01)
    g = torch.Generator().manual_seed(7)
    for it in range(4):
        grads = [torch.randn(*s, generator=g).to(_dev()) * (0.01 if it % 2 else 3.0) for s in shapes]
        for p, q, gr in zip(pa, pb, grads):
            p.grad, q.grad = gr.clone(), gr.clone()
        if it == 2:                        # a parameter without a gradient is skipped by both
            pa[4].grad = pb[4].grad = None
        na = torch.nn.utils.clip_grad_norm_(pa, max_norm)
        oa.step()
        nb = clip_grad_norm_and_step(pb, ob, max_norm, scale_grads=scale_grads)
        assert abs(float(na) - float(nb)) <= 1e-5 * float(na)
        for k, (p, q) in enumerate(zip(pa, pb)):
            err = float((p.detach() - q.detach()).abs().max())
            assert err <= 2e-6 * max(1.0, float(p.abs().max())), (it, k, err)
            if scale_grads and p.grad is not None:
                assert torch.allclose(p.grad, q.grad, rtol=1e-5, atol=1e-9)
    for p, q in zip(pa, pb):               # the optimizer state is torch.optim.Adam's own: same keys, same values
        sa, sb = oa.state[p], ob.state[q]
        assert set(sa) == set(sb) and float(sa["step"]) == float(sb["step"])
        for k in ("exp_avg", "exp_avg_sq"):    # (sums with cancellation: absolute tolerance relative to the largest entry)
            assert torch.allclose(sa[k], sb[k], rtol=1e-5, atol=2e-6 * float(sa[k].abs().max()))
    sd = ob.state_dict()                   # and a checkpoint round trip through torch.optim.Adam keeps stepping
    _, oc = _make(shapes, 1, wd=0.01)
    oc.load_state_dict(sd)


def test_other_optimizers_take_the_torch_calls():
    from show_edit_tell_amd.optim import clip_grad_norm_and_step
    p = torch.nn.Parameter(torch.ones(64, device=_dev()))
    q = torch.nn.Parameter(torch.ones(64, device=_dev()))
    p.grad, q.grad = torch.full_like(p, 2.0), torch.full_like(q, 2.0)
    oa, ob = torch.optim.SGD([p], lr=0.1), torch.optim.SGD([q], lr=0.1)
    torch.nn.utils.clip_grad_norm_([p], 0.25)
    oa.step()
    clip_grad_norm_and_step([q], ob, 0.25)
    assert torch.equal(p, q)


def test_many_tensors_span_several_launches():
    """more tensors than one kernel-argument table holds (40): the norm still covers all of them"""
    from show_edit_tell_amd.optim import clip_grad_norm_and_step
    shapes = [(257,)] * 95
    pa, oa = _make(shapes, 3)
    pb, ob = _make(shapes, 3)
    for p, q in zip(pa, pb):
        p.grad = torch.full_like(p, 0.5)
        q.grad = p.grad.clone()
    na = torch.nn.utils.clip_grad_norm_(pa, 0.25)
    oa.step()
    nb = clip_grad_norm_and_step(pb, ob, 0.25)
    assert abs(float(na) - float(nb)) <= 1e-5 * float(na)
    for p, q in zip(pa, pb):
        assert torch.allclose(p, q, rtol=1e-6, atol=1e-7)
