"""GPU: gradients of the XE loss through the grad-enabled path (HIP forward operators, autograd
backward) against the REFERENCE's own autograd gradients captured in tests/golden (eval-mode
dropout so they are deterministic)."""
import numpy as np
import pytest
import torch

import parity
from hip_adapter import editnet_modules, to_dev

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seq", [True, False], ids=["sequence-node", "per-operator"])
@pytest.mark.parametrize("deferred", [False, True])
@pytest.mark.parametrize("name", ["editnet_small", "editnet_full_b4", "editnet_full_v9490"])
def test_xe_gradients_vs_reference_autograd(name, deferred, seq, monkeypatch):
    """both grad-enabled routes of DecoderC.forward: the whole-sequence node (xe_sequence.py, the default) and the
    per-operator autograd loop it replaces (still used with scheduled sampling / adaptive features)"""
    import contextlib
    from show_edit_tell_amd import editnet
    from show_edit_tell_amd.autograd_ops import deferred_param_grads
    from show_edit_tell_amd.train import xe_loss_sum
    monkeypatch.setattr(editnet, "_XE_SEQUENCE", seq)
    d, xe, rl = editnet_modules(name)
    g = parity.load(name)
    xe.eval()                                         # dropout off; parameters still require grad
    pred, caps_s, dl, sort_ind = xe(to_dev(d["X"]), to_dev(d["caps"]), to_dev(d["clen"]), to_dev(d["prev"]),
                                    to_dev(d["plen"]), False, 0.0)
    assert pred.requires_grad
    loss_sum, n_tok, _, _ = xe_loss_sum(pred, caps_s, dl)
    loss = loss_sum / n_tok
    assert abs(float(loss.detach()) - float(g["grad_loss"])) < 1e-4
    with (deferred_param_grads() if deferred else contextlib.nullcontext()):   # time-batched weight gradients
        loss.backward()
    _check_grads(xe, g, name)


def test_xe_gradients_b128_vs_reference_autograd():
    """BASELINE.json config 1 (EditNet XE forward+backward, batch=128, 36x2048, prev-caption len 20): loss, every
    parameter's gradient norm and a 64-element strided slice of every gradient against the reference's autograd."""
    from show_edit_tell_amd.autograd_ops import deferred_param_grads
    from show_edit_tell_amd.train import xe_loss_sum
    name = "editnet_full_b128"
    d, xe, rl = editnet_modules(name)
    g = parity.load(name)
    xe.eval()
    pred, caps_s, dl, sort_ind = xe(to_dev(d["X"]), to_dev(d["caps"]), to_dev(d["clen"]), to_dev(d["prev"]),
                                    to_dev(d["plen"]), False, 0.0)
    loss_sum, n_tok, _, _ = xe_loss_sum(pred, caps_s, dl)
    loss = loss_sum / n_tok
    assert abs(float(loss.detach()) - float(g["grad_loss"])) < 1e-4
    with deferred_param_grads():
        loss.backward()
    _check_grads(xe, g, name)


@pytest.mark.parametrize("seq", [True, False], ids=["sequence-node", "per-operator"])
@pytest.mark.parametrize("name", ["dcnet_small", "dcnet_full_b4"])
def test_dcnet_xe_gradients_vs_reference_autograd(name, seq, monkeypatch):
    """DCNet (dcnet.py:353-402): all parameter gradients of the XE loss against the reference's autograd — through
    the packed BiLSTM encoder, the additive attention and both LSTM cells; immediate and time-batched weight gradients."""
    import contextlib
    from show_edit_tell_amd import editnet
    monkeypatch.setattr(editnet, "_XE_SEQUENCE", seq)
    from hip_adapter import dcnet_modules
    from show_edit_tell_amd.autograd_ops import deferred_param_grads
    from show_edit_tell_amd.train import xe_loss_sum
    g = parity.load(name)
    for deferred in (False, True):
        d, dae, _ = dcnet_modules(name)
        dae.eval()
        pred, caps_s, dl, _ = dae(to_dev(d["caps"]), to_dev(d["clen"]), to_dev(d["prev"]), to_dev(d["plen"]))
        assert pred.requires_grad
        loss_sum, n_tok, _, _ = xe_loss_sum(pred, caps_s, dl)
        loss = loss_sum / n_tok
        assert abs(float(loss.detach()) - float(g["grad_loss"])) < 1e-4
        with (deferred_param_grads() if deferred else contextlib.nullcontext()):
            loss.backward()
        _check_grads(dae, g, name)


def _check_grads(xe, g, name):
    # absolute floor: gradients that are mathematically zero (softmax shift invariance makes
    # d/d full_att.bias == 0) are pure rounding noise in both implementations
    floor = 1e-6 * max(float(g["gradnorm." + k]) for k, _ in xe.named_parameters())
    worst = 0.0
    for k, p in xe.named_parameters():
        got = p.grad.detach().cpu().numpy()
        gn = float(g["gradnorm." + k])
        mine = float(np.sqrt((got.astype(np.float64) ** 2).sum()))
        assert abs(mine - gn) <= 1e-4 * gn + floor, (k, mine, gn)
        if "grad." + k in g:
            ref = g["grad." + k]
            err = np.abs(got - ref).max()
            scale = max(np.abs(ref).max(), 1e-6)
            worst = max(worst, err / scale)
            assert err <= 1e-4 * scale + floor, (k, err, scale)
        else:
            ref = g["gradslice." + k]
            sl = got.reshape(-1)[:: max(1, got.size // 64)][:64]
            assert np.abs(sl - ref).max() <= 1e-4 * max(np.abs(ref).max(), gn / np.sqrt(got.size)) + floor, k
    print(name, "worst relative gradient error", worst)


def test_train_mode_step_runs_and_learns():
    """train() mode: dropout active, scheduled sampling on; optimizer steps reduce the (eval-mode) loss."""
    from show_edit_tell_amd.train import xe_loss_sum, xe_train_step
    d, xe, rl = editnet_modules("editnet_small")
    opt = torch.optim.Adam(xe.parameters(), lr=2e-3)
    args = (to_dev(d["X"]), to_dev(d["caps"]), to_dev(d["clen"]), to_dev(d["prev"]), to_dev(d["plen"]))

    def eval_loss():
        xe.eval()
        with torch.no_grad():
            pred, caps_s, dl, _ = xe(*args, False, 0.0)
        ls, n, _, _ = xe_loss_sum(pred, caps_s, dl)
        return float(ls) / n

    before = eval_loss()
    torch.manual_seed(0)
    losses = [xe_train_step(xe, opt, *args, use_ss=True, ss_prob=0.25)[0] for _ in range(30)]
    after = eval_loss()
    assert all(np.isfinite(losses))
    assert after < before - 0.2, (before, after)


def test_scst_rollout_pair():
    """SCST step of editnet_rl.py:663-679: greedy baseline (eval, no_grad, fused C path) + sampled
    rollout (train mode, autograd path) + RewardCriterion + backward."""
    from show_edit_tell_amd.editnet_rl import RewardCriterion
    d, xe, rl = editnet_modules("editnet_small")
    g = parity.load("editnet_small")
    wm = d["wm"]
    prev, plen, X = to_dev(d["prev"]), to_dev(d["plen"]), to_dev(d["X"])
    # grad-enabled greedy (eval mode) must reproduce the reference's greedy tokens / logprobs
    rl.eval()
    seq_g, logp_g = rl(wm, prev, plen, X, True, False)
    assert logp_g.requires_grad
    parity.check_greedy(seq_g.cpu().numpy(), logp_g.detach().cpu().numpy(), g)
    # the SCST pair
    with torch.no_grad():
        greedy_res, _ = rl(wm, prev, plen, X, sample_max=True, sample_rl=False)
    assert np.array_equal(greedy_res.cpu().numpy(), g["greedy_seq"])
    rl.train()
    torch.manual_seed(1)
    seq_s, logp_s = rl(wm, prev, plen, X, sample_max=False, sample_rl=True)
    assert seq_s.shape == (X.shape[0], 18) and logp_s.shape == (X.shape[0], 18)
    assert torch.isfinite(logp_s).all() and (logp_s <= 0).all()
    reward = torch.randn(X.shape[0], 1, device=X.device).repeat(1, 18)
    loss = RewardCriterion()(logp_s, seq_s, reward)
    rl.zero_grad()
    loss.backward()
    gn = sum(float(p.grad.norm()) for p in rl.parameters() if p.grad is not None)
    assert np.isfinite(gn) and gn > 0
    # RewardCriterion value against the oracle's restatement
    from oracle import editnet_np as EN
    ref = EN.reward_criterion(logp_s.detach().cpu().numpy().astype(np.float64), seq_s.cpu().numpy(),
                              reward.cpu().numpy().astype(np.float64))
    assert abs(float(loss.detach()) - ref) < 1e-5


def test_scst_train_step_with_ciderd_reward():
    """train.scst_train_step (editnet_rl.py:649-686): the policy gradient with the build-owned CIDEr-D
    reward raises the probability of the rewarded words: when every reference IS the current sample,
    the sampled captions' log-probability must increase over a few steps."""
    from show_edit_tell_amd import ciderd
    from show_edit_tell_amd.train import scst_train_step
    d, xe, rl = editnet_modules("editnet_small")
    wm = d["wm"]
    prev, plen, X = to_dev(d["prev"]), to_dev(d["plen"]), to_dev(d["X"])
    B, V = X.shape[0], len(wm)
    rng = np.random.default_rng(3)
    # references: 5 random captions per image over the real vocabulary, <start> w.. <end> <pad>..
    allcaps = np.zeros((B, 5, 12), dtype=np.int64)
    for b in range(B):
        for j in range(5):
            n = int(rng.integers(3, 9))
            allcaps[b, j, 0] = wm["<start>"]
            allcaps[b, j, 1:1 + n] = rng.integers(1, V - 4, n)
            allcaps[b, j, 1 + n] = wm["<end>"]
    gt = ciderd.ground_truth_lists(allcaps, wm)
    df, docs = ciderd.document_frequency([[ciderd.tokens_to_str(c) for c in caps] for caps in gt])
    scorer = ciderd.CiderD(df, max(docs, 2))
    opt = torch.optim.Adam(rl.parameters(), lr=1e-3)
    before = [p.detach().clone() for p in rl.parameters()]
    torch.manual_seed(4)
    for n_samples in (1, 3):
        reward, loss = scst_train_step(rl, opt, wm, X, prev, plen, gt, scorer, n_samples=n_samples)
        assert np.isfinite(reward) and np.isfinite(loss)
    assert any(not torch.equal(a, p.detach()) for a, p in zip(before, rl.parameters()))
    assert all(torch.isfinite(p).all() for p in rl.parameters())


def test_scst_full_size_five_samples():
    """BASELINE.json configs[4] shape (full-size model, 5 sampled rollouts per image, CIDEr-D reward) at a small batch:
    one self-critical step runs end to end — greedy baseline (fused loop) + 5·B sampled rows (device sampling epilogue)
    + native CIDEr-D + backward + clip + Adam — with finite numbers, and the sampled log-probs are the log-softmax of the
    step's logits at the drawn words (re-scored teacher-forced in eval mode is NOT comparable: dropout is active)."""
    from show_edit_tell_amd import ciderd
    from show_edit_tell_amd.train import scst_train_step
    d, xe, rl = editnet_modules("editnet_full_b4")
    wm = d["wm"]
    prev, plen, X = to_dev(d["prev"]), to_dev(d["plen"]), to_dev(d["X"])
    B, V = X.shape[0], len(wm)
    # references = five captions sampled from the model itself (fused sampled rollout, eval mode): random word
    # salad would share no n-gram with the rollouts and every reward would be exactly zero
    rl.eval()
    allcaps = np.zeros((B, 5, 20), dtype=np.int64)
    with torch.no_grad():
        for j in range(5):
            torch.manual_seed(100 + j)
            sj, _ = rl(wm, prev, plen, X, sample_max=False, sample_rl=True)
            sj = sj.cpu().numpy()
            for b in range(B):
                words = [int(w) for w in sj[b] if w > 0][:17]
                allcaps[b, j, 0] = wm["<start>"]
                allcaps[b, j, 1:1 + len(words)] = words
                allcaps[b, j, 1 + len(words)] = wm["<end>"]
    gt = ciderd.ground_truth_lists(allcaps, wm)
    df, docs = ciderd.document_frequency([[ciderd.tokens_to_str(c) for c in caps] for caps in gt])
    scorer = ciderd.CiderD(df, 50)                       # pretend the table came from a larger corpus (idf > 0)
    opt = torch.optim.Adam(rl.parameters(), lr=5e-5)
    before = rl.fc.weight.detach().clone()
    torch.manual_seed(9)
    reward, loss = scst_train_step(rl, opt, wm, X, prev, plen, gt, scorer, n_samples=5)
    assert np.isfinite(reward) and np.isfinite(loss)
    assert scorer._native, "the reward must come from the native CIDEr-D"
    assert not torch.equal(before, rl.fc.weight.detach())
    assert all(torch.isfinite(p).all() for p in rl.parameters())


def test_rollout_with_image_repeat_equals_explicitly_repeated_features():
    """configs[4]: n sampled rollouts per image.  `repeat_images=n` hands the decoder the B images once — the region
    embedding relu(att_embed(X)) and its weight gradient are contracted for B images, the rows repeated — and must give the
    rollout (same Philox streams: same words) and the gradients of the explicit n-fold copy of the features."""
    from show_edit_tell_amd import rng
    d, xe, rl = editnet_modules("editnet_full_b4")
    wm = d["wm"]
    prev, plen, X = to_dev(d["prev"]), to_dev(d["plen"]), to_dev(d["X"])
    n = 3
    rep = lambda t: t.repeat(n, *([1] * (t.dim() - 1)))
    rl.train()
    outs = []
    for explicit in (True, False):
        rl.zero_grad(set_to_none=True)
        torch.manual_seed(21)
        if explicit:
            seq, logp = rl(wm, rep(prev), rep(plen), rep(X), sample_max=False, sample_rl=True)
        else:
            seq, logp = rl(wm, rep(prev), rep(plen), X, sample_max=False, sample_rl=True, repeat_images=n)
        (logp * (seq > 0).float()).sum().backward()
        outs.append((seq.clone(), logp.detach().clone(), {k: p.grad.detach().clone() for k, p in rl.named_parameters()
                                                           if p.grad is not None}))
    (s0, l0, g0), (s1, l1, g1) = outs
    assert s0.shape[0] == n * X.shape[0]
    # the region embedding is the only operand that is produced differently: the same numbers up to the summation order of
    # another GEMM plan (64 x 36 rows instead of 192 x 36)
    from show_edit_tell_amd import _lib, autograd_ops as A
    va = rl.visual_attention
    with torch.no_grad():
        y1 = A.linear(X, va.att_embed[0].weight, va.att_embed[0].bias, _lib.ACT_RELU).repeat(n, 1, 1)
        y2 = A.linear(rep(X), va.att_embed[0].weight, va.att_embed[0].bias, _lib.ACT_RELU)
    assert float((y1 - y2).abs().max()) < 1e-5
    # a multinomial draw is a step function of the scores: a 1e-6 difference may move one draw across a CDF boundary, and
    # that row's trajectory then differs from there on.  All other rows: same words, same log-probs
    same = (s0 == s1).all(1)
    assert int((~same).sum()) <= 1, "rollouts differ in %d of %d rows" % (int((~same).sum()), s0.shape[0])
    assert float((l0 - l1)[same].abs().max()) < 2e-5
    assert set(g0) == set(g1)
    if bool(same.all()):
        for k in g0:
            scale = float(g0[k].abs().max()) + 1e-12
            assert float((g0[k] - g1[k]).abs().max()) <= 2e-4 * scale + 1e-7, k
    else:
        # (one trajectory differs: its gradient contribution differs too; the weight gradient of the region embedding
        # still has to be a gradient of the same size)
        k = "visual_attention.att_embed.0.weight"
        assert 0.5 < float(g1[k].norm() / g0[k].norm()) < 2.0


@pytest.mark.parametrize("seq", [True, False], ids=["sequence-node", "per-operator"])
def test_adaptive_xe_gradients_vs_reference_autograd(seq, monkeypatch):
    """Adaptive features (10-100 zero-padded regions): gradients of CE + MSE(decoder_last_hidden, gd_final_hidden)
    (adaptive_features/editnet_adaptive.py:584-598) through the masked visual attention, against the reference."""
    from hip_adapter import adaptive_module
    from show_edit_tell_amd import editnet
    from show_edit_tell_amd.autograd_ops import deferred_param_grads
    from show_edit_tell_amd.train import xe_loss_sum
    monkeypatch.setattr(editnet, "_XE_SEQUENCE", seq)
    name = "editnet_adaptive_small"
    d, xe = adaptive_module(name)
    g = parity.load(name)
    xe.eval()
    pred, caps_s, dl, sort_ind, gd_fh, last_h = xe(to_dev(d["X"]), to_dev(d["image_mean"]), to_dev(d["caps"]),
                                                   to_dev(d["clen"]), to_dev(d["prev"]), to_dev(d["plen"]), False, 0.0)
    assert pred.requires_grad and gd_fh.requires_grad and last_h.requires_grad
    assert np.abs(last_h.detach().cpu().numpy() - g["xe_last_hidden"]).max() < 2e-5
    assert np.abs(gd_fh.detach().cpu().numpy() - g["xe_gd_final"]).max() < 2e-5
    loss_sum, n_tok, _, _ = xe_loss_sum(pred, caps_s, dl)
    loss = loss_sum / n_tok + torch.nn.functional.mse_loss(last_h, gd_fh)
    assert abs(float(loss.detach()) - float(g["grad_loss"])) < 1e-4
    with deferred_param_grads():
        loss.backward()
    _check_grads(xe, g, name)
    # train mode (dropout on the region embedding) runs and masks the padded regions
    xe.train()
    xe.zero_grad()
    pred, caps_s, dl, *_ = xe(to_dev(d["X"]), to_dev(d["image_mean"]), to_dev(d["caps"]), to_dev(d["clen"]),
                              to_dev(d["prev"]), to_dev(d["plen"]), True, 0.25)
    ls, n, _, _ = xe_loss_sum(pred, caps_s, dl)
    (ls / n).backward()
    assert all(torch.isfinite(p.grad).all() for p in xe.parameters() if p.grad is not None)
    # train mode without scheduled sampling (the sequence node when enabled): CE + MSE, per-step region masks
    xe.zero_grad()
    pred, caps_s, dl, _, gd_fh, last_h = xe(to_dev(d["X"]), to_dev(d["image_mean"]), to_dev(d["caps"]), to_dev(d["clen"]),
                                            to_dev(d["prev"]), to_dev(d["plen"]), False, 0.0)
    ls, n, _, _ = xe_loss_sum(pred, caps_s, dl)
    (ls / n + torch.nn.functional.mse_loss(last_h, gd_fh)).backward()
    assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in xe.parameters())
