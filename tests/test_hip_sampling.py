"""GPU: the HIP multinomial sampling epilogue (csrc/epilogue.hip `sample_pick_k`, rows a8 / f3):
device RNG against the oracle's Philox, distribution of the draws (chi-square against the softmax of the
logits), the gathered log-prob / log-sum-exp, the <end> -> 0 / `unfinished` / early-break bookkeeping of
editnet_rl.py:529-547, its backward, and the fused sampled rollouts."""
import ctypes as C

import numpy as np
import pytest
import torch

from hip_adapter import dcnet_modules, editnet_modules, to_dev
from oracle import philox_np

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _lib():
    from show_edit_tell_amd import _lib
    return _lib, _lib.load()


def _pick(logits, t, max_len, end_idx, seed, offset, state=None):
    """one call of set_sample_pick_f32; returns (state, raw_ids, lse, step_logp) as numpy"""
    L, lib = _lib()
    B, V = logits.shape
    if state is None:
        state = dict(seq=torch.zeros(B, max_len, dtype=torch.long, device=DEV),
                     it=torch.zeros(B, dtype=torch.long, device=DEV),
                     unf=torch.zeros(B, dtype=torch.int32, device=DEV),
                     alive=torch.zeros(max_len + 2, dtype=torch.int32, device=DEV))
    raw = torch.empty(B, dtype=torch.long, device=DEV)
    lse = torch.empty(B, dtype=torch.float32, device=DEV)
    lp = torch.empty(B, dtype=torch.float32, device=DEV)
    L.check(lib.set_sample_pick_f32(L.ptr(logits), logits.stride(0), B, V, t, max_len, end_idx, seed, offset,
                                    L.ptr(state["seq"]), L.ptr(state["it"]), L.ptr(state["unf"]), L.ptr(state["alive"]),
                                    L.ptr(raw), L.ptr(lse), L.ptr(lp), L.stream_of(torch.device(DEV))), "set_sample_pick_f32")
    torch.cuda.synchronize()
    return state, raw.cpu().numpy(), lse.cpu().numpy(), lp.cpu().numpy()


def test_device_philox_matches_oracle():
    L, lib = _lib()
    n, seed, offset = 1000, 0x0123456789ABCDEF, 0x1122334455
    out = torch.empty(n, 4, dtype=torch.int32, device=DEV)
    L.check(lib.set_philox4x32(L.ptr(out), n, seed, offset, L.stream_of(torch.device(DEV))), "set_philox4x32")
    got = out.cpu().numpy().view(np.uint32)
    ctr = np.stack([np.arange(n), np.zeros(n), np.full(n, offset & 0xFFFFFFFF), np.full(n, offset >> 32)], 1).astype(np.uint32)
    want = philox_np.philox4x32_10(ctr, (seed & 0xFFFFFFFF, seed >> 32))
    assert np.array_equal(got, want)


@pytest.mark.parametrize("V,pad", [(203, 1), (10000, 0), (9490, 0), (13001, 3)])
def test_sampling_distribution_chi_square(V, pad):
    """>= 1e5 draws from one logit row: chi-square of the observed counts against softmax(logits) (bins merged to
    an expected count >= 8), p-value > 1e-4; covers the register path (V % 4 == 0, <= 12288 words) and the generic
    path (odd leading stride / long rows).  Also the exact inverse-CDF identity against the oracle's uniforms."""
    from scipy import stats
    rng = np.random.default_rng(V)
    row = (rng.standard_normal(V) * 2.0).astype(np.float32)
    row[rng.integers(0, V, 5)] += 4.0                        # a few heavy words
    B, reps = 8192, 13                                       # 106 496 draws
    logits = torch.from_numpy(np.repeat(row[None], B, 0).copy())
    if pad:                                                  # odd leading stride -> generic (non-register) kernel
        buf = torch.zeros(B, V + pad)
        buf[:, :V] = logits
        logits = buf.to(DEV)[:, :V]
    else:
        logits = logits.to(DEV)
    p = np.exp(row.astype(np.float64) - row.max())
    p /= p.sum()
    counts = np.zeros(V, np.int64)
    for r in range(reps):
        _, raw, lse, lp = _pick(logits, 0, 18, V - 1, 777, r)
        assert raw.min() >= 0 and raw.max() < V
        counts += np.bincount(raw, minlength=V)
        # log-prob of the drawn word and the row's log-sum-exp, against fp64
        ref_lse = row.max() + np.log(np.exp(row.astype(np.float64) - row.max()).sum())
        assert np.abs(lse - ref_lse).max() < 2e-5
        assert np.abs(lp - (row[raw].astype(np.float64) - ref_lse)).max() < 2e-5
    n = counts.sum()
    order = np.argsort(-p)
    exp_sorted, obs_sorted = p[order] * n, counts[order]
    bins_e, bins_o, ce, co = [], [], 0.0, 0
    for e, o in zip(exp_sorted, obs_sorted):
        ce += e
        co += o
        if ce >= 8.0:
            bins_e.append(ce)
            bins_o.append(co)
            ce, co = 0.0, 0
    if ce > 0:
        bins_e[-1] += ce
        bins_o[-1] += co
    chi2 = float((((np.array(bins_o) - np.array(bins_e)) ** 2) / np.array(bins_e)).sum())
    pval = float(stats.chi2.sf(chi2, len(bins_e) - 1))
    assert pval > 1e-4, (V, chi2, len(bins_e), pval)
    # a never-drawn word must have negligible probability
    assert p[counts == 0].max(initial=0.0) * n < 25


def test_sampling_is_reproducible_and_streams_differ():
    V, B = 1000, 512
    logits = torch.randn(B, V, device=DEV)
    st, a, _, _ = _pick(logits, 0, 18, V - 1, 42, 0)
    _, b, _, _ = _pick(logits, 0, 18, V - 1, 42, 0)
    _, c, _, _ = _pick(logits, 0, 18, V - 1, 42, 1)             # another offset = another stream
    _, e, _, _ = _pick(logits, 0, 18, V - 1, 43, 0)             # another seed
    _, d, _, _ = _pick(logits, 1, 18, V - 1, 42, 0, st)         # the next timestep of the same rollout
    assert np.array_equal(a, b) and a.min() >= 0
    assert (a != c).mean() > 0.9 and (a != d).mean() > 0.9 and (a != e).mean() > 0.9


def test_sampling_bookkeeping_matches_reference_loop():
    """<end> -> 0, `unfinished` latch, seq stores and the early `break` (editnet_rl.py:529-547), emulated in numpy
    from the raw draws."""
    V, B, max_len = 50, 64, 6
    end = V - 1
    rng = np.random.default_rng(5)
    state = None
    unf = None
    seq_ref = np.zeros((B, max_len), np.int64)
    broken_at = None
    for t in range(max_len):
        lg = rng.standard_normal((B, V)).astype(np.float32)
        lg[:, end] += 1.5 + (6.0 if t >= 2 else 0.0)          # most rows finish early; by t = 3 typically all have
        if t == 3:
            lg[:, end] += 50.0                                # force: every remaining row emits <end> here
        state, raw, lse, lp = _pick(to_dev(lg), t, max_len, end, 99, 7, state)
        if broken_at is not None:
            assert (raw == -1).all() and (lp == 0).all()
            continue
        it = raw.copy()
        it[it == end] = 0
        unf = (it > 0) if t == 0 else (unf & (it > 0))
        it = it * unf
        seq_ref[:, t] = it
        assert np.array_equal(state["it"].cpu().numpy(), it)
        assert np.array_equal(state["unf"].cpu().numpy().astype(bool), unf)
        assert int(state["alive"][t]) == int(unf.sum())
        if unf.sum() == 0:
            broken_at = t
    assert broken_at is not None and broken_at < max_len - 1
    assert np.array_equal(state["seq"].cpu().numpy(), seq_ref)


def test_sample_logp_backward_matches_autograd():
    from show_edit_tell_amd import autograd_ops as A
    V, B = 203, 9
    torch.manual_seed(3)
    logits = torch.randn(B, V, device=DEV, requires_grad=True)
    st = A.SampleState(B, 18, V - 2, V - 1, torch.device(DEV), seed=5)
    lp = A.sample_pick(logits, st, 0)
    w = torch.randn(B, device=DEV)
    (lp * w).sum().backward()
    raw_like = None
    # rebuild with torch: same ids via log_softmax gather
    lg2 = logits.detach().clone().requires_grad_(True)
    ls = torch.log_softmax(lg2, 1)
    ids = (ls.detach() - lp.detach().unsqueeze(1)).abs().argmin(1)       # the drawn word: its log-prob equals lp
    assert torch.allclose(ls.gather(1, ids.unsqueeze(1)).squeeze(1), lp.detach(), atol=1e-5)
    (ls.gather(1, ids.unsqueeze(1)).squeeze(1) * w).sum().backward()
    assert torch.allclose(logits.grad, lg2.grad, atol=1e-6), float((logits.grad - lg2.grad).abs().max())
    del raw_like


def _np(t):
    return t.detach().cpu().numpy()


def test_fused_sampled_rollout_editnet_logprobs_vs_oracle():
    """set_editnet_sample (eval, no grad): reproducible for a seed; the log-probs of the drawn words equal the
    oracle's log-softmax when the oracle is fed the same words (tolerance 1e-4)."""
    from oracle import editnet_np as EN
    d, xe, rl = editnet_modules("editnet_small")
    wm = d["wm"]
    prev, plen, X = to_dev(d["prev"]), to_dev(d["plen"]), to_dev(d["X"])
    rl.eval()
    with torch.no_grad():
        torch.manual_seed(11)
        seq, logp = rl(wm, prev, plen, X, sample_max=False, sample_rl=True)
        torch.manual_seed(11)
        seq2, logp2 = rl(wm, prev, plen, X, sample_max=False, sample_rl=True)
        torch.manual_seed(12)
        seq3, _ = rl(wm, prev, plen, X, sample_max=False, sample_rl=True)
    seq, logp = _np(seq), _np(logp)
    # (the second call runs with the folded token table, i.e. another summation order: log-probs agree to rounding)
    assert np.array_equal(seq, _np(seq2)) and np.abs(logp - _np(logp2)).max() < 1e-5
    assert not np.array_equal(seq, _np(seq3))
    assert seq.shape == (X.shape[0], 18) and (logp <= 0).all()
    P = EN.cast_params(d["sd"])
    S = EN.SeqState(P, d["X"], d["prev"], d["plen"])
    B = seq.shape[0]
    words = np.full((B,), wm["<start>"], np.int64)
    live = np.ones(B, bool)
    checked = 0
    for t in range(18):
        lg = EN.step(S, words, B)
        lsm = lg.astype(np.float64) - (lg.max(1, keepdims=True) + np.log(np.exp(lg.astype(np.float64) - lg.max(1, keepdims=True)).sum(1, keepdims=True)))
        for b in range(B):
            if not live[b]:
                continue
            w = int(seq[b, t])
            if w > 0:
                assert abs(lsm[b, w] - logp[b, t]) < 1e-4, (b, t, lsm[b, w], logp[b, t])
                checked += 1
            else:                                   # the row drew <end> (or <pad>): either explains the stored log-prob
                assert min(abs(lsm[b, wm["<end>"]] - logp[b, t]), abs(lsm[b, 0] - logp[b, t])) < 1e-4
                live[b] = False
        words = seq[:, t].copy()
        if not live.any():
            break
    assert checked >= B


def test_sampled_rollouts_train_mode_have_no_host_sync_path_and_gradients():
    """grad-enabled sampled rollout (SCST, editnet_rl.py:670 / dcnet_rl.py): sampling through set_sample_pick_f32,
    gradients reach every parameter, log-probs consistent with seq (0 after <end>), DCNet twin included."""
    from show_edit_tell_amd.editnet_rl import RewardCriterion
    d, xe, rl = editnet_modules("editnet_small")
    wm = d["wm"]
    prev, plen, X = to_dev(d["prev"]), to_dev(d["plen"]), to_dev(d["X"])
    rl.train()
    torch.manual_seed(2)
    seq, logp = rl(wm, prev, plen, X, sample_max=False, sample_rl=True)
    assert logp.requires_grad and seq.shape == (X.shape[0], 18)
    assert torch.isfinite(logp).all() and (logp <= 0).all()
    loss = RewardCriterion()(logp, seq, torch.ones_like(logp))
    rl.zero_grad()
    loss.backward()
    for k, p in rl.named_parameters():
        if "caption_encoder.embed" in k:
            continue
        assert p.grad is not None and torch.isfinite(p.grad).all(), k
    dd, dxe, drl = dcnet_modules("dcnet_small")
    drl.train()
    s2, l2 = drl(dd["wm"], to_dev(dd["prev"]), to_dev(dd["plen"]), False, True)
    assert l2.requires_grad and torch.isfinite(l2).all()
    with torch.no_grad():
        drl.eval()
        torch.manual_seed(1)
        a, _ = drl(dd["wm"], to_dev(dd["prev"]), to_dev(dd["plen"]), False, True)
        torch.manual_seed(1)
        b, _ = drl(dd["wm"], to_dev(dd["prev"]), to_dev(dd["plen"]), False, True)
    assert torch.equal(a, b)
