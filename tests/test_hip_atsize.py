"""GPU: the B = 64 / B = 128 configurations AT SIZE against fixtures the reference itself produced
(oracle/make_atsize_golden.py -> tests/golden/atsize_*.npz): adaptive features at batch 64 with 10-100 valid regions
(BASELINE.json configs[3]), DCNet at the metric batch 128, the gradients of the adaptive model at full dimensions, and the
self-critical rollout at batch 64 x 5 samples (configs[4]) against the numpy oracle's log-softmax at the drawn words."""
import numpy as np
import pytest
import torch

import parity
from hip_adapter import adaptive_module, dcnet_modules, editnet_modules, to_dev
from test_hip_train import _check_grads

pytestmark = pytest.mark.gpu


def _np(t):
    return t.detach().cpu().numpy()


def test_adaptive_b64_forward_vs_reference():
    """editnet_adaptive.py:489-562 at B = 64, R = 100 (10 ... 100 valid regions per image), full dimensions, fused
    no-grad loop: scores, gd_final_hidden and decoder_last_hidden per original sample"""
    name = "editnet_adaptive_full_b64"
    d, dec = adaptive_module(name)
    c, g = d["case"], parity.load("atsize_" + name)
    assert sorted(set(g["nvalid"].tolist()))[0] == 10 and g["nvalid"].max() == 100 and len(set(g["nvalid"].tolist())) > 30
    with torch.no_grad():
        for _ in range(2):                   # the second call runs with the folded token table
            pred, caps_s, dl, sort_ind, gd_fh, last_h = dec(to_dev(d["X"]), to_dev(d["image_mean"]), to_dev(d["caps"]),
                                                            to_dev(d["clen"]), to_dev(d["prev"]), to_dev(d["plen"]),
                                                            False, 0.0)
            parity.check_xe(_np(pred), dl, _np(sort_ind), g, c["V"], small=False)
            g_inv, inv = parity.unsort(g["xe_sort_ind"]), parity.unsort(_np(sort_ind))
            parity.assert_close(_np(gd_fh)[inv][:, :32], g["xe_gd_final_slice"][g_inv], parity.STATE_TOL, "gd_final_hidden")
            parity.assert_close(_np(last_h)[inv][:, :32], g["xe_last_hidden_slice"][g_inv], 5e-5, "decoder_last_hidden")
            parity.assert_close(_np(last_h).astype(np.float64).sum(1)[inv], g["xe_last_hidden_sum"][g_inv], parity.SUM_TOL,
                                "decoder_last_hidden sums")


@pytest.mark.parametrize("name", ["editnet_adaptive_full_b64", "editnet_adaptive_full_b4"])
def test_adaptive_full_gradients_vs_reference_autograd(name):
    """CE + MSE(decoder_last_hidden, gd_final_hidden) (editnet_adaptive.py:594-596) at full dimensions: loss, every
    parameter's gradient norm and a 64-element slice of every gradient against the reference's autograd"""
    from show_edit_tell_amd.autograd_ops import deferred_param_grads
    from show_edit_tell_amd.train import xe_loss_sum
    d, xe = adaptive_module(name)
    g = parity.load("atsize_" + name)
    xe.eval()
    pred, caps_s, dl, sort_ind, gd_fh, last_h = xe(to_dev(d["X"]), to_dev(d["image_mean"]), to_dev(d["caps"]),
                                                   to_dev(d["clen"]), to_dev(d["prev"]), to_dev(d["plen"]), False, 0.0)
    loss_sum, n_tok, _, _ = xe_loss_sum(pred, caps_s, dl)
    ce = loss_sum / n_tok
    assert abs(float(ce.detach()) - float(g["xe_loss"])) < 1e-4
    loss = ce + torch.nn.functional.mse_loss(last_h, gd_fh)
    assert abs(float(loss.detach()) - float(g["grad_loss"])) < 1e-4
    with deferred_param_grads():
        loss.backward()
    _check_grads(xe, g, name)


def test_dcnet_b128_vs_reference():
    """DCNet (dcnet.py:303-350, dcnet_rl.py:286-346) at the metric batch: encoder, teacher-forced scores, greedy decode
    (bit-exact ids on every row without a near-tie), all 27 gradients"""
    from show_edit_tell_amd.autograd_ops import deferred_param_grads
    from show_edit_tell_amd.train import xe_loss_sum
    name = "dcnet_full_b128"
    d, xe, rl = dcnet_modules(name)
    c, g = d["case"], parity.load("atsize_" + name)
    prev, plen, caps, clen = to_dev(d["prev"]), to_dev(d["plen"]), to_dev(d["caps"]), to_dev(d["clen"])
    with torch.no_grad():
        enc, fh, mask = xe.caption_encoder(prev, plen)
        parity.assert_close(_np(enc).astype(np.float64).sum(2), g["enc_out_sum"], parity.SUM_TOL, "dcnet encoder sums")
        parity.assert_close(_np(fh)[:, :32], g["enc_final_slice"], parity.STATE_TOL, "dcnet final_hidden")
        assert np.array_equal(_np(mask), g["enc_mask"])
        for _ in range(2):                   # second pass: token table
            pred, caps_s, dl, sort_ind = xe(caps, clen, prev, plen)
            parity.check_xe(_np(pred), dl, _np(sort_ind), g, c["V"], small=False)
            seq, logp = rl(d["wm"], prev, plen, True, False)
            namb = parity.check_greedy(_np(seq), _np(logp), g)
    print("dcnet b128 near-tie rows:", namb)
    xe.eval()
    pred, caps_s, dl, _ = xe(caps, clen, prev, plen)
    loss_sum, n_tok, _, _ = xe_loss_sum(pred, caps_s, dl)
    loss = loss_sum / n_tok
    assert abs(float(loss.detach()) - float(g["grad_loss"])) < 1e-4
    with deferred_param_grads():
        loss.backward()
    _check_grads(xe, g, name)


def test_scst_rollout_b64_x5_logprobs_vs_oracle():
    """BASELINE.json configs[4] shape: 64 images x 5 sampled rollouts = 320 rows in ONE rollout (train.scst_train_step's
    layout), full dimensions, eval-mode dropout so the oracle can follow: every stored log-prob equals the numpy oracle's
    log-softmax at the drawn word when the oracle is fed the same words; reproducible per seed; rows of one image differ"""
    from oracle import cases, editnet_np as EN, philox_np as PH
    from show_edit_tell_amd import rng, synth
    from show_edit_tell_amd.train import _repeater
    d, xe, rl = editnet_modules("editnet_full_b128")
    wm, n_img, n_s = d["wm"], 64, 5
    rep = _repeater(n_s)
    X, prev, plen = d["X"][:n_img], d["prev"][:n_img], d["plen"][:n_img]
    rl.eval()
    with rng.dropout_seed(0x5C57_0000_0064_0005):
        seq, logp = rl(wm, rep(to_dev(prev)), rep(to_dev(plen)), rep(to_dev(X)), sample_max=False, sample_rl=True)
    assert logp.requires_grad                 # the grad-enabled route SCST trains through (sequence node, rollout mode)
    with rng.dropout_seed(0x5C57_0000_0064_0005), torch.no_grad():
        seq2, logp2 = rl(wm, rep(to_dev(prev)), rep(to_dev(plen)), rep(to_dev(X)), sample_max=False, sample_rl=True)
    seq, logp = _np(seq), _np(logp)
    B = n_img * n_s
    assert seq.shape == (B, 18) and (logp <= 0).all()
    per_image = seq.reshape(n_s, n_img, 18)
    assert sum(int(not np.array_equal(per_image[0, i], per_image[1, i])) for i in range(n_img)) > n_img // 2
    P = EN.cast_params(d["sd"])
    rp = lambda a: np.concatenate([a] * n_s, 0)
    S = EN.SeqState(P, rp(X), rp(prev), rp(plen))
    words = np.full((B,), wm["<start>"], np.int64)
    live = np.ones(B, bool)
    checked = 0
    draw_margin, draw_alt = np.full((18, B), np.inf), np.zeros((18, B), np.int64)
    seed, off = 0x5C57_0000_0064_0005, PH.site_offset(PH.SITE_ROLLOUT)
    for t in range(18):
        lg32 = EN.step(S, words, B)
        # the draw itself: the device's word is the oracle's inverse-CDF draw of counter (row, t, offset) wherever the target
        # clears the CDF boundaries (the loop rewrites <end> to 0, editnet_rl.py:531)
        ids_o, mg, alt = PH.categorical_draw(lg32, seed, off, t, with_alt=True)
        ids_o = np.where(ids_o == wm["<end>"], 0, ids_o)
        clear = live & (mg > 1e-5)
        assert np.array_equal(seq[clear, t], ids_o[clear]), "sampled words differ from the oracle's draws at step %d" % t
        draw_margin[t, live] = mg[live]
        draw_alt[t] = np.where(alt == wm["<end>"], 0, alt)
        lg = lg32.astype(np.float64)
        m = lg.max(1, keepdims=True)
        lsm = lg - (m + np.log(np.exp(lg - m).sum(1, keepdims=True)))
        w = seq[:, t]
        pos = live & (w > 0)
        assert np.abs(lsm[pos, w[pos]] - logp[pos, t]).max() < 1e-4
        checked += int(pos.sum())
        ended = live & (w == 0)                 # the row drew <end> (or <pad>): either explains the stored log-prob
        if ended.any():
            e = np.minimum(np.abs(lsm[ended, wm["<end>"]] - logp[ended, t]), np.abs(lsm[ended, 0] - logp[ended, t]))
            assert e.max() < 1e-4
        live &= w > 0
        words = w.copy()
        if not live.any():
            break
    assert checked > B * 10
    # the fused no-grad loop draws from the same Philox stream; its scores differ in the last bits (token-table folding).  A
    # row may differ between the two routes only where the oracle shows its draw within 1e-5 of the mass from a CDF boundary,
    # and then by the neighbouring word (tests/parity.py check_sampled_paths_rows) — a handful of the 320 rows at most
    n_diff = parity.check_sampled_paths_rows(seq, logp, _np(seq2), _np(logp2), draw_margin, draw_alt)
    assert n_diff <= 16, n_diff
