"""CPU: the numpy oracle (oracle/*.py) against the golden vectors captured from the reference.

This is what pins the oracle: every fixture in tests/golden was produced by the
reference's own classes (oracle/make_golden.py) on inputs that are regenerated here from seeds.
"""
import numpy as np
import pytest

import parity
from oracle import cases, dcnet_np as DN, editnet_np as EN


def _editnet_inputs(name):
    d = cases.build_editnet(name)
    return d, d["case"], EN.cast_params(d["sd"]), parity.load(name)


@pytest.mark.parametrize("name", ["editnet_small", "editnet_full_b4", "editnet_full_v9490"])
def test_editnet_ops(name):
    d, c, P, g = _editnet_inputs(name)
    H, M, fh, mask = EN.caption_encoder(P, d["prev"], d["plen"])
    parity.assert_close(H, g["enc_H"], parity.STATE_TOL, "encoder H")
    parity.assert_close(M, g["enc_M"], parity.STATE_TOL, "encoder M")
    parity.assert_close(fh, g["enc_final"], parity.STATE_TOL, "encoder final_hidden")
    assert np.array_equal(mask, g["enc_mask"])
    p = d["probe"]
    emb = EN.embed(P, p["ids"])
    assert np.array_equal(emb, g["op_embed"])
    mean = d["X"].mean(1, dtype=np.float32)
    h1, c1 = EN.lstm_cell(P, "attention_lstm", np.concatenate([emb, fh, p["h2"], mean], 1), p["h1"], p["c1"])
    parity.assert_close(h1, g["op_h1"], parity.STATE_TOL, "attention_lstm h")
    parity.assert_close(c1, g["op_c1"], parity.STATE_TOL, "attention_lstm c")
    gated, alpha_c = EN.caption_attention(P, H, p["h1"], p["word"], mask)
    parity.assert_close(gated, g["op_gated"], parity.STATE_TOL, "caption_attention")
    parity.assert_close(alpha_c, g["op_alpha_c"], parity.STATE_TOL, "alpha_c")
    vis = EN.visual_attention(P, d["X"], p["h1"])
    parity.assert_close(vis, g["op_vis"], parity.STATE_TOL, "visual_attention")
    sel = EN.select(M, alpha_c)
    parity.assert_close(sel, g["op_sel"], parity.STATE_TOL, "select")
    h2, c2 = EN.copy_lstm(P, np.concatenate([p["h1"], gated, vis], 1), p["h2"], p["c2"], sel)
    parity.assert_close(h2, g["op_h2"], parity.STATE_TOL, "copy_lstm h")
    parity.assert_close(c2, g["op_c2"], parity.STATE_TOL, "copy_lstm c")
    logits = EN._linear(h2, P, "fc")
    if "op_logits" in g:
        parity.assert_close(logits, g["op_logits"], parity.LOGIT_TOL, "fc")
    else:
        parity.check_logit_summary(logits, g, "op_logits_", c["V"], what="fc")


@pytest.mark.parametrize("name", ["editnet_small", "editnet_full_b4"])
def test_editnet_xe(name):
    d, c, P, g = _editnet_inputs(name)
    pred, caps_s, dl, sort_ind = EN.xe_forward(P, d["X"], d["caps"], d["clen"], d["prev"], d["plen"])
    parity.check_xe(pred, dl, sort_ind, g, c["V"], small=c["D"] < 1024)
    assert np.array_equal(caps_s, g["xe_caps_sorted"])
    assert abs(EN.xe_loss(pred, caps_s, dl) - float(g["xe_loss"])) < 1e-5


@pytest.mark.parametrize("name", ["editnet_small", "editnet_small_end", "editnet_full_b4", "editnet_full_v9490"])
def test_editnet_greedy(name):
    d, c, P, g = _editnet_inputs(name)
    wm = d["wm"]
    tr = []
    seq, logp = EN.greedy_decode(P, wm["<start>"], wm["<end>"], d["prev"], d["plen"], d["X"], trace=tr)
    parity.check_greedy(seq, logp, g)
    assert len(tr) == int(g["greedy_nsteps"])
    reward = np.repeat(cases.synth.uniform(c["iseed"], "reward", (c["B"], 1), -1, 1), 18, 1)
    assert abs(EN.reward_criterion(logp.astype(np.float64), seq, reward) - float(g["reward_loss"])) < 1e-5
    if "greedy_h2" in g:                       # reduced-dimension case: every intermediate
        for k in ("h1", "c1", "h2", "c2", "attend_cap", "alpha_c", "attend_img", "sel"):
            parity.assert_close(np.stack([s[k] for s in tr]), g["greedy_" + k], parity.STATE_TOL * 5, "greedy " + k)
        parity.assert_close(np.stack([s["logits"] for s in tr]), g["greedy_logits"], parity.LOGIT_TOL, "greedy logits")
    else:
        lg = np.stack([s["logits"] for s in tr])
        S, B, V = lg.shape
        sub = {k: v.reshape(S * B, *v.shape[2:]) for k, v in g.items() if k.startswith("greedy_logits_")}
        parity.check_logit_summary(lg.reshape(S * B, V), sub, "greedy_logits_", V, what="greedy logits")
        for k in ("h1", "c1", "h2", "c2", "attend_cap", "attend_img", "sel"):
            full = np.stack([s[k] for s in tr])
            parity.assert_close(full[:, :8, :32], g["greedy_" + k + "_slice"], parity.STATE_TOL * 5, "greedy " + k)
            parity.assert_close(full.astype(np.float64).sum(2), g["greedy_" + k + "_sum"], parity.SUM_TOL, "greedy sum " + k)


@pytest.mark.parametrize("name", ["editnet_small", "editnet_small_end", "editnet_full_b4"])
def test_editnet_torch_restatement_greedy(name):
    """oracle/editnet_torch.py (the as-written torch-CPU restatement timed by bench.py's cpu_baseline)
    against the reference's own greedy decode: ids bit-exact, log-probs within 1e-4."""
    import torch
    from oracle import editnet_torch as ET
    d = cases.build_editnet(name)
    g = parity.load(name)
    wm = d["wm"]
    seq, logp = ET.greedy_decode(ET.params_from_numpy(d["sd"]), wm["<start>"], wm["<end>"], torch.from_numpy(d["prev"]),
                                 torch.from_numpy(d["plen"]), torch.from_numpy(d["X"]))
    parity.check_greedy(seq.numpy(), logp.numpy(), g)


def test_editnet_b128_greedy_and_xe():
    """BASELINE.json metric shape (B=128, 36x2048, prev-caption len 20)."""
    d, c, P, g = _editnet_inputs("editnet_full_b128")
    wm = d["wm"]
    seq, logp = EN.greedy_decode(P, wm["<start>"], wm["<end>"], d["prev"], d["plen"], d["X"])
    parity.check_greedy(seq, logp, g)
    pred, caps_s, dl, sort_ind = EN.xe_forward(P, d["X"], d["caps"], d["clen"], d["prev"], d["plen"])
    parity.check_xe(pred, dl, sort_ind, g, c["V"], small=False)


@pytest.mark.parametrize("name", ["editnet_adaptive_small", "editnet_adaptive_full_b4"])
def test_editnet_adaptive(name):
    d, c, P, g = _editnet_inputs(name)
    H, M, fh, mask = EN.caption_encoder(P, d["prev"], d["plen"])
    p = d["probe"]
    vis = EN.visual_attention_adaptive(P, d["X"], p["h1"])
    parity.assert_close(vis, g["op_vis"], parity.STATE_TOL, "adaptive visual_attention")
    pred, caps_s, dl, sort_ind = EN.xe_forward(P, d["X"], d["caps"], d["clen"], d["prev"], d["plen"],
                                               image_mean=d["image_mean"], adaptive=True)
    parity.check_xe(pred, dl, sort_ind, g, c["V"], small=c["D"] < 1024)


@pytest.mark.parametrize("name", ["dcnet_small", "dcnet_small_end", "dcnet_full_b4"])
def test_dcnet(name):
    d = cases.build_dcnet(name)
    c, wm, g = d["case"], d["wm"], parity.load(name)
    P = DN.cast_params(d["sd"])
    enc, fh, mask = DN.caption_encoder(P, d["prev"], d["plen"])
    parity.assert_close(enc, g["enc_out"], parity.STATE_TOL, "dcnet encoder outputs")
    parity.assert_close(fh, g["enc_final"], parity.STATE_TOL, "dcnet final_hidden")
    assert np.array_equal(mask, g["enc_mask"])
    parity.assert_close(DN.caption_attention(P, enc, d["probe"]["h1"], mask), g["op_ctx"], parity.STATE_TOL, "dcnet ctx")
    pred, caps_s, dl, sort_ind = DN.xe_forward(P, d["caps"], d["clen"], d["prev"], d["plen"])
    parity.check_xe(pred, dl, sort_ind, g, c["V"], small=c["D"] < 1024)
    seq, logp = DN.greedy_decode(P, wm["<start>"], wm["<end>"], d["prev"], d["plen"])
    parity.check_greedy(seq, logp, g)


def test_atsize_fixtures_vs_numpy_oracle():
    """the at-size fixtures (oracle/make_atsize_golden.py): the numpy oracle reproduces the reference's adaptive B = 64
    forward and DCNet's B = 128 forward + greedy decode"""
    name = "editnet_adaptive_full_b64"
    d = cases.build_editnet(name)
    g = parity.load("atsize_" + name)
    P = EN.cast_params(d["sd"])
    pred, caps_s, dl, sort_ind = EN.xe_forward(P, d["X"], d["caps"], d["clen"], d["prev"], d["plen"],
                                               image_mean=d["image_mean"], adaptive=True)
    parity.check_xe(pred, dl, sort_ind, g, d["case"]["V"], small=False)
    assert abs(EN.xe_loss(pred, caps_s, dl) - float(g["xe_loss"])) < 1e-4
    name = "dcnet_full_b128"
    d = cases.build_dcnet(name)
    g = parity.load("atsize_" + name)
    Pd = DN.cast_params(d["sd"])
    pred, caps_s, dl, sort_ind = DN.xe_forward(Pd, d["caps"], d["clen"], d["prev"], d["plen"])
    parity.check_xe(pred, dl, sort_ind, g, d["case"]["V"], small=False)
    seq, logp = DN.greedy_decode(Pd, d["wm"]["<start>"], d["wm"]["<end>"], d["prev"], d["plen"])
    parity.check_greedy(seq, logp, g)
