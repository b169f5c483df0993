"""GPU: the beam-search callers (evaluate() / evaluate_full() mirrors) over the HIP operator API
against the oracle's hand-stated numpy beam search."""
import numpy as np
import pytest

from hip_adapter import load_numpy_state, to_dev
from oracle import beam_np, cases, dcnet_np as DN, editnet_np as EN

pytestmark = pytest.mark.gpu


def _boosted(sd, V, boost):
    sd = dict(sd)
    sd["fc.bias"] = sd["fc.bias"].copy()
    sd["fc.bias"][V - 1] += np.float32(boost)
    return sd


def test_beam_search_editnet_and_ensemble():
    from show_edit_tell_amd import dcnet, editnet, evaluate
    d = cases.build_editnet("editnet_small")
    c, wm = d["case"], d["wm"]
    # <end> boosted so that some hypotheses finish early and others run to the step limit
    sd_e = _boosted(d["sd"], c["V"], 3.0)
    sd_d = _boosted(cases.synth.dcnet_state(17, c["V"], c["D"], c["A"], c["D"] // 2, c["D"], 3.0, 8.0, 3.0), c["V"], 3.0)
    xe = load_numpy_state(editnet.DecoderC(wm, c["D"], c["D"], c["D"], c["A"], c["F"]), sd_e)
    dae = load_numpy_state(dcnet.DAE(wm, None, c["D"], c["A"], c["D"] // 2, c["D"]), sd_d)
    Pe, Pd = EN.cast_params(sd_e), DN.cast_params(sd_d)
    finished = 0
    for b in range(c["B"]):
        X1, prev1, plen1 = d["X"][b:b + 1], d["prev"][b:b + 1], d["plen"][b:b + 1]
        for ens in (False, True):
            if ens:
                seq_o, sc_o, margin = beam_np.beam_ensemble(Pe, Pd, X1, prev1, plen1, wm["<start>"], wm["<end>"], 3)
                seq, sc = evaluate.beam_search_ensemble(xe, dae, to_dev(X1), to_dev(prev1), to_dev(plen1), wm, 3)
            else:
                seq_o, sc_o, margin = beam_np.beam_editnet(Pe, X1, prev1, plen1, wm["<start>"], wm["<end>"], 3)
                seq, sc = evaluate.beam_search_editnet(xe, to_dev(X1), to_dev(prev1), to_dev(plen1), wm, 3)
            if margin is None:                  # ran into the step limit (editnet.py:702-704,711)
                assert np.isnan(sc) and len(seq) == 18 and seq[:4] == seq_o[:4], (b, ens, seq, seq_o)
            else:
                assert abs(sc - sc_o) < 1e-3, (b, ens, sc, sc_o)
                if margin > 1e-3:
                    assert seq == seq_o, (b, ens, seq, seq_o)
                    finished += 1
            assert evaluate.sentence(seq, wm) == " ".join("w%d" % w for w in seq if 0 < w < c["V"] - 3)
    assert finished >= 2


def _same_search(batched_b, seq, sc, what):
    if np.isnan(sc):                       # step-limit path: the 50-step trajectory is chaotic, compare the head
        assert len(batched_b) == 18 and batched_b[:4] == seq[:4], (what, batched_b, seq)
        return 0
    assert batched_b == seq, (what, batched_b, seq)
    return 1


def test_batched_beam_matches_per_image_beam():
    """Row f2: all images of a batch searched at once, entirely on the device (fused step + set_beam_pick_f32 +
    set_beam_gather_f32) == the reference-style one-image-at-a-time search; EditNet, DCNet and the ensemble."""
    from show_edit_tell_amd import dcnet, editnet, evaluate
    d = cases.build_editnet("editnet_small")
    c, wm = d["case"], d["wm"]
    for boost in (3.0, 5.0):
        xe = load_numpy_state(editnet.DecoderC(wm, c["D"], c["D"], c["D"], c["A"], c["F"]), _boosted(d["sd"], c["V"], boost))
        sd_d = _boosted(cases.synth.dcnet_state(17, c["V"], c["D"], c["A"], c["D"] // 2, c["D"], 3.0, 8.0, 3.0), c["V"], boost)
        dae = load_numpy_state(dcnet.DAE(wm, None, c["D"], c["A"], c["D"] // 2, c["D"]), sd_d)
        X, prev, plen = to_dev(d["X"]), to_dev(d["prev"]), to_dev(d["plen"])
        fused_e = evaluate.beam_search_editnet_batched(xe, X, prev, plen, wm, 3)
        torch_e = evaluate.beam_search_editnet_batched_torch(xe, X, prev, plen, wm, 3)
        fused_d = evaluate.beam_search_dcnet_batched(dae, prev, plen, wm, 3)
        fused_x = evaluate.beam_search_ensemble_batched(xe, dae, X, prev, plen, wm, 3)
        agree = 0
        for b in range(c["B"]):
            one = (X[b:b + 1], prev[b:b + 1], plen[b:b + 1])
            seq, sc = evaluate.beam_search_editnet(xe, *one, wm, 3)
            agree += _same_search(fused_e[b], seq, sc, ("editnet", boost, b))
            _same_search(torch_e[b], seq, sc, ("editnet-torch", boost, b))
            seq, sc = evaluate.beam_search_dcnet(dae, one[1], one[2], wm, 3)
            agree += _same_search(fused_d[b], seq, sc, ("dcnet", boost, b))
            seq, sc = evaluate.beam_search_ensemble(xe, dae, *one, wm, 3)
            agree += _same_search(fused_x[b], seq, sc, ("ensemble", boost, b))
        assert agree >= 6
    # wider beams than the reference's 3 (k <= 8 in the kernel)
    for k in (1, 5, 8):
        fused = evaluate.beam_search_editnet_batched(xe, X, prev, plen, wm, k)
        ref = evaluate.beam_search_editnet_batched_torch(xe, X, prev, plen, wm, k)
        for b in range(c["B"]):
            assert fused[b][:4] == ref[b][:4] and (len(fused[b]) == 18 or fused[b] == ref[b]), (k, b, fused[b], ref[b])


def test_batched_beam_full_size_vs_oracle():
    """Full-size model (D=1024, V=10000, 36x2048 regions): the fused batched beam search against the oracle's
    numpy beam search, image by image."""
    from show_edit_tell_amd import editnet, evaluate
    d = cases.build_editnet("editnet_full_b4")
    c, wm = d["case"], d["wm"]
    sd = _boosted(d["sd"], c["V"], 4.0)
    xe = load_numpy_state(editnet.DecoderC(wm, c["D"], c["D"], c["D"], c["A"], c["F"]), sd)
    P = EN.cast_params(sd)
    X, prev, plen = to_dev(d["X"]), to_dev(d["prev"]), to_dev(d["plen"])
    got = evaluate.beam_search_editnet_batched(xe, X, prev, plen, wm, 3)
    checked = 0
    for b in range(c["B"]):
        seq_o, sc_o, margin = beam_np.beam_editnet(P, d["X"][b:b + 1], d["prev"][b:b + 1], d["plen"][b:b + 1],
                                                   wm["<start>"], wm["<end>"], 3)
        if margin is None:
            assert len(got[b]) == 18 and got[b][:4] == seq_o[:4]
        elif margin > 1e-3:
            assert got[b] == seq_o, (b, got[b], seq_o)
            checked += 1
    assert checked >= 1
