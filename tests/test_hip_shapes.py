"""GPU: awkward shapes through the fused C paths against the numpy oracle (no goldens needed: the
oracle is pinned to the reference separately): batch sizes that are not tile multiples, B=1, a
single region, T=1, vocabularies that are not multiples of 4/64, lengths all equal / all 1."""
import numpy as np
import pytest
import torch

import parity
from hip_adapter import load_numpy_state, to_dev
from oracle import editnet_np as EN
from show_edit_tell_amd import synth

pytestmark = pytest.mark.gpu

SHAPES = [
    # B,  T,  R,  D,  A,   F,   V
    (1, 1, 1, 64, 32, 64, 37),
    (3, 5, 2, 64, 32, 128, 203),
    (37, 7, 10, 64, 32, 128, 1003),
    (70, 9, 36, 128, 64, 256, 514),
    (130, 20, 5, 64, 64, 64, 9490 // 10),
    (200, 3, 100, 64, 32, 128, 300),
    (9, 6, 4, 512, 64, 128, 150),          # D = 512: the persistent encoder's third width
    (16, 4, 3, 2048, 64, 128, 211),        # D > 1024: more than one 1024-column pass per row in the pick's LSTM tail
]


@pytest.mark.parametrize("B,T,R,D,A,F,V", SHAPES)
def test_fused_paths_vs_oracle(B, T, R, D, A, F, V):
    from show_edit_tell_amd import editnet, editnet_rl
    wm = synth.word_map(V)
    sd = synth.editnet_state(3, V, D, A, F, emb_scale=3.0, fc_scale=8.0, gain=3.0)
    args = (wm, D, D, D, A, F)
    xe = load_numpy_state(editnet.DecoderC(*args), sd)
    rl = load_numpy_state(editnet_rl.DecoderC(*args), sd)
    X = synth.features(5, B, R, F)
    prev, plen = synth.prev_captions(5, B, T, V, min_len=1)
    caps, clen = synth.captions(5, B, V, L=20, min_len=3)
    P = EN.cast_params(sd)
    with torch.no_grad():
        seq, logp = rl(wm, to_dev(prev), to_dev(plen), to_dev(X), True, False)
        pred, caps_s, dl, sort_ind = xe(to_dev(X), to_dev(caps), to_dev(clen), to_dev(prev), to_dev(plen), False, 0.0)
    # XE forward: compare per original sample (sort ties may differ)
    xtr = []
    pred_o, _, dl_o, sort_o = EN.xe_forward(P, X, caps, clen, prev, plen, trace=xtr)
    inv, inv_o = parity.unsort(sort_ind.cpu().numpy()), parity.unsort(sort_o)
    assert sorted(dl) == sorted(dl_o)
    # SelectC is a HARD arg-max over alpha_c (editnet.py:410-416): a sample whose two largest copy weights are
    # closer than rounding noise may legitimately pick the other memory row, so such samples are not compared
    firm = np.ones(B, bool)
    for st_ in xtr:
        a = np.sort(st_["alpha_c"], 1)
        if a.shape[1] > 1:
            firm[:a.shape[0]] &= (a[:, -1] - a[:, -2]) > 1e-5
    assert firm.mean() > 0.9
    parity.assert_close(pred.cpu().numpy()[inv][firm[inv_o]], pred_o[inv_o][firm[inv_o]], parity.LOGIT_TOL,
                        "xe predictions")
    # greedy: rows whose oracle margin is comfortable must match bit-exactly; near-tie rows must match up to the
    # tie and pick one of the oracle's two best candidates there (parity.check_greedy_rows)
    tr = []
    seq_o, logp_o = EN.greedy_decode(P, wm["<start>"], wm["<end>"], prev, plen, X, trace=tr)
    lg = np.stack([s["logits"] for s in tr[:-1]]) if len(tr) > 1 else np.stack([s["logits"] for s in tr])
    srt = np.sort(lg, 2)
    margins = (srt[:, :, -1] - srt[:, :, -2]) if V > 1 else np.ones(lg.shape[:2], np.float32)
    top2 = np.argsort(-lg.astype(np.float64), axis=2, kind="stable")[:, :, :2]
    assert ((margins > 1e-3).all(0)).mean() > 0.8
    parity.check_greedy_rows(seq.cpu().numpy(), logp.cpu().numpy(), seq_o, logp_o, margins, top2, wm["<end>"], 1e-3)
