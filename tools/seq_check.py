"""Whole-sequence training node (xe_sequence.py) against the per-operator autograd path: outputs and every parameter
gradient, eval mode (no dropout) at several shapes, ragged and uniform lengths; train mode: runs, finite, loss falls."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from show_edit_tell_amd import editnet, synth
from show_edit_tell_amd.train import xe_loss_sum
from show_edit_tell_amd.autograd_ops import deferred_param_grads

dev = torch.device("cuda:0")


def build(V, D, A, F, seed=5):
    wm = synth.word_map(V)
    sd = synth.editnet_state(seed, V, D, A, F)
    sd["caption_encoder.embed.embedding.weight"] = sd["embed.embedding.weight"]
    m = editnet.DecoderC(wm, D, D, D, A, F)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return m.to(dev), wm


def run(m, inputs, seq, deferred):
    editnet._XE_SEQUENCE = seq
    m.zero_grad(set_to_none=True)
    X, caps, clen, prev, plen = inputs
    pred, caps_s, dl, _ = m(X, caps, clen, prev, plen, False, 0.0)
    ls, n, _, _ = xe_loss_sum(pred, caps_s, dl)
    loss = ls / n
    if deferred:
        with deferred_param_grads():
            loss.backward()
    else:
        loss.backward()
    return pred.detach().clone(), {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}


def compare(B, R, F, T, V, D, A, ragged, deferred):
    m, wm = build(V, D, A, F)
    m.eval()
    X = torch.from_numpy(synth.features(3, B, R, F)).to(dev)
    prev, plen = (torch.from_numpy(x).to(dev) for x in synth.prev_captions(3, B, T, V, 5))
    caps, clen = (torch.from_numpy(x).to(dev) for x in synth.captions(3, B, V, T, 8 if ragged else T))
    inputs = (X, caps, clen, prev, plen)
    p0, g0 = run(m, inputs, False, deferred)
    p1, g1 = run(m, inputs, True, deferred)
    err_p = (p0 - p1).abs().max().item()
    worst = 0.0
    gmax = max(g.abs().max().item() for g in g0.values())
    for k in g0:
        if k.endswith("full_att.bias"):        # mathematically zero (softmax is shift invariant): rounding noise only
            continue
        e = (g0[k] - g1[k]).abs().max().item() / max(g0[k].abs().max().item(), 1e-6 * gmax)
        worst = max(worst, e)
        if e > 1e-3:
            print("   ", k, e)
    assert set(g0) == set(g1), set(g0) ^ set(g1)
    print("B=%d V=%d D=%d ragged=%s deferred=%s: pred err %.2e, worst rel grad err %.2e" % (B, V, D, ragged, deferred, err_p, worst))
    assert err_p < 1e-4 and worst < 2e-3


compare(4, 36, 256, 20, 203, 64, 32, True, False)
compare(6, 36, 256, 20, 203, 64, 32, False, True)
compare(16, 36, 2048, 20, 1000, 256, 128, True, True)

# train mode at full size: timing of both paths
m, wm = build(10000, 1024, 512, 2048)
B = 128
X = torch.from_numpy(synth.features(3, B, 36, 2048)).to(dev)
prev, plen = (torch.from_numpy(x).to(dev) for x in synth.prev_captions(3, B, 20, 10000, 5))
caps, clen = (torch.from_numpy(x).to(dev) for x in synth.captions(3, B, 10000, 20, 20))
opt = torch.optim.Adam(m.parameters(), lr=5e-4)
for seq in (False, True):
    editnet._XE_SEQUENCE = seq
    m.train()
    losses = []
    for i in range(9):
        if i == 3:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        opt.zero_grad()
        pred, caps_s, dl, _ = m(X, caps, clen, prev, plen, False, 0.0)
        ls, n, _, _ = xe_loss_sum(pred, caps_s, dl)
        with deferred_param_grads():
            (ls / n).backward()
        torch.nn.utils.clip_grad_norm_(m.parameters(), 0.25)
        opt.step()
        losses.append(float(ls / n))
    torch.cuda.synchronize()
    print("train mode, sequence node %s: %.2f ms/step, loss %.4f -> %.4f" % (seq, 1e3 * (time.perf_counter() - t0) / 6, losses[0], losses[-1]))
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]
