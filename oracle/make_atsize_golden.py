"""ORACLE tooling (authoring container only): at-size fixtures for the B = 64 / B = 128 configurations.

    python -m oracle.make_atsize_golden           # rewrites tests/golden/atsize_*.npz

BASELINE.json configs[3] (adaptive features, 10-100 regions padded to 100, batch 64) and DCNet at the metric batch (128)
were timed but golden-checked at B = 4 only.  Here the reference's own classes (oracle/ref_slice.py) run those shapes
on CPU; stored are summaries only: per-row top-8 scores + log-sum-exp of the teacher-forced forward, the loss, every
parameter's gradient norm and a 64-element strided slice of every gradient (eval-mode dropout; CE + MSE for the adaptive
model, editnet_adaptive.py:594-596), the greedy decode of DCNet with its per-step top-1/top-2 margins; plus the
gradients of `editnet_adaptive_full_b4`, whose fixture carried none.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch
from torch.nn.utils.rnn import pack_padded_sequence as pps

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from oracle import cases, ref_slice  # noqa: E402
from oracle.make_golden import OUT, T_, _np, _summ_logits, _trace_greedy_dcnet  # noqa: E402


def _grads(module, out):
    for k, p in module.named_parameters():
        g = _np(p.grad)
        out["gradnorm." + k] = np.float64(np.sqrt((g.astype(np.float64) ** 2).sum()))
        out["gradslice." + k] = g.reshape(-1)[:: max(1, g.size // 64)][:64].copy()


def _pred_summary(pred, V, out):
    flat = pred.reshape(-1, V)
    for k, v in _summ_logits(flat, V).items():
        if k != "cols":
            out["xe_pred_" + k] = v.reshape(pred.shape[0], pred.shape[1], *v.shape[1:])


def make_adaptive(name, forward_summary):
    d = cases.build_editnet(name)
    c, wm = d["case"], d["wm"]
    dec = ref_slice.load_state(ref_slice.editnet_adaptive()["DecoderC"](wm, c["D"], c["D"], c["D"], c["A"], c["F"]),
                               d["sd"]).eval()
    X, mean, prev, plen, caps, clen = (T_(d[k]) for k in ("X", "image_mean", "prev", "plen", "caps", "clen"))
    dec.zero_grad()
    pred, caps_s, dl, sort_ind, gd_fh, last_h = dec(X, mean, caps, clen, prev, plen, False, 0.0)
    ce = torch.nn.CrossEntropyLoss()(pps(pred, dl, batch_first=True).data, pps(caps_s[:, 1:], dl, batch_first=True).data)
    loss = ce + torch.nn.MSELoss()(last_h, gd_fh)
    loss.backward()
    out = {"grad_loss": np.float64(loss.item()), "xe_loss": np.float64(ce.item())}
    _grads(dec, out)
    if forward_summary:
        out.update(xe_sort_ind=_np(sort_ind), xe_decode_lengths=np.asarray(dl, np.int64), xe_caps_sorted=_np(caps_s),
                   xe_gd_final_slice=_np(gd_fh)[:, :32], xe_last_hidden_slice=_np(last_h)[:, :32],
                   xe_gd_final_sum=_np(gd_fh).astype(np.float64).sum(1), xe_last_hidden_sum=_np(last_h).astype(np.float64).sum(1),
                   nvalid=np.asarray(d["nvalid"], np.int64))
        _pred_summary(_np(pred), c["V"], out)
    return out


def make_dcnet(name):
    d = cases.build_dcnet(name)
    c, wm = d["case"], d["wm"]
    args = (wm, None, c["D"], c["A"], c["C"], c["E"])
    xe = ref_slice.load_state(ref_slice.dcnet_xe()["DAE"](*args), d["sd"]).eval()
    rl = ref_slice.load_state(ref_slice.dcnet_rl()["DAE"](*args), d["sd"]).eval()
    prev, plen, caps, clen = (T_(d[k]) for k in ("prev", "plen", "caps", "clen"))
    out = {}
    xe.zero_grad()
    pred, caps_s, dl, sort_ind = xe(caps, clen, prev, plen)
    loss = torch.nn.CrossEntropyLoss()(pps(pred, dl, batch_first=True).data, pps(caps_s[:, 1:], dl, batch_first=True).data)
    loss.backward()
    out["grad_loss"] = out["xe_loss"] = np.float64(loss.item())
    _grads(xe, out)
    out.update(xe_sort_ind=_np(sort_ind), xe_decode_lengths=np.asarray(dl, np.int64), xe_caps_sorted=_np(caps_s))
    _pred_summary(_np(pred), c["V"], out)
    with torch.no_grad():
        enc, fh, mask = xe.caption_encoder(prev, plen)
        out.update(enc_out_sum=_np(enc).astype(np.float64).sum(2), enc_final_slice=_np(fh)[:, :32], enc_mask=_np(mask))
        seq, logp = rl(wm, prev, plen, True, False)
        out.update(greedy_seq=_np(seq), greedy_logp=_np(logp))
        steps = _trace_greedy_dcnet(rl, wm, prev, plen)
        out["greedy_nsteps"] = np.int64(len(steps))
        lg = np.stack([s["logits"] for s in steps])
        srt = np.sort(lg, axis=2)
        out["greedy_margin"] = (srt[:, :, -1] - srt[:, :, -2]).astype(np.float32)
        summ = _summ_logits(lg.reshape(-1, c["V"]), c["V"])
        for k in ("top_idx", "top_val", "lse"):
            out["greedy_logits_" + k] = summ[k].reshape(lg.shape[0], lg.shape[1], *summ[k].shape[1:])
        for k in ("h1", "c1", "h2", "c2", "attend_cap"):
            out["greedy_" + k + "_sum"] = np.stack([s[k] for s in steps]).astype(np.float64).sum(2)
    return out


def main(argv):
    assert ref_slice.have_reference(), "needs /root/reference (authoring container only)"
    torch.manual_seed(0)
    want = set(argv[1:])
    jobs = [("editnet_adaptive_full_b64", lambda: make_adaptive("editnet_adaptive_full_b64", True)),
            ("editnet_adaptive_full_b4", lambda: make_adaptive("editnet_adaptive_full_b4", False)),
            ("dcnet_full_b128", lambda: make_dcnet("dcnet_full_b128"))]
    for name, fn in jobs:
        if want and name not in want:
            continue
        o = fn()
        path = os.path.join(OUT, "atsize_" + name + ".npz")
        np.savez_compressed(path, **o)
        print("%-36s %8.1f KiB  loss %.6f" % ("atsize_" + name, os.path.getsize(path) / 1024, o["grad_loss"]))


if __name__ == "__main__":
    main(sys.argv)
