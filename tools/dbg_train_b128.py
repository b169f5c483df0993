"""debug (round 4): full-tensor comparison of selected train-mode B=128 gradients against the reference's (gpurun_dbg/)"""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.join(os.getcwd(), "tests")); sys.path.insert(0, os.getcwd())
import parity
from hip_adapter import editnet_modules, to_dev
from show_edit_tell_amd import rng, editnet
from show_edit_tell_amd.train import xe_loss_sum
ref = dict(np.load("gpurun_dbg/train_b128_full.npz"))
g = parity.load("train_editnet_full_b128")
res = {}
for seq in (True, False):
    editnet._XE_SEQUENCE = seq
    for rep in range(2):
        d, xe, _ = editnet_modules("editnet_full_b128")
        xe.train()
        with rng.dropout_seed(int(g["train.seed"])):
            pred, caps_s, dl, sort_ind = xe(to_dev(d["X"]), to_dev(d["caps"]), to_dev(d["clen"]), to_dev(d["prev"]), to_dev(d["plen"]), False, 0.0)
        loss_sum, n_tok, _, _ = xe_loss_sum(pred, caps_s, dl)
        (loss_sum / n_tok).backward()
        torch.cuda.synchronize()
        res[(seq, rep)] = {k: p.grad.detach().cpu().numpy().copy() for k, p in xe.named_parameters()}
        print("route", "node" if seq else "per-op", "rep", rep, "sum(bts)/(T*B) =", sum(dl) / (max(dl) * len(dl)))
for k in [x[len("train.fullgrad."):] for x in ref if "fullgrad" in x]:
    r = ref["train.fullgrad." + k]
    sc = np.abs(r).max()
    for key, gr in res.items():
        e = np.abs(gr[k] - r)
        bad = e > 1e-4 * sc + 3e-6
        print("%-45s %-12s max err %.3e (scale %.3e)  bad elements %d" % (k, key, e.max(), sc, bad.sum()))
        if bad.sum() and r.ndim == 2:
            rows, cols = np.nonzero(bad)
            print("      rows:", np.unique(rows)[:20], "cols:", np.unique(cols)[:20], " n rows", len(np.unique(rows)), "n cols", len(np.unique(cols)))
    a, b = res[(True, 0)][k], res[(True, 1)][k]
    print("      node run-to-run identical:", np.array_equal(a, b), " node vs per-op max diff %.3e" % np.abs(res[(True, 0)][k] - res[(False, 0)][k]).max())
