"""ORACLE tooling (authoring container only): golden vectors for the reference-present branches of two sub-modules that
the reference's own loops never take but a drop-in must still compute (missing #3):

    python -m oracle.make_boundary_golden        # writes tests/golden/boundary_ops.npz

* `SelectC.forward(..., soft=True)` (/root/reference/editnet.py:403-421, the branch at :419-420): output and both input
  gradients for a seeded upstream gradient, on the `editnet_small` case (memory rows from the reference's own caption
  encoder, weights = a seeded softmax over the valid positions).
* a differentiable DIRECT call of the adaptive `VisualAttentionC` (/root/reference/adaptive_features/editnet_adaptive.py:
  438-457), eval mode: context, d(context)/d(decoder_hidden) and every parameter gradient of the sub-module for a seeded
  upstream gradient, on the `editnet_adaptive_small` case (ragged region counts).

Only numbers the reference computed are stored; inputs and weights are regenerated from seeds by the tests
(`boundary_inputs` below is the shared recipe).
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from oracle import cases, ref_slice  # noqa: E402
from show_edit_tell_amd import synth  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
T_ = torch.from_numpy


def boundary_inputs(d, T):
    """Seeded probe inputs shared by this script and tests/test_hip_boundary.py: attention weights over T positions (a
    softmax, so rows sum to one like the reference's sim_weights) and upstream gradients."""
    c = d["case"]
    B, D, F = c["B"], c["D"], c["F"]
    sc = synth.uniform(c["iseed"], "boundary.scores", (B, T), -2, 2).astype(np.float64)
    e = np.exp(sc - sc.max(1, keepdims=True))
    alpha = (e / e.sum(1, keepdims=True)).astype(np.float32)
    return dict(alpha=alpha, dsel=synth.uniform(c["iseed"], "boundary.dsel", (B, D), -1, 1),
                dctx=synth.uniform(c["iseed"], "boundary.dctx", (B, F), -1, 1))


def main():
    out = {}
    # ---- SelectC, soft = True
    d = cases.build_editnet("editnet_small")
    c, wm = d["case"], d["wm"]
    cls = ref_slice.editnet_xe()
    dec = ref_slice.load_state(cls["DecoderC"](wm, c["D"], c["D"], c["D"], c["A"], c["F"]), d["sd"]).eval()
    with torch.no_grad():
        _, M, _, _ = dec.caption_encoder(T_(d["prev"]), T_(d["plen"]))
    p = boundary_inputs(d, M.shape[1])
    Mg = M.clone().requires_grad_(True)
    ag = T_(p["alpha"]).requires_grad_(True)
    sel = dec.select(Mg, ag, soft=True)
    sel.backward(T_(p["dsel"]))
    out.update(soft_sel=sel.detach().numpy(), soft_dM=Mg.grad.numpy(), soft_dalpha=ag.grad.numpy())
    # ---- adaptive VisualAttentionC, differentiable direct call (eval mode: Dropout is the identity)
    d = cases.build_editnet("editnet_adaptive_small")
    c, wm = d["case"], d["wm"]
    acls = ref_slice.editnet_adaptive()
    adec = ref_slice.load_state(acls["DecoderC"](wm, c["D"], c["D"], c["D"], c["A"], c["F"]), d["sd"]).eval()
    va = adec.visual_attention
    p = boundary_inputs(d, 1)
    h1 = T_(d["probe"]["h1"]).requires_grad_(True)
    va.zero_grad()
    ctx = va(T_(d["X"]), h1)
    # (the reference trims the context's region axis to the longest sample only; the feature axis is F)
    ctx.backward(T_(p["dctx"]))
    out.update(ada_ctx=ctx.detach().numpy(), ada_dh1=h1.grad.numpy())
    for k, q in va.named_parameters():
        out["ada_grad." + k] = q.grad.numpy()
    np.savez_compressed(os.path.join(OUT, "boundary_ops.npz"), **out)
    print("boundary_ops.npz %.1f KiB: %s" % (os.path.getsize(os.path.join(OUT, "boundary_ops.npz")) / 1024, sorted(out)))


if __name__ == "__main__":
    main()
