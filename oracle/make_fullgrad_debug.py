"""ORACLE tooling (authoring container only), debugging aid: the reference's FULL train-mode gradients of a few parameters at
the benchmarked training shape (editnet_full_b128), written to gpurun_dbg/train_b128_full.npz (git-ignored, 95 MB; travels to
the GPU box with the snapshot) for tools/dbg_train_b128.py, which compares them element by element with both HIP routes.
Used in round 4 to show that the two sampled gradient elements outside 1e-4 are flipped ReLU kinks of the additive attention
(rank-one terms), not a defect.    python -m oracle.make_fullgrad_debug"""
import sys, numpy as np, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle.make_train_golden as M
orig = M._grads
def grads_full(module, small, out, prefix):
    orig(module, small, out, prefix)
    for k, p in module.named_parameters():
        if k in ("visual_attention.att_embed.0.weight", "visual_attention.features_att.weight", "visual_attention.att_embed.0.bias",
                 "caption_attention.cap_features_att.weight", "embed.embedding.weight", "fc.weight"):
            out[prefix + "fullgrad." + k] = p.grad.detach().numpy().copy()
M._grads = grads_full
name, seed, ss = [c for c in M.TRAIN_CASES if c[0] == "editnet_full_b128"][0]
out = M.make_editnet(name, seed, ss)
os.makedirs('gpurun_dbg', exist_ok=True)
np.savez('gpurun_dbg/train_b128_full.npz', **{k: v for k, v in out.items() if "fullgrad" in k or k in ("train.seed", "train.loss")})
print("saved", [k for k in out if "fullgrad" in k])
