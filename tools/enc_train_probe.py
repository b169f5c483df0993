#!/usr/bin/env python
"""Time of the caption encoder's grad-enabled path (forward + backward) inside a training step.  GPU box only."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from show_edit_tell_amd import editnet, synth
from show_edit_tell_amd.autograd_ops import deferred_param_grads
B, R, F, T, V, D, A = 128, 36, 2048, 20, 10000, 1024, 512
dev = torch.device("cuda:0")
wm = synth.word_map(V)
dec = editnet.DecoderC(wm, D, D, D, A, F)
sd = synth.editnet_state(14, V, D, A, F, emb_scale=3.0, fc_scale=8.0, gain=3.0)
sd["caption_encoder.embed.embedding.weight"] = sd["embed.embedding.weight"]
dec.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}); dec = dec.to(dev).train()
prev, plen = (torch.from_numpy(x).to(dev) for x in synth.prev_captions(25, B, T, V, 5))
def step():
    for p in dec.parameters(): p.grad = None
    with deferred_param_grads():
        H, M, fh, mask = dec._encoder_autograd(prev, plen)
        (H.sum() + M.sum() + fh.sum()).backward()
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): step()
torch.cuda.synchronize()
print("encoder fwd+bwd (train mode): %.2f ms" % (1e2 * (time.perf_counter() - t0)))
