#!/usr/bin/env python
"""DCNet greedy decode / XE forward timing (BASELINE.json configs[0] shape B=4 and B=128)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from show_edit_tell_amd import dcnet, dcnet_rl, editnet_rl, synth
V, T = 10000, 20
wm = synth.word_map(V)
dev = torch.device("cuda:0")
def tm(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n
with torch.no_grad():
    sd = synth.dcnet_state(18, V, emb_scale=3.0, fc_scale=8.0, gain=3.0); sd["caption_encoder.embed.embedding.weight"] = sd["embed.embedding.weight"]
    xe = dcnet.DAE(wm, None); xe.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}); xe = xe.to(dev).eval()
    rl = dcnet_rl.DAE(wm, None); rl.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}); rl = rl.to(dev).eval()
    sde = synth.editnet_state(14, V, emb_scale=3.0, fc_scale=8.0, gain=3.0); sde["caption_encoder.embed.embedding.weight"] = sde["embed.embedding.weight"]
    er = editnet_rl.DecoderC(wm); er.load_state_dict({k: torch.from_numpy(v) for k, v in sde.items()}); er = er.to(dev).eval()
    for B in (4, 128):
        prev, plen = (torch.from_numpy(x).to(dev) for x in synth.prev_captions(3, B, T, V, 5))
        caps, clen = (torch.from_numpy(x).to(dev) for x in synth.captions(3, B, V, 20, 20))
        X = torch.from_numpy(synth.features(3, B, 36, 2048)).to(dev)
        t1 = tm(lambda: xe(caps, clen, prev, plen)); t2 = tm(lambda: rl(wm, prev, plen, True, False)); t3 = tm(lambda: er(wm, prev, plen, X, True, False))
        print("B=%3d  DCNet XE forward %.3f ms (%.0f steps/s) | DCNet greedy %.3f ms (%.0f steps/s) | EditNet greedy %.3f ms (%.0f steps/s)"
              % (B, 1e3 * t1, 19 / t1, 1e3 * t2, 19 / t2, 1e3 * t3, 19 / t3))
