#!/usr/bin/env python
"""Secondary benchmark (BASELINE.json configs[3]): EditNet with adaptive bottom-up features (10-100 valid regions,
zero padded to R=100, `image_mean` supplied) at B=64: XE forward (eval), XE training step, on one MI355X."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from show_edit_tell_amd import editnet_adaptive, synth
from show_edit_tell_amd.autograd_ops import deferred_param_grads
from show_edit_tell_amd.train import xe_loss_sum
B, R, F, T, V, D, A = 64, 100, 2048, 20, 10000, 1024, 512
dev = torch.device("cuda:0")
wm = synth.word_map(V)
dec = editnet_adaptive.DecoderC(wm, D, D, D, A, F)
sd = synth.editnet_state(14, V, D, A, F, emb_scale=3.0, fc_scale=8.0, gain=3.0)
sd["caption_encoder.embed.embedding.weight"] = sd["embed.embedding.weight"]
dec.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}); dec = dec.to(dev)
Xn, mean_n, nvalid = synth.adaptive_features(33, B, R, F, 10)
X, mean = torch.from_numpy(Xn).to(dev), torch.from_numpy(mean_n).to(dev)
prev, plen = (torch.from_numpy(x).to(dev) for x in synth.prev_captions(33, B, T, V, 5))
caps, clen = (torch.from_numpy(x).to(dev) for x in synth.captions(33, B, V, 20, 20))
def tm(fn, n=10, w=3):
    for _ in range(w): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n
dec.eval()
with torch.no_grad():
    t_fwd = tm(lambda: dec(X, mean, caps, clen, prev, plen, False, 0.0))
opt = torch.optim.Adam(dec.parameters(), lr=5e-4)
def train_step():
    dec.train(); opt.zero_grad()
    pred, caps_s, dl, _, gd_fh, last_h = dec(X, mean, caps, clen, prev, plen, False, 0.0)
    ls, n, _, _ = xe_loss_sum(pred, caps_s, dl)
    loss = ls / n + torch.nn.functional.mse_loss(last_h, gd_fh)          # editnet_adaptive.py:594-596
    with deferred_param_grads():
        loss.backward()
    torch.nn.utils.clip_grad_norm_(dec.parameters(), 0.25); opt.step()
t_train = tm(train_step, n=5, w=2)
print(json.dumps({"config": "adaptive features B=64, R=100 (valid %d..%d)" % (int(nvalid.min()), int(nvalid.max())),
                  "xe_forward_ms": round(1e3 * t_fwd, 3), "xe_forward_decode_steps_per_sec": round(19 / t_fwd, 1),
                  "train_step_ms": round(1e3 * t_train, 2), "train_decode_steps_per_sec": round(19 / t_train, 1)}))
