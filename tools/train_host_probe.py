#!/usr/bin/env python
"""Is the XE training step bound by the host's launch rate or by the GPU?  Times the enqueue (no synchronisation) and the
whole step (synchronised) of tools/bench_train.py's workload."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from show_edit_tell_amd import editnet, synth
from show_edit_tell_amd.train import xe_train_step
dev = torch.device("cuda", 0)
B, R, F, T, V, D, A = 128, 36, 2048, 20, 10000, 1024, 512
wm = synth.word_map(V)
dec = editnet.DecoderC(wm, D, D, D, A, F)
sd = synth.editnet_state(14, V, D, A, F, emb_scale=3.0, fc_scale=8.0, gain=3.0)
sd["caption_encoder.embed.embedding.weight"] = sd["embed.embedding.weight"]
dec.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}); dec = dec.to(dev)
opt = torch.optim.Adam(dec.parameters(), lr=5e-4)
X = torch.from_numpy(synth.features(25, B, R, F)).to(dev)
prev, plen = (torch.from_numpy(x).to(dev) for x in synth.prev_captions(25, B, T, V, 5))
caps, clen = (torch.from_numpy(x).to(dev) for x in synth.captions(25, B, V, 20, 20))
step = lambda: xe_train_step(dec, opt, X, caps, clen, prev, plen, False, 0.0)
for _ in range(4): step()
torch.cuda.synchronize()
enq, tot = [], []
for _ in range(10):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    step(); t1 = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    enq.append(t1 - t0); tot.append(t2 - t0)
enq.sort(); tot.sort()
print("SET_SLAB_DIRECT=%s: host enqueue %.2f ms, whole step %.2f ms (medians of 10; back-to-back steps overlap the two)" % (
    os.environ.get("SET_SLAB_DIRECT", "1"), 1e3 * enq[5], 1e3 * tot[5]))
