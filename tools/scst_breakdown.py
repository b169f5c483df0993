#!/usr/bin/env python
"""Where an SCST step's wall time goes (BASELINE.json configs[4] shape): greedy baseline, sampled rollout forward,
reward on the host, backward, all-reduce/clip/Adam.  GPU box only."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from show_edit_tell_amd import ciderd, editnet_rl, synth
from show_edit_tell_amd.autograd_ops import deferred_param_grads
from show_edit_tell_amd.train import reward_loss_sum

B, R, F, T, V, D, A, NS = 64, 36, 2048, 20, 10000, 1024, 512, 5
dev = torch.device("cuda:0")
wm = synth.word_map(V)
dec = editnet_rl.DecoderC(wm, D, D, D, A, F)
sd = synth.editnet_state(14, V, D, A, F, emb_scale=3.0, fc_scale=8.0, gain=3.0)
sd["caption_encoder.embed.embedding.weight"] = sd["embed.embedding.weight"]
dec.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}); dec = dec.to(dev)
opt = torch.optim.Adam(dec.parameters(), lr=5e-5)
X = torch.from_numpy(synth.features(41, B, R, F)).to(dev)
prev, plen = (torch.from_numpy(x).to(dev) for x in synth.prev_captions(41, B, T, V, 5))
rng = np.random.default_rng(41)
allcaps = np.zeros((B, 5, 20), dtype=np.int64)
for b in range(B):
    for j in range(5):
        n = int(rng.integers(6, 17))
        allcaps[b, j, 0] = wm["<start>"]; allcaps[b, j, 1:1 + n] = rng.integers(1, V - 4, n); allcaps[b, j, 1 + n] = wm["<end>"]
gt = ciderd.ground_truth_lists(allcaps, wm)
df, docs = ciderd.document_frequency([[ciderd.tokens_to_str(c) for c in caps] for caps in gt])
scorer = ciderd.CiderD(df, docs)
rep = lambda t: t.repeat(NS, *([1] * (t.dim() - 1)))
acc = {}
def lap(name, t0):
    torch.cuda.synchronize(); acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0; return time.perf_counter()
for it in range(4):
    if it == 1: acc.clear()
    t = time.perf_counter()
    for p in dec.parameters(): p.grad = None
    dec.eval()
    with torch.no_grad():
        greedy, _ = dec(wm, prev, plen, X, True, False)
    t = lap("greedy (fused, eval)", t)
    dec.train()
    with deferred_param_grads():
        seq, logp = dec(wm, rep(prev), rep(plen), rep(X), sample_max=False, sample_rl=True)
        t = lap("sampled rollout forward (320 rows, autograd ops)", t)
        rewards = ciderd.self_critical_reward(scorer, seq, rep(greedy), list(gt) * NS, 1.0)
        t = lap("CIDEr-D reward (host)", t)
        num, cnt = reward_loss_sum(logp, seq, torch.from_numpy(rewards).to(dev))
        (num / cnt).backward()
        t = lap("backward (activation part)", t)
    t = lap("deferred weight gradients", t)
    torch.nn.utils.clip_grad_norm_(dec.parameters(), 0.25); opt.step()
    t = lap("clip + Adam", t)
tot = sum(acc.values())
for k, v in acc.items():
    print("%-52s %7.2f ms  %5.1f %%" % (k, 1e3 * v / 3, 100 * v / tot))
print("total %.2f ms" % (1e3 * tot / 3))
