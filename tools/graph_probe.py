#!/usr/bin/env python
"""Probe: does capturing the whole greedy decode (prologue + 19 timesteps, ~160 launches) in a HIP graph beat
stream launches?  (GPU box only)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from show_edit_tell_amd import editnet_rl, synth
dev = torch.device("cuda", 0)
B, R, F, T, V, D, A = 128, 36, 2048, 20, 10000, 1024, 512
wm = synth.word_map(V)
dec = editnet_rl.DecoderC(wm, D, D, D, A, F)
sd = synth.editnet_state(14, V, D, A, F, emb_scale=3.0, fc_scale=8.0, gain=3.0)
sd["caption_encoder.embed.embedding.weight"] = sd["embed.embedding.weight"]
dec.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}); dec = dec.to(dev).eval()
X = torch.from_numpy(synth.features(25, B, R, F)).to(dev)
prev, plen = (torch.from_numpy(x).to(dev) for x in synth.prev_captions(25, B, T, V, 5))
with torch.no_grad():
    for _ in range(4):
        seq0, _ = dec(wm, prev, plen, X, True, False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        dec(wm, prev, plen, X, True, False)
    torch.cuda.synchronize()
    print("stream launches: %.3f ms / decode" % (1e3 * (time.perf_counter() - t0) / 20))
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        dec(wm, prev, plen, X, True, False)
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        seq_g, logp_g = dec(wm, prev, plen, X, True, False)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    print("graph replay:    %.3f ms / decode" % (1e3 * (time.perf_counter() - t0) / 20), "same tokens:", bool(torch.equal(seq_g, seq0)))
