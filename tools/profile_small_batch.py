#!/usr/bin/env python
"""Per-kernel HIP-event breakdown of one EditNet greedy decode at a small batch (BASELINE.json configs[0] shape)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from show_edit_tell_amd import _lib, editnet_rl, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
R, F, T, V, D, A = 36, 2048, 20, 10000, 1024, 512
dev = torch.device("cuda", 0)
wm = synth.word_map(V)
dec = editnet_rl.DecoderC(wm, D, D, D, A, F)
sd = synth.editnet_state(14, V, D, A, F, emb_scale=3.0, fc_scale=8.0, gain=3.0)
sd["caption_encoder.embed.embedding.weight"] = sd["embed.embedding.weight"]
dec.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}); dec = dec.to(dev).eval()
X = torch.from_numpy(synth.features(25, B, R, F)).to(dev)
prev, plen = (torch.from_numpy(x).to(dev) for x in synth.prev_captions(25, B, T, V, 5))
with torch.no_grad():
    for _ in range(5): dec(wm, prev, plen, X, True, False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): dec(wm, prev, plen, X, True, False)
    torch.cuda.synchronize(); ms = 1e3 * (time.perf_counter() - t0) / 20
    lib = _lib.load(); lib.set_profile_enable(1)
    for _ in range(10): dec(wm, prev, plen, X, True, False)
    torch.cuda.synchronize()
    prof = _lib.profile_report(); lib.set_profile_enable(0)
print("B=%d: %.3f ms per decode (%.0f decode-steps/s); weights streamed per step ~263 MB -> %.2f TB/s" % (B, ms, 19e3 / ms, 0.263 * 19 / ms))
for p in sorted(prof, key=lambda p: -p["ms"]):
    print("  %-28s %5.1f launches/decode  %8.1f us/launch  %7.1f GB/s  %6.2f TFLOP/s" % (p["tag"], p["launches"] / 10, 1e3 * p["ms"] / p["launches"], p["bytes"] / max(p["ms"], 1e-9) / 1e6, p["flops"] / max(p["ms"], 1e-9) / 1e9))
