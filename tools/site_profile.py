#!/usr/bin/env python
"""Per-call-site timing of the greedy decode (single stream): HIP events around every launch, aggregated by
kernel family AND by GEMM call site (SET_PROFILE_SITES=1).  GPU box only.

    SET_PROFILE_SITES=1 python tools/site_profile.py [--batch 128] [--decodes 20]
"""
import argparse, json, os, sys
os.environ.setdefault("SET_PROFILE_SITES", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--decodes", type=int, default=20)
    a = ap.parse_args()
    from show_edit_tell_amd import _lib, editnet_rl, synth
    B, R, F, T, V, D, A = a.batch, 36, 2048, 20, 10000, 1024, 512
    dev = torch.device("cuda:0")
    wm = synth.word_map(V)
    dec = editnet_rl.DecoderC(wm, D, D, D, A, F)
    sd = synth.editnet_state(14, V, D, A, F, emb_scale=3.0, fc_scale=8.0, gain=3.0)
    sd["caption_encoder.embed.embedding.weight"] = sd["embed.embedding.weight"]
    dec.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    dec = dec.to(dev).eval()
    X = torch.from_numpy(synth.features(25, B, R, F)).to(dev)
    prev, plen = (torch.from_numpy(x).to(dev) for x in synth.prev_captions(25, B, T, V, 5))
    lib = _lib.load()
    with torch.no_grad():
        for _ in range(5):
            dec(wm, prev, plen, X, True, False)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.decodes):
            dec(wm, prev, plen, X, True, False)
        e1.record()
        torch.cuda.synchronize()
        print("plain: %.3f ms per decode" % (e0.elapsed_time(e1) / a.decodes))
        lib.set_profile_enable(1)
        for _ in range(a.decodes):
            dec(wm, prev, plen, X, True, False)
        torch.cuda.synchronize()
        prof = _lib.profile_report(96)
        lib.set_profile_enable(0)
    tot = 0.0
    for p in sorted(prof, key=lambda p: -p["ms"]):
        n = p["launches"] / a.decodes
        us = 1e3 * p["ms"] / max(p["launches"], 1)
        if not p["tag"].startswith("gemm:"):
            tot += p["ms"] / a.decodes
        print("%-28s %6.1f launches/decode  %7.2f us/launch  %7.3f ms/decode  %7.2f TFLOP/s  %7.1f GB/s" % (
            p["tag"], n, us, p["ms"] / a.decodes, p["flops"] / max(p["ms"], 1e-9) / 1e9, p["bytes"] / max(p["ms"], 1e-9) / 1e6))
    print("sum of kernel families: %.3f ms per decode" % tot)


if __name__ == "__main__":
    main()
