#!/usr/bin/env python
"""Secondary benchmark: beam search (k=3) over a batch of images, EditNet and the
EditNet+DCNet ensemble, fused on-device bookkeeping vs. the torch-bookkeeping variant vs. the
reference-style one-image-at-a-time loop.

    python tools/bench_beam.py [--images 128] [--beam 3]
"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--images", type=int, default=128); ap.add_argument("--beam", type=int, default=3)
    ap.add_argument("--reps", type=int, default=3); ap.add_argument("--per-image", type=int, default=8)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    from show_edit_tell_amd import dcnet, editnet, evaluate, synth
    NI, R, F, T, V, D, A = a.images, 36, 2048, 20, 10000, 1024, 512
    wm = synth.word_map(V)
    sd = synth.editnet_state(14, V, D, A, F, emb_scale=3.0, fc_scale=8.0, gain=3.0)
    sd["fc.bias"] = sd["fc.bias"].copy(); sd["fc.bias"][wm["<end>"]] += 4.0        # captions end after ~10-20 words
    sd["caption_encoder.embed.embedding.weight"] = sd["embed.embedding.weight"]
    dec = editnet.DecoderC(wm, D, D, D, A, F); dec.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}); dec = dec.to(dev).eval()
    sdd = synth.dcnet_state(17, V, D, A, D // 2, D, 3.0, 8.0, 3.0)
    sdd["fc.bias"] = sdd["fc.bias"].copy(); sdd["fc.bias"][wm["<end>"]] += 4.0
    dae = dcnet.DAE(wm, None, D, A, D // 2, D)
    miss = dae.load_state_dict({k: torch.from_numpy(v) for k, v in sdd.items()}, strict=False); dae = dae.to(dev).eval()
    X = torch.from_numpy(synth.features(31, NI, R, F)).to(dev)
    prev, plen = (torch.from_numpy(x).to(dev) for x in synth.prev_captions(31, NI, T, V, 5))
    def timed(fn, reps):
        fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps): out = fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps, out
    res = {}
    t, out_f = timed(lambda: evaluate.beam_search_editnet_batched(dec, X, prev, plen, wm, a.beam), a.reps)
    res["editnet_fused_ms"] = round(1e3 * t, 2)
    n1 = min(a.per_image, NI)
    t, _ = timed(lambda: [evaluate.beam_search_editnet(dec, X[i:i + 1], prev[i:i + 1], plen[i:i + 1], wm, a.beam) for i in range(n1)], 1)
    res["editnet_one_image_per_call_ms_per_image"] = round(1e3 * t / n1, 2)
    t, out_x = timed(lambda: evaluate.beam_search_ensemble_batched(dec, dae, X, prev, plen, wm, a.beam), a.reps)
    res["ensemble_fused_ms"] = round(1e3 * t, 2)
    t, _ = timed(lambda: [evaluate.beam_search_ensemble(dec, dae, X[i:i + 1], prev[i:i + 1], plen[i:i + 1], wm, a.beam) for i in range(n1)], 1)
    res["ensemble_one_image_per_call_ms_per_image"] = round(1e3 * t / n1, 2)
    res["mean_caption_len"] = float(np.mean([len(s) for s in out_f]))
    res["images"], res["beam"] = NI, a.beam
    res["images_per_sec_fused"] = round(NI / (res["editnet_fused_ms"] / 1e3), 1)
    print(json.dumps(res))
if __name__ == "__main__":
    main()
