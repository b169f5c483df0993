"""round 3 probe: one B=128 greedy decode as ONE chain vs as 2 / 4 row groups decoded concurrently on side streams
(same rows, same weights; the groups are independent samples).  Prints ms per complete B=128 decode."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from show_edit_tell_amd import editnet_rl, synth

V, D, A, F, R, T, B = 10000, 1024, 512, 2048, 36, 20, 128
dev = torch.device("cuda:0")
wm = synth.word_map(V)
dec = editnet_rl.DecoderC(wm, D, D, D, A, F)
sd = synth.editnet_state(14, V, D, A, F, emb_scale=3.0, fc_scale=8.0, gain=3.0)
sd["caption_encoder.embed.embedding.weight"] = sd["embed.embedding.weight"]
dec.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
dec = dec.to(dev).eval()
X = torch.from_numpy(synth.features(25, B, R, F)).to(dev)
prev_np, plen_np = synth.prev_captions(25, B, T, V, 5)
prev, plen = torch.from_numpy(prev_np).to(dev), torch.from_numpy(plen_np).to(dev)
side = [torch.cuda.Stream(dev) for _ in range(8)]


def decode(groups):
    if groups == 1:
        return dec(wm, prev, plen, X, True, False)
    cur = torch.cuda.current_stream(dev)
    n = B // groups
    outs = []
    for g in range(groups):
        s = side[g]
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            sl = slice(g * n, (g + 1) * n)
            outs.append(dec(wm, prev[sl], plen[sl], X[sl], True, False))
    for g in range(groups):
        cur.wait_stream(side[g])
    return torch.cat([o[0] for o in outs]), torch.cat([o[1] for o in outs])


with torch.no_grad():
    ref = None
    for groups in (1, 2, 4, 1, 2, 4, 3):
        if B % groups:
            continue
        for _ in range(4):
            out = decode(groups)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 60
        for _ in range(n):
            out = decode(groups)
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / n
        if ref is None:
            ref = out
        same = bool(torch.equal(out[0], ref[0]))
        print("groups %d: %.3f ms per B=128 decode = %.0f decode-steps/s   tokens identical to one chain: %s  max |dlogp| %.2e"
              % (groups, ms, 19e3 / ms, same, float((out[1] - ref[1]).abs().max())))
