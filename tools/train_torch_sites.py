#!/usr/bin/env python
"""Which Python lines launch the torch-side kernels of a training step (fills, copies, reductions, cat ...)?  The library's
own kernels are named in the rocprofv3 summaries; the ~100 small aten launches per step are not attributable there.  Runs
the B = 128 XE training step of tools/bench_train.py under torch.profiler (with_stack) and prints, per aten operator, the
source lines (innermost frame inside this repository) with their launch counts and device time per step.

    python tools/train_torch_sites.py [--steps 3]
"""
import argparse, collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--steps", type=int, default=3); ap.add_argument("--batch", type=int, default=128)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    from show_edit_tell_amd import editnet, synth
    from show_edit_tell_amd.train import xe_train_step
    B, R, F, T, V, D, A = a.batch, 36, 2048, 20, 10000, 1024, 512
    wm = synth.word_map(V)
    dec = editnet.DecoderC(wm, D, D, D, A, F)
    sd = synth.editnet_state(14, V, D, A, F, emb_scale=3.0, fc_scale=8.0, gain=3.0)
    sd["caption_encoder.embed.embedding.weight"] = sd["embed.embedding.weight"]
    dec.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}); dec = dec.to(dev)
    opt = torch.optim.Adam(dec.parameters(), lr=5e-4)
    X = torch.from_numpy(synth.features(25, B, R, F)).to(dev)
    prev, plen = (torch.from_numpy(x).to(dev) for x in synth.prev_captions(25, B, T, V, 5))
    caps, clen = (torch.from_numpy(x).to(dev) for x in synth.captions(25, B, V, 20, 20))
    clen_host = clen.cpu()
    step = lambda: xe_train_step(dec, opt, X, caps, clen, prev, plen, False, 0.0, caplens_host=clen_host)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True,
                 experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as prof:
        for _ in range(a.steps):
            step()
        torch.cuda.synchronize()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sites = collections.defaultdict(lambda: [0, 0.0])
    for ev in prof.events():
        dt = getattr(ev, "self_device_time_total", None)
        if dt is None:
            dt = getattr(ev, "self_cuda_time_total", 0.0)
        if not ev.name.startswith("aten::") or dt <= 0:
            continue
        where = "?"
        for fr in ev.stack or ():
            if "show-edit-tell_amd/" in fr or "show_edit_tell_amd/" in fr:
                where = fr[fr.index("show"):]
                break
        if where == "?" and ev.stack:
            where = " < ".join(os.path.basename(f.split(":")[0]) + ":" + f.split(":")[-1][:28] for f in ev.stack[:3])
        s = sites[(ev.name, where)]
        s[0] += 1; s[1] += dt
    rows = sorted(sites.items(), key=lambda kv: -kv[1][1])
    tot = sum(v[1] for _, v in rows)
    print("# aten operators with device time, per training step (%d steps profiled): %.1f us, %d launches"
          % (a.steps, tot / a.steps, sum(v[0] for _, v in rows) // a.steps))
    print("%-34s %7s %9s  %s" % ("operator", "n/step", "us/step", "innermost repository frame"))
    for (name, where), (n, us) in rows[:60]:
        print("%-34s %7.1f %9.1f  %s" % (name, n / a.steps, us / a.steps, where))


if __name__ == "__main__":
    main()
