#!/usr/bin/env python
"""Secondary benchmark (BASELINE.json configs[4]): EditNet self-critical (SCST) step at B=64 per GPU,
5 sampled rollouts per image + greedy baseline, build-owned CIDEr-D reward, gradient all-reduce when
launched with torchrun.

    python tools/bench_scst.py [--steps K] [--warmup W] [--samples 5]
"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--steps", type=int, default=3); ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=64); ap.add_argument("--samples", type=int, default=5)
    a = ap.parse_args()
    world, rank, lr = int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(lr); dist.init_process_group("nccl", device_id=torch.device("cuda", lr))
    dev = torch.device("cuda", lr); torch.cuda.set_device(dev)
    from show_edit_tell_amd import ciderd, editnet_rl, synth
    from show_edit_tell_amd.train import scst_train_step
    B, R, F, T, V, D, A = a.batch, 36, 2048, 20, 10000, 1024, 512
    wm = synth.word_map(V)
    dec = editnet_rl.DecoderC(wm, D, D, D, A, F)
    sd = synth.editnet_state(14, V, D, A, F, emb_scale=3.0, fc_scale=8.0, gain=3.0)
    sd["caption_encoder.embed.embedding.weight"] = sd["embed.embedding.weight"]
    dec.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}); dec = dec.to(dev)
    opt = torch.optim.Adam(dec.parameters(), lr=5e-5)
    seed = 41 + rank
    X = torch.from_numpy(synth.features(seed, B, R, F)).to(dev)
    prev, plen = (torch.from_numpy(x).to(dev) for x in synth.prev_captions(seed, B, T, V, 5))
    rng = np.random.default_rng(seed)
    allcaps = np.zeros((B, 5, 20), dtype=np.int64)
    for b in range(B):
        for j in range(5):
            n = int(rng.integers(6, 17))
            allcaps[b, j, 0] = wm["<start>"]; allcaps[b, j, 1:1 + n] = rng.integers(1, V - 4, n); allcaps[b, j, 1 + n] = wm["<end>"]
    gt = ciderd.ground_truth_lists(allcaps, wm)
    df, docs = ciderd.document_frequency([[ciderd.tokens_to_str(c) for c in caps] for caps in gt])
    scorer = ciderd.CiderD(df, docs)
    def step(): return scst_train_step(dec, opt, wm, X, prev, plen, gt, scorer, n_samples=a.samples)
    for _ in range(a.warmup): step()
    torch.cuda.synchronize()
    if dist: dist.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps): reward, loss = step()
    torch.cuda.synchronize()
    if dist: dist.barrier()
    el = time.perf_counter() - t0
    if rank == 0:
        print(json.dumps({"metric": "SCST step (greedy + %d sampled rollouts + CIDEr-D reward + bwd + clip + Adam)" % a.samples,
                          "n_gpus": world, "batch_per_gpu": B, "ms_per_scst_step": round(1e3 * el / a.steps, 2),
                          "decode_steps_per_sec": round(world * a.steps * 19 * (a.samples + 1) / el, 2),
                          "mean_reward": reward, "loss": loss,
                          "note": "reward from the build-owned CIDEr-D (parity unpinned: the reference's scorer is external)"}))
    if dist: dist.destroy_process_group()
if __name__ == "__main__":
    main()
