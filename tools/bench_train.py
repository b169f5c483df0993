#!/usr/bin/env python
"""Secondary benchmark (BASELINE.json configs[1]/[2]): EditNet XE training step at B=128 per GPU —
forward (HIP operators, train mode: dropout active, nothing hoisted) + backward (PyTorch autograd)
+ gradient all-reduce (RCCL, when launched with torchrun) + clip + Adam.

    python tools/bench_train.py [--steps K] [--warmup W]
    python -m torch.distributed.run --nproc-per-node N tools/bench_train.py
"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--steps", type=int, default=5); ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=128)
    a = ap.parse_args()
    world, rank, lr = int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(lr); dist.init_process_group("nccl", device_id=torch.device("cuda", lr))
    dev = torch.device("cuda", lr); torch.cuda.set_device(dev)
    from show_edit_tell_amd import editnet, synth
    from show_edit_tell_amd.train import xe_train_step
    B, R, F, T, V, D, A = a.batch, 36, 2048, 20, 10000, 1024, 512
    wm = synth.word_map(V)
    dec = editnet.DecoderC(wm, D, D, D, A, F)
    sd = synth.editnet_state(14, V, D, A, F, emb_scale=3.0, fc_scale=8.0, gain=3.0)
    sd["caption_encoder.embed.embedding.weight"] = sd["embed.embedding.weight"]
    dec.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}); dec = dec.to(dev)
    opt = torch.optim.Adam(dec.parameters(), lr=5e-4)
    seed = 25 + rank
    X = torch.from_numpy(synth.features(seed, B, R, F)).to(dev)
    prev, plen = (torch.from_numpy(x).to(dev) for x in synth.prev_captions(seed, B, T, V, 5))
    caps, clen = (torch.from_numpy(x).to(dev) for x in synth.captions(seed, B, V, 20, 20))
    clen_host = clen.cpu()        # the data loader's host copy of the lengths
    def step(): return xe_train_step(dec, opt, X, caps, clen, prev, plen, False, 0.0, caplens_host=clen_host)
    for _ in range(a.warmup): step()
    torch.cuda.synchronize()
    if dist: dist.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps): loss, ntok = step()
    torch.cuda.synchronize()
    if dist: dist.barrier()
    el = time.perf_counter() - t0
    if rank == 0:
        print(json.dumps({"metric": "XE train step (fwd+bwd+allreduce+clip+Adam), 19 timesteps", "n_gpus": world, "batch_per_gpu": B,
                          "ms_per_train_step": round(1e3 * el / a.steps, 2),
                          "decode_steps_per_sec": round(world * a.steps * 19 / el, 2), "loss": loss,
                          "note": "forward = HIP operators through the C ABI (train mode, un-hoisted); backward = HIP pointwise/attention kernels + set_gemm_f32 (fp32 MFMA) contractions, time-batched weight gradients"}))
    if dist: dist.destroy_process_group()
if __name__ == "__main__":
    main()
