#!/usr/bin/env python
"""Headline benchmark: EditNet decode-steps/sec at B=128 (36x2048 features, prev-caption len 20).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One bench "step" = one pass of the hot path over one batch: the free-running greedy decode of
the reference's `editnet_rl.py:485-549` for a batch of 128 images — the per-sequence prologue
(caption encoder + hoisted projections) plus 19 decode timesteps — entirely on the GPU through
the C ABI (no host sync inside).  The metric counts decode timesteps:
    value = n_gpus * K * 19 / wall_time          [decode-steps/sec, B=128 rows each]
Inputs and weights are synthetic (seeded generator, random-init weights of the reference's
architecture, V = 10 000) and are resident in HBM before the timed region.  Multi-GPU: the path
shards by batch with replicated weights and no data-path collective: each rank
decodes its own 128-image batch (weak scaling); timing is barrier / sync bracketed, max over ranks.
Each decode is a strictly sequential chain of ~190 small kernels over 128 rows, so by default
`--streams 3` independent B=128 batches are kept in flight per GPU, each on its own HIP stream and
workspace (the K timed steps are still K complete B=128 decodes; nothing is skipped or shared);
`config.single_stream_*` reports the one-batch-at-a-time figure measured in the same run.

Extra objects on the JSON line:
  roofline      dominant kernel = the fp32-MFMA grouped GEMM `gemm_nt_f32<128,64>` (six launches per
                timestep); achieved = algorithmic FLOPs of its launches / their summed duration,
                measured with HIP events on the launch stream in a second, identical, profiled pass
                (`value` comes from the un-instrumented pass; both ms_per_step are reported)
  kernels       the same event timing for every kernel family (ms per bench step)
  cpu_baseline  the numpy oracle (oracle/editnet_np.py, a port of the reference's CPU path) timed
                on this box's host cores for a bounded sample of the same workload
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

B, R, F, T, V, D, A = 128, 36, 2048, 20, 10000, 1024, 512
MAX_LEN = 18
STEPS_PER_DECODE = MAX_LEN + 1          # editnet_rl.py:503 runs max_len + 1 timesteps
PEAK_FP32_MFMA_TFLOPS = 157.3           # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_HBM_GBS = 8000.0


def pmc_traffic():
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes
    (tools/pmc_bench.sh -> tools/pmc_traffic.py); counters cannot be read from inside this process."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_bench_traffic.json")
    try:
        with open(path) as f:
            return json.load(f)
    except OSError:
        return None


def cpu_baseline(seconds_budget=20.0):
    """Oracle (numpy port of the reference CPU path) on the host cores, bounded sample."""
    from oracle import editnet_np as EN
    from show_edit_tell_amd import synth
    threads = int(os.environ.get("SET_CPU_THREADS", "0")) or min(os.cpu_count() or 1, 32)
    try:
        from threadpoolctl import threadpool_limits
        limiter = threadpool_limits(limits=threads)
    except Exception:
        limiter, threads = None, os.cpu_count()
    sd = synth.editnet_state(14, V, D, A, F, emb_scale=3.0, fc_scale=8.0, gain=3.0)
    P = EN.cast_params(sd)
    X = synth.features(25, B, R, F)
    prev, plen = synth.prev_captions(25, B, T, V, 5)
    wm = synth.word_map(V)
    EN.greedy_decode(P, wm["<start>"], wm["<end>"], prev[:8], plen[:8], X[:8])     # warm BLAS threads
    n, t0 = 0, time.time()
    while True:
        EN.greedy_decode(P, wm["<start>"], wm["<end>"], prev, plen, X)
        n += 1
        el = time.time() - t0
        if el > seconds_budget or n >= 10:
            break
    if limiter is not None:
        limiter.restore_original_limits()
    return dict(value=round(n * STEPS_PER_DECODE / el, 3), unit="decode-steps/sec", cores=threads,
                kind="port", sample="%d full greedy decodes (prologue + 19 timesteps) of the B=128 workload, numpy fp32 "
                "(OpenBLAS, %d threads of %d host cpus), %.1f s" % (n, threads, os.cpu_count(), el))


def experimental_split(args):
    """Secondary, NOT the headline: the same bench with SET_GEMM_SPLIT=1 (csrc/gemm_f32.hip, gemm_nt_split_bf16:
    every fp32 operand split exactly into 3 bf16, 6 partial products, fp32 accumulation -- fp32-level accuracy,
    all parity tests pass with it) in a child process, because the switch is read once per process."""
    import subprocess
    env = dict(os.environ, SET_GEMM_SPLIT="1")
    cmd = [sys.executable, os.path.abspath(__file__), "--steps", str(args.steps), "--warmup", str(args.warmup),
           "--streams", str(args.streams), "--no-cpu-baseline", "--no-profile", "--no-experimental"]
    try:
        out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
        d = json.loads(out.stdout.strip().splitlines()[-1])
        return {"split_bf16x3_gemm": {"decode_steps_per_sec": d["value"],
                                      "single_stream_decode_steps_per_sec": d["config"]["single_stream_decode_steps_per_sec"],
                                      "switch": "SET_GEMM_SPLIT=1 (off by default)",
                                      "arithmetic": "fp32 operands split exactly into 3 bf16 (8+8+8 significand bits), "
                                                    "6 of 9 partial products on v_mfma_f32_32x32x16_bf16, fp32 accumulate; "
                                                    "dropped terms <= 2^-24 relative; parity suite green"}}
    except Exception as e:          # never let the experiment break the bench line
        return {"split_bf16x3_gemm": {"error": repr(e)[:200]}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--no-experimental", action="store_true",
                    help="skip the opt-in split-precision (bf16x3) GEMM figure reported under 'experimental'")
    ap.add_argument("--streams", type=int, default=int(os.environ.get("SET_BENCH_STREAMS", "3")),
                    help="independent batches in flight per GPU (each on its own HIP stream + workspace)")
    args = ap.parse_args()

    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the decode path has no CPU fallback")
    # SET_BENCH_BACKEND=gloo + SET_BENCH_ONE_DEVICE=1: exercise the multi-rank code path on a 1-GPU box
    backend = os.environ.get("SET_BENCH_BACKEND", "nccl")
    if os.environ.get("SET_BENCH_ONE_DEVICE"):
        local_rank = 0
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    from show_edit_tell_amd import _lib, editnet_rl, synth
    wm = synth.word_map(V)
    dec = editnet_rl.DecoderC(wm, D, D, D, A, F)
    sd = synth.editnet_state(14, V, D, A, F, emb_scale=3.0, fc_scale=8.0, gain=3.0)
    sd["caption_encoder.embed.embedding.weight"] = sd["embed.embedding.weight"]
    dec.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    dec = dec.to(dev).eval()
    seed = 25 + rank
    X = torch.from_numpy(synth.features(seed, B, R, F)).to(dev)
    prev_np, plen_np = synth.prev_captions(seed, B, T, V, 5)
    prev, plen = torch.from_numpy(prev_np).to(dev), torch.from_numpy(plen_np).to(dev)

    streams = [torch.cuda.Stream(dev) for _ in range(max(1, args.streams))] if args.streams > 1 else None

    def run(k):
        out = None
        if streams is None:
            for _ in range(k):
                out = dec(wm, prev, plen, X, True, False)
            return out
        cur = torch.cuda.current_stream(dev)
        for s in streams:
            s.wait_stream(cur)
        for i in range(k):
            with torch.cuda.stream(streams[i % len(streams)]):
                out = dec(wm, prev, plen, X, True, False)
        for s in streams:
            cur.wait_stream(s)
        return out

    def barrier():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    with torch.no_grad():
        run(args.warmup)
        barrier()
        t0 = time.perf_counter()
        seq, _ = run(args.steps)
        barrier()
        elapsed = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())

        single = None
        if streams is not None and rank == 0:
            torch.cuda.synchronize(dev)
            keep, streams = streams, None
            ts = time.perf_counter()
            run(args.steps)
            torch.cuda.synchronize(dev)
            single = time.perf_counter() - ts
            streams = keep
        # secondary figure: teacher-forced XE forward (editnet.py:479-548, eval mode), same batch, 19 timesteps
        xe_rate = None
        if rank == 0:
            from show_edit_tell_amd import editnet
            caps_np, clen_np = synth.captions(seed, B, V, 20, 20)
            caps, clen = torch.from_numpy(caps_np).to(dev), torch.from_numpy(clen_np).to(dev)
            xe_fwd = lambda: editnet.DecoderC.forward(dec, X, caps, clen, prev, plen, False, 0.0)
            for _ in range(2):
                xe_fwd()
            torch.cuda.synchronize(dev)
            tx = time.perf_counter()
            nx = max(3, args.steps // 4)
            for _ in range(nx):
                xe_fwd()
            torch.cuda.synchronize(dev)
            xe_rate = nx * STEPS_PER_DECODE / (time.perf_counter() - tx)
        prof = None
        if rank == 0 and not args.no_profile:
            keep, streams = streams, None                 # per-kernel timing is taken on ONE stream
            lib = _lib.load()
            lib.set_profile_enable(1)
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            run(args.steps)
            torch.cuda.synchronize(dev)
            prof_elapsed = time.perf_counter() - t1
            prof = _lib.profile_report()
            lib.set_profile_enable(0)
            streams = keep

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    n_gpus = world
    total_steps = n_gpus * args.steps * STEPS_PER_DECODE
    line = {
        "metric": "decode-steps/sec (B=128, 36x2048 feats, seqlen=20)",
        "value": round(total_steps / elapsed, 2),
        "unit": "decode-steps/sec",
        "n_gpus": n_gpus,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "EditNet greedy decode (editnet_rl.py:485-549): prologue + 19 timesteps per bench step",
                   "batch_per_gpu": B, "regions": R, "feat_dim": F, "prev_caption_len": T, "vocab": V,
                   "decoder_dim": D, "attention_dim": A, "timesteps_per_bench_step": STEPS_PER_DECODE, "batches_in_flight_per_gpu": max(1, args.streams),
                   "mode": "eval (loop-invariant projections hoisted)", "parallelism": "dp%d (no collective)" % n_gpus,
                   "us_per_timestep_incl_prologue": round(1e6 * elapsed / (args.steps * STEPS_PER_DECODE), 2),
                   "single_stream_decode_steps_per_sec": (None if single is None else
                                                          round(args.steps * STEPS_PER_DECODE / single, 2)),
                   "single_stream_ms_per_step": None if single is None else round(1e3 * single / args.steps, 4),
                   "xe_forward_single_stream_decode_steps_per_sec": None if xe_rate is None else round(xe_rate, 2),
                   "distinct_tokens_in_last_batch": int(torch.unique(seq).numel())},
    }
    if prof is not None:
        by = {p["tag"]: p for p in prof}
        gemm_tags = sorted((t for t in by if t.startswith("gemm_nt_f32")), key=lambda t: -by[t]["ms"])
        g = by[gemm_tags[0]] if gemm_tags else None      # the dominant kernel = the GEMM tile variant with most time
        if g and g["ms"] > 0:
            tf = g["flops"] / (g["ms"] * 1e-3) / 1e12
            line["roofline"] = {
                "kernel": gemm_tags[0].replace(">", ",2,2>") + " (v_mfma_f32_32x32x2_f32)", "bound": "mfma",
                "achieved": round(tf, 2), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": round(tf / PEAK_FP32_MFMA_TFLOPS, 4),
                "traffic": (lambda t: None if t is None else round(t["traffic_bytes_per_launch"] / 1e6, 2))(pmc_traffic()),
                "traffic_unit": "MB HBM per launch (rocprofv3 PMC FETCH_SIZE x2 + WRITE_SIZE, profiles/r01_pmc_bench_traffic.json)",
                "algorithmic_MB_per_launch": round(g["bytes"] / g["launches"] / 1e6, 2),
                "launches": g["launches"], "avg_us_per_launch": round(1e3 * g["ms"] / g["launches"], 2),
                "algorithmic_gflop_per_launch": round(g["flops"] / g["launches"] / 1e9, 4),
                "algorithmic_GBs": round(g["bytes"] / (g["ms"] * 1e-3) / 1e9, 1),
                "timing": "HIP events on the launch stream, second identical pass (profiled ms_per_step %.3f)"
                          % (1e3 * prof_elapsed / args.steps)}
        line["kernels"] = {p["tag"]: {"launches_per_step": round(p["launches"] / args.steps, 2),
                                      "ms_per_step": round(p["ms"] / args.steps, 4),
                                      "GBs": round(p["bytes"] / max(p["ms"], 1e-9) / 1e6, 1),
                                      "TFLOPs": round(p["flops"] / max(p["ms"], 1e-9) / 1e9, 2)} for p in prof}
    split_env = os.environ.get("SET_GEMM_SPLIT", "0") not in ("", "0")
    if split_env:      # opt-in experimental kernel: say so in the line, never pass it off as the fp32-MFMA figure
        line["dtype"] = "f32 emulated as bf16x3 (6 partial products on v_mfma_f32_32x32x16_bf16, fp32 accumulate)"
        line.pop("roofline", None)
    if n_gpus == 1 and not args.no_experimental and not split_env:
        line["experimental"] = experimental_split(args)
    if not args.no_cpu_baseline and n_gpus == 1:
        line["cpu_baseline"] = cpu_baseline()
    print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
