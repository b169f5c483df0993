#!/usr/bin/env python
"""The exchange step of XE training (hook between /root/reference/editnet.py:579 and :580) on a ONE-rank RCCL group — the
only form of it a one-GPU box can run.  The same calls as an N-rank run (communicator creation, the global token count
before the forward, the in-place asynchronous all-reduce of every flat gradient bucket launched as the deferred weight
gradients finish, `work.wait()` before clip + Adam); a SUM over one rank leaves the gradients as they are, so what is
measured is what the exchange costs BEFORE any byte crosses xGMI: the baseline an N > 1 run's `allreduce_exposed_ms` /
`allreduce_ms` (bench.py train object) is compared against.

    python tools/rccl_one_rank.py [--steps K]       -> one JSON line

bench.py (N = 1) runs this in a child process with a timeout (a collective library that hangs must not take the bench line
down) and files the result under train.one_rank_rccl.
"""
import argparse, datetime, json, os, socket, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--batch", type=int, default=128)
    a = ap.parse_args()
    import torch.distributed as dist
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1,
                            timeout=datetime.timedelta(seconds=120), device_id=dev)
    from show_edit_tell_amd import editnet, synth, train
    from show_edit_tell_amd.train import xe_train_step, _all_reduce_sum
    B, R, F, T, V, D, A = a.batch, 36, 2048, 20, 10000, 1024, 512
    wm = synth.word_map(V)
    dec = editnet.DecoderC(wm, D, D, D, A, F)
    sd = synth.editnet_state(14, V, D, A, F, emb_scale=3.0, fc_scale=8.0, gain=3.0)
    sd["caption_encoder.embed.embedding.weight"] = sd["embed.embedding.weight"]
    dec.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    dec = dec.to(dev)
    opt = torch.optim.Adam(dec.parameters(), lr=5e-4)
    X = torch.from_numpy(synth.features(25, B, R, F)).to(dev)
    prev, plen = (torch.from_numpy(x).to(dev) for x in synth.prev_captions(25, B, T, V, 5))
    caps, clen = (torch.from_numpy(x).to(dev) for x in synth.captions(25, B, V, 20, 20))
    clen_host = clen.cpu()
    train.MIN_WORLD_FOR_EXCHANGE = 1            # a one-rank group goes through the collectives (train.py)

    def timed(reduce):
        for _ in range(3):
            xe_train_step(dec, opt, X, caps, clen, prev, plen, False, 0.0, reduce=reduce, caplens_host=clen_host)
        ts = []
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                xe_train_step(dec, opt, X, caps, clen, prev, plen, False, 0.0, reduce=reduce, caplens_host=clen_host)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / a.steps)
        return sorted(ts)[1]

    t_dp = timed(True)
    t_local = timed(False)
    fb = dec.__dict__.get("_grad_buckets")
    ar_ms, buckets = None, None
    if fb is not None:
        buckets = [round(f.numel() * 4 / 1e6, 1) for f in fb.flat]
        for _ in range(2):
            for f in fb.flat:
                _all_reduce_sum(dist, f, None)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            works = [_all_reduce_sum(dist, f, None, async_op=True) for f in fb.flat]
            for w_ in works:
                if w_ is not None:
                    w_.wait()
        torch.cuda.synchronize()
        ar_ms = round(1e3 * (time.perf_counter() - t0) / 5, 3)
    try:
        ver = ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception as e:
        ver = "unavailable: %r" % (e,)
    print(json.dumps({
        "what": "XE training step on a ONE-rank RCCL group (backend nccl): every call of the exchange step runs, nothing crosses "
                "xGMI — the baseline for an N > 1 run's allreduce_exposed_ms / allreduce_ms",
        "backend": dist.get_backend(), "rccl_version": ver, "batch": B, "steps": a.steps,
        "ms_per_train_step_with_exchange": round(1e3 * t_dp, 3), "ms_per_train_step_no_exchange": round(1e3 * t_local, 3),
        "allreduce_exposed_ms": round(1e3 * (t_dp - t_local), 3),
        "allreduce_alone_ms": ar_ms, "allreduce_overlapped_ms": None if ar_ms is None else round(ar_ms - 1e3 * (t_dp - t_local), 3),
        "allreduce_buckets_MB": buckets}))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
