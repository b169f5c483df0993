"""GPU, world_size 2: the data-parallel exchange step of the XE training path with the HIP model
(show_edit_tell_amd.train.xe_backward: global-token normalisation, deferred weight-gradient
contractions handing finished gradients to the bucketed all-reduce, SUM reduction).

Two ranks share cuda:0 (the gpurun box has one GPU), so the process group is gloo and the reducer
stages its flat buckets through host memory; on a multi-GPU node the same code runs over RCCL
(`backend "nccl"`, one device per rank — tests/test_hip_dp.py::test_dp_nccl_two_devices).
Asserted: the reduced gradients on every rank equal the gradients of ONE process running the
concatenated batch (reference semantics: CrossEntropyLoss(mean) over all packed rows,
editnet.py:575-577), and the returned loss is the same global number on both ranks.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _grads_of(xe):
    return {k: p.grad.detach().cpu().double().numpy().copy() for k, p in xe.named_parameters() if p.grad is not None}


def _worker(rank, world, port, backend, one_device, name, bucket_bytes, ret):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (here, os.path.dirname(here)):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dev_index = 0 if one_device else rank
    torch.cuda.set_device(dev_index)
    dev = "cuda:%d" % dev_index
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(dev))
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    from hip_adapter import editnet_modules, to_dev
    from show_edit_tell_amd import train
    train.BUCKET_BYTES = bucket_bytes               # small buckets -> several collectives in flight
    d, xe, _ = editnet_modules(name, dev)
    xe.eval()                                       # dropout off: deterministic, parameters still require grad
    B = d["X"].shape[0]
    full = tuple(to_dev(d[k], dev) for k in ("X", "caps", "clen", "prev", "plen"))
    # single-process big batch (the reference's semantics)
    loss_ref, n_ref, _ = train.xe_backward(xe, *full, reduce=False)
    ref = _grads_of(xe)
    # this rank's ragged shard (rank 0 gets fewer rows: mean-of-means would be wrong); more than two ranks: shards of
    # 1, 2, 3, 1, 2, 3, ... rows scaled to the batch
    if world == 2:
        cut = max(1, B // 3)
        cuts = [0, cut, B]
    else:
        w = [1 + (r % 3) for r in range(world)]
        cuts = [0]
        for r in range(world):
            cuts.append(min(B - (world - 1 - r), max(cuts[-1] + 1, round(B * sum(w[:r + 1]) / sum(w)))))
        cuts[-1] = B
    sl = slice(cuts[rank], cuts[rank + 1])
    assert sl.stop > sl.start
    shard = tuple(t[sl].contiguous() for t in full)
    loss, n_tok, reducer = train.xe_backward(xe, *shard)
    got = _grads_of(xe)
    assert set(got) == set(ref)
    # absolute floor: gradients that are mathematically zero (softmax shift invariance makes d/d full_att.bias == 0)
    # are pure rounding noise in both runs
    floor = 1e-6 * max(float(np.sqrt((r ** 2).sum())) for r in ref.values())
    worst = 0.0
    for k in ref:
        scale = max(np.abs(ref[k]).max(), 1e-6)
        worst = max(worst, float(max(np.abs(got[k] - ref[k]).max() - floor, 0.0) / scale))
    ret[rank] = dict(worst=worst, loss=loss, loss_ref=loss_ref, n_tok=n_tok, n_ref=n_ref,
                     buckets=reducer.n_buckets, bytes=reducer.bytes)
    dist.barrier()
    dist.destroy_process_group()


def _run(backend, one_device, name, bucket_bytes, world=2):
    import torch.multiprocessing as mp
    port = 29700 + (os.getpid() % 2000) + world
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, backend, one_device, name, bucket_bytes, ret), nprocs=world, join=True)
    assert len(ret) == world
    r0, r1 = ret[0], ret[1]
    # gradients: summation order differs (shards vs one batch) -> fp32 rounding only
    assert max(ret[r]["worst"] for r in range(world)) < 2e-4, dict(ret)
    assert sum(ret[r]["n_tok"] for r in range(world)) == r0["n_ref"]
    assert all(abs(ret[r]["loss"] - r0["loss"]) < 1e-12 for r in range(world)), "the returned loss must be the GLOBAL mean on every rank"
    assert abs(r0["loss"] - r0["loss_ref"]) < 1e-5 * max(1.0, abs(r0["loss_ref"]))
    assert r0["buckets"] >= 2 and r0["bytes"] > 0
    return dict(ret)


def test_dp_hip_model_two_ranks_one_device_gloo():
    _run("gloo", True, "editnet_small", 64 << 10)


def test_dp_hip_model_full_size_two_ranks_one_device_gloo():
    """BASELINE.json config 2 dims (D=1024, 36x2048, V=10000): 354.7 MB of gradients in 64 MB buckets."""
    r = _run("gloo", True, "editnet_full_b4", 64 << 20)
    assert r[0]["bytes"] > 300e6


def test_dp_hip_model_eight_ranks_one_device_gloo():
    """world size 8 (BASELINE.json configs[2] is an 8-GPU job; the box has one GPU, so the eight ranks share it over gloo):
    eight ragged shards of a 24-row batch, reduced gradients == the single-process big-batch gradients on EVERY rank."""
    r = _run("gloo", True, "editnet_small_b24", 16 << 10, world=8)
    assert len(r) == 8 and all(r[k]["buckets"] >= 2 for k in range(8))


def _nccl_one_rank_worker(rank, port, ret):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (here, os.path.dirname(here)):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    from hip_adapter import editnet_modules, to_dev
    from show_edit_tell_amd import train
    train.BUCKET_BYTES = 64 << 10
    train.MIN_WORLD_FOR_EXCHANGE = 1                # a one-rank group runs the collectives too
    d, xe, _ = editnet_modules("editnet_small", "cuda:0")
    xe.eval()
    full = tuple(to_dev(d[k], "cuda:0") for k in ("X", "caps", "clen", "prev", "plen"))
    loss_ref, n_ref, _ = train.xe_backward(xe, *full, reduce=False)
    ref = _grads_of(xe)
    loss, n_tok, reducer = train.xe_backward(xe, *full)            # token-count exchange + bucketed all-reduce on RCCL
    got = _grads_of(xe)
    # (the flat-bucket backward accumulates into zeroed views where the single-rank one writes fresh tensors: parameters with
    # several contributions may differ by the order of their additions)
    scale = max(float(np.abs(r).max()) for r in ref.values())
    worst = max(float(np.abs(got[k] - ref[k]).max()) for k in ref) / scale
    ret[0] = dict(worst=worst, loss=loss, loss_ref=loss_ref, n_tok=n_tok, n_ref=n_ref, buckets=reducer.n_buckets,
                  backend=dist.get_backend())
    dist.barrier()
    dist.destroy_process_group()


def test_dp_exchange_runs_on_rccl_with_one_rank():
    """The box has one GPU, so a two-rank RCCL group cannot form (RCCL refuses two ranks on one device); a ONE-rank group
    still goes through the same calls — communicator creation, the int64 token-count all-reduce before the forward, the
    in-place async all-reduce of every flat gradient bucket on the `nccl` backend, `work.wait()` — and must leave the
    gradients as they were (SUM over one rank)."""
    import torch.multiprocessing as mp
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_nccl_one_rank_worker, args=(29650 + (os.getpid() % 300), ret), nprocs=1, join=True)
    r = ret[0]
    assert r["backend"] == "nccl" and r["buckets"] >= 2
    assert r["worst"] < 1e-6, r
    assert r["n_tok"] == r["n_ref"] and abs(r["loss"] - r["loss_ref"]) < 1e-12


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL over xGMI)")
def test_dp_nccl_two_devices():
    _run("nccl", False, "editnet_small", 64 << 10)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL over xGMI)")
def test_bench_gpus_two_devices_nccl():
    """`python bench.py --gpus 2` on two real devices over RCCL: the line must say so (ranks_seen: two distinct devices,
    backend nccl) and carry the collective's own time next to the exposed part"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "SET_BENCH_ONE_DEVICE", "SET_BENCH_BACKEND"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2",
                          "--repeat", "1", "--no-profile", "--no-cpu-baseline", "--no-secondary", "--train-steps", "2"],
                         env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    tr = line["train"]
    assert line["n_gpus"] == 2 and tr["n_gpus"] == 2
    assert sorted(r["device"] for r in tr["ranks_seen"]) == [0, 1] and all(r["backend"] == "nccl" for r in tr["ranks_seen"])
    assert tr["allreduce_ms"] > 0 and "allreduce_exposed_ms" in tr


def test_bench_gpus_flag_spawns_ranks():
    """`python bench.py --gpus 2` (no torchrun around it) must launch two ranks and report n_gpus = 2 — the command the
    driver uses for its scaling run.  On the one-GPU test box both ranks share cuda:0 over gloo."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SET_BENCH_ONE_DEVICE="1", SET_BENCH_BACKEND="gloo")
    env.pop("WORLD_SIZE", None)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--repeat", "1", "--no-profile", "--no-cpu-baseline", "--no-secondary", "--train-steps", "2"],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["scaling"] == "weak"
    assert line["value"] > 0 and line["config"]["parallelism"].startswith("dp2")
    tr = line["train"]
    assert tr["n_gpus"] == 2 and tr["ms_per_train_step"] > 0 and "allreduce_exposed_ms" in tr
    assert len(tr["ranks_seen"]) == 2 and all(r["backend"] == "gloo" for r in tr["ranks_seen"])
    assert tr["allreduce_ms"] > 0 and sum(tr["allreduce_buckets"]) > 300        # 354.7 MB in flat buckets


def test_bench_gpus_eight_ranks_one_device_gloo():
    """`python bench.py --gpus 8` as the driver's scaling run launches it, on the one-GPU box (eight ranks share cuda:0,
    gloo): eight ranks must rendezvous, decode, run the exchange step and report — n_gpus 8, eight entries in ranks_seen."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SET_BENCH_ONE_DEVICE="1", SET_BENCH_BACKEND="gloo")
    env.pop("WORLD_SIZE", None)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1",
                          "--repeat", "1", "--streams", "1", "--no-profile", "--no-cpu-baseline", "--no-secondary",
                          "--train-steps", "2"], env=env, capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 8 and line["config"]["parallelism"].startswith("dp8") and line["value"] > 0
    tr = line["train"]
    assert tr["n_gpus"] == 8 and len(tr["ranks_seen"]) == 8
    assert sorted(r["rank"] for r in tr["ranks_seen"]) == list(range(8))


def test_flat_buckets_leave_unused_parameters_without_gradient():
    """with the persistent flat buckets every parameter's `.grad` is a zeroed view, so a parameter that takes no
    part in a step would look "updated with zeros" to the optimizer (Adam step count, momentum-only / weight-decay update)
    while the single-rank path leaves `.grad = None` and torch.optim.Adam skips it.  After the exchange the views of such
    parameters are detached again; and a `first` parameter that is not trainable does not shift the first bucket's boundary."""
    import torch.distributed as dist
    from show_edit_tell_amd import train
    from show_edit_tell_amd.train import BucketedAllReduce, FlatGradBuckets, allreduce_gradients
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(29400 + (os.getpid() % 200))
    dist.init_process_group("gloo", rank=0, world_size=1)                    # a one-rank group makes the exchange path live
    old_min, train.MIN_WORLD_FOR_EXCHANGE = train.MIN_WORLD_FOR_EXCHANGE, 1
    try:
        _flat_bucket_body(train, BucketedAllReduce, FlatGradBuckets, allreduce_gradients)
    finally:
        train.MIN_WORLD_FOR_EXCHANGE = old_min
        dist.destroy_process_group()


def _flat_bucket_body(train, BucketedAllReduce, FlatGradBuckets, allreduce_gradients):
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    a, b, c = (torch.nn.Linear(64, 64).to(dev) for _ in range(3))
    frozen = torch.nn.Parameter(torch.zeros(8, device=dev), requires_grad=False)
    params = [p for m in (a, b, c) for p in m.parameters()]
    fb = FlatGradBuckets(params, 1 << 20, first=(c.weight, frozen))         # `frozen` is not among the parameters
    assert fb.members[0] == 1 and sum(fb.members) == len(params)
    fb.attach()
    x = torch.randn(5, 64, device=dev)
    c(a(x)).sum().backward()                                                 # b is not part of this step's graph
    red = BucketedAllReduce(None, enabled=True, flat=fb)
    allreduce_gradients(params, reducer=red)
    assert b.weight.grad is None and b.bias.grad is None
    assert a.weight.grad is not None and c.bias.grad is not None and float(a.weight.grad.abs().sum()) > 0
    assert a.weight.grad.data_ptr() == fb.view[id(a.weight)].data_ptr()
    fb.attach()                                                              # the next step starts from views again
    assert b.weight.grad is not None and not b.weight.grad.any()
