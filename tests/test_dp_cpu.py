"""CPU, world_size 2 over gloo: the data-parallel exchange step of the training path
(show_edit_tell_amd.train): bucketed SUM all-reduce of gradients + global token normalisation
reproduce single-process big-batch gradients exactly."""
import os

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from show_edit_tell_amd.train import allreduce_gradients, global_token_count
    torch.manual_seed(0)
    lin = torch.nn.Linear(16, 8)
    emb = torch.nn.Embedding(11, 16)
    params = list(lin.parameters()) + list(emb.parameters())
    ids = torch.arange(12) % 11
    tgt = torch.arange(12) % 8
    # big-batch reference (same on every rank)
    loss = torch.nn.functional.cross_entropy(lin(emb(ids)), tgt, reduction="sum") / 12
    ref = torch.autograd.grad(loss, params)
    # sharded: rank 0 gets 5 rows, rank 1 gets 7 (ragged shards -> mean-of-means would be wrong); eight ranks: 1 or 2 rows each
    cuts = [0, 5, 12] if world == 2 else [0, 1, 3, 4, 6, 7, 9, 10, 12]
    sl = slice(cuts[rank], cuts[rank + 1])
    n_glob = global_token_count(sl.stop - sl.start, torch.device("cpu"))
    assert n_glob == 12
    for p in params:
        p.grad = None
    (torch.nn.functional.cross_entropy(lin(emb(ids[sl])), tgt[sl], reduction="sum") / n_glob).backward()
    nb = allreduce_gradients(params, bucket_bytes=256)        # tiny buckets -> several collectives
    assert nb >= 2
    err = max(float((p.grad - r).abs().max()) for p, r in zip(params, ref))
    # overlapped variant: gradients handed to the reducer one by one (as the backward finishes them), the rest
    # at the end; same sums, each tensor reduced exactly once
    from show_edit_tell_amd.train import BucketedAllReduce
    for p in params:
        p.grad = None
    (torch.nn.functional.cross_entropy(lin(emb(ids[sl])), tgt[sl], reduction="sum") / n_glob).backward()
    red = BucketedAllReduce(bucket_bytes=256)
    red.add(params[0].grad)
    red.add(params[0].grad)                                    # a repeated hand-over must not double count
    assert allreduce_gradients(params, reducer=red) >= 2
    err = max(err, max(float((p.grad - r).abs().max()) for p, r in zip(params, ref)))
    ret[rank] = err
    dist.destroy_process_group()


import pytest


@pytest.mark.parametrize("world", [2, 8])
def test_dp_gradient_allreduce_matches_big_batch(world):
    """(eight ranks: rank / port / bucket-order mistakes show up at 8, not at 2)"""
    port = 29500 + (os.getpid() % 2000) + world
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert len(ret) == world
    assert max(ret.values()) < 1e-6, dict(ret)


def test_flat_grad_buckets_touched_set_equals_single_rank_grads_and_hooks_are_removed():
    """FlatGradBuckets (the persistent flat gradient buffers of the data-parallel step): the set of parameters it reports as
    touched by a backward == the set of parameters that get a gradient in the plain single-rank step (the others keep
    `.grad = None` after release_untouched, as Adam expects), and rebuilding the buckets removes the accumulate hooks of
    the old ones instead of piling new ones on top."""
    from show_edit_tell_amd.train import FlatGradBuckets, flat_grad_buckets
    torch.manual_seed(1)
    used = torch.nn.Linear(6, 4)
    unused = torch.nn.Linear(3, 3)                       # takes no part in the graph
    mod = torch.nn.ModuleDict({"a": used, "b": unused})
    params = list(mod.parameters())
    x = torch.randn(5, 6)
    # single-rank reference: which parameters get a gradient
    for p in params:
        p.grad = None
    used(x).sum().backward()
    ref = {id(p) for p in params if p.grad is not None}
    ref_grads = {id(p): p.grad.clone() for p in params if p.grad is not None}
    fb = flat_grad_buckets(mod, params, bucket_bytes=64)
    fb.attach()
    assert all(p.grad is not None and p.grad.data_ptr() == fb.view[id(p)].data_ptr() for p in params)
    used(x).sum().backward()
    assert fb.touched == ref
    assert fb.release_untouched() == len(params) - len(ref)
    assert {id(p) for p in params if p.grad is not None} == ref
    for p in params:
        if id(p) in ref:
            assert torch.equal(p.grad, ref_grads[id(p)])
    hooks = lambda p: len(getattr(p, "_post_accumulate_grad_hooks", None) or {})
    assert all(hooks(p) == 1 for p in params)
    # another bucket size -> the buckets are rebuilt: old hooks gone, one hook per parameter again after attach()
    fb2 = flat_grad_buckets(mod, params, bucket_bytes=4096)
    assert fb2 is not fb and all(hooks(p) == 0 for p in params)
    fb2.attach()
    assert all(hooks(p) == 1 for p in params)
    used(x).sum().backward()
    assert fb2.touched == ref
    fb2.close()
    assert all(hooks(p) == 0 and p.grad is None for p in params)
