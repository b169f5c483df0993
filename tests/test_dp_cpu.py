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
