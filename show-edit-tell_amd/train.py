"""XE training step of the reference's `train()` (editnet.py:551-593) with data-parallel gradient
all-reduce: one process per GPU, `torch.distributed` backend "nccl" (= RCCL over
xGMI), weights replicated, batch sharded.

Single-process big-batch semantics are reproduced exactly: every rank normalises its summed token
loss by the GLOBAL token count (one scalar all-reduce) and gradients are SUM-reduced, so the
reduced gradient equals the gradient of CrossEntropyLoss(mean) over the concatenated batch; then
`clip_grad_norm_(0.25)` and the optimizer step run identically on every rank.
"""
from __future__ import annotations

import os

import torch
import torch.nn.functional as F
from torch.nn.utils.rnn import pack_padded_sequence

from . import loss as _loss
from .optim import clip_grad_norm_and_step

GRAD_CLIP = 0.25                     # editnet.py:580
BUCKET_BYTES = 64 << 20              # xGMI rings are per-link bound: few, large buckets


def xe_loss_sum(scores, caps_sorted, decode_lengths):
    """Summed token cross-entropy over the packed rows + token count (editnet.py:571-577).  Returns (loss sum, token
    count, packed scores, packed targets); the packed tensors are None when the loss ran on the library's kernels."""
    targets = caps_sorted[:, 1:]
    if _loss.fusable(scores, targets, decode_lengths):       # device scores: the library's loss kernels, nothing is packed
        return _loss.xe_loss_sum(scores, targets, decode_lengths), int(sum(int(l) for l in decode_lengths)), None, None
    sc = pack_padded_sequence(scores, decode_lengths, batch_first=True).data
    tg = pack_padded_sequence(targets, decode_lengths, batch_first=True).data
    return F.cross_entropy(sc, tg, reduction="sum"), sc.shape[0], sc, tg


def _all_reduce_sum(dist, t, group=None, async_op=False):
    """SUM all-reduce of `t` in place.  RCCL ("nccl") reduces device tensors directly; with the gloo
    backend (CPU tests, two ranks sharing one GPU) device tensors are staged through host memory here so the
    same code path runs on any backend build."""
    if t.is_cuda and dist.get_backend(group) == "gloo":
        h = t.detach().cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
        t.copy_(h)
        return None
    return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group, async_op=async_op)


class BucketedAllReduce:
    """SUM all-reduce of gradients in flat buckets.  A bucket's collective is started asynchronously as soon as
    the bucket is full (RCCL runs it on its own stream) and the sums are copied back in `finish()`.  Gradients
    arrive through `add()`: inside `deferred_param_grads(on_ready=...)` every parameter is handed over the moment its
    time-batched weight-gradient contraction has been enqueued, so a bucket's collective overlaps the contractions of
    the parameters that follow (NOT the activation backward, which has finished by then).
    xGMI rings are per-link bound, so buckets are few and large (64 MB)."""

    def __init__(self, group=None, bucket_bytes=None, enabled=True):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.active = (enabled and dist.is_available() and dist.is_initialized()
                       and dist.get_world_size(group) > 1)
        self.bucket_bytes = bucket_bytes or BUCKET_BYTES
        self.bucket, self.size, self.pending, self.seen = [], 0, [], set()
        self.n_buckets = 0
        self.bytes = 0

    def add(self, grad, flush=False):
        """queue one final gradient tensor (each tensor once); flush: start the (partial) bucket's collective now — used
        for gradients that are final long before the rest (fc in the sequence nodes: its all-reduce then runs underneath
        the back-propagation through time)"""
        if not self.active or grad is None or id(grad) in self.seen:
            return
        self.seen.add(id(grad))
        nbytes = grad.numel() * grad.element_size()
        if self.bucket and self.size + nbytes > self.bucket_bytes:
            self._launch()
        self.bucket.append(grad)
        self.size += nbytes
        if self.size >= self.bucket_bytes or (flush and self.size >= (8 << 20)):
            self._launch()

    def _launch(self):
        if not self.bucket:
            return
        flat = torch.cat([g.reshape(-1) for g in self.bucket])
        work = _all_reduce_sum(self.dist, flat, self.group, async_op=True)
        self.pending.append((work, flat, self.bucket))
        self.bytes += flat.numel() * flat.element_size()
        self.bucket, self.size = [], 0
        self.n_buckets += 1

    def finish(self):
        """launch the last partial bucket, wait for every collective and scatter the sums back; returns #buckets"""
        if not self.active:
            return 0
        self._launch()
        for work, flat, bucket in self.pending:
            if work is not None:
                work.wait()
            off = 0
            for g in bucket:
                g.copy_(flat[off:off + g.numel()].view_as(g))
                off += g.numel()
        self.pending = []
        return self.n_buckets


def allreduce_gradients(params, group=None, bucket_bytes=None, reducer=None):
    """SUM all-reduce of .grad over the process group in flat buckets (no-op without a group).  With a `reducer`
    that already received some gradients during backward, only the remaining ones are added."""
    r = reducer if reducer is not None else BucketedAllReduce(group, bucket_bytes)
    for p in params:
        r.add(p.grad)
    return r.finish()


def _dist_active(group=None):
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1


def global_token_count(n_local, device, group=None):
    """Sum of the ranks' token counts (the CE normaliser of the concatenated batch, editnet.py:577)."""
    import torch.distributed as dist
    if not _dist_active(group):
        return n_local
    t = torch.tensor([float(n_local)], dtype=torch.float64, device="cpu" if dist.get_backend(group) == "gloo" else device)
    _all_reduce_sum(dist, t, group)
    return int(t.item())


def _global_loss(loss_sum_local, n_glob, group=None):
    """Global mean loss = (sum over ranks of the local summed loss) / global count: the same number on every rank."""
    import torch.distributed as dist
    t = loss_sum_local.detach().double().reshape(1).clone()
    if _dist_active(group):
        _all_reduce_sum(dist, t, group)
    return float(t.item()) / max(n_glob, 1)


def xe_backward(decoder, image_features, caps, caplens, previous_caption, prev_caplen, use_ss=False, ss_prob=0.0,
                group=None, reduce=True):
    """Forward + backward + gradient exchange of editnet.py:558-579 on this rank's shard; leaves the SUM-reduced
    gradients of the GLOBAL mean loss in `.grad`.  Returns (global mean loss, local tokens, reducer).
    `reduce=False` skips every collective (single-rank timing of the same step)."""
    from .autograd_ops import deferred_param_grads
    grp_on = reduce and _dist_active(group)
    # the token count only depends on the caption lengths: exchange it BEFORE the forward is enqueued so the
    # host never waits on the device between forward and backward
    n_tok = int((caplens.reshape(-1) - 1).sum().item())
    n_glob = global_token_count(n_tok, image_features.device, group) if grp_on else n_tok
    scores, caps_sorted, decode_lengths, _ = decoder(image_features, caps, caplens, previous_caption, prev_caplen, use_ss,
                                                     ss_prob)
    loss_sum, n_chk, _, _ = xe_loss_sum(scores, caps_sorted, decode_lengths)
    assert n_chk == n_tok, (n_chk, n_tok)
    loss = loss_sum / n_glob
    for p in decoder.parameters():
        p.grad = None
    reducer = BucketedAllReduce(group, enabled=grp_on)
    # one weight-gradient contraction per parameter over all timesteps; every finished gradient goes straight
    # into an all-reduce bucket, so the collectives overlap the remaining contractions
    with deferred_param_grads(on_ready=lambda p, eager=False: reducer.add(p.grad, flush=eager)):
        loss.backward()
    params = [p for p in decoder.parameters() if p.requires_grad]
    allreduce_gradients(params, group, reducer=reducer)
    return (_global_loss(loss_sum, n_glob, group) if grp_on else float(loss_sum.detach()) / max(n_glob, 1)), n_tok, reducer


def xe_train_step(decoder, optimizer, image_features, caps, caplens, previous_caption, prev_caplen, use_ss=False,
                  ss_prob=0.0, group=None, reduce=True):
    """One step of editnet.py:558-581 on this rank's shard.  Returns (GLOBAL mean loss — identical on every
    rank —, local tokens)."""
    decoder.train()
    loss, n_tok, _ = xe_backward(decoder, image_features, caps, caplens, previous_caption, prev_caplen, use_ss, ss_prob,
                                 group, reduce)
    params = [p for p in decoder.parameters() if p.requires_grad]
    clip_grad_norm_and_step(params, optimizer, GRAD_CLIP)
    return loss, n_tok


def dcnet_xe_backward(dae, caps, caplens, previous_caption, prev_caplen, group=None, reduce=True):
    """DCNet twin of `xe_backward` (dcnet.py:352-400): the denoising auto-encoder has no image input."""
    from .autograd_ops import deferred_param_grads
    grp_on = reduce and _dist_active(group)
    n_tok = int((caplens.reshape(-1) - 1).sum().item())
    n_glob = global_token_count(n_tok, caps.device, group) if grp_on else n_tok
    scores, caps_sorted, decode_lengths, _ = dae(caps, caplens, previous_caption, prev_caplen)
    loss_sum, n_chk, _, _ = xe_loss_sum(scores, caps_sorted, decode_lengths)
    assert n_chk == n_tok, (n_chk, n_tok)
    loss = loss_sum / n_glob
    for p in dae.parameters():
        p.grad = None
    reducer = BucketedAllReduce(group, enabled=grp_on)
    with deferred_param_grads(on_ready=lambda p, eager=False: reducer.add(p.grad, flush=eager)):
        loss.backward()
    params = [p for p in dae.parameters() if p.requires_grad]
    allreduce_gradients(params, group, reducer=reducer)
    return (_global_loss(loss_sum, n_glob, group) if grp_on else float(loss_sum.detach()) / max(n_glob, 1)), n_tok, reducer


def dcnet_xe_train_step(dae, optimizer, caps, caplens, previous_caption, prev_caplen, group=None, reduce=True):
    """One step of dcnet.py:352-402 on this rank's shard; same global-token-count normalisation and gradient
    all-reduce as `xe_train_step`.  Returns (GLOBAL mean loss, local tokens)."""
    dae.train()
    loss, n_tok, _ = dcnet_xe_backward(dae, caps, caplens, previous_caption, prev_caplen, group, reduce)
    params = [p for p in dae.parameters() if p.requires_grad]
    clip_grad_norm_and_step(params, optimizer, GRAD_CLIP)
    return loss, n_tok


def reward_loss_sum(sample_logprobs, seq, reward):
    """Numerator and mask count of RewardCriterion (editnet_rl.py:557-573): the mask keeps every
    sampled word plus the <end> position (shifted `seq > 0`)."""
    mask = (seq > 0).float()
    mask = torch.cat([mask.new_ones(mask.size(0), 1), mask[:, :-1]], 1)
    return torch.sum(-sample_logprobs * reward * mask), mask.sum()


_SCST_OVERLAP = os.environ.get("SET_SCST_OVERLAP", "1") != "0"
_side_streams = {}


def _side_stream(dev):
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
    s = _side_streams.get(key)
    if s is None:
        s = _side_streams[key] = torch.cuda.Stream(dev)
    return s


def _scst_step(model, optimizer, greedy_fn, sample_fn, rep, ground_truth, scorer, n_samples, cider_weight, dev, group):
    """Shared body of the self-critical step (editnet_rl.py:649-686, dcnet_rl.py:451-493): greedy baseline in eval
    mode under no_grad (fused device loop), `n_samples` multinomial rollouts in train mode (autograd operators, the
    sampling epilogue runs on the device), reward = CIDEr-D(sample) - CIDEr-D(greedy) from `scorer` (host work, per
    sample), RewardCriterion normalised by the GLOBAL mask count, backward, gradient all-reduce, clip, optimizer."""
    from . import ciderd
    from .autograd_ops import deferred_param_grads
    for p in model.parameters():
        p.grad = None
    # the greedy baseline (a latency-bound chain of small kernels) and the sampled rollout are independent: the baseline is
    # enqueued on a side stream and runs underneath the rollout's forward (SET_SCST_OVERLAP=0: one after the other)
    cur = torch.cuda.current_stream(dev) if dev.type == "cuda" else None
    side = _side_stream(dev) if (cur is not None and _SCST_OVERLAP) else None
    model.eval()
    if side is not None:
        side.wait_stream(cur)
        with torch.cuda.stream(side), torch.no_grad():
            greedy, _ = greedy_fn()
    else:
        with torch.no_grad():
            greedy, _ = greedy_fn()
    model.train()
    reducer = BucketedAllReduce(group)
    with deferred_param_grads(on_ready=lambda p, eager=False: reducer.add(p.grad, flush=eager)):
        seq, logp = sample_fn()
        if side is not None:
            cur.wait_stream(side)
            greedy.record_stream(cur)
        rewards = ciderd.self_critical_reward(scorer, seq, rep(greedy), list(ground_truth) * n_samples, cider_weight)
        num, cnt = reward_loss_sum(logp, seq, torch.from_numpy(rewards).to(dev))
        n_glob = global_token_count(int(cnt.item()), dev, group)
        loss = num / n_glob
        loss.backward()
    reward_mean, loss_val = float(rewards[:, 0].mean()), _global_loss(num, n_glob, group)
    params = [p for p in model.parameters() if p.requires_grad]
    allreduce_gradients(params, group, reducer=reducer)
    clip_grad_norm_and_step(params, optimizer, GRAD_CLIP)
    return reward_mean, loss_val


def _repeater(n_samples):
    # the n_samples rollouts of one image are independent rows (own dropout masks, own draws): run them as ONE
    # rollout over a batch of n_samples * B rows -- bigger GEMM tiles, one loop
    return lambda t: t if n_samples == 1 else t.repeat(n_samples, *([1] * (t.dim() - 1)))


def scst_train_step(decoder, optimizer, word_map, image_features, previous_caption, prev_caplen, ground_truth,
                    scorer, n_samples=1, cider_weight=1.0, group=None):
    """One self-critical step of editnet_rl.py:649-686 on this rank's shard.  The reference draws one sample per
    image; BASELINE.json config 5 asks for 5: all samples enter one RewardCriterion over n_samples * B rows.  As in
    the XE step the loss is normalised by the GLOBAL mask count so that N ranks reproduce one big batch.
    Returns (mean reward of the samples on this rank, GLOBAL loss value)."""
    rep = _repeater(n_samples)
    return _scst_step(
        decoder, optimizer,
        lambda: decoder(word_map, previous_caption, prev_caplen, image_features, sample_max=True, sample_rl=False),
        lambda: decoder(word_map, rep(previous_caption), rep(prev_caplen), rep(image_features), sample_max=False,
                        sample_rl=True),
        rep, ground_truth, scorer, n_samples, cider_weight, image_features.device, group)


def dcnet_scst_train_step(dae, optimizer, word_map, previous_caption, prev_caplen, ground_truth, scorer, n_samples=1,
                          cider_weight=1.0, group=None):
    """The text-only twin (dcnet_rl.py:451-493): `dae` is a dcnet_rl.DAE or the DAEWithAR wrapper the reference
    trains (its forward delegates to `.dae`; `affine_hidden` receives no gradient from this loss, as in the reference).
    Returns (mean reward of the samples on this rank, GLOBAL loss value)."""
    rep = _repeater(n_samples)
    return _scst_step(
        dae, optimizer,
        lambda: dae(word_map, previous_caption, prev_caplen, sample_max=True, sample_rl=False),
        lambda: dae(word_map, rep(previous_caption), rep(prev_caplen), sample_max=False, sample_rl=True),
        rep, ground_truth, scorer, n_samples, cider_weight, previous_caption.device, group)
