"""XE training step of the reference's `train()` (editnet.py:551-593) with data-parallel gradient
all-reduce: one process per GPU, `torch.distributed` backend "nccl" (= RCCL over
xGMI), weights replicated, batch sharded.

Single-process big-batch semantics are reproduced exactly: every rank normalises its summed token
loss by the GLOBAL token count (one scalar all-reduce) and gradients are SUM-reduced, so the
reduced gradient equals the gradient of CrossEntropyLoss(mean) over the concatenated batch; then
`clip_grad_norm_(0.25)` and the optimizer step run identically on every rank.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F
from torch.nn.utils.rnn import pack_padded_sequence

GRAD_CLIP = 0.25                     # editnet.py:580
BUCKET_BYTES = 64 << 20              # xGMI rings are per-link bound: few, large buckets


def xe_loss_sum(scores, caps_sorted, decode_lengths):
    """Summed token cross-entropy over the packed rows + token count (editnet.py:571-577)."""
    targets = caps_sorted[:, 1:]
    sc = pack_padded_sequence(scores, decode_lengths, batch_first=True).data
    tg = pack_padded_sequence(targets, decode_lengths, batch_first=True).data
    return F.cross_entropy(sc, tg, reduction="sum"), sc.shape[0], sc, tg


class BucketedAllReduce:
    """SUM all-reduce of gradients in flat buckets, started asynchronously as buckets fill (RCCL runs them on its own
    stream while the remaining backward kernels keep the compute stream busy) and copied back in `finish()`.
    xGMI rings are per-link bound, so buckets are few and large (64 MB)."""

    def __init__(self, group=None, bucket_bytes=BUCKET_BYTES):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.active = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
        self.bucket_bytes = bucket_bytes
        self.bucket, self.size, self.pending, self.seen = [], 0, [], set()
        self.n_buckets = 0

    def add(self, grad):
        """queue one final gradient tensor (each tensor once)"""
        if not self.active or grad is None or id(grad) in self.seen:
            return
        self.seen.add(id(grad))
        nbytes = grad.numel() * grad.element_size()
        if self.bucket and self.size + nbytes > self.bucket_bytes:
            self._launch()
        self.bucket.append(grad)
        self.size += nbytes
        if self.size >= self.bucket_bytes:
            self._launch()

    def _launch(self):
        if not self.bucket:
            return
        flat = torch.cat([g.reshape(-1) for g in self.bucket])
        work = self.dist.all_reduce(flat, op=self.dist.ReduceOp.SUM, group=self.group, async_op=True)
        self.pending.append((work, flat, self.bucket))
        self.bucket, self.size = [], 0
        self.n_buckets += 1

    def finish(self):
        """launch the last partial bucket, wait for every collective and scatter the sums back; returns #buckets"""
        if not self.active:
            return 0
        self._launch()
        for work, flat, bucket in self.pending:
            work.wait()
            off = 0
            for g in bucket:
                g.copy_(flat[off:off + g.numel()].view_as(g))
                off += g.numel()
        self.pending = []
        return self.n_buckets


def allreduce_gradients(params, group=None, bucket_bytes=BUCKET_BYTES, reducer=None):
    """SUM all-reduce of .grad over the process group in flat buckets (no-op without a group).  With a `reducer`
    that already received some gradients during backward, only the remaining ones are added."""
    r = reducer if reducer is not None else BucketedAllReduce(group, bucket_bytes)
    for p in params:
        r.add(p.grad)
    return r.finish()


def global_token_count(n_local, device, group=None):
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return n_local
    t = torch.tensor([float(n_local)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return int(t.item())


def xe_train_step(decoder, optimizer, image_features, caps, caplens, previous_caption, prev_caplen, use_ss=False,
                  ss_prob=0.0, group=None):
    """One step of editnet.py:558-581 on this rank's shard.  Returns (global mean loss, local tokens)."""
    decoder.train()
    scores, caps_sorted, decode_lengths, _ = decoder(image_features, caps, caplens, previous_caption, prev_caplen,
                                                     use_ss, ss_prob)
    loss_sum, n_tok, _, _ = xe_loss_sum(scores, caps_sorted, decode_lengths)
    n_glob = global_token_count(n_tok, image_features.device, group)
    loss = loss_sum / n_glob
    optimizer.zero_grad()
    from .autograd_ops import deferred_param_grads
    reducer = BucketedAllReduce(group)
    # one weight-gradient contraction per parameter over all timesteps; every finished gradient goes straight
    # into an all-reduce bucket, so the collectives overlap the remaining contractions
    with deferred_param_grads(on_ready=lambda p: reducer.add(p.grad)):
        loss.backward()
    params = [p for p in decoder.parameters() if p.requires_grad]
    allreduce_gradients(params, group, reducer=reducer)
    torch.nn.utils.clip_grad_norm_(params, GRAD_CLIP)
    optimizer.step()
    return float(loss.detach()) , n_tok


def dcnet_xe_train_step(dae, optimizer, caps, caplens, previous_caption, prev_caplen, group=None):
    """One step of dcnet.py:352-402 (the denoising auto-encoder has no image input) on this rank's shard; same
    global-token-count normalisation and gradient all-reduce as `xe_train_step`."""
    from .autograd_ops import deferred_param_grads
    dae.train()
    scores, caps_sorted, decode_lengths, _ = dae(caps, caplens, previous_caption, prev_caplen)
    loss_sum, n_tok, _, _ = xe_loss_sum(scores, caps_sorted, decode_lengths)
    n_glob = global_token_count(n_tok, caps.device, group)
    loss = loss_sum / n_glob
    optimizer.zero_grad()
    reducer = BucketedAllReduce(group)
    with deferred_param_grads(on_ready=lambda p: reducer.add(p.grad)):
        loss.backward()
    params = [p for p in dae.parameters() if p.requires_grad]
    allreduce_gradients(params, group, reducer=reducer)
    torch.nn.utils.clip_grad_norm_(params, GRAD_CLIP)
    optimizer.step()
    return float(loss.detach()), n_tok


def reward_loss_sum(sample_logprobs, seq, reward):
    """Numerator and mask count of RewardCriterion (editnet_rl.py:557-573): the mask keeps every
    sampled word plus the <end> position (shifted `seq > 0`)."""
    mask = (seq > 0).float()
    mask = torch.cat([mask.new_ones(mask.size(0), 1), mask[:, :-1]], 1)
    return torch.sum(-sample_logprobs * reward * mask), mask.sum()


def scst_train_step(decoder, optimizer, word_map, image_features, previous_caption, prev_caplen, ground_truth,
                    scorer, n_samples=1, cider_weight=1.0, group=None):
    """One self-critical step of editnet_rl.py:649-686 on this rank's shard: greedy baseline in eval mode
    under no_grad (fused device loop), `n_samples` multinomial rollouts in train mode (autograd operators),
    reward = CIDEr-D(sample) - CIDEr-D(greedy) from `scorer` (ciderd.CiderD; host work, per sample),
    RewardCriterion, backward, gradient all-reduce, clip, optimizer.  The reference draws one sample per
    image; BASELINE.json config 5 asks for 5: all samples enter one RewardCriterion over n_samples * B rows.  As in the XE step the
    loss is normalised by the GLOBAL mask count so that N ranks reproduce one big batch.
    Returns (mean reward of the samples on this rank, loss value)."""
    from . import ciderd
    from .autograd_ops import deferred_param_grads
    dev = image_features.device
    optimizer.zero_grad()
    decoder.eval()
    with torch.no_grad():
        greedy, _ = decoder(word_map, previous_caption, prev_caplen, image_features, sample_max=True, sample_rl=False)
    decoder.train()
    # the n_samples rollouts of one image are independent rows (own dropout masks, own multinomial draws):
    # run them as ONE rollout over a batch of n_samples * B rows -- bigger GEMM tiles, one loop
    rep = lambda t: t if n_samples == 1 else t.repeat(n_samples, *([1] * (t.dim() - 1)))
    reducer = BucketedAllReduce(group)
    with deferred_param_grads(on_ready=lambda p: reducer.add(p.grad)):
        seq, logp = decoder(word_map, rep(previous_caption), rep(prev_caplen), rep(image_features),
                            sample_max=False, sample_rl=True)
        rewards = ciderd.self_critical_reward(scorer, seq, rep(greedy), list(ground_truth) * n_samples, cider_weight)
        num, cnt = reward_loss_sum(logp, seq, torch.from_numpy(rewards).to(dev))
        n_glob = global_token_count(int(cnt.item()), dev, group)
        loss = num / n_glob
        loss.backward()
    reward_mean, loss_val = float(rewards[:, 0].mean()), float(loss.detach())
    params = [p for p in decoder.parameters() if p.requires_grad]
    allreduce_gradients(params, group, reducer=reducer)
    torch.nn.utils.clip_grad_norm_(params, GRAD_CLIP)
    optimizer.step()
    return reward_mean, loss_val
