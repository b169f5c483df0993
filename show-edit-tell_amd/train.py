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
_FLAT_BUCKETS = os.environ.get("SET_FLAT_GRAD_BUCKETS", "1") != "0"   # data-parallel: `.grad` as views of persistent flat buckets


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


class FlatGradBuckets:
    """Persistent flat fp32 gradient buffers of one module: every parameter's `.grad` is a VIEW into a bucket, so a
    bucket's all-reduce runs IN PLACE — no `torch.cat` into a fresh 64 MB buffer before every collective, no copy back
    after it (2 x 355 MB of traffic and the allocator churn per step otherwise).  Buckets follow the order in which the
    backward finishes gradients: `first` (fc, whose gradient is final when the sequence node's backward starts) gets a
    bucket of its own so that its collective runs underneath the whole back-propagation through time; the rest are
    packed in parameter order into <= bucket_bytes buckets (xGMI rings are per-link bound: few, large collectives).
    `attach()` zeroes the buffers and points every `.grad` at its view; the backward then accumulates into them."""

    ALIGN = 64                      # floats: every view starts on a 256-byte boundary (16-byte kernel loads, Adam kernel)

    def __init__(self, params, bucket_bytes, first=()):
        first_ids = {id(p) for p in first}
        order = [p for p in params if id(p) in first_ids] + [p for p in params if id(p) not in first_ids]
        n_first = sum(1 for p in params if id(p) in first_ids)       # (`first` parameters that are frozen / absent do not count)
        assert order and all(p.dtype == torch.float32 for p in order)
        self.params = order
        self.sig = tuple((id(p), p.numel(), p.device) for p in params)
        dev = order[0].device
        plan, cur, cur_n, cap = [], [], 0, max(1, bucket_bytes // 4)
        for i, p in enumerate(order):
            n = -(-p.numel() // self.ALIGN) * self.ALIGN
            boundary = cur and (cur_n + n > cap or (n_first and i == n_first))
            if boundary:
                plan.append((cur, cur_n))
                cur, cur_n = [], 0
            cur.append((p, cur_n))
            cur_n += n
        plan.append((cur, cur_n))
        self.flat = [torch.zeros(n, dtype=torch.float32, device=dev) for _, n in plan]
        self.view, self.bucket_of, self.members = {}, {}, []
        for b, (items, _) in enumerate(plan):
            self.members.append(len(items))
            for p_, off in items:
                self.view[id(p_)] = self.flat[b][off:off + p_.numel()].view_as(p_)
                self.bucket_of[id(p_)] = b
        self.by_ptr = {v.data_ptr(): pid for pid, v in self.view.items()}
        self.bytes = sum(f.numel() * 4 for f in self.flat)

    def attach(self):
        for f in self.flat:
            f.zero_()
        self.touched = set()
        if not getattr(self, "_hooks", None):
            # gradients produced by plain autograd accumulation (not announced through deferred_param_grads) are seen here.
            # The handles are kept: close() removes the hooks when the buckets are rebuilt (they would otherwise pile up on
            # the same parameters and keep the old buckets alive)
            self._hooks = [p.register_post_accumulate_grad_hook(lambda t, _self=self: _self.touched.add(id(t)))
                           for p in self.params]
        for p in self.params:
            p.grad = self.view[id(p)]

    def close(self):
        """detach from the parameters: hooks removed, `.grad` views of these buckets dropped"""
        for h in getattr(self, "_hooks", None) or ():
            h.remove()
        self._hooks = None
        for p in self.params:
            if p.grad is not None and p.grad.data_ptr() == self.view[id(p)].data_ptr():
                p.grad = None

    def touch(self, grad):
        """the backward announces a parameter whose gradient it wrote (BucketedAllReduce.add)"""
        pid = self.by_ptr.get(grad.data_ptr())
        if pid is not None:
            self.touched.add(pid)

    def release_untouched(self):
        """A parameter that took no part in this step's graph keeps `.grad = None` in the single-rank path, and
        torch.optim.Adam / the fused clip + Adam skip it (no step count, no momentum-only or weight-decay update).  Its flat
        view here is all zeros — needed so that every rank reduces the same buckets —, so once the exchange is done the view
        is detached again: both paths treat the parameter alike.  (The set is a property of the graph, equal on all ranks.)"""
        n = 0
        for p in self.params:
            if id(p) not in self.touched:
                p.grad = None
                n += 1
        return n


def flat_grad_buckets(module, params, bucket_bytes=None, first=()):
    """the module's FlatGradBuckets (built once, rebuilt when the parameter set / device changes)"""
    params = list(params)
    fb = module.__dict__.get("_grad_buckets")
    sig = tuple((id(p), p.numel(), p.device) for p in params)
    if fb is None or fb.sig != sig or fb.bucket_bytes != (bucket_bytes or BUCKET_BYTES):
        if fb is not None:
            fb.close()
        fb = FlatGradBuckets(params, bucket_bytes or BUCKET_BYTES, first)
        fb.bucket_bytes = bucket_bytes or BUCKET_BYTES
        module.__dict__["_grad_buckets"] = fb
    return fb


class BucketedAllReduce:
    """SUM all-reduce of gradients in flat buckets.  A bucket's collective is started asynchronously as soon as
    the bucket is complete (RCCL runs it on its own stream); `finish()` waits for all of them.  Gradients
    arrive through `add()`: inside `deferred_param_grads(on_ready=...)` every parameter is handed over the moment its
    time-batched weight-gradient contraction has been enqueued, so a bucket's collective overlaps the contractions of
    the parameters that follow (NOT the activation backward, which has finished by then) — except fc, whose gradient
    is final at the start of the backward and whose bucket therefore runs underneath all of it.
    With `flat` (FlatGradBuckets: `.grad` tensors are views of persistent flat buffers) the collective runs in place on
    the bucket; without it, finished gradients are packed into a fresh flat buffer and the sums copied back in `finish()`.
    xGMI rings are per-link bound, so buckets are few and large (64 MB)."""

    def __init__(self, group=None, bucket_bytes=None, enabled=True, flat=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.active = enabled and _dist_active(group)
        self.bucket_bytes = bucket_bytes or BUCKET_BYTES
        self.bucket, self.size, self.pending, self.seen = [], 0, [], set()
        self.n_buckets = 0
        self.bytes = 0
        self.flat = flat if self.active else None
        if self.flat is not None:
            self.arrived = [0] * len(self.flat.flat)
            self.launched = [False] * len(self.flat.flat)
            self.strays = []

    def add(self, grad, flush=False, from_backward=True):
        """queue one final gradient tensor (each tensor once); flush: start the (partial) bucket's collective now — used
        for gradients that are final long before the rest (fc in the sequence nodes: its all-reduce then runs underneath
        the back-propagation through time)"""
        if not self.active or grad is None or id(grad) in self.seen:
            return
        self.seen.add(id(grad))
        if self.flat is not None:
            if from_backward:
                self.flat.touch(grad)            # (the final sweep of allreduce_gradients announces nothing)
            pid = self.flat.by_ptr.get(grad.data_ptr())
            if pid is None:                  # a gradient that does not live in its view (replaced by the caller): pack it the old way
                self.strays.append(grad)
                return
            b = self.flat.bucket_of[pid]
            self.arrived[b] += 1
            if self.arrived[b] == self.flat.members[b]:
                self._launch_flat(b)
            return
        nbytes = grad.numel() * grad.element_size()
        if self.bucket and self.size + nbytes > self.bucket_bytes:
            self._launch()
        self.bucket.append(grad)
        self.size += nbytes
        if self.size >= self.bucket_bytes or (flush and self.size >= (8 << 20)):
            self._launch()

    def _launch_flat(self, b):
        if self.launched[b]:
            return
        self.launched[b] = True
        flat = self.flat.flat[b]
        work = _all_reduce_sum(self.dist, flat, self.group, async_op=True)
        self.pending.append((work, None, None))
        self.bytes += flat.numel() * flat.element_size()
        self.n_buckets += 1

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
        """launch what has not been launched, wait for every collective (and scatter the sums back where a bucket was
        packed); returns #buckets"""
        if not self.active:
            return 0
        if self.flat is not None:
            for b in range(len(self.flat.flat)):      # buckets with a member that never announced itself (unused parameter)
                self._launch_flat(b)
            for g in self.strays:
                self.bucket.append(g)
        self._launch()
        for work, flat, bucket in self.pending:
            if work is not None:
                work.wait()
            if bucket is None:
                continue
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
        r.add(p.grad, from_backward=False)
    n = r.finish()
    if r.flat is not None:
        r.flat.release_untouched()       # parameters outside this step's graph: `.grad = None`, as on a single rank
    return n


# A process group of ONE rank needs no exchange and the collectives are skipped.  Setting this to 1 makes a one-rank group
# go through them anyway (tests: the only way to run the RCCL calls of the exchange step on a one-GPU box).
MIN_WORLD_FOR_EXCHANGE = 2


def _dist_active(group=None):
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized() and dist.get_world_size(group) >= MIN_WORLD_FOR_EXCHANGE


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


def _token_count(caplens, caplens_host=None):
    """sum of the decode lengths (caption length - 1) WITHOUT a device->host synchronisation when a host copy of the
    lengths is at hand (the data loader's tensor before `.to(device)`, editnet.py:560-563)"""
    src = caplens_host if caplens_host is not None else caplens
    if torch.is_tensor(src):
        return int((src.reshape(-1).to("cpu") - 1).sum().item()) if src.is_cuda else int((src.reshape(-1) - 1).sum())
    return int(sum(int(l) - 1 for l in src))


def _begin_backward(module, group, grp_on, first=()):
    """gradient storage of one backward: data-parallel -> persistent flat buckets with `.grad` as views (zeroed; the
    backward accumulates into them and every bucket is all-reduced in place); single rank -> `.grad = None` (the fused
    weight-gradient contractions then write fresh tensors without reading anything)"""
    params = [p for p in module.parameters() if p.requires_grad]
    if grp_on and _FLAT_BUCKETS and all(p.is_cuda and p.dtype == torch.float32 for p in params):
        fb = flat_grad_buckets(module, params, first=first)
        fb.attach()
        return BucketedAllReduce(group, enabled=True, flat=fb)
    for p in module.parameters():
        p.grad = None
    return BucketedAllReduce(group, enabled=grp_on)


def xe_backward(decoder, image_features, caps, caplens, previous_caption, prev_caplen, use_ss=False, ss_prob=0.0,
                group=None, reduce=True, caplens_host=None):
    """Forward + backward + gradient exchange of editnet.py:558-579 on this rank's shard; leaves the SUM-reduced
    gradients of the GLOBAL mean loss in `.grad`.  Returns (global mean loss, local tokens, reducer).
    `reduce=False` skips every collective (single-rank timing of the same step).  `caplens_host`: the caption lengths
    as the data loader produced them (host tensor / list) — with it the token count costs no device round trip."""
    from .autograd_ops import deferred_param_grads
    grp_on = reduce and _dist_active(group)
    # the token count only depends on the caption lengths: exchange it BEFORE the forward is enqueued so the
    # host never waits on the device between forward and backward
    n_tok = _token_count(caplens, caplens_host)
    n_glob = global_token_count(n_tok, image_features.device, group) if grp_on else n_tok
    if caplens_host is not None and hasattr(decoder, "with_host_lengths"):
        decoder.with_host_lengths(caplens_host)      # sort order / decode lengths without a device round trip
    try:
        scores, caps_sorted, decode_lengths, _ = decoder(image_features, caps, caplens, previous_caption, prev_caplen, use_ss,
                                                         ss_prob)
    finally:
        decoder.__dict__.pop("_caplens_host", None)
    loss_sum, n_chk, _, _ = xe_loss_sum(scores, caps_sorted, decode_lengths)
    assert n_chk == n_tok, (n_chk, n_tok)
    loss = loss_sum / n_glob
    reducer = _begin_backward(decoder, group, grp_on, first=(decoder.fc.weight, decoder.fc.bias))
    # one weight-gradient contraction per parameter over all timesteps; every finished gradient goes straight
    # into an all-reduce bucket, so the collectives overlap the remaining contractions
    with deferred_param_grads(on_ready=lambda p, eager=False: reducer.add(p.grad, flush=eager)):
        loss.backward()
    params = [p for p in decoder.parameters() if p.requires_grad]
    allreduce_gradients(params, group, reducer=reducer)
    return (_global_loss(loss_sum, n_glob, group) if grp_on else float(loss_sum.detach()) / max(n_glob, 1)), n_tok, reducer


def _refuse_non_finite(loss):
    """The token cross-entropy kernel marks a target id outside [0, V) with a NaN row loss (csrc/loss.hip) and the fused
    clip + Adam deliberately lets a NaN norm through (as torch.clamp(max=1) does): one corrupted caption would turn every
    weight and Adam moment NaN for good.  The reference's F.cross_entropy raises before any update (editnet.py:577); so does
    the train step here, at the host read of the loss that it makes anyway — before the optimizer runs.  (The loss is the
    GLOBAL mean: every rank raises together.)"""
    import math
    if not math.isfinite(loss):
        raise _lib.SetError("non-finite XE loss (%r): NaN scores, or a target id outside [0, V) (csrc/loss.hip poisons that row); "
                            "no optimizer step was taken" % (loss,))


def xe_train_step(decoder, optimizer, image_features, caps, caplens, previous_caption, prev_caplen, use_ss=False,
                  ss_prob=0.0, group=None, reduce=True, caplens_host=None):
    """One step of editnet.py:558-581 on this rank's shard.  Returns (GLOBAL mean loss — identical on every
    rank —, local tokens)."""
    decoder.train()
    loss, n_tok, _ = xe_backward(decoder, image_features, caps, caplens, previous_caption, prev_caplen, use_ss, ss_prob,
                                 group, reduce, caplens_host)
    _refuse_non_finite(loss)
    params = [p for p in decoder.parameters() if p.requires_grad]
    clip_grad_norm_and_step(params, optimizer, GRAD_CLIP)
    return loss, n_tok


def dcnet_xe_backward(dae, caps, caplens, previous_caption, prev_caplen, group=None, reduce=True, caplens_host=None):
    """DCNet twin of `xe_backward` (dcnet.py:352-400): the denoising auto-encoder has no image input."""
    from .autograd_ops import deferred_param_grads
    grp_on = reduce and _dist_active(group)
    n_tok = _token_count(caplens, caplens_host)
    n_glob = global_token_count(n_tok, caps.device, group) if grp_on else n_tok
    scores, caps_sorted, decode_lengths, _ = dae(caps, caplens, previous_caption, prev_caplen)
    loss_sum, n_chk, _, _ = xe_loss_sum(scores, caps_sorted, decode_lengths)
    assert n_chk == n_tok, (n_chk, n_tok)
    loss = loss_sum / n_glob
    reducer = _begin_backward(dae, group, grp_on, first=(dae.fc.weight, dae.fc.bias))
    with deferred_param_grads(on_ready=lambda p, eager=False: reducer.add(p.grad, flush=eager)):
        loss.backward()
    params = [p for p in dae.parameters() if p.requires_grad]
    allreduce_gradients(params, group, reducer=reducer)
    return (_global_loss(loss_sum, n_glob, group) if grp_on else float(loss_sum.detach()) / max(n_glob, 1)), n_tok, reducer


def dcnet_xe_train_step(dae, optimizer, caps, caplens, previous_caption, prev_caplen, group=None, reduce=True,
                        caplens_host=None):
    """One step of dcnet.py:352-402 on this rank's shard; same global-token-count normalisation and gradient
    all-reduce as `xe_train_step`.  Returns (GLOBAL mean loss, local tokens)."""
    dae.train()
    loss, n_tok, _ = dcnet_xe_backward(dae, caps, caplens, previous_caption, prev_caplen, group, reduce, caplens_host)
    _refuse_non_finite(loss)
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
        if len(_side_streams) >= 32:          # keyed by the caller's stream: bounded, not a process-lifetime leak
            _side_streams.clear()
        s = _side_streams[key] = torch.cuda.Stream(dev)
    return s


def _scst_step(model, optimizer, greedy_fn, sample_fn, rep, ground_truth, scorer, n_samples, cider_weight, dev, group):
    """Shared body of the self-critical step (editnet_rl.py:649-686, dcnet_rl.py:451-493): greedy baseline in eval
    mode under no_grad (fused device loop), `n_samples` multinomial rollouts in train mode (autograd operators, the
    sampling epilogue runs on the device), reward = CIDEr-D(sample) - CIDEr-D(greedy) from `scorer` (host work, per
    sample), RewardCriterion normalised by the GLOBAL mask count, backward, gradient all-reduce, clip, optimizer."""
    from . import ciderd
    from .autograd_ops import deferred_param_grads
    core = getattr(model, "dae", model)              # DAEWithAR wraps the DAE whose fc finishes first
    reducer = _begin_backward(model, group, _dist_active(group), first=(core.fc.weight, core.fc.bias))
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
    with deferred_param_grads(on_ready=lambda p, eager=False: reducer.add(p.grad, flush=eager)):
        seq, logp = sample_fn()
        if side is not None:
            cur.wait_stream(side)
            greedy.record_stream(cur)
        # ONE device->host copy of the sampled ids serves the scorer AND the mask count of RewardCriterion (the shifted
        # `seq > 0` of editnet_rl.py:563-566 summed): no second round trip (`cnt.item()`) between the rollout and the backward
        seq_host = seq.cpu().numpy()
        rewards = ciderd.self_critical_reward(scorer, seq_host, rep(greedy), list(ground_truth) * n_samples, cider_weight)
        num, _ = reward_loss_sum(logp, seq, torch.from_numpy(rewards).to(dev, non_blocking=True))
        n_mask = int(seq_host.shape[0] + (seq_host[:, :-1] > 0).sum())
        n_glob = global_token_count(n_mask, dev, group)
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
        # (the image features go in ONCE with the repeat count: relu(att_embed(X)) has no dropout before it, so it is
        # contracted for the B images and its rows repeated — not for n_samples * B copies of the same regions, forward and
        # weight gradient alike; the rollout's region stream still gets its own row per sample)
        lambda: decoder(word_map, rep(previous_caption), rep(prev_caplen), image_features, sample_max=False,
                        sample_rl=True, repeat_images=n_samples),
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
