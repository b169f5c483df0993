"""Caller-side coalescing of small greedy-decode requests (a serving helper around the reference's eval-time call
`decoder(word_map, prev_caption, prev_len, features, sample_max=True)`, editnet_rl.py:485-549 / dcnet_rl.py:286-346).

Why: up to 16 rows (DCNet 8) the whole greedy loop is ONE persistent launch whose grid owns the chip, so the launches of a
process run one after the other — four callers with a 4-row request each get 790 requests/s from four launches, LESS than
the per-step kernels on four streams (1 070), but 2 280 when the four requests ride ONE 16-row launch (a 16-row timestep
costs 76 us, a 4-row one 52: DESIGN.md 3.6).  The library cannot do that for its callers (a C entry point sees one request);
this class does:

    co = RequestCoalescer(lambda prev, plen, X: decoder(word_map, prev, plen, X, True, False), max_rows=16)
    fut = co.submit(prev, plen, X)          # any thread; tensors on the device; returns at once
    seq, logp = fut.result()                # rows of THIS request; the caller's current stream is ordered after the decode
    co.close()

A worker thread takes the oldest waiting request, adds every request that is waiting at that moment (and, for at most
`window_s`, those that still arrive) up to `max_rows` rows, pads the previous captions to a common length with <pad> = 0
(the decode is invariant to trailing pad columns: tests/test_hip_properties.py), runs ONE decode on its own stream and hands
every request its rows.  Requests are never reordered within the batch, a request larger than `max_rows` runs alone, an
exception of the decode reaches every future of that batch.  Rows of one call are independent (row b's caption depends on
row b's inputs only), so a request's result does not depend on what it was batched with — bit for bit on the persistent
launch, whose per-row arithmetic does not depend on the row count.
"""
from __future__ import annotations

import threading
import time
from concurrent.futures import Future

import torch


class _Result:
    """(tensors, event): `.get()` orders the caller's current stream after the decode and returns the tensors"""

    def __init__(self, tensors, event):
        self.tensors, self.event = tensors, event

    def get(self):
        if self.event is not None:
            torch.cuda.current_stream(self.tensors[0].device).wait_event(self.event)
        return self.tensors


class CoalescedFuture(Future):
    def result(self, timeout=None):
        r = super().result(timeout)
        return r.get() if isinstance(r, _Result) else r


def plan_batches(row_counts, max_rows):
    """Host logic, testable without a GPU: greedy first-fit IN ARRIVAL ORDER — a batch is the longest prefix of the waiting
    requests whose rows fit `max_rows`; a request that alone exceeds `max_rows` forms its own batch.  Returns a list of
    index lists."""
    out, cur, rows = [], [], 0
    for i, n in enumerate(row_counts):
        if cur and rows + n > max_rows:
            out.append(cur)
            cur, rows = [], 0
        cur.append(i)
        rows += n
    if cur:
        out.append(cur)
    return out


def pad_and_cat(requests, t_multiple=1):
    """requests: [(prev (n_i, T_i) int64, plen (n_i,) or (n_i, 1), *rest)] -> (prev (N, Tmax) zero-padded, plen (N,), *rest
    concatenated along dim 0), and the row offsets.  Tmax is rounded up to a multiple of `t_multiple` (fewer distinct
    workspace shapes behind the decoder; the persistent launch wants an even caption length)."""
    tmax = max(r[0].shape[1] for r in requests)
    tmax = -(-tmax // t_multiple) * t_multiple
    prevs = []
    for r in requests:
        p = r[0]
        if p.shape[1] < tmax:
            p = torch.cat([p, p.new_zeros(p.shape[0], tmax - p.shape[1])], 1)
        prevs.append(p)
    prev = torch.cat(prevs, 0)
    plen = torch.cat([r[1].reshape(-1) for r in requests], 0)
    rest = [torch.cat([r[k] for r in requests], 0) for k in range(2, len(requests[0]))]
    offs = [0]
    for r in requests:
        offs.append(offs[-1] + r[0].shape[0])
    return (prev, plen, *rest), offs


class RequestCoalescer:
    def __init__(self, decode_fn, max_rows=16, window_s=0.0, device=None, t_multiple=2):
        """decode_fn(prev, plen, *rest) -> tuple of tensors whose dim 0 is the batch (e.g. (seq, seq_logp)); called under
        torch.no_grad() on the worker's own stream.  window_s: how long the worker waits for more requests once it holds
        one (0 = take what is waiting right now: adds no latency to a lone request).  t_multiple: previous captions are
        padded with <pad> to a multiple of this length (2: the persistent launch takes even lengths only; a larger value
        bounds the number of distinct workspace shapes a long-running server accumulates)."""
        self.decode_fn, self.max_rows, self.window_s = decode_fn, int(max_rows), float(window_s)
        self.t_multiple = max(1, int(t_multiple))
        self.device = device
        self._q, self._cv, self._stop = [], threading.Condition(), False
        self.batches, self.requests = 0, 0            # counters: how many decodes served how many requests
        self._th = threading.Thread(target=self._run, daemon=True, name="set-coalescer")
        self._th.start()

    def submit(self, prev, plen, *rest):
        fut = CoalescedFuture()
        ev = None
        if prev.is_cuda:                               # the request's tensors may still be in flight on the caller's stream
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(prev.device))
        with self._cv:
            if self._stop:
                raise RuntimeError("RequestCoalescer is closed")
            self._q.append(((prev, plen) + tuple(rest), ev, fut))
            self._cv.notify()
        return fut

    def close(self):
        with self._cv:
            self._stop = True
            self._cv.notify()
        self._th.join()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _take(self):
        """the next batch: the oldest request + what fits behind it, waiting up to window_s for more"""
        with self._cv:
            while not self._q and not self._stop:
                self._cv.wait()
            if not self._q:
                return None
            deadline = time.monotonic() + self.window_s
            while self.window_s > 0 and not self._stop:
                rows = sum(r[0][0].shape[0] for r in self._q)
                left = deadline - time.monotonic()
                if rows >= self.max_rows or left <= 0:
                    break
                self._cv.wait(left)
            first = plan_batches([r[0][0].shape[0] for r in self._q], self.max_rows)[0]
            batch = [self._q[i] for i in first]
            del self._q[:len(first)]
            return batch

    def _run(self):
        stream = None
        while True:
            batch = self._take()
            if batch is None:
                return
            try:
                reqs = [b[0] for b in batch]
                cuda = reqs[0][0].is_cuda
                if cuda and stream is None:
                    stream = torch.cuda.Stream(reqs[0][0].device)
                ctx = torch.cuda.stream(stream) if cuda else _null()
                with ctx, torch.no_grad():
                    for _, ev, _ in batch:
                        if ev is not None:
                            stream.wait_event(ev)
                    if len(reqs) == 1 and reqs[0][0].shape[1] % self.t_multiple == 0:
                        args, offs = (reqs[0][0], reqs[0][1].reshape(-1)) + tuple(reqs[0][2:]), [0, reqs[0][0].shape[0]]
                    else:
                        args, offs = pad_and_cat(reqs, self.t_multiple)
                    outs = self.decode_fn(*args)
                    outs = tuple(outs) if isinstance(outs, (tuple, list)) else (outs,)
                    done = None
                    if cuda:
                        done = torch.cuda.Event()
                        done.record(stream)
                self.batches += 1
                self.requests += len(batch)
                for i, (_, _, fut) in enumerate(batch):
                    fut.set_result(_Result(tuple(o[offs[i]:offs[i + 1]] for o in outs), done))
            except BaseException as e:                 # every caller of the batch sees the failure
                for _, _, fut in batch:
                    if not fut.done():
                        fut.set_exception(e)


class _null:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False
