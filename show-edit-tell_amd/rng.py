"""Random streams of the train-mode path: ONE 64-bit seed per forward call, every consumer at its own Philox offset.

The reference draws from torch's global generator at five kinds of site (`nn.Dropout` in `EmbeddingC` editnet.py:300-304 —
called by the caption encoder :329 and by every timestep :513 —, in `att_embed` :430-432,441, before `fc` :545; the
scheduled-sampling coin and draw :509,517; the multinomial rollout editnet_rl.py:524).  Here all of them are
Philox4x32-10 streams (csrc/philox.h) addressed by (seed, offset(site, t), row, column group): reproducible, independent
of launch geometry and of the route (whole-sequence node or per-operator loop) — and reproducible OUTSIDE this package:
oracle/philox_np.py regenerates every mask, which is how the train-mode forward and backward are pinned to the
reference's own autograd (oracle/make_train_golden.py injects these masks into the reference's Dropout modules).

Addressing (dropout_k / embed_relu_dropout_k in csrc/train_seq.hip): element (row r, column c) of a site's 2-D operand
uses counter (r, c // 4, offset_lo, offset_hi), key (seed_lo, seed_hi), word c % 4; keep iff (word >> 8) * 2^-24 >= p.
Rows are the operand's rows in the decoder's length-sorted batch order:

  site                      t            operand rows
  SITE_ENC_EMBED   (0)      0            b * Tmax + l   embedding of the previous caption, (B * Tmax, E)
  SITE_EMBED       (1)      timestep     b              embedding of the step's input word, (bt, E)
  SITE_REGION      (2)      timestep     b * R + r      relu(att_embed.0(X)), (bt * R, D)
  SITE_OUT         (3)      timestep     b              h2 before fc, (bt, D)
  SITE_SS_DRAW     (4)      timestep     b              scheduled-sampling word draw (sample_pick_k's uniform)
  SITE_ENC2_EMBED  (5)      0            b * Tmax + l   embedding of the ground-truth caption (adaptive model's 2nd encoder pass)
  SITE_SS_COIN     (6)      0            t * B + b      scheduled-sampling coin: replace the word iff u < ss_prob
  SITE_ROLLOUT     (7)      0            b              multinomial rollout (sample_pick_k: counter (b, timestep, offset))
"""
from __future__ import annotations

import contextlib

import torch

SITE_ENC_EMBED, SITE_EMBED, SITE_REGION, SITE_OUT, SITE_SS_DRAW, SITE_ENC2_EMBED, SITE_SS_COIN, SITE_ROLLOUT = range(8)

_forced = None


def offset(site, t=0):
    return (int(site) << 40) | int(t)


def next_seed():
    """the seed of one forward call: drawn from torch's CPU generator (so torch.manual_seed() makes training runs
    reproducible) unless a `dropout_seed(...)` context pins it"""
    if _forced is not None:
        return _forced
    return int(torch.randint(0, 2 ** 62, (1,)).item())


@contextlib.contextmanager
def dropout_seed(seed):
    """pin the seed of every forward call inside the context (tests; bit-reproducible debugging)"""
    global _forced
    old, _forced = _forced, int(seed)
    try:
        yield
    finally:
        _forced = old


def uniforms(n, seed, off, device):
    """(n,) fp32 uniforms in [0, 1): element i = (word 0 of counter (i, 0, off) >> 8) * 2^-24 (set_philox4x32)"""
    from . import _lib
    lib = _lib.load()
    w = torch.empty(n, 4, dtype=torch.int32, device=device)
    _lib.check(lib.set_philox4x32(w.data_ptr(), n, int(seed), int(off), _lib.stream_of(device)), "set_philox4x32")
    return ((w[:, 0].to(torch.int64) & 0xFFFFFFFF) >> 8).to(torch.float32) * (1.0 / 16777216.0)
