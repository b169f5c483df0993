"""ORACLE (test infrastructure): Philox4x32-10 counter-based RNG in numpy (Salmon, Moraes, Dror, Shaw:
"Parallel random numbers: as easy as 1, 2, 3", SC'11) — the generator behind the HIP sampling epilogue
(csrc/epilogue.hip `sample_pick_k`).  Pinned by the published known-answer vectors of the Random123
distribution (tests/test_sampling_cpu.py) and compared word for word with the device implementation
(tests/test_hip_sampling.py).  Only tests/ import this module."""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(counters, key):
    """counters (n,4) uint32, key (k0,k1) -> (n,4) uint32"""
    c = np.asarray(counters, dtype=np.uint64).copy()
    k0, k1 = int(key[0]) & 0xFFFFFFFF, int(key[1]) & 0xFFFFFFFF
    for _ in range(10):
        p0 = M0 * c[:, 0]
        p1 = M1 * c[:, 2]
        hi0, lo0 = p0 >> np.uint64(32), p0 & MASK
        hi1, lo1 = p1 >> np.uint64(32), p1 & MASK
        n0 = hi1 ^ c[:, 1] ^ np.uint64(k0)
        n2 = hi0 ^ c[:, 3] ^ np.uint64(k1)
        c = np.stack([n0, lo1, n2, lo0], 1)
        k0 = (k0 + W0) & 0xFFFFFFFF
        k1 = (k1 + W1) & 0xFFFFFFFF
    return c.astype(np.uint32)


def sample_uniform(seed, offset, rows, t):
    """the epilogue's uniform for (row, timestep): counter (row, t, offset_lo, offset_hi), key (seed_lo, seed_hi),
    u = (word0 >> 8) * 2^-24"""
    rows = np.asarray(rows, dtype=np.uint64)
    ctr = np.stack([rows, np.full_like(rows, t), np.full_like(rows, offset & 0xFFFFFFFF),
                    np.full_like(rows, (offset >> 32) & 0xFFFFFFFF)], 1)
    w = philox4x32_10(ctr, (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF))
    return ((w[:, 0] >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0))


# ---- the train-mode streams of show_edit_tell_amd/rng.py (dropout masks, scheduled-sampling coin and draw) ----------
SITE_ENC_EMBED, SITE_EMBED, SITE_REGION, SITE_OUT, SITE_SS_DRAW, SITE_ENC2_EMBED, SITE_SS_COIN, SITE_ROLLOUT = range(8)


def site_offset(site, t=0):
    return (int(site) << 40) | int(t)


def _key(seed):
    return (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)


def dropout_keep(seed, offset, rows, cols, p):
    """keep mask (rows, cols) bool of csrc/train_seq.hip `dropout_k`: element (r, c) uses counter
    (r, c // 4, offset_lo, offset_hi), key seed, word c % 4; kept iff (word >> 8) * 2^-24 >= p."""
    assert cols % 4 == 0
    c4 = cols // 4
    r = np.repeat(np.arange(rows, dtype=np.uint64), c4)
    c = np.tile(np.arange(c4, dtype=np.uint64), rows)
    ctr = np.stack([r, c, np.full_like(r, offset & 0xFFFFFFFF), np.full_like(r, (offset >> 32) & 0xFFFFFFFF)], 1)
    w = philox4x32_10(ctr, _key(seed))                       # (rows * c4, 4)
    u = (w >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)
    return (u >= np.float32(p)).reshape(rows, cols)


def uniforms(n, seed, offset):
    """element i: (word 0 of counter (i, 0, offset) >> 8) * 2^-24 — show_edit_tell_amd.rng.uniforms"""
    i = np.arange(n, dtype=np.uint64)
    ctr = np.stack([i, np.zeros_like(i), np.full_like(i, offset & 0xFFFFFFFF), np.full_like(i, (offset >> 32) & 0xFFFFFFFF)], 1)
    w = philox4x32_10(ctr, _key(seed))
    return (w[:, 0] >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)


def categorical_draw(logits, seed, offset, t=0, with_alt=False):
    """the sampling epilogue's draw (csrc/epilogue.hip `sample_pick_k`) for every row of `logits` (B, V): inverse CDF of
    softmax(logits) over the kernel's fixed enumeration of the vocabulary (thread tid of 256 owns the words
    (tid + 256 q) * 4 + e, q = 0.., e = 0..3, in that order), uniform of counter (row, t, offset).  Returns (ids (B,),
    margin (B,)): margin = distance of the target from the nearest CDF boundary, relative to the total mass — draws with
    a margin below ~1e-5 may legitimately differ between the fp32 device scan and this fp64 one.  with_alt: also return
    alt (B,) = the word on the other side of that nearest boundary — the only word such a draw can legitimately become."""
    B, V = logits.shape
    nq = -(-V // 1024)
    order = np.array([(tid + 256 * q) * 4 + e for tid in range(256) for q in range(nq) for e in range(4)])
    order = order[order < V]
    lg = logits.astype(np.float64)
    pe = np.exp(lg - lg.max(1, keepdims=True))[:, order]
    cdf = np.cumsum(pe, 1)
    u = sample_uniform(seed, offset, np.arange(B), t).astype(np.float64)
    target = u * cdf[:, -1]
    ids, margin, alt = np.zeros(B, np.int64), np.zeros(B), np.zeros(B, np.int64)
    for b in range(B):
        j = int(np.searchsorted(cdf[b], target[b], side="right"))
        j = min(j, len(order) - 1)
        ids[b] = order[j]
        lo = cdf[b, j - 1] if j else 0.0
        margin[b] = min(target[b] - lo, cdf[b, j] - target[b]) / cdf[b, -1]
        below = (target[b] - lo) < (cdf[b, j] - target[b])
        alt[b] = order[max(j - 1, 0)] if below else order[min(j + 1, len(order) - 1)]
    return (ids, margin, alt) if with_alt else (ids, margin)
