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
