"""CPU: the oracle's Philox4x32-10 against the known-answer vectors published with Random123
(examples/kat_vectors: philox4x32 10 rounds)."""
import numpy as np

from oracle import philox_np


def test_philox4x32_10_known_answers():
    kat = [
        ((0x00000000, 0x00000000, 0x00000000, 0x00000000), (0x00000000, 0x00000000),
         (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
        ((0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff), (0xffffffff, 0xffffffff),
         (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
        ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
         (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
    ]
    for ctr, key, want in kat:
        got = philox_np.philox4x32_10(np.array([ctr], dtype=np.uint32), key)[0]
        assert tuple(int(x) for x in got) == want, (ctr, key, [hex(int(x)) for x in got])


def test_sample_uniform_range_and_independence():
    u = philox_np.sample_uniform(1234567, 3, np.arange(20000), 5)
    assert u.min() >= 0.0 and u.max() < 1.0
    assert abs(u.mean() - 0.5) < 0.01 and abs(u.var() - 1 / 12) < 0.005
    v = philox_np.sample_uniform(1234567, 4, np.arange(20000), 5)
    assert abs(np.corrcoef(u, v)[0, 1]) < 0.03
