"""CPU: the oracle's numpy beam search (oracle/beam_np.py) against the golden vectors produced by the
REFERENCE's own beam loops (`editnet.py:603-713`, `dcnet.py:413-514`, `eval/eval xe/eval_full.py:96-210`,
run through oracle/ref_beam.py in the authoring container).  This is what pins beam-search parity; the GPU
tests (tests/test_hip_beam.py) compare the HIP search with the same fixtures."""
import numpy as np
import pytest

import beam_parity
from oracle import beam_np, cases, dcnet_np as DN, editnet_np as EN


@pytest.mark.parametrize("name", ["beam_small_e3", "beam_small_e5"])
def test_oracle_beam_vs_reference_beam(name):
    d = cases.build_beam(name)
    g = beam_parity.load(name)
    wm, B = d["wm"], d["case"]["B"]
    Pe, Pd = EN.cast_params(d["sd_e"]), DN.cast_params(d["sd_d"])
    firm = 0
    for k in d["beams"]:
        for b in range(B):
            X1, prev1, plen1 = d["X"][b:b + 1], d["prev"][b:b + 1], d["plen"][b:b + 1]
            res = {"editnet": beam_np.beam_editnet(Pe, X1, prev1, plen1, wm["<start>"], wm["<end>"], k),
                   "dcnet": beam_np.beam_dcnet(Pd, prev1, plen1, wm["<start>"], wm["<end>"], k),
                   "ensemble": beam_np.beam_ensemble(Pe, Pd, X1, prev1, plen1, wm["<start>"], wm["<end>"], k)}
            for model, (seq, score, _) in res.items():
                firm += beam_parity.check_one(g, k, model, b, seq, score)
    assert firm >= 6


def test_oracle_beam_full_size_vs_reference_beam():
    d = cases.build_beam("beam_full_b4")
    g = beam_parity.load("beam_full_b4")
    wm = d["wm"]
    Pe = EN.cast_params(d["sd_e"])
    firm = 0
    for b in range(d["case"]["B"]):
        seq, score, _ = beam_np.beam_editnet(Pe, d["X"][b:b + 1], d["prev"][b:b + 1], d["plen"][b:b + 1], wm["<start>"],
                                             wm["<end>"], 3)
        firm += beam_parity.check_one(g, 3, "editnet", b, seq, score)
    assert firm >= 2
