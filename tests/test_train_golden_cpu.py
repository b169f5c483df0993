"""CPU: the train-mode goldens (tests/golden/train_*.npz) and the numpy statement of the Philox streams they were made
with.  When /root/reference is present (authoring container) the small cases are regenerated from the reference's own
classes and must come out bit-identical."""
import numpy as np
import pytest

import parity
from oracle import philox_np as PH, ref_slice


def test_dropout_streams_are_distinct_per_site_and_timestep():
    seed = 0x1234_5678_9ABC
    a = PH.dropout_keep(seed, PH.site_offset(PH.SITE_REGION, 0), 64, 256, 0.5)
    assert abs(a.mean() - 0.5) < 0.02
    for other in (PH.site_offset(PH.SITE_REGION, 1), PH.site_offset(PH.SITE_EMBED, 0), PH.site_offset(PH.SITE_OUT, 0)):
        b = PH.dropout_keep(seed, other, 64, 256, 0.5)
        assert abs((a == b).mean() - 0.5) < 0.03                  # independent masks agree on half the elements
    assert np.array_equal(a, PH.dropout_keep(seed, PH.site_offset(PH.SITE_REGION, 0), 64, 256, 0.5))
    # geometry independence: the first rows of a taller operand carry the same mask
    assert np.array_equal(a[:10], PH.dropout_keep(seed, PH.site_offset(PH.SITE_REGION, 0), 10, 256, 0.5))
    for p in (0.1, 0.8):
        assert abs(PH.dropout_keep(seed, 5, 128, 128, p).mean() - (1 - p)) < 0.02


def test_categorical_draw_follows_the_distribution():
    rng = np.random.default_rng(0)
    lg = rng.normal(size=(1, 37)).astype(np.float32) * 2
    p = np.exp(lg[0] - lg[0].max())
    p /= p.sum()
    counts = np.zeros(37)
    for t in range(4000):
        ids, _ = PH.categorical_draw(lg, 99, PH.site_offset(PH.SITE_SS_DRAW, t))
        counts[ids[0]] += 1
    chi2 = ((counts - 4000 * p) ** 2 / (4000 * p)).sum()
    assert chi2 < 80, chi2                                        # 36 degrees of freedom


@pytest.mark.parametrize("name", ["editnet_small", "editnet_full_b4", "editnet_full_b128", "editnet_adaptive_small", "dcnet_small",
                                  "dcnet_full_b4"])
def test_train_goldens_present_and_consistent(name):
    g = parity.load("train_" + name)
    assert int(g["train.seed"]) > 2 ** 32                         # exercises the high key word
    norms = [k for k in g if k.startswith("train.gradnorm.")]
    assert len(norms) >= 27 and all(np.isfinite(g[k]) for k in norms)
    assert float(g["train.loss"]) > 0
    # train mode differs from the eval-mode golden of the same case (the masks really acted)
    ge = parity.load(name)
    assert abs(float(g["train.loss"]) - float(ge["grad_loss"])) > 1e-3
    if name == "editnet_full_b128":
        # tied caption lengths: the reference's unstable sort and the package's stable one order the rows differently inside
        # a group of equal lengths (the golden stores both; scores are in the package's order)
        a, b = g["train.sort_ind"], g["train.ref_sort_ind"]
        assert not np.array_equal(a, b) and sorted(a.tolist()) == sorted(b.tolist()) == list(range(128))
    if name == "editnet_small":
        assert int(g["train_ss.n_replaced"]) >= 10 and float(g["train_ss.draw_margin_min"]) > 2e-5
        assert g["train_ss.fed_tokens"].shape == (19, 6)


@pytest.mark.skipif(not ref_slice.have_reference(), reason="needs /root/reference (authoring container)")
@pytest.mark.parametrize("name", ["editnet_small", "dcnet_small"])
def test_train_goldens_regenerate_bit_identically(name):
    from oracle import cases, make_train_golden as M
    g = parity.load("train_" + name)
    out = {}
    for n, seed, ss in M.TRAIN_CASES:
        if n == name:
            out.update(M.make_dcnet(n, seed) if n in cases.DCNET_CASES else M.make_editnet(n, seed, ss))
    assert set(out) == set(g)
    for k in g:
        assert np.array_equal(np.asarray(out[k]), g[k]), k
