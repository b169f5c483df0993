"""Shared checker for beam-search results against tests/golden/beam_*.npz (outputs of the reference's own
evaluate() / evaluate_full() loops, oracle/make_beam_golden.py)."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SCORE_TOL = 1e-3          # a sum of <= 50 log-probabilities, each within 1e-4 / 2
MARGIN_MIN = 2e-3         # the two best completed hypotheses may swap below 2 * SCORE_TOL


def load(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


def check_one(g, k, model, b, seq, score=None):
    """One image's result (token list incl. <start>/<end>, optional score) against the reference's.
    Finished searches with a comfortable margin between the two best completed hypotheses: the sequence must be
    IDENTICAL and the score within SCORE_TOL.  Searches that ran into the 50-step limit (editnet.py:702-704,711:
    the answer is seqs[0][:18], a chaotic 50-step trajectory): length 18 and the same first 4 tokens.
    Returns 1 if the strict comparison was made."""
    pre = "k%d.%s." % (k, model)
    ref = g[pre + "seq"][b][: int(g[pre + "n"][b])].tolist()
    seq = [int(w) for w in seq]
    if bool(g[pre + "infinite"][b]):
        assert len(seq) == 18 and seq[:4] == ref[:4], (model, k, b, seq, ref)
        return 0
    if score is not None and not (isinstance(score, float) and np.isnan(score)):
        assert abs(float(score) - float(g[pre + "score"][b])) < SCORE_TOL, (model, k, b, score, float(g[pre + "score"][b]))
    if float(g[pre + "margin"][b]) <= MARGIN_MIN:
        return 0
    assert seq == ref, (model, k, b, seq, ref)
    return 1
