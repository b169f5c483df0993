"""GPU: the two ways the free-running loops stop doing work (csrc/set_common.h RowGate).

* the reference's `break` (editnet_rl.py:546, dcnet_rl.py:343) happens in TIME: after every row has finished, the kernels of
  the remaining timesteps return at once.  Outputs are unchanged — the goldens with early-finishing rows
  (tests/test_hip_editnet.py / test_hip_dcnet.py `*_small_end`) run with the gate on; here the gate is also switched off in a
  child process and the two results must be bit-identical.
* opt-in `decoder.skip_finished_rows = True` (set_decode_options): rows whose caption has ended leave the computation — row
  kernels skip them, the GEMMs walk the compacted list of unfinished rows.  Token ids are bit-identical, and so is every
  log-prob up to and including a row's <end>; behind it the reference records the log-prob of whatever the row's ghost
  decode (fed word 0) would pick — masked by RewardCriterion (editnet_rl.py:563-566) — and the skipping loop leaves 0."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from hip_adapter import dcnet_modules, editnet_modules, to_dev

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _per_step_loop(monkeypatch):
    """These tests compare the per-step loop with and without row skipping BIT for bit; the persistent small-batch launch
    (tests/test_hip_persistent_decode.py) adds the same products in another order and never skips rows."""
    monkeypatch.setenv("SET_DEC_PERSISTENT", "0")


def _boosted(rl, wm, boost):
    with torch.no_grad():
        rl.fc.bias[int(wm["<end>"])] += boost
    return rl


def _lengths(seq):
    """decode steps a row needs: position of its first 0 (= <end>) + 1, or max_len"""
    z = (seq == 0)
    first = torch.where(z.any(1), z.float().argmax(1), torch.full((seq.shape[0],), seq.shape[1] - 1, device=seq.device))
    return first + 1


def _check_skip_vs_exact(exact, skip):
    seq_e, lp_e = exact
    seq_s, lp_s = skip
    assert torch.equal(seq_e, seq_s), "token ids must be bit-identical"
    n = _lengths(seq_e)
    pos = torch.arange(seq_e.shape[1], device=seq_e.device)[None, :]
    live = pos < n[:, None]                                   # up to and including each row's <end>
    assert torch.equal(lp_e[live], lp_s[live]), "log-probs up to each row's <end> must be bit-identical"
    assert not lp_s[~live].any(), "behind a row's <end> the skipping loop leaves 0"
    return n


@pytest.mark.parametrize("name,boost", [("editnet_full_b128", None), ("editnet_small", 6.0), ("editnet_full_b4", 5.5)])
def test_editnet_skip_finished_rows(name, boost):
    d, xe, rl = editnet_modules(name)
    wm = d["wm"]
    args = (wm, to_dev(d["prev"]), to_dev(d["plen"]), to_dev(d["X"]), True, False)
    if boost is None:
        # the metric batch: find the <end> bias that spreads the finish times (greedy <end> is an arg-max event, so the
        # spread is a property of the random weights: scan for the boost whose mean is closest to half the steps)
        best = None
        with torch.no_grad():
            for b10 in range(40, 70, 2):
                _boosted(rl, wm, b10 / 10.0)
                m = float(_lengths(rl(*args)[0]).float().mean())
                _boosted(rl, wm, -b10 / 10.0)
                if best is None or abs(m - 9.5) < abs(best[1] - 9.5):
                    best = (b10 / 10.0, m)
        boost = best[0]
        print("boost", best)
    _boosted(rl, wm, boost)
    with torch.no_grad():
        rl(*args)
        rl(*args)                                             # (the folded token table is active from here on: both modes
        for rep in range(2):                                  #  are compared under the same summation order)
            rl.skip_finished_rows = False
            exact = rl(*args)
            rl.skip_finished_rows = True
            skip = rl(*args)
            torch.cuda.synchronize()
            n = _check_skip_vs_exact(exact, skip)
    n = n.cpu().numpy()
    print(name, "steps per row: mean %.1f min %d max %d of %d" % (n.mean(), n.min(), n.max(), exact[0].shape[1]))
    if d["case"]["B"] >= 64:                                  # the case must exercise mixed finish times and shrinking tiles
        assert n.min() <= 6 and n.max() >= 12 and 0.2 < (n <= 9).mean() < 0.9, n
    # multinomial sampling through the same gate (Philox draws are addressed by (row, timestep): skipping rows moves nothing)
    from show_edit_tell_amd import rng
    with torch.no_grad():
        outs = []
        for flag in (False, True):
            rl.skip_finished_rows = flag
            torch.manual_seed(5)
            outs.append(rl(wm, args[1], args[2], args[3], False, True))
        torch.cuda.synchronize()
    _check_skip_vs_exact(outs[0], outs[1])


@pytest.mark.parametrize("name,boost", [("dcnet_full_b128", 5.5), ("dcnet_small", 6.0)])
def test_dcnet_skip_finished_rows(name, boost):
    d, xe, rl = dcnet_modules(name)
    wm = d["wm"]
    _boosted(rl, wm, boost)
    args = (wm, to_dev(d["prev"]), to_dev(d["plen"]), True, False)
    with torch.no_grad():
        rl(*args)
        rl(*args)
        for rep in range(2):
            rl.skip_finished_rows = False
            exact = rl(*args)
            rl.skip_finished_rows = True
            skip = rl(*args)
            torch.cuda.synchronize()
            n = _check_skip_vs_exact(exact, skip)
    print(name, "steps per row: mean %.1f max %d" % (float(n.float().mean()), int(n.max())))


_GATE_SCRIPT = r"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.getcwd(), "tests")); sys.path.insert(0, os.getcwd())
from hip_adapter import editnet_modules, to_dev
d, xe, rl = editnet_modules("editnet_small_end")
args = (d["wm"], to_dev(d["prev"]), to_dev(d["plen"]), to_dev(d["X"]), True, False)
with torch.no_grad():
    for _ in range(3):
        seq, lp = rl(*args)
torch.cuda.synchronize()
np.savez(sys.argv[1], seq=seq.cpu().numpy(), lp=lp.cpu().numpy())
"""


def test_loop_gate_changes_no_output(tmp_path):
    """SET_LOOP_GATE=0 (the kernels of the timesteps behind the break run and discard, as before round 4) vs the default:
    bit-identical seq / seqLogprobs on the case where every row finishes within a few steps."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for gate in ("1", "0"):
        f = os.path.join(str(tmp_path), "g%s.npz" % gate)
        r = subprocess.run([sys.executable, "-c", _GATE_SCRIPT, f], env=dict(os.environ, SET_LOOP_GATE=gate), cwd=root,
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(dict(np.load(f)))
    assert np.array_equal(outs[0]["seq"], outs[1]["seq"]) and np.array_equal(outs[0]["lp"], outs[1]["lp"])
    assert (outs[0]["seq"][:, -1] == 0).all()
