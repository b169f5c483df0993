"""GPU: the persistent small-batch decode (csrc/decode_persistent.hip — the whole greedy loop as ONE launch, flag-in-data
exchange between workgroups) against the reference's golden and against the per-step loop it replaces."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import parity
from hip_adapter import dcnet_modules, editnet_modules, to_dev
from show_edit_tell_amd import _lib

pytestmark = pytest.mark.gpu


def _np(t):
    return t.detach().cpu().numpy()


def _tags(fn):
    lib = _lib.load()
    lib.set_profile_enable(1)
    fn()
    torch.cuda.synchronize()
    names = [r["tag"] for r in _lib.profile_report()]
    lib.set_profile_enable(0)
    return names


def _random_prev(B, T, seed, V=9000):
    rs = np.random.RandomState(seed)
    plen = rs.randint(1, T + 1, size=(B, 1)).astype(np.int64)
    prev = rs.randint(4, V, size=(B, T)).astype(np.int64)
    for i in range(B):
        prev[i, plen[i, 0]:] = 0
    return to_dev(prev), to_dev(plen)


def _with_env(key, val, fn):
    old = os.environ.get(key)
    os.environ[key] = val
    try:
        return fn()
    finally:
        if old is None:
            del os.environ[key]
        else:
            os.environ[key] = old


def test_dcnet_persistent_decode_matches_golden_and_per_step():
    """BASELINE.json configs[0]'s shape (DCNet greedy, B = 4, full dimensions): with the token table active the decode is the
    persistent launch (profile tag `persistent_decode`); ids / log-probs equal the reference's golden, and ids equal the
    per-step loop's (SET_DEC_PERSISTENT=0) with log-probs within 1e-5 — the two paths add the same products in a
    different order."""
    d, xe, rl = dcnet_modules("dcnet_full_b4")
    g = parity.load("dcnet_full_b4")
    prev, plen = to_dev(d["prev"]), to_dev(d["plen"])
    with torch.no_grad():
        rl(d["wm"], prev, plen, True, False)
        rl(d["wm"], prev, plen, True, False)                      # token table from the second call on
        names = _tags(lambda: rl(d["wm"], prev, plen, True, False))
        assert "persistent_decode" in names, names
        seq, logp = rl(d["wm"], prev, plen, True, False)
        torch.cuda.synchronize()
        parity.check_greedy(_np(seq), _np(logp), g)
        ref = _with_env("SET_DEC_PERSISTENT", "0", lambda: rl(d["wm"], prev, plen, True, False))
        names = _with_env("SET_DEC_PERSISTENT", "0", lambda: _tags(lambda: rl(d["wm"], prev, plen, True, False)))
        assert "persistent_decode" not in names
    assert torch.equal(seq, ref[0])
    assert float((logp - ref[1]).abs().max()) < 1e-5


def _check_against_oracle_and_per_step(model, d, prev, plen, X, got, ref):
    """`got` = the persistent launch, `ref` = the per-step loop, both (seq, logp) device tensors.  The numpy oracle decodes the
    same inputs on the host (its per-step logits give every row's top-1 / top-2 margin): rows without a near-tie must be
    bit-identical to the oracle AND between the two HIP paths; a differing row must be a demonstrated near-tie whose token is
    one of the oracle's two best candidates (parity.check_greedy_rows)."""
    seq_o, logp_o, margins, top2 = parity.oracle_greedy_reference(model, d["sd"], d["wm"], _np(prev), _np(plen),
                                                                  None if X is None else _np(X))
    end = int(d["wm"]["<end>"])
    n_amb = parity.check_greedy_rows(_np(got[0]), _np(got[1]), seq_o, logp_o, margins, top2, end)
    parity.check_greedy_rows(_np(ref[0]), _np(ref[1]), seq_o, logp_o, margins, top2, end)
    parity.check_two_paths_rows(_np(got[0]), _np(got[1]), _np(ref[0]), _np(ref[1]), margins)
    return n_amb


@pytest.mark.parametrize("B", [1, 3, 5, 6, 7, 8])
def test_dcnet_persistent_decode_other_batch_sizes(B):
    """1 .. 8 rows (one 16-row MFMA tile, 1 - 2 rows per wave in the attention phase, ragged previous captions incl. length
    1), all T = 19 timesteps: the persistent launch against the NUMPY ORACLE and the per-step loop on the same inputs.  No
    row is excused without evidence: rows whose oracle margins stay >= 2.5e-4 are bit-identical everywhere."""
    d, xe, rl = dcnet_modules("dcnet_full_b4")
    prev, plen = _random_prev(B, d["prev"].shape[1], 100 + B)
    with torch.no_grad():
        for _ in range(2):
            rl(d["wm"], prev, plen, True, False)
        names = _tags(lambda: rl(d["wm"], prev, plen, True, False))
        assert "persistent_decode" in names, names
        seq, logp = rl(d["wm"], prev, plen, True, False)
        again = rl(d["wm"], prev, plen, True, False)
        ref = _with_env("SET_DEC_PERSISTENT", "0", lambda: rl(d["wm"], prev, plen, True, False))
        torch.cuda.synchronize()
    assert torch.equal(seq, again[0]) and torch.equal(logp, again[1]), "the persistent decode must be run-to-run deterministic"
    _check_against_oracle_and_per_step("dcnet", d, prev, plen, None, (seq, logp), ref)
    assert torch.isfinite(logp).all()


@pytest.mark.parametrize("model", ["dcnet", "editnet"])
def test_persistent_decode_many_random_inputs(model):
    """64 rows of random inputs (16 batches of 4): the persistent launch scores the attention with tanh = 1 - 2 / (1 + e^2x)
    on the hardware exp2 / rcp and adds every product in another order than the per-step loop.  Every batch is decoded by the
    numpy oracle too: greedy ids identical on every row without a demonstrated near-tie (oracle margins), log-probs within
    1e-5 between the two HIP paths and 1e-4 against the oracle; near-tie rows stay a small minority."""
    if model == "dcnet":
        d, xe, rl = dcnet_modules("dcnet_full_b4")
    else:
        d, xe, rl = editnet_modules("editnet_full_b4")
    T = d["prev"].shape[1]
    n_amb = 0
    with torch.no_grad():
        for i in range(16):
            prev, plen = _random_prev(4, T, 1000 + i)
            X = None
            if model == "dcnet":
                args = (d["wm"], prev, plen, True, False)
            else:
                X = to_dev(np.abs(np.random.RandomState(2000 + i).randn(4, d["X"].shape[1], d["X"].shape[2])).astype(np.float32))
                args = (d["wm"], prev, plen, X, True, False)
            rl(*args)                                              # (first call: builds / keeps the token table)
            got = rl(*args)
            ref = _with_env("SET_DEC_PERSISTENT", "0", lambda: rl(*args))
            torch.cuda.synchronize()
            n_amb += _check_against_oracle_and_per_step(model, d, prev, plen, X, got, ref)
    assert n_amb <= 3, "near-tie rows should be rare among 64 random rows: %d" % n_amb


def test_dcnet_persistent_decode_on_concurrent_streams():
    """Eight decodes on four streams: the library serialises persistent launches with its event chain (their workgroups
    must all be resident), nothing hangs, no exchange times out and every result equals the single-stream decode."""
    d, xe, rl = dcnet_modules("dcnet_full_b4")
    prev, plen = to_dev(d["prev"]), to_dev(d["plen"])
    with torch.no_grad():
        for _ in range(2):
            rl(d["wm"], prev, plen, True, False)
        ref = rl(d["wm"], prev, plen, True, False)
        torch.cuda.synchronize()
        streams = [torch.cuda.Stream() for _ in range(4)]
        outs = []
        for rep in range(2):
            for s in streams:
                s.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(s):
                    outs.append(rl(d["wm"], prev, plen, True, False))
        torch.cuda.synchronize()
    for seq, lp in outs:
        assert torch.equal(seq, ref[0]) and torch.equal(lp, ref[1])


def test_dcnet_persistent_xe_forward_matches_golden_and_per_step():
    """BASELINE.json configs[0] names the teacher-forced forward too: with the token table active `DAE.forward` under no_grad
    at B <= 8 is the same persistent launch in teacher-forced mode (words from the captions, the scores of the first
    batch_size_t rows written per timestep, dcnet.py:333-348).  Against the golden, against the per-step loop (same scores
    within 2e-5, identical zero pattern behind each caption's length), ragged lengths included."""
    d, xe, rl = dcnet_modules("dcnet_full_b4")
    c, g = d["case"], parity.load("dcnet_full_b4")
    prev, plen, caps, clen = (to_dev(d[k]) for k in ("prev", "plen", "caps", "clen"))
    with torch.no_grad():
        xe(caps, clen, prev, plen)
        xe(caps, clen, prev, plen)
        names = _tags(lambda: xe(caps, clen, prev, plen))
        assert "persistent_decode" in names, names
        pred, caps_s, dl, sort_ind = xe(caps, clen, prev, plen)
        parity.check_xe(_np(pred), dl, _np(sort_ind), g, c["V"], small=False)
        ref = _with_env("SET_DEC_PERSISTENT", "0", lambda: xe(caps, clen, prev, plen))
        torch.cuda.synchronize()
    assert dl == ref[2] and torch.equal(sort_ind, ref[3])
    assert torch.equal(pred == 0, ref[0] == 0), "rows behind a caption's length stay zero in both paths"
    assert float((pred - ref[0]).abs().max()) < 2e-5 * max(1.0, float(ref[0].abs().max()))


@pytest.mark.parametrize("name", ["editnet_full_b4", "editnet_full_v9490"])
def test_editnet_persistent_decode_matches_golden_and_per_step(name):
    """EditNet greedy at full dimensions (csrc/decode_persistent_wide.hip): B = 4 (one row per wave, its attention rows
    resident in registers) and the B = 5 / V = 9490 fixture (two rows on one wave, streamed attention rows, a vocabulary that is
    no multiple of the 256 workgroups): the persistent launch equals the reference's golden and the per-step loop (ids
    bit-identical, log-probs within 1e-5)."""
    d, xe, rl = editnet_modules(name)
    g = parity.load(name)
    args = (d["wm"], to_dev(d["prev"]), to_dev(d["plen"]), to_dev(d["X"]), True, False)
    with torch.no_grad():
        rl(*args)
        rl(*args)
        names = _tags(lambda: rl(*args))
        assert "persistent_decode" in names, names
        seq, logp = rl(*args)
        torch.cuda.synchronize()
        parity.check_greedy(_np(seq), _np(logp), g)
        ref = _with_env("SET_DEC_PERSISTENT", "0", lambda: rl(*args))
        names = _with_env("SET_DEC_PERSISTENT", "0", lambda: _tags(lambda: rl(*args)))
        assert "persistent_decode" not in names
    assert torch.equal(seq, ref[0])
    assert float((logp - ref[1]).abs().max()) < 1e-5


@pytest.mark.parametrize("name", ["editnet_full_b4", "editnet_full_v9490"])
def test_editnet_persistent_xe_forward_matches_golden_and_per_step(name):
    """The teacher-forced forward of EditNet under no_grad at B <= 16 (editnet.py:505-546) on the persistent launch: golden,
    per-step loop (scores within 2e-5, the same zero pattern behind each caption's length).  B = 4: the <= 4-row kernel;
    the B = 5 / V = 9490 fixture: the wide variant in teacher-forced mode."""
    d, xe, rl = editnet_modules(name)
    c, g = d["case"], parity.load(name)
    args = (to_dev(d["X"]), to_dev(d["caps"]), to_dev(d["clen"]), to_dev(d["prev"]), to_dev(d["plen"]), False, 0.0)
    with torch.no_grad():
        xe(*args)
        xe(*args)
        names = _tags(lambda: xe(*args))
        assert "persistent_decode" in names, names
        pred, caps_s, dl, sort_ind = xe(*args)
        parity.check_xe(_np(pred), dl, _np(sort_ind), g, c["V"], small=False)
        ref = _with_env("SET_DEC_PERSISTENT", "0", lambda: xe(*args))
        torch.cuda.synchronize()
    assert dl == ref[2] and torch.equal(sort_ind, ref[3])
    assert torch.equal(pred == 0, ref[0] == 0)
    assert float((pred - ref[0]).abs().max()) < 2e-5 * max(1.0, float(ref[0].abs().max()))


@pytest.mark.parametrize("B", [1, 2, 3, 5, 6, 7, 8, 9, 12, 13, 16])
def test_editnet_persistent_decode_other_batch_sizes(B):
    """1 .. 16 rows, random features and ragged previous captions, all T = 19 timesteps: the persistent launch against the NUMPY
    ORACLE and the per-step loop, row by row with the oracle's margins (see the DCNet twin above), and run-to-run
    determinism.  From 5 rows on the launch is the wide variant (csrc/decode_persistent_wide.hip: one activation buffer,
    attention scores spread over the grid, a full 16-row MFMA tile at B = 16)."""
    d, xe, rl = editnet_modules("editnet_full_b4")
    prev, plen = _random_prev(B, d["prev"].shape[1], 200 + B)
    X = to_dev(np.abs(np.random.RandomState(300 + B).randn(B, d["X"].shape[1], d["X"].shape[2])).astype(np.float32))
    args = (d["wm"], prev, plen, X, True, False)
    with torch.no_grad():
        for _ in range(2):
            rl(*args)
        names = _tags(lambda: rl(*args))
        assert "persistent_decode" in names, names
        seq, logp = rl(*args)
        again = rl(*args)
        ref = _with_env("SET_DEC_PERSISTENT", "0", lambda: rl(*args))
        torch.cuda.synchronize()
    assert torch.equal(seq, again[0]) and torch.equal(logp, again[1])
    _check_against_oracle_and_per_step("editnet", d, prev, plen, X, (seq, logp), ref)
    assert torch.isfinite(logp).all()


def test_editnet_persistent_decode_early_finish_and_begun():
    """Rows that emit <end> early (the `editnet_small_end`-style bookkeeping at full size: fc bias pushed towards <end>):
    finished rows are fed word 0 and keep emitting zeros, the loop leaves once every row has finished, and the prologue /
    decode split (`set_editnet_greedy_begun`, DevicePrefetcher begin_ahead) takes the persistent launch too."""
    d, xe, rl = editnet_modules("editnet_full_b4")
    args = (d["wm"], to_dev(d["prev"]), to_dev(d["plen"]), to_dev(d["X"]), True, False)
    end = int(d["wm"]["<end>"])
    with torch.no_grad():
        rl.fc.bias[end] += 6.0                                    # captions now end after a few words
        for _ in range(2):
            rl(*args)
        seq, logp = rl(*args)
        ref = _with_env("SET_DEC_PERSISTENT", "0", lambda: rl(*args))
        torch.cuda.synchronize()
        assert torch.equal(seq, ref[0]) and float((logp - ref[1]).abs().max()) < 1e-5
        assert (seq == 0).any(1).all(), "every row should have finished inside max_len in this set-up"
        rl.begin_ahead(args[1], args[2], args[3])
        hits = rl.__dict__.get("_ahead_hits", 0)
        names = _tags(lambda: rl(*args))
        assert rl.__dict__.get("_ahead_hits", 0) == hits + 1 and "persistent_decode" in names
        again = rl(*args)
        assert torch.equal(again[0], seq)


_FAULT_SCRIPT = r"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.getcwd(), "tests")); sys.path.insert(0, os.getcwd())
import parity
from hip_adapter import dcnet_modules, editnet_modules, to_dev
from show_edit_tell_amd import _lib
d, xe, rl = dcnet_modules("dcnet_full_b4")
g = parity.load("dcnet_full_b4")
args = (d["wm"], to_dev(d["prev"]), to_dev(d["plen"]), True, False)
mode = sys.argv[1]
with torch.no_grad():
    rl(*args)                                      # (no token table yet: the per-step loop)
    if mode == "stall":
        seq, logp = rl(*args)                      # workgroup 0 never publishes h1: every poll times out
        torch.cuda.synchronize()
        assert torch.isnan(logp).all() and not seq.any(), "a timed-out persistent decode must poison its result"
        try:
            rl(*args)
            raise SystemExit("the call after a timeout must raise SetError")
        except _lib.SetError as e:
            assert "code 5" in str(e), str(e)
    for _ in range(3):                             # from now on the per-step kernels: parity with the golden
        seq, logp = rl(*args)
        torch.cuda.synchronize()
        parity.check_greedy(seq.cpu().numpy(), logp.cpu().numpy(), g)
    lib = _lib.load()
    lib.set_profile_enable(1)
    rl(*args); torch.cuda.synchronize()
    names = [r["tag"] for r in _lib.profile_report()]
    lib.set_profile_enable(0)
    assert "persistent_decode" not in names, names
print("OK", mode)
"""


@pytest.mark.parametrize("mode,env", [("stall", {"SET_PENC_TEST_STALL": "1", "SET_PENC_TIMEOUT_US": "20000", "SET_ENC_PERSISTENT": "0"}),
                                      ("capacity", {"SET_PENC_TEST_CAPACITY": "100"})])
def test_dcnet_persistent_decode_failure_is_loud(mode, env, tmp_path):
    """Same two failure modes as the persistent encoder (tests/test_hip_boundary.py): a grid that is not admitted whole is
    refused up front (per-step loop, golden parity); a workgroup that never publishes makes every bounded poll time out, the
    kernel overwrites seq_logp with NaN, the NEXT call raises SetError(SET_ERR_FAULT) once and the library keeps to the
    per-step kernels afterwards."""
    e = dict(os.environ)
    e.update(env)
    e["SET_PERSISTENT_LOCK_DIR"] = str(tmp_path)       # this pytest process owns the device's persistent launches; the child gets its own lock
    r = subprocess.run([sys.executable, "-c", _FAULT_SCRIPT, mode], env=e, capture_output=True, text=True, timeout=600,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0 and ("OK " + mode) in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_request_coalescer_serves_concurrent_small_requests_on_one_persistent_launch():
    """show_edit_tell_amd/serving.py: four callers with a 4-row request each, submitted from four
    threads; the coalescer runs them as shared decodes of <= 16 rows (the persistent launch) and every caller gets the rows a
    decode of its request alone produces — ids bit-identical (the persistent kernel's per-row arithmetic does not depend on the
    row count), log-probs within 2e-5 — including requests whose previous captions have different padded lengths."""
    import threading
    from show_edit_tell_amd import serving
    d, xe, rl = editnet_modules("editnet_full_b128")
    wm = d["wm"]
    X, prev, plen = to_dev(d["X"]), to_dev(d["prev"]), to_dev(d["plen"])
    reqs = []
    for i in range(6):
        lo = 4 * i
        p = prev[lo:lo + 4]
        if i % 2:                                            # a shorter padded length: the coalescer pads it back with <pad>
            tmax = int(plen[lo:lo + 4].max())
            p = p[:, :max(tmax, 2)].contiguous()
        reqs.append((p.contiguous(), plen[lo:lo + 4].contiguous(), X[lo:lo + 4].contiguous()))
    with torch.no_grad():
        for _ in range(3):
            rl(wm, *reqs[0], True, False)                    # token table
        alone = [tuple(t.clone() for t in rl(wm, *r, True, False)) for r in reqs]
    torch.cuda.synchronize()
    outs = {}
    with serving.RequestCoalescer(lambda p, l, x: rl(wm, p, l, x, True, False), max_rows=16, window_s=0.01) as co:
        def caller(i):
            f = co.submit(*reqs[i])
            seq, logp = f.result(timeout=120)
            torch.cuda.synchronize()
            outs[i] = (seq.clone(), logp.clone())
        ths = [threading.Thread(target=caller, args=(i,)) for i in range(6)]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        assert co.requests == 6 and co.batches < 6, (co.requests, co.batches)
    for i in range(6):
        assert torch.equal(outs[i][0], alone[i][0]), i
        assert float((outs[i][1] - alone[i][1]).abs().max()) < 2e-5
