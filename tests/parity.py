"""Shared parity checker: compares an implementation of the EditNet / DCNet path (the numpy oracle
on CPU, the HIP modules on the GPU) against the golden vectors captured from the reference.

Tolerances (north_star in BASELINE.json): token ids bit-exact for greedy decode, logits within
1e-4 absolute in fp32.  Bit-exactness of ids is asserted on every row whose reference top-1/top-2
logit margin stays above MARGIN_MIN for the whole sequence; rows with a near-tie are "ambiguous"
(a 1e-5 logit difference may legitimately flip them), must be a small minority, and are still
checked: bit-exact prefix up to the near-tie step, and the token taken there must be one of the
reference's two best candidates (check_greedy).
"""
import os

import numpy as np

from oracle import cases

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
LOGIT_TOL = 1e-4
STATE_TOL = 2e-5          # h/c/attention outputs are O(1) and sit upstream of the logits
SUM_TOL = 2e-3            # sums over 1024 state elements
MARGIN_MIN = 2.5e-4      # two logits each within LOGIT_TOL can swap only below 2e-4


def load(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


def maxerr(a, b):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max()) if np.size(a) else 0.0


def assert_close(a, b, tol, what):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, "%s: shape %s vs %s" % (what, a.shape, b.shape)
    e = maxerr(a, b)
    assert e <= tol, "%s: max abs err %.3e > %.1e" % (what, e, tol)
    return e


def check_logit_summary(logits, g, prefix, V, tol=LOGIT_TOL, what=""):
    """logits (N,V) against a stored summary (top-8, lse, 64-column slice)."""
    N = logits.shape[0]
    ti = g[prefix + "top_idx"].reshape(N, 8)
    tv = g[prefix + "top_val"].reshape(N, 8)
    got = np.take_along_axis(logits, ti.astype(np.int64), 1)
    assert_close(got, tv, tol, what + " top-8 values")
    lg = logits.astype(np.float64)
    m = lg.max(1, keepdims=True)
    lse = m[:, 0] + np.log(np.exp(lg - m).sum(1))
    assert_close(lse, g[prefix + "lse"].reshape(N), tol, what + " logsumexp")
    if prefix + "cols" in g:
        assert_close(logits[:, cases.logit_slice_cols(V)], g[prefix + "cols"].reshape(N, 64), tol, what + " column slice")
    # the reference's argmax must be our argmax wherever its margin is comfortable
    margin = tv[:, 0] - tv[:, 1]
    ok = margin > MARGIN_MIN
    assert np.array_equal(logits.argmax(1)[ok], ti[ok, 0]), what + " argmax"


def unsort(sort_ind):
    inv = np.empty_like(sort_ind)
    inv[sort_ind] = np.arange(len(sort_ind))
    return inv


def check_xe(pred, dl, sort_ind, g, V, small):
    """XE predictions compared per ORIGINAL sample (the reference's descending sort is unstable, so
    rows of equal length may be permuted between implementations)."""
    g_inv, inv = unsort(g["xe_sort_ind"]), unsort(np.asarray(sort_ind))
    assert sorted(dl) == sorted(g["xe_decode_lengths"].tolist())
    assert list(dl) == sorted(dl, reverse=True)
    B, Tm = pred.shape[0], pred.shape[1]
    assert Tm == int(max(g["xe_decode_lengths"]))
    mine = pred[inv]                      # original sample order
    if small:
        assert_close(mine, g["xe_pred"][g_inv], LOGIT_TOL, "xe predictions")
        return
    sub = {k: v[g_inv] for k, v in g.items() if k.startswith("xe_pred_")}
    dl_orig = np.asarray(g["xe_decode_lengths"])[g_inv]
    rows = [(b, t) for b in range(B) for t in range(Tm) if t < dl_orig[b]]
    bi, ti = np.array([r[0] for r in rows]), np.array([r[1] for r in rows])
    gs = {k: v[bi, ti] for k, v in sub.items()}
    check_logit_summary(mine[bi, ti], gs, "xe_pred_", V, what="xe predictions")
    # rows past a sample's decode length stay exactly zero (editnet.py:499,546)
    pad = np.array([[t >= dl_orig[b] for t in range(Tm)] for b in range(B)])
    assert not mine[pad].any(), "xe predictions beyond decode_lengths must be zero"


def ambiguous_rows(g):
    """rows whose greedy trajectory passes a near-tie in the reference."""
    return (g["greedy_margin"] < MARGIN_MIN).any(0)


def check_greedy_rows(seq, logp, ref_seq, ref_logp, margins, top2, end_idx=None, margin_min=MARGIN_MIN):
    """Row-wise greedy comparison against a reference trajectory with per-step top-1/top-2 `margins` (S,B) and
    candidate indices `top2` (S,B,2) (None: candidates unknown).

    Rows that never pass a near-tie (margin >= margin_min at every step): ids bit-exact and log-probs within
    LOGIT_TOL over the whole row.  Near-tie rows are NOT dropped: up to the first near-tie step t* the row must
    still be bit-exact (ids and log-probs), and at t* the token must be one of the reference's two best
    candidates (only those two can swap under a <= LOGIT_TOL perturbation; `end_idx` tells the checker which
    candidate the loop rewrites to 0, editnet_rl.py:531).  If the same candidate as the reference was taken at
    t*, the check continues to the next near-tie; after a legitimate swap the trajectories diverge and nothing
    more can be asserted about that row.  Returns the number of near-tie rows."""
    amb = (margins < margin_min).any(0)
    ok = ~amb
    assert seq.dtype == np.int64 and seq.shape == ref_seq.shape
    assert np.array_equal(seq[ok], ref_seq[ok]), "greedy token ids differ on non-ambiguous rows"
    assert_close(logp[ok], ref_logp[ok], LOGIT_TOL, "greedy seqLogprobs")
    for b in np.nonzero(amb)[0]:
        gs, gl = ref_seq[b], ref_logp[b]
        for t in range(min(seq.shape[1], margins.shape[0])):
            if margins[t, b] >= margin_min:
                assert seq[b, t] == gs[t], "near-tie row %d: token differs at step %d before any near-tie" % (b, t)
                assert abs(float(logp[b, t]) - float(gl[t])) <= LOGIT_TOL, ("near-tie row logp", b, t)
                if gs[t] == 0:
                    break
                continue
            if top2 is not None:
                cand = set(int(x) for x in top2[t, b])
                # the loop rewrites <end> to 0 and finished rows to 0 (editnet_rl.py:531-540)
                allowed = cand | {0} if (gs[t] == 0 or end_idx is None or end_idx in cand) else cand
                assert int(seq[b, t]) in allowed, "near-tie row %d: token %d at step %d not in the reference top-2 %s" % (
                    b, int(seq[b, t]), t, sorted(cand))
            if seq[b, t] != gs[t]:
                break               # legitimate swap: trajectories diverge from here
    return int(amb.sum())


def check_greedy(seq, logp, g, max_ambiguous_frac=0.05, end_idx=None):
    """Greedy token ids / log-probs against the reference fixture `g` (see check_greedy_rows)."""
    amb = ambiguous_rows(g)
    assert amb.mean() <= max_ambiguous_frac, "too many near-tie rows in fixture: %.3f" % amb.mean()
    return check_greedy_rows(seq, logp, g["greedy_seq"], g["greedy_logp"], g["greedy_margin"], _greedy_top2(g), end_idx)


def _greedy_top2(g):
    """(S,B,2) indices of the reference's two largest logits per greedy step, or None if the fixture keeps neither
    the full logits nor the top-8 summary."""
    if "greedy_logits_top_idx" in g:
        return g["greedy_logits_top_idx"][:, :, :2]
    if "greedy_logits" in g:
        return np.argsort(-g["greedy_logits"].astype(np.float64), axis=2, kind="stable")[:, :, :2]
    return None


def oracle_greedy_reference(model, sd, wm, prev, plen, X=None, max_len=18):
    """The numpy oracle's greedy trajectory on arbitrary inputs, with what check_greedy_rows needs to judge a differing row:
    (seq, logp, margins (S,B), top2 (S,B,2)) — margins / candidates from the oracle's own per-step logits (its trace).  Rows
    whose margin stays >= MARGIN_MIN must then match bit for bit; a differing row must be a DEMONSTRATED near-tie."""
    from oracle import dcnet_np as DN, editnet_np as EN
    tr = []
    if model == "editnet":
        P = EN.cast_params(sd)
        seq, logp = EN.greedy_decode(P, wm["<start>"], wm["<end>"], prev, plen, X, max_len=max_len, trace=tr)
    else:
        P = DN.cast_params(sd)
        seq, logp = DN.greedy_decode(P, wm["<start>"], wm["<end>"], prev, plen, max_len=max_len, trace=tr)
    logits = np.stack([s["logits"] for s in tr], 0).astype(np.float64)           # (S, B, V)
    top2 = np.argsort(-logits, axis=2, kind="stable")[:, :, :2]
    tv = np.take_along_axis(logits, top2, 2)
    margins = tv[:, :, 0] - tv[:, :, 1]
    # behind a row's <end> every implementation feeds word 0 whatever the arg-max says (editnet_rl.py:531-540): a near-tie
    # there cannot make trajectories diverge (the recorded max log-prob moves by less than the margin)
    for b in range(seq.shape[0]):
        z = np.nonzero(seq[b] == 0)[0]
        if len(z):
            margins[z[0] + 1:, b] = np.inf
    return seq, logp, margins, top2


def check_two_paths_rows(seq_a, logp_a, seq_b, logp_b, margins, tol=2e-5, margin_min=MARGIN_MIN):
    """Two implementations of the same decode (e.g. the persistent launch and the per-step loop): on every row that never
    passes a near-tie of the reference trajectory (`margins` (S,B)) ids are bit-identical and log-probs within `tol`; a
    near-tie row must agree up to its first near-tie step.  Returns the number of near-tie rows."""
    amb = (margins < margin_min).any(0)
    ok = ~amb
    assert np.array_equal(seq_a[ok], seq_b[ok]), "ids differ on rows without a near-tie"
    if ok.any():
        e = float(np.abs(logp_a[ok] - logp_b[ok]).max())
        assert e < tol, "log-probs of the two paths differ by %.3e (tol %.1e; each is within 1e-4 of the oracle)" % (e, tol)
    for b in np.nonzero(amb)[0]:
        t_star = int(np.argmax(margins[:, b] < margin_min))
        n = min(t_star, seq_a.shape[1])
        assert np.array_equal(seq_a[b, :n], seq_b[b, :n]), "near-tie row %d differs before its near-tie step %d" % (b, t_star)
        if n:
            assert float(np.abs(logp_a[b, :n] - logp_b[b, :n]).max()) < tol
    return int(amb.sum())


def check_sampled_paths_rows(seq_a, logp_a, seq_b, logp_b, draw_margin, draw_alt, tol=2e-5, margin_max=1e-5):
    """Two implementations of the same SAMPLED rollout (same Philox stream), judged row by row with the ORACLE's evidence
    along trajectory a: `draw_margin` (S,B) = distance of every draw's target from the nearest CDF boundary relative to the
    total mass, `draw_alt` (S,B) = the word on the other side of that boundary (oracle/philox_np.categorical_draw).  With
    ~V boundaries in [0,1) a few of the B*S draws DO land within fp32 noise of one, so rows may differ — but only like this:
    identical ids and log-probs (tol) up to the first differing step t*, the draw at t* demonstrably close (margin <
    margin_max) and path b's word there = the oracle's neighbouring word (or 0 for <end>).  A row-indexing bug produces
    differing rows whose draws are nowhere near a boundary.  Returns the number of differing rows."""
    S = min(draw_margin.shape[0], seq_a.shape[1])
    differing = 0
    for b in range(seq_a.shape[0]):
        diff = np.nonzero(seq_a[b, :S] != seq_b[b, :S])[0]
        n = int(diff[0]) if len(diff) else S
        if n:
            e = float(np.abs(logp_a[b, :n] - logp_b[b, :n]).max())
            assert e < tol, "row %d: log-probs of the two sampled paths differ by %.3e before step %d" % (b, e, n)
        if len(diff):
            differing += 1
            assert draw_margin[n, b] < margin_max, ("row %d differs at step %d although its draw clears the CDF boundaries by "
                                                    "%.2e of the mass" % (b, n, draw_margin[n, b]))
            assert int(seq_b[b, n]) in (int(draw_alt[n, b]), 0), "row %d step %d: not the neighbouring word" % (b, n)
    return differing
