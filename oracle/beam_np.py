"""ORACLE (test infrastructure): numpy restatement of the reference's beam search
(`editnet.py:595-718`, ensemble `eval/eval xe/eval_full.py:88-218`) over the oracle's step functions.
Pinned: tests/test_beam_golden.py checks it against tests/golden/beam_*.npz, which hold the outputs of the
reference's OWN per-image loops (AST-sliced with the single `top_k_words / vocab_size` -> `//` patch that
torch >= 1.5 needs, oracle/ref_beam.py + oracle/make_beam_golden.py).  Only tests/ import this module."""
import numpy as np

from . import dcnet_np as DN
from . import editnet_np as EN


class EditNetBeam:
    def __init__(self, P, X1, prev1, plen1, k):
        self.S = EN.SeqState(P, np.repeat(X1, k, 0), np.repeat(prev1, k, 0), np.repeat(plen1, k, 0))

    def step(self, words):
        return EN.step(self.S, words, len(words))

    def reindex(self, idx):
        S = self.S
        for n in ("X", "H", "M", "final_hidden", "mask", "image_mean", "att1", "att1_c", "h1", "c1", "h2", "c2"):
            setattr(S, n, getattr(S, n)[idx])


class DcnetBeam:
    def __init__(self, P, prev1, plen1, k):
        self.S = DN.SeqState(P, np.repeat(prev1, k, 0), np.repeat(plen1, k, 0))

    def step(self, words):
        return DN.step(self.S, words, len(words))

    def reindex(self, idx):
        S = self.S
        for n in ("enc", "final_hidden", "mask", "att1_c", "h1", "c1", "h2", "c2"):
            setattr(S, n, getattr(S, n)[idx])


def beam_loop(states, combine, start, end, V, k, max_steps=50):
    words = np.full((k,), start, np.int64)
    seqs = words[:, None]
    top = np.zeros((k, 1), np.float32)
    done, done_scores = [], []
    step = 1
    while True:
        scores = top + combine([s.step(words) for s in states])
        flat = scores[0] if step == 1 else scores.reshape(-1)
        order = np.argsort(-flat, kind="stable")[:k]
        top_s = flat[order]
        parent, nxt = order // V, order % V
        seqs = np.concatenate([seqs[parent], nxt[:, None]], 1)
        inc = [i for i, w in enumerate(nxt) if w != end]
        comp = [i for i in range(len(nxt)) if i not in inc]
        for i in comp:
            done.append(seqs[i].tolist())
            done_scores.append(float(top_s[i]))
        k -= len(comp)
        if k == 0:
            break
        seqs = seqs[inc]
        for s in states:
            s.reindex(parent[inc])
        top = top_s[inc][:, None].astype(np.float32)
        words = nxt[inc]
        if step > max_steps:
            return seqs[0][:18].tolist(), float("nan"), None
        step += 1
    i = int(np.argmax(done_scores))
    # margin between the two best finished hypotheses (tests skip near-ties)
    srt = sorted(done_scores, reverse=True)
    return done[i], done_scores[i], (srt[0] - srt[1] if len(srt) > 1 else np.inf)


def beam_editnet(P, X1, prev1, plen1, start, end, k=3):
    V = P["fc.weight"].shape[0]
    return beam_loop([EditNetBeam(P, X1, prev1, plen1, k)], lambda ls: EN._log_softmax(ls[0], 1), start, end, V, k)


def beam_dcnet(Pd, prev1, plen1, start, end, k=3):
    V = Pd["fc.weight"].shape[0]
    return beam_loop([DcnetBeam(Pd, prev1, plen1, k)], lambda ls: EN._log_softmax(ls[0], 1), start, end, V, k)


def beam_ensemble(Pe, Pd, X1, prev1, plen1, start, end, k=3):
    V = Pe["fc.weight"].shape[0]
    comb = lambda ls: np.log((EN._softmax(ls[0], 1) + EN._softmax(ls[1], 1)) / 2)
    return beam_loop([EditNetBeam(Pe, X1, prev1, plen1, k), DcnetBeam(Pd, prev1, plen1, k)], comb, start, end, V, k)
