"""Build-owned CIDEr-D scorer for the SCST reward.

The reference does not contain its scorer: `editnet_rl.py:576-578` imports `CiderD` from the un-vendored
`ruotianluo/cider` checkout (version unpinned), so reward VALUES are "parity unpinned" (SURVEY.md §8c.3).
This module restates the published CIDEr-D metric (Vedantam et al., CVPR 2015, §"CIDEr-D"; the variant
with a precomputed document-frequency table that self-critical training uses):

    g_n(s)      tf-idf vector over the n-grams of s, n = 1..4:  tf(ngram) * (log N_docs - log max(1, df(ngram)))
    sim_n(c, r) = sum_ngram min(g(c), g(r)) * g(r) / (|g(c)| |g(r)|) * exp(-(l_c - l_r)^2 / (2 sigma^2)),  sigma = 6
    CIDEr-D(c)  = 10 / (4 |refs|) * sum_n sum_r sim_n(c, r)

with l = number of bigrams of the sentence (the length the public implementations use).  The document
frequency table has the format `preprocess_rl.py:7-55` writes: {n-gram tuple of token strings: number of
images whose reference set contains it} plus `ref_len` = number of images.

Host-side, per-sample work (strings and dictionaries): it shards with the batch and never touches the GPU.
`compute_score` runs on the native implementation in csrc/ciderd_host.hip (`set_ciderd_*`, host pointers): in Python
the scorer was half of the SCST step's wall time at 5 samples per image.  The pure-Python `score` below is the
readable statement of the metric and the cross-check of the native one (tests/test_ciderd.py); identical
(reference set, caption) pairs of a batch — e.g. the greedy baseline repeated for every sample — are scored once.
"""
from __future__ import annotations

import math
from collections import Counter

import numpy as np


def ngram_counts(sentence, n=4):
    """Counter of all 1..n-grams (tuples of token strings) of a whitespace-tokenised sentence."""
    words = sentence.split()
    c = Counter()
    for k in range(1, n + 1):
        for i in range(len(words) - k + 1):
            c[tuple(words[i:i + k])] += 1
    return c


def document_frequency(reference_sets, n=4):
    """df table over a corpus: reference_sets = iterable of lists of reference sentences (one list per
    image).  Returns (df dict, number of images) -- the two fields `preprocess_rl.py` pickles."""
    df, docs = Counter(), 0
    for refs in reference_sets:
        seen = set()
        for r in refs:
            seen.update(ngram_counts(r, n).keys())
        for g in seen:
            df[g] += 1
        docs += 1
    return dict(df), docs


class CiderD:
    def __init__(self, df, ref_len, n=4, sigma=6.0):
        """df: n-gram -> document count; ref_len: number of documents the table was built from"""
        self.df = df
        self.ref_len = float(ref_len)
        self.log_ref_len = math.log(float(ref_len))
        self.n = n
        self.sigma = sigma
        self._tok = {}           # token string -> int id (interned for the native scorer)
        self._native = None

    # ---- native scorer (csrc/ciderd_host.hip) -------------------------------------------------------------
    def _ids(self, sentence):
        tok = self._tok
        return [tok.setdefault(w, len(tok)) for w in sentence.split()]

    def _handle(self):
        """Build the native df table once (None if the library is unavailable: pure-Python scoring then)."""
        if self._native is None:
            try:
                from . import _lib
                lib = _lib.load()
                if "set_ciderd_create" in _lib.MISSING:
                    raise _lib.SetError("library without CIDEr-D")
            except Exception:
                self._native = False
                return None
            grams = [g for g in self.df if 1 <= len(g) <= 4]
            toks = np.zeros((max(1, len(grams)), 4), dtype=np.int64)
            lens = np.zeros(max(1, len(grams)), dtype=np.int32)
            cnt = np.zeros(max(1, len(grams)), dtype=np.float64)
            tok = self._tok
            for i, g in enumerate(grams):
                lens[i] = len(g)
                cnt[i] = self.df[g]
                for j, w in enumerate(g):
                    toks[i, j] = tok.setdefault(w, len(tok))
            h = lib.set_ciderd_create(toks.ctypes.data, lens.ctypes.data, cnt.ctypes.data, len(grams), self.ref_len,
                                      self.n, self.sigma)
            if not h:
                self._native = False
                return None
            self._native = (lib, h)
        return self._native or None

    def __del__(self):
        if isinstance(getattr(self, "_native", None), tuple):
            lib, h = self._native
            try:
                lib.set_ciderd_destroy(h)
            except Exception:
                pass

    def __getstate__(self):
        st = dict(self.__dict__)
        st["_native"] = None
        return st

    def _vector(self, counts):
        vec = [dict() for _ in range(self.n)]
        norm = [0.0] * self.n
        length = 0
        for g, tf in counts.items():
            k = len(g) - 1
            w = float(tf) * (self.log_ref_len - math.log(max(1.0, self.df.get(g, 0.0))))
            vec[k][g] = w
            norm[k] += w * w
            if k == 1:
                length += tf
        return vec, [math.sqrt(x) for x in norm], length

    def _similarity(self, hyp, ref):
        (vh, nh, lh), (vr, nr, lr) = hyp, ref
        delta = float(lh - lr)
        penalty = math.exp(-(delta * delta) / (2.0 * self.sigma * self.sigma))
        out = np.zeros(self.n)
        for k in range(self.n):
            s = 0.0
            for g, w in vh[k].items():
                wr = vr[k].get(g)
                if wr is not None:
                    s += min(w, wr) * wr
            if nh[k] != 0.0 and nr[k] != 0.0:
                s /= nh[k] * nr[k]
            out[k] = s * penalty
        return out

    def score(self, hypothesis, references):
        """CIDEr-D of one sentence against its reference sentences."""
        hyp = self._vector(ngram_counts(hypothesis, self.n))
        tot = np.zeros(self.n)
        for r in references:
            tot += self._similarity(hyp, self._vector(ngram_counts(r, self.n)))
        return float(tot.mean() / max(1, len(references)) * 10.0)

    def compute_score(self, gts, res):
        """Same call shape as the external scorer at editnet_rl.py:636: gts = {image_id: [ref strings]},
        res = [{'image_id': id, 'caption': [string]}].  Returns (mean score, per-entry scores)."""
        nat = self._handle()
        if nat is None:
            return self._compute_score_py(gts, res)
        lib, h = nat
        # reference sets and (set, caption) pairs are de-duplicated by content
        set_index, sets = {}, []
        pair_index, pairs = {}, []
        which = np.zeros(len(res), dtype=np.int64)
        for i, r in enumerate(res):
            refs = gts[r['image_id']]
            key = tuple(refs)
            si = set_index.get(key)
            if si is None:
                si = set_index[key] = len(sets)
                sets.append(refs)
            pk = (si, r['caption'][0])
            pi = pair_index.get(pk)
            if pi is None:
                pi = pair_index[pk] = len(pairs)
                pairs.append(pk)
            which[i] = pi
        ref_tok, ref_off, set_off = [], [0], [0]
        for refs in sets:
            for sref in refs:
                ref_tok += self._ids(sref)
                ref_off.append(len(ref_tok))
            set_off.append(len(ref_off) - 1)
        hyp_tok, hyp_off, set_of = [], [0], []
        for si, cap in pairs:
            hyp_tok += self._ids(cap)
            hyp_off.append(len(hyp_tok))
            set_of.append(si)
        a = lambda x, dt: np.ascontiguousarray(np.asarray(x if len(x) else [0], dtype=dt))
        ht, ho, so = a(hyp_tok, np.int64), a(hyp_off, np.int64), a(set_of, np.int32)
        rt, ro, rs = a(ref_tok, np.int64), a(ref_off, np.int64), a(set_off, np.int64)
        out = np.zeros(max(1, len(pairs)), dtype=np.float64)
        rc = lib.set_ciderd_score(h, ht.ctypes.data, ho.ctypes.data, len(pairs), so.ctypes.data, rt.ctypes.data,
                                  ro.ctypes.data, rs.ctypes.data, len(sets), out.ctypes.data)
        if rc != 0:
            raise RuntimeError("set_ciderd_score failed with code %d" % rc)
        scores = out[which] if len(res) else np.zeros(0)
        return float(scores.mean()) if len(res) else 0.0, scores

    def _lut_for(self, max_id):
        """vocabulary id -> interned index of the word str(id) (the reward plumbing stringifies token ids,
        editnet_rl.py:601-609), as a numpy lookup table grown on demand"""
        lut = getattr(self, "_lut", None)
        if lut is None or lut.shape[0] <= max_id:
            n0 = 0 if lut is None else lut.shape[0]
            n1 = max(max_id + 1, 2 * n0, 1024)
            new = np.empty(n1, dtype=np.int64)
            if n0:
                new[:n0] = lut
            tok = self._tok
            for i in range(n0, n1):
                new[i] = tok.setdefault(str(i), len(tok))
            self._lut = lut = new
        return lut

    def score_token_ids(self, hyps, hyp_set, ref_sets):
        """Integer fast path of compute_score for the self-critical reward: `hyps` (N, L) int array of decoded captions
        (0 = <end>, everything after the first 0 is ignored, the 0 itself is a word), `hyp_set` (N,) index of each
        caption's reference set, `ref_sets` = list of reference sets, each a list of id lists (as ground_truth_lists
        returns them).  Same scores as compute_score on the stringified ids; needs the native scorer."""
        nat = self._handle()
        if nat is None:
            raise RuntimeError("score_token_ids needs the native CIDEr-D scorer")
        lib, h = nat
        hyps = np.ascontiguousarray(hyps, dtype=np.int64)
        N, L = hyps.shape
        is0 = hyps == 0
        first0 = np.where(is0.any(1), is0.argmax(1), L - 1)          # keep the first 0; no 0 -> the whole row
        lens = first0 + 1
        keep = np.arange(L)[None, :] < lens[:, None]
        # references are cut after their first 0 as well (tokens_to_str / array_to_str do the same on the string path;
        # lists from ground_truth_lists carry no interior 0)
        def _cut(c):
            c = list(c)
            return c[:c.index(0) + 1] if 0 in c else c
        flat_refs = [_cut(c) for refs in ref_sets for c in refs]
        ref_len = np.fromiter((len(c) for c in flat_refs), dtype=np.int64, count=len(flat_refs))
        ref_flat = np.fromiter((w for c in flat_refs for w in c), dtype=np.int64, count=int(ref_len.sum()))
        max_id = int(max(hyps.max(initial=0), ref_flat.max(initial=0)))
        lut = self._lut_for(max_id)
        ht = np.ascontiguousarray(lut[hyps[keep]])
        ho = np.zeros(N + 1, dtype=np.int64)
        np.cumsum(lens, out=ho[1:])
        so = np.ascontiguousarray(hyp_set, dtype=np.int32)
        rt = np.ascontiguousarray(lut[ref_flat]) if ref_flat.size else np.zeros(1, dtype=np.int64)
        ro = np.zeros(len(flat_refs) + 1, dtype=np.int64)
        np.cumsum(ref_len, out=ro[1:])
        rs = np.zeros(len(ref_sets) + 1, dtype=np.int64)
        np.cumsum(np.fromiter((len(r) for r in ref_sets), dtype=np.int64, count=len(ref_sets)), out=rs[1:])
        out = np.zeros(max(1, N), dtype=np.float64)
        rc = lib.set_ciderd_score(h, ht.ctypes.data, ho.ctypes.data, N, so.ctypes.data, rt.ctypes.data, ro.ctypes.data,
                                  rs.ctypes.data, len(ref_sets), out.ctypes.data)
        if rc != 0:
            raise RuntimeError("set_ciderd_score failed with code %d" % rc)
        return out[:N]

    def _compute_score_py(self, gts, res):
        """pure-Python twin of compute_score (cross-check; also the fallback when the library is not built)"""
        cache = {}
        scores = np.zeros(len(res))
        for i, r in enumerate(res):
            key = r['image_id']
            refs = gts[key]
            rk = tuple(refs)
            if rk not in cache:
                cache[rk] = [self._vector(ngram_counts(s, self.n)) for s in refs]
            hyp = self._vector(ngram_counts(r['caption'][0], self.n))
            tot = np.zeros(self.n)
            for rv in cache[rk]:
                tot += self._similarity(hyp, rv)
            scores[i] = tot.mean() / max(1, len(refs)) * 10.0
        return float(scores.mean()) if len(res) else 0.0, scores


# ---- the reward plumbing of editnet_rl.py:584-646 ------------------------------------------------------
def tokens_to_str(ids):
    """ids of one caption -> space-joined string, cut after the first 0 (the decoder writes 0 for <end>
    and everything after it; the 0 itself is kept as the end-of-sentence token, editnet_rl.py:601-609)."""
    out = []
    for w in ids:
        out.append(str(int(w)))
        if int(w) == 0:
            break
    return ' '.join(out)


def ground_truth_lists(allcaps, word_map):
    """(B, n_refs, L) ids -> per image list of id lists without <start>/<pad>, <end> rewritten to 0
    (editnet_rl.py:584-599)."""
    drop = {int(word_map['<start>']), int(word_map['<pad>'])}
    end = int(word_map['<end>'])
    out = []
    for caps in np.asarray(allcaps.cpu() if hasattr(allcaps, 'cpu') else allcaps).tolist():
        out.append([[0 if w == end else w for w in c if w not in drop] for c in caps])
    return out


def self_critical_reward(scorer, sampled, greedy, ground_truth, cider_weight=1.0):
    """reward[b, :] = CIDEr-D(sampled_b) - CIDEr-D(greedy_b), repeated over the max_len columns
    (editnet_rl.py:611-646).  sampled/greedy: (B, max_len) integer arrays/tensors; returns float32 numpy."""
    sampled = np.asarray(sampled.cpu() if hasattr(sampled, 'cpu') else sampled)
    greedy = np.asarray(greedy.cpu() if hasattr(greedy, 'cpu') else greedy)
    B = sampled.shape[0]
    if getattr(scorer, "_handle", None) is not None and scorer._handle() is not None:
        # integer fast path (no strings): the reference sets of the distinct images once, all 2B captions in one call
        sets, index = [], {}
        hyp_set = np.empty(2 * B, dtype=np.int32)
        for i in range(B):
            caps = ground_truth[i]
            k = id(caps)
            if k not in index:
                index[k] = len(sets)
                sets.append(caps)
            hyp_set[i] = hyp_set[B + i] = index[k]
        s = scorer.score_token_ids(np.concatenate([sampled, greedy], 0), hyp_set, sets)
        diff = cider_weight * (s[:B] - s[B:])
        return np.repeat(diff[:, None], sampled.shape[1], 1).astype(np.float32)
    memo = {}                    # the same image's reference lists are repeated once per sample: stringify them once

    def ref_strings(caps):
        k = id(caps)
        if k not in memo:
            memo[k] = [tokens_to_str(c) for c in caps]
        return memo[k]

    refs = [ref_strings(caps) for caps in ground_truth]
    gts = {i: refs[i % B] for i in range(2 * B)}
    res = [{'image_id': i, 'caption': [tokens_to_str(sampled[i])]} for i in range(B)]
    res += [{'image_id': B + i, 'caption': [tokens_to_str(greedy[i])]} for i in range(B)]
    _, s = scorer.compute_score(gts, res)
    diff = cider_weight * (s[:B] - s[B:])
    return np.repeat(diff[:, None], sampled.shape[1], 1).astype(np.float32)
