"""CPU: the build-owned CIDEr-D scorer (show_edit_tell_amd.ciderd) against hand-stated values of the
published metric.  The reference's own scorer is an un-vendored external checkout, so these values are
NOT pinned to the reference ("parity unpinned")."""
import math

import pytest

import numpy as np

from show_edit_tell_amd import ciderd


def _scorer():
    corpus = [["1 2 3 0"], ["4 5 6 0"]]
    df, docs = ciderd.document_frequency(corpus)
    assert docs == 2 and df[("0",)] == 2 and df[("1",)] == 1 and df[("1", "2", "3", "0")] == 1
    return ciderd.CiderD(df, docs)


def test_identical_and_disjoint():
    s = _scorer()
    assert abs(s.score("1 2 3 0", ["1 2 3 0"]) - 10.0) < 1e-9
    # the only shared n-gram ("0") occurs in every document: idf 0
    assert s.score("4 5 6 0", ["1 2 3 0"]) == 0.0


def test_length_penalty_and_partial_overlap():
    s = _scorer()
    # unigram cosine 2/sqrt(6), bigram cosine 1/sqrt(6), no shared 3-/4-grams; bigram lengths 2 vs 3
    want = 10.0 * math.exp(-1.0 / 72.0) * (2 / math.sqrt(6) + 1 / math.sqrt(6)) / 4.0
    assert abs(s.score("1 2 0", ["1 2 3 0"]) - want) < 1e-9


def test_count_clipping():
    s = _scorer()
    # hypothesis repeats a word three times: min(g_c, g_r) clips it to the reference count
    want = 10.0 * (1.0 / (3.0 * math.sqrt(3.0))) / 4.0
    assert abs(s.score("1 1 1 0", ["1 2 3 0"]) - want) < 1e-9


def test_mean_over_references_and_compute_score():
    s = _scorer()
    a = s.score("1 2 3 0", ["1 2 3 0", "4 5 6 0"])
    assert abs(a - 5.0) < 1e-9
    mean, per = s.compute_score({0: ["1 2 3 0"], 1: ["1 2 3 0"]},
                                [{"image_id": 0, "caption": ["1 2 3 0"]}, {"image_id": 1, "caption": ["4 5 6 0"]}])
    assert np.allclose(per, [10.0, 0.0]) and abs(mean - 5.0) < 1e-9


def test_reward_plumbing():
    s = _scorer()
    wm = {"<pad>": 0, "<start>": 8, "<end>": 9}
    allcaps = np.array([[[8, 1, 2, 3, 9, 0]], [[8, 4, 5, 6, 9, 0]]])
    gt = ciderd.ground_truth_lists(allcaps, wm)
    assert gt == [[[1, 2, 3, 0]], [[4, 5, 6, 0]]]
    assert ciderd.tokens_to_str([1, 2, 0, 0, 0]) == "1 2 0"
    assert ciderd.tokens_to_str([1, 2, 3]) == "1 2 3"
    sampled = np.array([[1, 2, 3, 0], [1, 2, 3, 0]])
    greedy = np.array([[4, 5, 6, 0], [4, 5, 6, 0]])
    r = ciderd.self_critical_reward(s, sampled, greedy, gt)
    assert r.shape == (2, 4) and r.dtype == np.float32
    assert np.allclose(r[0], 10.0) and np.allclose(r[1], -10.0)


def test_native_scorer_matches_python_statement():
    """csrc/ciderd_host.hip (set_ciderd_*) against the pure-Python statement of the metric on a random corpus:
    ragged lengths, repeated words (count clipping), unseen n-grams, duplicated (image, caption) pairs."""
    rng = np.random.default_rng(0)
    V, NI = 40, 12
    refs = [[" ".join(str(int(w)) for w in rng.integers(1, V, rng.integers(3, 12))) + " 0" for _ in range(5)] for _ in range(NI)]
    df, docs = ciderd.document_frequency(refs)
    s = ciderd.CiderD(df, docs)
    res, gts = [], {}
    for i in range(60):
        img = int(rng.integers(0, NI))
        gts[i] = refs[img]
        if i % 7 == 0:
            cap = refs[img][int(rng.integers(0, 5))]                       # an exact reference
        elif i % 7 == 1 and res:
            cap = res[-1]["caption"][0]                                   # duplicate of the previous caption
            gts[i] = gts[i - 1]
        else:
            cap = " ".join(str(int(w)) for w in rng.integers(1, V + 5, rng.integers(0, 14))) + " 0"
        res.append({"image_id": i, "caption": [cap]})
    mean_n, per_n = s.compute_score(gts, res)
    assert s._native, "the native scorer must be in use (libset_hip.so exports set_ciderd_*)"
    mean_p, per_p = s._compute_score_py(gts, res)
    assert np.allclose(per_n, per_p, rtol=0, atol=1e-10), float(np.abs(per_n - per_p).max())
    assert abs(mean_n - mean_p) < 1e-10
    for i in (0, 7, 14):
        assert abs(per_n[i] - s.score(res[i]["caption"][0], gts[i])) < 1e-10
    assert per_n[0] > 1.0


def test_integer_fast_path_matches_string_path():
    """CiderD.score_token_ids / the integer route of self_critical_reward (no strings, one native call) against the
    string route (compute_score on stringified ids): ragged captions, captions without an <end>, empty captions (a lone 0),
    ids the scorer has never seen, several samples per image sharing one reference set."""
    rng = np.random.default_rng(1)
    V, B, L, n_samples = 60, 7, 12, 3
    gt = [[list(map(int, rng.integers(1, V, rng.integers(3, 10)))) + [0] for _ in range(5)] for _ in range(B)]
    df, docs = ciderd.document_frequency([[ciderd.tokens_to_str(c) for c in caps] for caps in gt])
    s = ciderd.CiderD(df, docs)
    N = B * n_samples

    def captions():
        arr = rng.integers(1, V + 20, (N, L))
        for i in range(N):
            if i % 5 == 0:
                continue                                   # no <end>: the whole row counts
            n = int(rng.integers(0, L))
            arr[i, n:] = 0
        arr[3] = np.array(gt[3 % B][0][:L] + [0] * max(0, L - len(gt[3 % B][0])))[:L]      # an exact reference
        return arr

    sampled, greedy = captions(), captions()
    gts_rep = list(gt) * n_samples
    fast = ciderd.self_critical_reward(s, sampled, greedy, gts_rep, cider_weight=0.7)
    assert s._native
    # string route, explicitly
    gts = {i: [ciderd.tokens_to_str(c) for c in gts_rep[i % N]] for i in range(2 * N)}
    res = [{"image_id": i, "caption": [ciderd.tokens_to_str(sampled[i])]} for i in range(N)]
    res += [{"image_id": N + i, "caption": [ciderd.tokens_to_str(greedy[i])]} for i in range(N)]
    _, per = s.compute_score(gts, res)
    _, per_py = s._compute_score_py(gts, res)
    want = 0.7 * (per[:N] - per[N:])
    assert np.allclose(per, per_py, atol=1e-10)
    assert fast.shape == (N, L) and np.allclose(fast[:, 0], want, atol=1e-6) and np.allclose(fast, fast[:, :1])
    direct = s.score_token_ids(np.concatenate([sampled, greedy]), np.arange(2 * N) % B, gt)
    assert np.allclose(direct, per, atol=1e-10)


def test_token_id_path_cuts_references_after_their_first_zero():
    """references that were not produced by ground_truth_lists (interior 0 = <end>): the integer path cuts them after the
    first 0, as tokens_to_str does on the string path"""
    from show_edit_tell_amd import ciderd
    refs = [[[5, 6, 7, 0, 9, 9], [5, 6, 8, 0]], [[1, 2, 3, 0, 4]]]
    cut = [[[5, 6, 7, 0], [5, 6, 8, 0]], [[1, 2, 3, 0]]]
    df, docs = ciderd.document_frequency([[ciderd.tokens_to_str(c) for c in caps] for caps in cut])
    s = ciderd.CiderD(df, docs)
    if s._handle() is None:                       # (the handle is built lazily: ask for it, do not peek at `_native`)
        pytest.skip("native scorer not built")
    hyps = np.array([[5, 6, 7, 0, 0, 0], [1, 2, 4, 0, 0, 0]], np.int64)
    a = s.score_token_ids(hyps, np.array([0, 1]), refs)
    b = s.score_token_ids(hyps, np.array([0, 1]), cut)
    assert np.array_equal(a, b)
    # and the cut matters: scoring against the UNCUT token lists through the string route gives something else
    gts = {i: [" ".join(str(t) for t in c) for c in caps] for i, caps in enumerate(refs)}
    res = [{"image_id": i, "caption": [ciderd.tokens_to_str(h)]} for i, h in enumerate(hyps)]
    cut_gts = {i: [ciderd.tokens_to_str(c) for c in caps] for i, caps in enumerate(cut)}
    _, per_cut = s.compute_score(cut_gts, res)
    assert np.allclose(a, per_cut, atol=1e-10)
