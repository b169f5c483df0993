"""CPU: the build-owned CIDEr-D scorer (show_edit_tell_amd.ciderd) against hand-stated values of the
published metric.  The reference's own scorer is an un-vendored external checkout, so these values are
NOT pinned to the reference ("parity unpinned")."""
import math

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
