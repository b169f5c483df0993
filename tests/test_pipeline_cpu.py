"""CPU: the reader stage of the input pipeline (show_edit_tell_amd.pipeline.AdaptiveFeatureReader, row f4)
reproduces the reference's `collate_fn_train` (adaptive_features/editnet_adaptive.py:58-80) on synthetic per-image
files: zero padding to 100 regions, float64 / float32 files, fp32 output, batch order, ring-slot reuse."""
import os

import numpy as np
import torch

from show_edit_tell_amd import pipeline


def _write_dataset(root, n_images, F=64, R=100, seed=0):
    rng = np.random.default_rng(seed)
    att, fc = os.path.join(root, "cocobu_att"), os.path.join(root, "cocobu_fc")
    os.makedirs(att)
    os.makedirs(fc)
    feats = {}
    for i in range(n_images):
        img_id = 1000 + 7 * i
        n = int(rng.integers(10, R + 1))
        dt = np.float64 if i % 2 else np.float32           # the bottom-up dumps come in both precisions
        f = np.maximum(rng.standard_normal((n, F)), 0).astype(dt)
        m = f.mean(0).astype(np.float32)
        np.savez_compressed(os.path.join(att, "%d.npz" % img_id), feat=f)
        np.save(os.path.join(fc, "%d.npy" % img_id), m)
        feats[img_id] = (f, m)
    return att, fc, feats


def _reference_collate(feats, ids, R, F):
    # restated from editnet_adaptive.py:66-79 + the later .float(): zeros (B,100,F) float64, rows [:n] = feat
    images = np.zeros((len(ids), R, F))
    means = np.zeros((len(ids), F))
    for i, img_id in enumerate(ids):
        f, m = feats[img_id]
        images[i, :f.shape[0]] = f
        means[i] = m
    return images.astype(np.float32), means.astype(np.float32)


def test_reader_matches_reference_collate(tmp_path):
    F, R = 64, 100
    att, fc, feats = _write_dataset(str(tmp_path), 23, F, R)
    ids = sorted(feats)
    batches = [ids[0:6], ids[6:12], ids[12:18], ids[18:23], ids[0:6], ids[3:9], ids[9:10]]    # ragged last batches, > ring depth
    extras = [(torch.full((len(b), 1), k, dtype=torch.long),) for k, b in enumerate(batches)]
    reader = pipeline.AdaptiveFeatureReader(att, fc, batches, extras=extras, max_regions=R, feat_dim=F, workers=4,
                                            depth=3, pin=False)
    seen = 0
    for k, batch in enumerate(reader):
        images, means, tag = batch
        assert images.dtype == torch.float32 and images.shape == (len(batches[k]), R, F)
        want_i, want_m = _reference_collate(feats, batches[k], R, F)
        assert np.array_equal(images.numpy(), want_i)
        assert np.array_equal(means.numpy(), want_m)
        assert int(tag[0, 0]) == k
        reader.release(batch.slot)                     # what DevicePrefetcher does once the copy has finished
        seen += 1
    assert seen == len(batches)


def test_reader_reports_bad_files(tmp_path):
    att, fc, feats = _write_dataset(str(tmp_path), 3, 32, 20)
    ids = sorted(feats)
    reader = pipeline.AdaptiveFeatureReader(att, fc, [ids, [ids[0], 999999]], max_regions=20, feat_dim=32, workers=2,
                                            pin=False)
    it = iter(reader)
    b = next(it)
    reader.release(b.slot)
    try:
        next(it)
    except (FileNotFoundError, OSError):
        return
    raise AssertionError("a missing feature file must surface as an error, not as zeros")


def test_prefetcher_depth_is_clamped_to_the_reader_ring():
    """DevicePrefetcher(depth) >= reader.depth would leave the producer waiting for a free pinned slot and the prefetcher
    waiting for a batch (slots come back only when a staged batch is consumed): the depth is clamped to ring - 1"""
    class FakeReader:
        depth = 3
        def __iter__(self):
            return iter(())
    if not torch.cuda.is_available():
        # the constructor creates a copy stream: exercise the clamp rule itself
        ring, asked = FakeReader.depth, 8
        assert max(1, min(asked, ring - 1)) == 2
        return
    p = pipeline.DevicePrefetcher(FakeReader(), "cuda:0", depth=8)
    assert p.depth == 2
