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


def _fixed_dataset(tmp_path, n_train=17, n_val=9, R=36, F=32, cpi=5, seed=3):
    rng = np.random.default_rng(seed)
    tr = np.maximum(rng.standard_normal((n_train, R, F)), 0).astype(np.float32)
    va = np.maximum(rng.standard_normal((n_val, R, F)), 0).astype(np.float32)
    tp, vp = os.path.join(str(tmp_path), "train36.npy"), os.path.join(str(tmp_path), "val36.npy")
    np.save(tp, tr)
    np.save(vp, va)
    n_img = 12
    objdet = [["v" if i % 3 == 0 else "t", int(rng.integers(0, n_val if i % 3 == 0 else n_train))] for i in range(n_img)]
    names = ["img%03d" % i for i in range(n_img)]
    captions = [[int(x) for x in rng.integers(1, 50, 20)] for _ in range(n_img * cpi)]
    caplens = [int(rng.integers(7, 21)) for _ in range(n_img * cpi)]
    util = {n: {"encoded_previous_caption": [int(x) for x in rng.integers(1, 50, 18)],
                "previous_caption_length": [int(rng.integers(1, 19))]} for n in names}
    return dict(tr=tr, va=va, tp=tp, vp=vp, objdet=objdet, names=names, captions=captions, caplens=caplens, util=util, cpi=cpi)


def test_fixed_reader_matches_the_reference_getitem(tmp_path):
    """FixedFeatureReader against COCOTrainDataset.__getitem__ (editnet.py:46-74): the split test on objdet[0], the row
    objdet[1], fp32 features; when /root/reference is present the reference's OWN __getitem__ (AST-sliced, its HDF5 datasets
    replaced by the same arrays) produces the expected tuples, otherwise its restatement below does."""
    from oracle import ref_slice
    ds = _fixed_dataset(tmp_path)
    cpi = ds["cpi"]
    n = len(ds["captions"])

    def restated(i):
        od = ds["objdet"][i // cpi]
        return torch.from_numpy((ds["va"] if od[0] == "v" else ds["tr"])[od[1]].copy())

    getitem = restated
    if ref_slice.have_reference():
        cls = ref_slice.load_classes("editnet.py", ("COCOTrainDataset",))["COCOTrainDataset"]
        obj = cls.__new__(cls)
        obj.train_features, obj.val_features, obj.cpi = ds["tr"], ds["va"], cpi
        obj.captions, obj.caplens, obj.names = ds["captions"], ds["caplens"], ds["names"]
        obj.caption_util, obj.objdet, obj.dataset_size = ds["util"], ds["objdet"], n
        getitem = lambda i: obj[i][0]
        assert torch.equal(obj[7][0], restated(7))
    order = np.random.default_rng(0).permutation(n)
    batches = [order[i:i + 8].tolist() for i in range(0, n, 8)]                     # ragged last batch, > ring depth
    refs = [[tuple(ds["objdet"][i // cpi][:2]) for i in b] for b in batches]
    extras = [(torch.tensor([ds["caplens"][i] for i in b]).view(-1, 1),) for b in batches]
    reader = pipeline.FixedFeatureReader({"t": ds["tp"], "v": ds["vp"]}, refs, extras=extras, workers=3, depth=3, pin=False)
    assert len(reader) == len(batches) and (reader.R, reader.F) == (36, 32)
    for k, batch in enumerate(reader):
        img, caplen = batch
        assert img.dtype == torch.float32 and img.shape == (len(batches[k]), 36, 32)
        want = torch.stack([getitem(i) for i in batches[k]])
        assert torch.equal(img, want)
        assert torch.equal(caplen, extras[k][0])
        reader.release(batch.slot)
    # array-likes are accepted as stores too (an open h5py dataset is one)
    r2 = pipeline.FixedFeatureReader({"t": ds["tr"], "v": ds["va"]}, refs[:1], pin=False)
    b0 = next(iter(r2))
    assert torch.equal(b0[0], torch.stack([getitem(i) for i in batches[0]]))


def test_fixed_reader_needs_h5py_for_hdf5(tmp_path):
    import pytest
    try:
        import h5py  # noqa: F401
        pytest.skip("h5py present")
    except ImportError:
        pass
    p = os.path.join(str(tmp_path), "train36.hdf5")
    open(p, "wb").write(b"x")
    with pytest.raises(RuntimeError, match="convert"):
        pipeline.FixedFeatureReader({"t": p}, [[("t", 0)]], pin=False)


def test_fixed_reader_ring_is_allocated_once_and_survives_stale_and_missing_releases(tmp_path):
    """the pinned ring is allocated once and reused by every epoch; a plain `for` loop (nobody calls release)
    runs through more batches than the ring has slots; a batch of an earlier iteration that is released late never gives the
    ring a second token for a slot; samples that name a split without a store are refused at construction."""
    import pytest
    ds = _fixed_dataset(tmp_path)
    cpi = ds["cpi"]
    n = len(ds["captions"])
    batches = [list(range(i, min(i + 4, n))) for i in range(0, n, 4)]
    refs = [[tuple(ds["objdet"][i // cpi][:2]) for i in b] for b in batches]
    assert len(batches) > 3
    reader = pipeline.FixedFeatureReader({"t": ds["tp"], "v": ds["vp"]}, refs, workers=2, depth=3, pin=False)
    seen = [b[0].clone() for b in reader]                       # plain iteration, no release() anywhere: must not stall
    assert len(seen) == len(batches)
    slots = [t.data_ptr() for t in reader._slots]
    again = [b[0].clone() for b in reader]
    assert [t.data_ptr() for t in reader._slots] == slots, "the ring is allocated once"
    assert all(torch.equal(a, b) for a, b in zip(seen, again))
    # manual release (what DevicePrefetcher does): break early holding one batch, start a new epoch, release late
    reader.manual_release_next()
    it = iter(reader)
    held = next(it)
    it.close()
    # close() stops AND joins the old producer: no thread of the abandoned iteration can still be filling a slot
    # when the next iteration's producer starts taking slots from the rebuilt free list
    import threading
    assert sum(1 for t in threading.enumerate() if t.name.startswith("set-reader")) == 0
    reader.manual_release_next()
    it2 = iter(reader)
    first = next(it2)
    assert held.slot in reader._lent and first.slot != held.slot
    reader.release(held.slot)
    reader.release(held.slot)                                   # twice: harmless
    reader.release(first.slot)
    rest = []
    for b in it2:
        rest.append(b[0].clone())
        reader.release(b.slot)
    assert len(rest) == len(batches) - 1
    assert reader._free.qsize() + len(reader._lent) == reader.depth
    # the manual mode held for those iterations only: a plain loop over the same reader releases per batch again and runs
    # through more batches than the ring has slots (the flag used to stick to the reader)
    assert len([b[0].clone() for b in reader]) == len(batches)
    with pytest.raises(KeyError):
        pipeline.FixedFeatureReader({"t": ds["tp"]}, [[("v", 0)]], pin=False)
    with pytest.raises(IndexError):
        pipeline.FixedFeatureReader({"t": ds["tp"]}, [[("t", 10 ** 6)]], pin=False)
