"""Feature input pipeline: reader workers -> pinned staging ring -> asynchronous H2D
copies on a side HIP stream, so that reading + transferring the features of step i+1 overlaps the decode of
step i.  The reference reads synchronously in the training process (`DataLoader(num_workers=0)`,
`collate_fn_train` of `adaptive_features/editnet_adaptive.py:58-80`: per image one `np.load` of
`data/cocobu_att/<id>.npz['feat']` (n x 2048, n = 10..100) into a zero `(B,100,2048)` float64 array plus
`data/cocobu_fc/<id>.npy` (2048), then `.to(device)` / `.float()` at `:573-574`).

    reader = AdaptiveFeatureReader(att_dir, fc_dir, id_batches, extras=per_batch_tensors, workers=8)
    for images, images_mean, *extras in DevicePrefetcher(reader, device):       # device tensors, fp32
        scores, *_ = decoder(images, images_mean, caps, caplens, prev, prevlen)

Stages
  1. `AdaptiveFeatureReader`: a producer thread walks the id batches; the per-image loads of a batch are spread
     over a thread pool (np.load / zlib / the f64->f32 conversion release the GIL) and write straight into one
     slot of a ring of PINNED host buffers: rows >= n are zeroed, values are converted to fp32 once (the
     reference converts the whole padded float64 batch on the device).
  2. `DevicePrefetcher`: issues the H2D copies of up to `depth` batches ahead on its own stream, hands a batch to
     the consumer after making the consumer's stream wait on the copy event (no host sync on the compute
     stream), and returns the pinned slot to the reader once the copy has completed.
The fixed 36-region features of `editnet.py:24-74` (`train36.hdf5` / `val36.hdf5`, dataset `image_features`
(I,36,2048)) go through `FixedFeatureReader`: it reads the HDF5 datasets themselves when h5py imports, or `.npy` files
converted from them once (`FixedFeatureReader.convert_hdf5`; memory-mapped here, h5py is not part of this image), and
gathers a batch's rows with a thread pool straight into the pinned ring.
"""
from __future__ import annotations

import os
import queue
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch


class _PinnedRing:
    """The ring of pinned host slots both readers produce into.  Allocated ONCE (pinning 37 MB buffers is slow), shared by
    every iteration of the reader.  A slot is `lent` from the moment a batch in it is handed out until release(): a new
    iteration starts with the slots that are not lent (a batch of an earlier iteration whose H2D copy is still in flight
    keeps its slot until it is released — late releases just add the slot to the current free list, never a second token for a
    slot that is already free).  Plain `for batch in reader:` loops release the previous batch's slot when the next one is
    asked for; a DevicePrefetcher (which keeps several batches in flight and releases each after its copy) switches that off
    through `manual_release_next()`, which holds for the ONE iteration started next (a later plain loop over the same reader
    releases per batch again).  An abandoned iteration (`break`) stops AND JOINS its producer before the generator returns, so
    no worker of an old iteration can still be filling a slot that the next iteration hands to its own producer."""

    def _ring_init(self):
        self._slots, self._free, self._lent, self._ring_lock = None, None, set(), threading.Lock()
        self._manual_next = False

    def manual_release_next(self):
        """the iteration started by the next iter(reader) leaves every slot lent until release(slot) is called for it"""
        self._manual_next = True

    def _take_manual(self):
        manual, self._manual_next = self._manual_next, False
        return manual

    def _iterate(self, work, fill, view, manual):
        """producer thread: for (bi, item) in enumerate(work): take a free slot, fill(slot, item) -> n rows, queue it;
        consumer (this generator): view(bi, slot, n) -> HostBatch."""
        ready = queue.Queue(maxsize=self.depth)
        stop = threading.Event()

        def producer():
            try:
                with ThreadPoolExecutor(self.workers) as pool:
                    for bi, item in enumerate(work):
                        slot = None
                        while slot is None and not stop.is_set():
                            try:
                                slot = self._free.get(timeout=0.1)
                            except queue.Empty:
                                pass
                        if stop.is_set():
                            return
                        n = fill(pool, slot, item)
                        while not stop.is_set():
                            try:
                                ready.put((bi, slot, n), timeout=0.1)
                                break
                            except queue.Full:
                                pass
                ready.put(None)
            except BaseException as e:            # surface reader errors in the consumer
                ready.put(e)

        th = threading.Thread(target=producer, daemon=True, name="set-reader-producer")
        th.start()
        prev = None
        try:
            while True:
                item = ready.get()
                if item is None:
                    return
                if isinstance(item, BaseException):
                    raise item
                bi, slot, n = item
                batch = view(bi, slot, n)
                batch.slot, batch.owner = slot, self
                if not manual and prev is not None:
                    self.release(prev)                   # plain iteration: the previous batch's slot goes back
                self._lend(slot)
                prev = slot
                yield batch
        finally:
            stop.set()
            while th.is_alive():                         # the old producer (and its pool) is gone before anyone iterates again
                try:
                    ready.get_nowait()
                except queue.Empty:
                    pass
                th.join(timeout=0.05)
            if not manual and prev is not None:
                self.release(prev)

    def _ring_ensure(self, make_slot):
        if self._slots is None:
            self._slots = [make_slot() for _ in range(self.depth)]
        with self._ring_lock:
            self._free = queue.Queue()
            for i in range(self.depth):
                if i not in self._lent:
                    self._free.put(i)

    def _lend(self, slot):
        with self._ring_lock:
            self._lent.add(slot)

    def release(self, slot):
        """hand a pinned slot back (DevicePrefetcher: once its H2D copy has completed); releasing twice is harmless"""
        if slot is None or self._free is None:
            return
        with self._ring_lock:
            if slot in self._lent:
                self._lent.discard(slot)
                self._free.put(slot)


class AdaptiveFeatureReader(_PinnedRing):
    """Iterable of `(images (B,R,F) fp32, images_mean (B,F) fp32, *extras)` host batches in pinned memory.

    att_dir / fc_dir : directories holding `<image_id>.npz` (array 'feat', (n,F), n <= R) and `<image_id>.npy` (F,)
    id_batches       : sequence of sequences of image ids (one inner sequence per batch)
    extras           : optional sequence (same length) of tuples of tensors passed through unchanged
    """

    def __init__(self, att_dir, fc_dir, id_batches, extras=None, max_regions=100, feat_dim=2048, workers=8, depth=4,
                 pin=None):
        self.att_dir, self.fc_dir = att_dir, fc_dir
        self.id_batches = [[int(i) for i in b] for b in id_batches]
        self.extras = extras
        self.R, self.F = max_regions, feat_dim
        self.workers, self.depth = max(1, workers), max(2, depth)
        self.pin = torch.cuda.is_available() if pin is None else pin
        self._ring_init()

    def __len__(self):
        return len(self.id_batches)

    def _alloc(self):
        bmax = max(len(b) for b in self.id_batches)
        mk = lambda *shape: (torch.empty(*shape, dtype=torch.float32).pin_memory() if self.pin
                             else torch.empty(*shape, dtype=torch.float32))
        self._ring_ensure(lambda: (mk(bmax, self.R, self.F), mk(bmax, self.F)))

    def _load_one(self, img_np, mean_np, row, image_id):
        with np.load(os.path.join(self.att_dir, "%d.npz" % image_id)) as z:
            feat = z["feat"]
        n = feat.shape[0]
        if n > self.R or feat.shape[1] != self.F:
            raise ValueError("image %d: features %s do not fit (%d,%d)" % (image_id, feat.shape, self.R, self.F))
        img_np[row, :n] = feat                       # converts to fp32 on the way (float64 files included)
        img_np[row, n:] = 0.0                        # zero padding = the reference's np.zeros((B,100,2048))
        mean_np[row] = np.load(os.path.join(self.fc_dir, "%d.npy" % image_id))

    def __iter__(self):
        self._alloc()

        def fill(pool, slot, ids):
            img, mean = self._slots[slot]
            img_np, mean_np = img.numpy(), mean.numpy()
            list(pool.map(lambda a: self._load_one(img_np, mean_np, *a), enumerate(ids)))
            return len(ids)

        def view(bi, slot, n):
            img, mean = self._slots[slot]
            return HostBatch((img[:n], mean[:n]) + tuple(self.extras[bi] if self.extras is not None else ()))

        return self._iterate(self.id_batches, fill, view, self._take_manual())


class FixedFeatureReader(_PinnedRing):
    """Iterable of `(images (B,R,F) fp32, *extras)` host batches in pinned memory for the fixed-region feature files of the
    reference's datasets (`COCOTrainDataset.__getitem__`, editnet.py:46-74: `objdet = self.objdet[i // cpi]`; the row
    `objdet[1]` of `val_features` if `objdet[0] == "v"` else of `train_features`).

    stores      : {"t": path_or_array, "v": path_or_array}: per split an `.npy` file (memory-mapped), an HDF5 file (needs
                  h5py; dataset `image_features`) or any array-like indexable by row with shape (I,R,F)
    ref_batches : sequence of batches, each a sequence of `(split, row)` pairs (`split` = "v" -> the "v" store, anything
                  else -> the "t" store, exactly the reference's test) — i.e. `objdet[:2]` of every sample of the batch
    extras      : optional per-batch tuples of tensors passed through unchanged (captions, lengths, previous captions:
                  the reference's other `__getitem__` fields, collated by the caller)
    A batch's rows are gathered by `workers` threads (the memory-mapped / HDF5 reads and the copies release the GIL) into
    one slot of a ring of pinned buffers; DevicePrefetcher copies from there and returns the slot."""

    DATASET = "image_features"

    def __init__(self, stores, ref_batches, extras=None, workers=8, depth=4, pin=None):
        self._files = []
        self.stores = {k: self._open(v) for k, v in stores.items()}
        if "t" not in self.stores:
            raise ValueError('stores needs at least the "t" (train) split')
        self.ref_batches = [[(str(sp), int(row)) for sp, row in b] for b in ref_batches]
        self.extras = extras
        any_store = next(iter(self.stores.values()))
        self.R, self.F = int(any_store.shape[1]), int(any_store.shape[2])
        self.workers, self.depth = max(1, workers), max(2, depth)
        self.pin = torch.cuda.is_available() if pin is None else pin
        self._ring_init()
        # the splits and rows every batch names are checked here, not in a worker thread half way through an epoch
        for b in self.ref_batches:
            for sp, row in b:
                key = "v" if sp == "v" else "t"
                if key not in self.stores:
                    raise KeyError('a sample names the "%s" split but stores has no "%s" entry' % (sp, key))
                if not 0 <= row < int(self.stores[key].shape[0]):
                    raise IndexError("row %d of split %r is outside its store (%d rows)" % (row, sp, int(self.stores[key].shape[0])))

    def _open(self, src):
        if isinstance(src, (str, os.PathLike)):
            path = os.fspath(src)
            if path.endswith(".npy"):
                return np.load(path, mmap_mode="r")
            try:
                import h5py
            except ImportError as e:
                raise RuntimeError("%s is an HDF5 file and h5py is not installed: convert it once with "
                                   "FixedFeatureReader.convert_hdf5(path, path_npy) where h5py is available" % path) from e
            f = h5py.File(path, "r")
            self._files.append(f)
            return f[self.DATASET]
        return src

    @staticmethod
    def convert_hdf5(h5_path, npy_path, dataset="image_features", chunk=1024):
        """one-off: copy `dataset` of an HDF5 feature file into a `.npy` file that this reader memory-maps"""
        import h5py
        with h5py.File(h5_path, "r") as f:
            d = f[dataset]
            out = np.lib.format.open_memmap(npy_path, mode="w+", dtype=np.float32, shape=tuple(d.shape))
            for i in range(0, d.shape[0], chunk):
                out[i:i + chunk] = d[i:i + chunk]
            out.flush()
        return npy_path

    def __len__(self):
        return len(self.ref_batches)

    def _alloc(self):
        bmax = max(len(b) for b in self.ref_batches)
        mk = lambda *shape: (torch.empty(*shape, dtype=torch.float32).pin_memory() if self.pin
                             else torch.empty(*shape, dtype=torch.float32))
        self._ring_ensure(lambda: mk(bmax, self.R, self.F))

    def _load_one(self, img_np, i, ref):
        split, row = ref
        store = self.stores["v"] if split == "v" else self.stores["t"]        # editnet.py:59-62
        img_np[i] = store[row]                                                # (R,F) -> fp32 (FloatTensor in the reference)

    def __iter__(self):
        self._alloc()

        def fill(pool, slot, refs):
            img_np = self._slots[slot].numpy()
            list(pool.map(lambda a: self._load_one(img_np, *a), enumerate(refs)))
            return len(refs)

        def view(bi, slot, n):
            return HostBatch((self._slots[slot][:n],) + tuple(self.extras[bi] if self.extras is not None else ()))

        return self._iterate(self.ref_batches, fill, view, self._take_manual())


class HostBatch(tuple):
    """a tuple of host tensors that remembers which pinned ring slot it lives in"""
    slot = None
    owner = None


class DevicePrefetcher:
    """Asynchronous H2D staging of an iterable of host batches (tuples of tensors) on a side stream.

    `timeline` (when record_timing=True) collects, per batch, timing events around its H2D copy on the copy stream
    (tests use them to show the copy of batch i+1 running while batch i is being decoded)."""

    def __init__(self, iterable, device, depth: int = 2, record_timing: bool = False, begin_ahead=None, streams: int = 1,
                 stream_priority=None, side_streams=None):
        """begin_ahead: optional callable(batch_on_device), run on the copy stream right after a batch's H2D copies — e.g.
        `lambda b: decoder.begin_ahead(b.prev, b.plen, b.features)` (editnet_rl.DecoderC.begin_ahead): the per-sequence
        prologue of batch i+1 then runs underneath the timestep loop of batch i, and the `decoder(...)` call for batch i+1
        finds it done.  What a caller that issues one decode after the other (the reference's train() / evaluate()) gains."""
        self.begin_ahead = begin_ahead
        if isinstance(iterable, _PinnedRing):
            iterable.manual_release_next()       # several batches in flight here: each slot is returned after ITS copy
        self.it = iter(iterable)
        self.device = torch.device(device)
        # `streams` side streams, used round-robin per staged batch (1 = the copy stream alone).  With
        # begin_ahead = decoder.decode_ahead and streams = depth, `depth` whole decodes are in flight while the caller walks
        # the batches one by one.
        # stream_priority: priority of the side streams (torch.cuda.Stream: lower = more urgent; None = default)
        # side_streams: the caller's own stream objects instead (e.g. torch.cuda.ExternalStream of a CU-masked HIP stream)
        kw = {} if stream_priority is None else {"priority": int(stream_priority)}
        self.streams = list(side_streams) if side_streams else [torch.cuda.Stream(self.device, **kw) for _ in range(max(1, streams))]
        self.stream = self.streams[0]
        self._staged = 0
        self.depth = max(1, depth)
        # a reader that lends out pinned ring slots (AdaptiveFeatureReader: `depth` slots, one returned per consumed batch)
        # must keep at least one slot to produce into: staging `depth` batches ahead of a `depth`-slot ring would leave the
        # producer waiting for a slot and this side waiting for a batch
        ring = getattr(iterable, "depth", None)
        if isinstance(ring, int) and ring >= 1:
            self.depth = max(1, min(self.depth, ring - 1))
        self.queue = []
        self.record_timing = record_timing
        self.timeline = []

    def _stage(self):
        try:
            batch = next(self.it)
        except StopIteration:
            return False
        slot, owner = getattr(batch, "slot", None), getattr(batch, "owner", None)
        if not isinstance(batch, (tuple, list)):
            batch = (batch,)
        stream = self.streams[self._staged % len(self.streams)]
        self._staged += 1
        if any(torch.is_tensor(t) and t.is_cuda for t in batch):
            # tensors that already live on the device were produced on the caller's stream: order the side stream after it
            stream.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(stream):
            t0 = None
            if self.record_timing:
                t0 = torch.cuda.Event(enable_timing=True)
                t0.record(stream)
            dev = []
            for t in batch:
                if torch.is_tensor(t):
                    if not t.is_cuda and not t.is_pinned():
                        t = t.pin_memory()
                    t = t.to(self.device, non_blocking=True)
                dev.append(t)
            ev = torch.cuda.Event(enable_timing=self.record_timing)
            ev.record(stream)
            if self.record_timing:
                self.timeline.append((t0, ev))
            if self.begin_ahead is not None:               # (after `ev`: the consumer's copy dependency stays the copy alone)
                self.begin_ahead(tuple(dev))
        self.queue.append((tuple(dev), ev, slot, owner))
        return True

    def __iter__(self):
        while len(self.queue) < self.depth and self._stage():
            pass
        while self.queue:
            batch, ev, slot, owner = self.queue.pop(0)
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(ev)                         # device-side dependency only: the host does not block the compute stream
            for t in batch:
                if torch.is_tensor(t):
                    t.record_stream(cur)
            if owner is not None:                      # the pinned slot is reusable once its copy has finished
                ev.synchronize()
                owner.release(slot)
            self._stage()
            yield batch
