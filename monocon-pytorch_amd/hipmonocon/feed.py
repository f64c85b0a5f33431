"""Host -> device feed of the train loop (the reference moves every batch with ``move_data_device`` on the compute
stream right before the forward, engine/monocon_engine.py:86-88, utils/engine_utils.py:58-66).

At 47 ms per step a 32-frame batch of float32 CHW images is 189 MB: copied from pageable memory on the compute stream it costs
more than a quarter of the step, with the GPU idle.  ``DevicePrefetcher`` wraps the DataLoader (built with ``pin_memory=True``,
so a loader thread page-locks the batch) and keeps ONE batch ahead: batch i+1 is uploaded on a copy stream while step i runs,
the compute stream waits on the upload's event only.  The labels are checked on the host before they leave it
(train.labels_ok_on_host), so the forward does not read a verdict back from the device; a batch that fails the host check is
handed on unmarked and the device-side check raises as before.

``RingLoader`` is the DataLoader for batches of this size.  A worker of ``torch.utils.data.DataLoader`` stacks its 32 frames
(a 189 MB copy), moves the stack into a fresh shared-memory segment (another one) and the pin thread of the training process
copies it into page-locked memory (a third, over freshly mapped pages): measured 75-105 ms per batch with 8-14 workers, more
than a train step.  Here every worker writes each frame ONCE, straight into its place in a ring of batch slots that lives in
shared memory for the life of the loader and is page-locked (hipHostRegister) in the training process: the upload is one
asynchronous copy out of the ring, and only the labels travel through the workers' queues (9 ms per batch on the same box).

``DeferredScalars`` reads the per-step loss back one step late: the reference calls ``total_loss.item()`` inside the step
(monocon_engine.py:89), which drains the stream before the optimizer is even enqueued.  The values, their order and the lists
they go into are the same; they arrive a step later, except on the iterations that print them.
"""
import collections

import torch

from . import train as _train


WORKER_CONTEXT = "forkserver"      # how loaders beside a HIP device start their workers (see RingLoader)


def prepare_worker_context():
    """the fork server imports torch ONCE and the workers are forked from it with the modules in place (otherwise every worker of
    every loader imports torch by itself: seconds each, much longer on a box whose image is not in the page cache yet).  Has to
    run before the first worker of the process is started; a no-op afterwards."""
    if WORKER_CONTEXT == "forkserver":
        import multiprocessing
        multiprocessing.set_forkserver_preload(["torch", "numpy", "torch.utils.data", __name__])


def _upload(obj, device):
    if isinstance(obj, torch.Tensor):
        return obj.to(device, non_blocking=True)
    if isinstance(obj, dict):
        return {k: _upload(v, device) for k, v in obj.items()}
    return obj          # metadata (lists of shapes, calibration objects) stays on the host, as in move_data_device


def _record(obj, stream):
    if isinstance(obj, torch.Tensor):
        if obj.is_cuda:
            obj.record_stream(stream)
    elif isinstance(obj, dict):
        for v in obj.values():
            _record(v, stream)


class DevicePrefetcher:
    """iterate ``loader``; yield its batches with every tensor (top level and ``label``) on ``device``, uploaded one batch
    ahead on a copy stream.  ``detector``: the module whose forward validates the labels (hipmonocon.train._require_objects);
    batches that pass the host-side check are marked as validated on it.  Without a HIP device the batches pass through
    unchanged (the CPU plumbing run of SURVEY 8d config 1)."""

    def __init__(self, loader, device, detector=None, num_classes=3):
        self.loader, self.detector, self.num_classes = loader, detector, num_classes
        self.device = torch.device(device) if device is not None else torch.device("cpu")
        self.on_device = self.device.type == "cuda" and torch.cuda.is_available()
        self.copy_stream = torch.cuda.Stream(self.device) if self.on_device else None

    def __len__(self):
        return len(self.loader)

    def _stage(self, batch):
        ok = False
        if self.detector is not None and "label" in batch and isinstance(batch.get("img"), torch.Tensor):
            img = batch["img"]              # (B, 3, Hp, Wp) float32, or transforms.DeferredImage's raw (B, Hp, Wp, 3) uint8 frames
            pad_hw = tuple(img.shape[1:3]) if "img_aug" in batch else tuple(img.shape[-2:])
            ok = _train.labels_ok_on_host(batch["label"], pad_hw, self.num_classes)
        with torch.cuda.stream(self.copy_stream):
            dev = _upload(batch, self.device)
            ev = torch.cuda.Event()
            ev.record(self.copy_stream)
        return dev, ev, ok

    def __iter__(self):
        if not self.on_device:
            yield from self.loader
            return
        staged = None
        ring = hasattr(self.loader, "host_batches")      # RingLoader: frames are uploaded straight out of its pinned ring
        for batch in (self.loader.host_batches() if ring else self.loader):
            nxt = self._stage(batch)
            if ring:
                self.loader.note_upload(nxt[1])          # the slot must not be refilled before this event
            if staged is not None:
                yield self._hand_over(staged)
            staged = nxt
        if staged is not None:
            yield self._hand_over(staged)

    def _hand_over(self, staged):
        dev, ev, ok = staged
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(ev)
        _record(dev, cur)          # allocated on the copy stream, used on this one: the allocator must not recycle it early
        if ok:
            _train.note_labels_validated(self.detector, dev["label"])
        return dev


class DeferredScalars:
    """0-dim device tensors read back without draining the stream: ``push`` enqueues a copy into pinned memory plus an event,
    ``ready(keep)`` returns, in order, the values of all but the ``keep`` newest entries (their events have long fired when
    the next step has been enqueued), ``ready(0)`` all of them (a synchronisation on the newest)."""

    def __init__(self):
        self.pending = collections.deque()

    def push(self, t):
        t = t.detach()
        if not t.is_cuda:
            self.pending.append((float(t), None))
            return
        host = torch.empty((), dtype=t.dtype, pin_memory=True)
        host.copy_(t, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(t.device))
        self.pending.append((host, ev))

    def ready(self, keep=1):
        out = []
        while len(self.pending) > keep:
            host, ev = self.pending.popleft()
            if ev is not None:
                ev.synchronize()
                host = float(host)
            out.append(host)
        return out


class _RingBatchSampler(torch.utils.data.Sampler):
    """the batches of ``base`` (lists of sample indices) with the ring slot of the batch and the sample's place in it attached
    to every index: the DATASET (in a worker) writes the frame there.  ENDLESS: one pass over ``base`` after the other, so that
    the workers fill slots for the next epoch while the tail of this one (and whatever the loop does between two epochs) is
    still being consumed; a sampler with ``set_epoch`` (DistributedSampler) is moved on by one per pass, which is what the
    engine does at the start of every epoch (RingLoader's _EpochGuard notices a caller that does otherwise)."""

    def __init__(self, base, nslots, sampler):
        self.base, self.nslots, self.count, self.sampler = base, nslots, 0, sampler
        self.epoch_of_next_pass = None

    def __len__(self):
        return len(self.base)

    def __iter__(self):
        paced = hasattr(self.sampler, "set_epoch") and hasattr(self.sampler, "epoch")
        self.epoch_of_next_pass = self.sampler.epoch if paced else None
        while True:
            if paced:
                self.sampler.set_epoch(self.epoch_of_next_pass)
                self.epoch_of_next_pass += 1
            for idxs in self.base:
                slot = self.count % self.nslots
                self.count += 1                 # runs on over passes: slots are reused in dispatch order
                yield [(int(i), slot, pos) for pos, i in enumerate(idxs)]


class _EpochGuard:
    """what RingLoader shows as its ``sampler``: the caller's sampler, with ``set_epoch`` watched -- the passes the workers
    are already filling slots for assumed epoch, epoch + 1, ...; a caller that sets anything else (a resume, a replay) makes
    the loader start over from that epoch instead of handing out batches of the wrong permutation"""

    def __init__(self, loader, sampler):
        object.__setattr__(self, "_loader", loader)
        object.__setattr__(self, "_sampler", sampler)

    def __getattr__(self, name):
        return getattr(self._sampler, name)

    def __setattr__(self, name, value):
        setattr(self._sampler, name, value)

    def __iter__(self):
        return iter(self._sampler)

    def __len__(self):
        return len(self._sampler)

    def set_epoch(self, epoch):
        self._loader._epoch_wanted(epoch)


class _RingDataset(torch.utils.data.Dataset):
    def __init__(self, base, ring):
        self.base, self.ring = base, ring

    def __len__(self):
        return len(self.base)

    def __getitem__(self, key):
        idx, slot, pos = key
        d = dict(self.base[idx])
        img = d["img"]
        if isinstance(img, torch.Tensor) and tuple(img.shape) == tuple(self.ring.shape[2:]) and img.dtype == self.ring.dtype:
            self.ring[slot, pos].copy_(img)
            d["img"] = img.new_empty((0,))          # the frame is in the ring; the sample's collate sees a placeholder
            d["_ring"] = (slot, pos)
        return d                                    # (another shape / dtype: the frame travels with the sample, as in a DataLoader)


class _RingCollate:
    """the dataset's collate over samples whose frames are already in the ring (a module-level callable: the workers are
    started by a fork server, their arguments are pickled)"""

    def __init__(self, collate_fn, ring):
        self.collate_fn, self.ring = collate_fn, ring

    def __call__(self, samples):
        where = [d.pop("_ring", None) for d in samples]
        in_ring = all(w is not None for w in where)
        if not in_ring:             # frames of another shape among them: every frame travels with its sample (and the
            for d, w in zip(samples, where):        # dataset's collate decides what a mixed batch is, as under a DataLoader)
                if w is not None:
                    d["img"] = self.ring[w[0], w[1]].clone()
        out = self.collate_fn(samples)
        if in_ring:
            out["img"] = ("ring", where[0][0], len(samples))
        return out


class RingLoader:
    """DataLoader for (img, label, ...) sample dicts whose frames all have one shape: same batches, same order and the same
    collated dict as ``DataLoader(dataset, batch_size, sampler=..., collate_fn=...)``, the frames written by the workers into a
    shared, page-locked ring of batch slots instead of travelling through the workers' queues.

    Iterating it yields host batches (``img`` a COPY of the slot); under a DevicePrefetcher the frames are uploaded straight out
    of the ring and a slot is refilled only after its upload has completed: a batch is dispatched to a worker when the consumer
    takes one, at most ``prefetch_factor * num_workers`` are in flight, and the ring has three slots more than that -- the loader
    waits for the upload of the batch handed out three batches ago before it asks for the next one.

    One iterator of the underlying DataLoader serves all epochs (the batch sampler is endless, an epoch is ``len(self)`` batches
    of it): the first batch of an epoch is already in the ring when the previous epoch ends -- filling it takes a worker
    32 decodes, as long as dozens of steps."""

    EXTRA_SLOTS = 3

    def __init__(self, dataset, batch_size, num_workers, shuffle=False, sampler=None, drop_last=False, collate_fn=None,
                 worker_init_fn=None, prefetch_factor=2, image_shape=None, image_dtype=torch.float32, pin=None, generator=None,
                 mp_context=None, timeout=0):
        from torch.utils.data import BatchSampler, DataLoader, RandomSampler, SequentialSampler
        if num_workers < 1:
            raise ValueError("RingLoader needs worker processes (num_workers >= 1); use a DataLoader without them")
        self.dataset, self.batch_size, self.num_workers = dataset, int(batch_size), int(num_workers)
        self.collate_fn = collate_fn if collate_fn is not None else dataset.collate_fn
        if sampler is None:
            sampler = RandomSampler(dataset, generator=generator) if shuffle else SequentialSampler(dataset)
        self._sampler = sampler
        self.sampler = _EpochGuard(self, sampler) if hasattr(sampler, "set_epoch") else sampler
        if image_shape is None:                     # one sample tells the frames' shape and type (float32 CHW under the host
            probe = dataset[0]["img"]               # transforms, uint8 HWC when the image work is deferred to the device)
            image_shape, image_dtype = tuple(probe.shape), probe.dtype
        self.nslots = prefetch_factor * self.num_workers + self.EXTRA_SLOTS
        self.ring = torch.empty((self.nslots, self.batch_size) + tuple(image_shape), dtype=image_dtype).share_memory_()
        self.ring.zero_()                           # touch every page once, here
        self.pinned = False
        if pin is None:
            pin = torch.cuda.is_available()
        if pin:
            try:
                rc = torch.cuda.cudart().cudaHostRegister(self.ring.data_ptr(), self.ring.numel() * self.ring.element_size(), 0)
                self.pinned = int(rc) == 0
            except Exception as e:      # noqa: BLE001  (no such entry point in this build)
                rc = e
            if not self.pinned:
                import warnings
                warnings.warn("RingLoader: hipHostRegister of the %.1f GB ring failed (%s): uploads will be staged copies"
                              % (self.ring.numel() * self.ring.element_size() / 1e9, rc))
        self._uploads = collections.deque()
        self._bs = _RingBatchSampler(BatchSampler(sampler, self.batch_size, drop_last), self.nslots, sampler)
        self._it, self._midway, self._epochs_out, self._epoch0 = None, False, 0, None
        # The workers live as long as the loader and come from a FORK SERVER, not from a fork of this process: children forked
        # from a process that holds page-locked memory (the ring, torch's pinned blocks) slow every launch of the parent's
        # GPU work down for as long as they live -- measured on this platform: 45 -> 600 ms per train step (COW
        # write-protection of the parent's pages against the driver's MMU notifiers on its pinned ranges).  Dataset, collate
        # and worker_init_fn therefore have to be picklable.
        if (mp_context or WORKER_CONTEXT) == "forkserver":
            prepare_worker_context()
        self._dl = DataLoader(_RingDataset(dataset, self.ring), batch_sampler=self._bs, num_workers=self.num_workers,
                              collate_fn=_RingCollate(self.collate_fn, self.ring), worker_init_fn=worker_init_fn,
                              prefetch_factor=prefetch_factor, multiprocessing_context=mp_context or WORKER_CONTEXT,
                              persistent_workers=True, timeout=timeout)       # timeout: seconds to wait for a batch (0 = for ever)

    def __len__(self):
        return len(self._bs)

    def _epoch_wanted(self, epoch):
        """set_epoch of the caller, before an epoch: fine when it is the epoch that pass was (or will be) sampled for"""
        running = self._it is not None and not self._midway
        if running and self._epoch0 is not None and epoch == self._epoch0 + self._epochs_out:
            return
        self._sampler.set_epoch(epoch)
        self._it = None                         # (re)start from that epoch: what the workers have queued belongs to others

    def _batches(self, copies):
        if self._it is None or self._midway:
            # (re)start: nothing may still be on its way out of the ring when the new iterator dispatches its first
            # prefetch_factor * num_workers batches
            while self._uploads:
                self._uploads.popleft().synchronize()
            self._epoch0 = getattr(self._sampler, "epoch", None) if hasattr(self._sampler, "set_epoch") else None
            self._it = iter(self._dl)           # persistent workers: resets them, stale batches are dropped
            self._epochs_out = 0
        self._midway = True                     # an epoch abandoned half-way is not continued by the next one
        for _ in range(len(self._bs)):
            while len(self._uploads) > self.EXTRA_SLOTS - 1:
                self._uploads.popleft().synchronize()
            batch = next(self._it)              # (hands one more batch, i.e. one more slot, to a worker)
            tag = batch.get("img")
            if isinstance(tag, tuple) and tag and tag[0] == "ring":
                view = self.ring[tag[1], :tag[2]]
                batch["img"] = view.clone() if copies else view
            yield batch
        self._midway = False
        self._epochs_out += 1

    def host_batches(self):
        """the batches with ``img`` a VIEW of the ring slot: the consumer must report the event of each upload (note_upload)
        before it asks for the next batch (DevicePrefetcher does)"""
        return self._batches(copies=False)

    def note_upload(self, event):
        self._uploads.append(event)

    def __iter__(self):
        return self._batches(copies=True)

    def close(self):
        if self.pinned:
            torch.cuda.cudart().cudaHostUnregister(self.ring.data_ptr())
            self.pinned = False

    def __del__(self):
        try:
            self.close()
        except Exception:       # noqa: BLE001  (interpreter shutdown)
            pass
