"""Batches for Trainer.train / Trainer.validate, assembled ON THE DEVICE.

Worker processes decode (JPEG + annotation masks) and draw the augmentation plan; the collate function packs the uint8
photos and labels of a batch back to back; the main process uploads the two packed buffers (~0.9 MB per VOC sample instead of
the 23 MB of float crops and target planes the reference's loader ships) and runs cl4_input_batch
(+ cl4_target_generation for the fully supervised step) on a side stream so the next batch is prepared while the current
training step runs. Yields the reference's batch tuples (dataset/dataset.py:199,221):
  train: (images f32 [b,3,S,S], seg_map int64 [b,S,S], center [b,C,S,S], offset [b,2,S,S], weight [b,1,S,S], l1h f64 [b,C])
  val:   (images f32 [1,3,h,w], seg int64 [1,h,w], masks [1,R,H,W], mask_label [1,R], [path])
In weakly supervised phases the dense targets are not read by the step (train.py:243-248) and are skipped (zero placeholders)."""
import os

import numpy as np
import torch
import torch.utils.data as data

from .. import ops


class PackedBatch:
    __slots__ = ("images", "labels", "descs", "out_hw", "l1h", "mean", "std", "masks", "mask_label", "paths")


def collate_raw(samples):
    pb = PackedBatch()
    pb.out_hw = samples[0].plan.out_hw
    if any(s.plan.out_hw != pb.out_hw for s in samples):
        raise RuntimeError("all samples of a batch must share the output size (use batch_size 1 for Resize(int) stacks)")
    io = lo = 0
    descs = []
    for s in samples:
        descs.append(s.plan.desc(io, lo))
        io += s.image.size
        lo += s.label.size
    pb.descs = descs
    pb.images = torch.from_numpy(np.concatenate([s.image.reshape(-1) for s in samples]))
    pb.labels = torch.from_numpy(np.concatenate([s.label.reshape(-1) for s in samples]))
    pb.mean, pb.std = samples[0].plan.mean, samples[0].plan.std
    pb.l1h = torch.from_numpy(np.stack([s.l1h for s in samples])) if samples[0].l1h is not None else None
    pb.masks = [s.masks for s in samples]
    pb.mask_label = [s.mask_label for s in samples]
    pb.paths = [s.path for s in samples]
    return pb


def worker_options(num_workers):
    """DataLoader keywords for `num_workers` host-side worker processes next to a GPU-initialised parent: spawn, persistent."""
    if not num_workers:
        return {}
    return {"multiprocessing_context": os.environ.get("CL4_WORKER_START", "spawn"), "persistent_workers": True}


class DeviceLoader:
    """DataLoader whose batches are finished on the GPU. `supervised` = build centre / offset / weight targets (step 0)."""

    def __init__(self, dataset, batch_size, device, sampler=None, shuffle=False, num_workers=0, drop_last=False,
                 supervised=False):
        self.dataset, self.device, self.supervised = dataset, device, supervised
        # Worker processes are SPAWNED and kept for the life of the loader. The default (fork, again at every epoch) copies the page
        # tables of a process that has the HIP runtime's device mappings in its address space: measured ~190 s per fork on an MI355X box
        # once a few models had lived in the parent (FINDINGS.md finding 42), against ~3 s for a fresh interpreter.
        self.loader = data.DataLoader(dataset, batch_size=batch_size, sampler=sampler, shuffle=shuffle and sampler is None,
                                      num_workers=num_workers, drop_last=drop_last, collate_fn=collate_raw,
                                      pin_memory=False, **worker_options(num_workers))
        self.sampler = sampler if sampler is not None else self.loader.sampler
        self.side = torch.cuda.Stream(device=device)
        self.lut = torch.from_numpy(dataset.lut).to(device)

    def __len__(self):
        return len(self.loader)

    def _finish(self, pb):
        """Upload + device pipeline on the side stream; returns (batch tuple, ready event)."""
        with torch.cuda.stream(self.side):
            images = pb.images.pin_memory().to(self.device, non_blocking=True)
            labels = pb.labels.pin_memory().to(self.device, non_blocking=True)
            train = pb.l1h is not None
            img, seg, mask = ops.input_batch(images, pb.descs, pb.out_hw, labels=labels, lut=self.lut,
                                             want_mask=train and self.supervised, mean=pb.mean, std=pb.std)
            if train:
                l1h = pb.l1h.to(self.device, non_blocking=True)
                if self.supervised:
                    center, offset, weight = ops.target_generation(seg, mask, self.dataset.total_classes, self.dataset.sigma)
                else:
                    center = offset = weight = torch.zeros(1, device=self.device)
                batch = (img, seg, center, offset, weight, l1h)
            else:
                batch = (img, seg, torch.from_numpy(np.stack(pb.masks)), torch.from_numpy(np.stack(pb.mask_label)), pb.paths)
            for t in (images, labels):
                t.record_stream(self.side)
            ev = torch.cuda.Event()
            ev.record(self.side)
        return batch, ev

    def __iter__(self):
        pending = None
        for pb in self.loader:
            nxt = self._finish(pb)                       # queued on the side stream before the current batch is consumed
            if pending is not None:
                yield self._hand_over(pending)
            pending = nxt
        if pending is not None:
            yield self._hand_over(pending)

    def _hand_over(self, pending):
        batch, ev = pending
        torch.cuda.current_stream(self.device).wait_event(ev)
        for t in batch:
            if torch.is_tensor(t) and t.is_cuda:
                t.record_stream(torch.cuda.current_stream(self.device))
        return batch
