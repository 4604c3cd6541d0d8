"""Packed host->device transfer of one training batch.

The reference's collate returns six ragged pieces (`data/SROIE_dataset.py:165-208`: a tuple of images, tuples of
per-document `seg_indices` / `token_classes` / `ocr_coors`, the padded `ocr_corpus` and its `mask`) and the training loop
moves every tensor with its own blocking `.to(device)` (`pipeline/train_val_utils.py:257-262`: 4*B + 2 pageable copies per
step).  `PackedBatch` lays the same tensors out in ONE pinned host buffer (16-byte aligned slots) so the step needs one
asynchronous H2D copy; the device-side arguments of `ViBERTgridNet.forward` are views into the single device buffer, with the
same dtypes and shapes as the reference's.  The integer pieces the model needs on the host to build its index tables (token ids,
mask, segment indices) stay reachable through a `_vbg_host` attribute on the device tensors, which saves the model's two
device->host copies per step.
"""
from typing import Sequence, Tuple

import numpy as np
import torch

_ALIGN = 16


def host_mirror(t: torch.Tensor):
    """numpy view of the host copy a tensor was uploaded from, or None"""
    return getattr(t, "_vbg_host", None)


class PackedBatch:
    def __init__(self, buf: torch.Tensor, table, extras=()):
        self.buf = buf                  # uint8, pinned when possible
        self.table = table              # [(group, offset, dtype, shape)]
        self.extras = tuple(extras)     # whatever followed the six model arguments in the collate output (eval mode)

    @staticmethod
    def pack(image_list: Sequence[torch.Tensor], seg_indices: Sequence[torch.Tensor], token_classes: Sequence[torch.Tensor],
             ocr_coors: Sequence[torch.Tensor], ocr_corpus: torch.Tensor, mask: torch.Tensor, *extras, pin: bool = True) -> "PackedBatch":
        groups = [tuple(image_list), tuple(seg_indices), tuple(token_classes), tuple(ocr_coors), (ocr_corpus,), (mask,)]
        table, off = [], 0
        for gi, g in enumerate(groups):
            for t in g:
                t = t.detach()
                table.append((gi, off, t.dtype, tuple(t.shape), t))
                off += (t.numel() * t.element_size() + _ALIGN - 1) // _ALIGN * _ALIGN
        pin = pin and torch.cuda.is_available()
        buf = torch.empty((max(off, _ALIGN),), dtype=torch.uint8, pin_memory=pin)
        out = []
        for gi, o, dt, shape, t in table:
            n = t.numel() * t.element_size()
            if n:
                buf[o:o + n].view(dt).view(shape).copy_(t)
            out.append((gi, o, dt, shape))
        return PackedBatch(buf, out, extras)

    def to(self, device, non_blocking: bool = True) -> Tuple:
        """one H2D copy -> (image_list, seg_indices, token_classes, ocr_coors, ocr_corpus, mask) on `device`"""
        dbuf = self.buf.to(device, non_blocking=non_blocking)
        groups = [[] for _ in range(6)]
        for gi, o, dt, shape in self.table:
            n = int(np.prod(shape)) * torch.empty((), dtype=dt).element_size()
            v = dbuf[o:o + n].view(dt).view(shape) if n else torch.empty(shape, dtype=dt, device=dbuf.device)
            if gi in (1, 4, 5):          # seg_indices, corpus, mask: the model derives its index tables from these on the host
                v._vbg_host = self.buf[o:o + n].view(dt).view(shape).numpy() if n else np.zeros(shape, dtype=np.int64)
            groups[gi].append(v)
        return (tuple(groups[0]), tuple(groups[1]), tuple(groups[2]), tuple(groups[3]), groups[4][0], groups[5][0])

    def nbytes(self) -> int:
        return int(self.buf.numel())


def packed_collate(collate_fn):
    """wrap the reference's collate function: same samples in, a PackedBatch out (DataLoader(collate_fn=packed_collate(ds._ViBERTgrid_coll_func)))"""
    def fn(samples):
        return PackedBatch.pack(*collate_fn(samples))
    return fn


class AsyncScalar:
    """`loss.item()` without draining the stream (pipeline/train_val_utils.py:270 reads the loss value between forward and
    backward, which parks the GPU until the host has started to enqueue the backward pass).  The value is copied to pinned host
    memory on a side stream that waits only for the kernels enqueued so far; `get()` blocks on that copy alone, so the caller can
    enqueue the backward pass first and read the value afterwards -- same number, no bubble."""
    _side = {}

    def __init__(self, t: torch.Tensor):
        dev = t.device
        side = AsyncScalar._side.get(dev.index)
        if side is None:
            side = AsyncScalar._side[dev.index] = torch.cuda.Stream(device=dev)
        self.host = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(dev))
        src = t.detach()
        with torch.cuda.stream(side):
            side.wait_event(ready)
            self.host.copy_(src, non_blocking=True)
            src.record_stream(side)
            self.done = torch.cuda.Event()
            self.done.record(side)

    def get(self) -> float:
        self.done.synchronize()
        return float(self.host.reshape(-1)[0])
