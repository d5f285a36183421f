"""Input side of the training loop: image batching with padding masks and a host->device prefetcher.

`nested_tensor_from_tensor_list` is util/misc.py:326-343 (zero-pad every image to the batch maximum, mask = True on the
padding); `DataPrefetcher` plays the role of data_utils/data_prefetcher.py:22-78 (the next batch travels to the GPU on a
second stream while the current step runs).  Plumbing only: no arithmetic of the hot path lives here."""
from __future__ import annotations

from typing import Iterable, List, Optional, Sequence

import torch

from .modules import NestedTensor


def nested_tensor_from_tensor_list(images: Sequence[torch.Tensor]) -> NestedTensor:
    """images: list of (C, h_i, w_i) -> NestedTensor((N, C, H, W) zero padded at the right / bottom, (N, H, W) bool mask)."""
    if not len(images):
        raise ValueError("empty image list")
    if any(im.dim() != 3 for im in images):
        raise ValueError("not supported")                     # (the reference only handles CHW images, util/misc.py:342)
    c = int(images[0].shape[0])
    H = max(int(im.shape[1]) for im in images)
    W = max(int(im.shape[2]) for im in images)
    batch = torch.zeros((len(images), c, H, W), dtype=images[0].dtype, device=images[0].device)
    mask = torch.ones((len(images), H, W), dtype=torch.bool, device=images[0].device)
    for i, im in enumerate(images):
        batch[i, :, : im.shape[1], : im.shape[2]].copy_(im)
        mask[i, : im.shape[1], : im.shape[2]] = False
    return NestedTensor(batch, mask)


def _to_device(obj, device):
    if torch.is_tensor(obj):
        if device.type != "cuda":
            return obj.to(device)
        src = obj if (obj.is_cuda or obj.is_pinned()) else obj.pin_memory()      # pinned source: the copy is truly asynchronous
        return src.to(device, non_blocking=True)
    if isinstance(obj, NestedTensor):
        return NestedTensor(None if obj.tensors is None else _to_device(obj.tensors, device), _to_device(obj.mask, device))
    if isinstance(obj, dict):
        return {k: _to_device(v, device) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_to_device(v, device) for v in obj)
    return obj


def _record(obj, stream):
    if torch.is_tensor(obj):
        if obj.is_cuda:
            obj.record_stream(stream)
    elif isinstance(obj, NestedTensor):
        _record(obj.tensors, stream)
        _record(obj.mask, stream)
    elif isinstance(obj, dict):
        for v in obj.values():
            _record(v, stream)
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            _record(v, stream)


class DataPrefetcher:
    """Iterates `(samples, targets)` batches of a loader with the NEXT batch's host->device copies already enqueued on a
    side stream.  `next()` returns `(None, None)` when the loader is exhausted (the reference's protocol, engine.py:46-52);
    the object is also a normal iterator.  `keep_on_host` names target fields that stay where they are (the query assembly
    and the matcher read `boxes` / `labels` on the host: leaving them there saves a device->host round trip per step)."""

    def __init__(self, loader: Iterable, device, prefetch: bool = True, keep_on_host: Sequence[str] = ()):
        self.it, self.device, self.prefetch = iter(loader), torch.device(device), prefetch
        self.keep = set(keep_on_host)
        self.stream = torch.cuda.Stream(device=self.device) if (prefetch and self.device.type == "cuda") else None
        self._next = None
        if prefetch:
            self._preload()

    def _move(self, samples, targets):
        samples = _to_device(samples, self.device)
        if targets is not None:
            targets = [{k: (v if k in self.keep else _to_device(v, self.device)) for k, v in t.items()} for t in targets]
        return samples, targets

    def _preload(self):
        try:
            samples, targets = next(self.it)
        except StopIteration:
            self._next = None
            return
        if self.stream is None:
            self._next = self._move(samples, targets)
            return
        with torch.cuda.stream(self.stream):
            self._next = self._move(samples, targets)

    def next(self):
        if not self.prefetch:
            try:
                return self._move(*next(self.it))
            except StopIteration:
                return None, None
        if self._next is None:
            return None, None
        cur = torch.cuda.current_stream(self.device) if self.stream is not None else None
        if cur is not None:
            cur.wait_stream(self.stream)
            _record(self._next, cur)                      # the caching allocator must not recycle these while `cur` uses them
        out = self._next
        self._preload()
        return out

    def __iter__(self):
        return self

    def __next__(self):
        s, t = self.next()
        if s is None:
            raise StopIteration
        return s, t
