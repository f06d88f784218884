"""Input side of the hot path: the reference's on-disk MultiMNIST format and a device-resident batcher.

The reference stores ``(uint8 tensor (N,50,50), list of digit lists)`` in ``<root>/processed/{training,test}.pt``
(multimnist/datasets.py:180-190) and turns one sample at a time into fp32 with PIL + ``ToTensor`` on the host
(``:57-71``), the labels with ``charlist_tensor`` (multimnist/utils.py:34-37).  Here the uint8 pixels go to the GPU as
they are (4x fewer H2D bytes, from pinned memory, on a copy stream, double-buffered) and ``ToTensor`` is the device
kernel ``mmvae_u8_to_f32``; labels are padded once up front.
"""
from __future__ import annotations

import os
from typing import Iterator, List, Sequence, Tuple

import numpy as np
import torch

from ._lib import MMVAEError, call, ptr
from .utils import FILL, charlist_tensor, max_length

PROCESSED = "processed"
TRAIN_FILE, TEST_FILE = "training.pt", "test.pt"      # multimnist/datasets.py:28-30


def save_multimnist(root: str, train: bool, images_u8: torch.Tensor, labels: Sequence[Sequence[int]]) -> str:
    """Write a split in the reference's format (used by tests and by the synthetic generator)."""
    assert images_u8.dtype == torch.uint8 and images_u8.dim() == 3
    d = os.path.join(os.path.expanduser(root), PROCESSED)
    os.makedirs(d, exist_ok=True)
    path = os.path.join(d, TRAIN_FILE if train else TEST_FILE)
    torch.save((images_u8, [list(map(int, l)) for l in labels]), path)
    return path


def load_multimnist(root: str, train: bool) -> Tuple[torch.Tensor, torch.Tensor]:
    """-> (uint8 (N,H,W), int64 (N,4) FILL-padded) from the reference's file."""
    path = os.path.join(os.path.expanduser(root), PROCESSED, TRAIN_FILE if train else TEST_FILE)
    if not os.path.exists(path):
        raise RuntimeError("Dataset not found at %s (the reference builds it from MNIST: multimnist/datasets.py:163-190)" % path)
    images, labels = torch.load(path, weights_only=False)
    text = torch.stack([charlist_tensor(l) for l in labels]) if len(labels) else torch.zeros(0, max_length, dtype=torch.long)
    return images.contiguous(), text


def synthetic_multimnist(n: int, seed: int = 0, size: int = 50) -> Tuple[torch.Tensor, List[List[int]]]:
    """MultiMNIST-shaped stand-in (no MNIST files in this environment): 0..4 blobs per canvas, one label digit each."""
    g = torch.Generator().manual_seed(seed)
    images = torch.zeros(n, size, size, dtype=torch.uint8)
    labels: List[List[int]] = []
    for i in range(n):
        k = int(torch.randint(0, max_length + 1, (1,), generator=g))
        digs = []
        for _ in range(k):
            d = int(torch.randint(0, 10, (1,), generator=g))
            y, x = (int(v) for v in torch.randint(0, size - 14, (2,), generator=g))
            w = 6 + d                                   # blob extent encodes the digit
            patch = (torch.rand(w, w, generator=g) * 255).to(torch.uint8)
            images[i, y:y + w, x:x + w] = torch.maximum(images[i, y:y + w, x:x + w], patch)
            digs.append(d)
        labels.append(digs)
    return images, labels


class DeviceBatcher:
    """Iterates ``(image fp32, second modality)`` device batches over a uint8 image dataset.

    ``images_u8``: (N,H,W) -> batches (B,1,H,W) (MultiMNIST), or (N,C,H,W) -> (B,C,H,W) (CelebA / COCO pixels).
    ``text``: (N, ...) of any dtype, copied as is: int64 tokens (B,4), fp32 attributes (B,18), fp32 caption vectors
    (B,102,300) -- the asynchronous H2D image + caption pipeline of the COCO configuration.

    Host: one index gather per batch into a pinned staging buffer (two of them, alternating).  Copy stream: async H2D of
    the uint8 pixels + the second modality.  Compute stream: waits for the copy event, then the u8->f32 kernel (ToTensor
    on the device).  Two hazards, kept apart so the host never waits for the GPU's compute stream: (1) the pinned staging
    buffer of a slot is free once that slot's previous H2D copy has finished (host waits on ``ready[slot]``, a copy-stream
    event two batches old); (2) the device buffers of a slot are free once the step that read them is done -- a
    device-side edge (``copy_stream.wait_event(consumed[slot])``), no host synchronisation.  So the host runs a full
    step ahead: batch i+1 is gathered and copied while batch i trains.  ``drop_last`` because the fused plans are built
    for a fixed batch size."""

    def __init__(self, images_u8: torch.Tensor, text: torch.Tensor, batch_size: int, device: torch.device, shuffle: bool = True,
                 seed: int = 0):
        assert images_u8.dtype == torch.uint8 and images_u8.dim() in (3, 4) and len(images_u8) == len(text)
        self.images, self.text, self.B, self.device = images_u8, text, int(batch_size), device
        self.shuffle, self.seed, self.epoch = shuffle, seed, 0
        ishape = tuple(images_u8.shape[1:])
        oshape = (1,) + ishape if images_u8.dim() == 3 else ishape
        tshape = tuple(text.shape[1:])
        self.hw = ishape[-2:]
        self.stage_u8 = [torch.empty((self.B,) + ishape, dtype=torch.uint8).pin_memory() for _ in range(2)]
        self.stage_tx = [torch.empty((self.B,) + tshape, dtype=text.dtype).pin_memory() for _ in range(2)]
        self.dev_u8 = [torch.empty((self.B,) + ishape, dtype=torch.uint8, device=device) for _ in range(2)]
        self.dev_tx = [torch.empty((self.B,) + tshape, dtype=text.dtype, device=device) for _ in range(2)]
        self.dev_f32 = [torch.empty((self.B,) + oshape, dtype=torch.float32, device=device) for _ in range(2)]
        # the copy stream is one more default-priority stream next to the step's: the engine's side streams must then run at
        # default priority too (1.93 vs 1.10 ms per step measured with this loader)
        try:
            call("mmvae_set_stream_policy", 1)
        except MMVAEError:
            import warnings
            warnings.warn("DeviceBatcher created after the engine's side streams: if they were created with the opt-in "
                          "lowest priority (mmvae_set_stream_policy(0)) every kernel runs slower while the copy stream "
                          "is active; create the loader first or keep the default (flat) priorities")
        self._images_np, self._text_np = images_u8.numpy(), text.numpy()
        self._stage_u8_np = [t.numpy() for t in self.stage_u8]
        self._stage_tx_np = [t.numpy() for t in self.stage_tx]
        self.copy_stream = torch.cuda.Stream(device=device)
        self.ready = [torch.cuda.Event() for _ in range(2)]
        self.consumed = [torch.cuda.Event() for _ in range(2)]
        self._used = [False, False]

    def __len__(self) -> int:
        return len(self.images) // self.B

    def _stage(self, slot: int, idx: torch.Tensor) -> None:
        if self._used[slot]:
            self.ready[slot].synchronize()                  # hazard 1: this slot's previous H2D copy left the pinned buffer
        # numpy's single-threaded take, NOT torch.index_select: next to a running HIP process the OpenMP team of a torch
        # CPU op took 8 ms per 640 kB gather on the GPU box (0.03 ms when nothing else runs) -- 8x the training step
        ix = idx.numpy()
        np.take(self._images_np, ix, axis=0, out=self._stage_u8_np[slot])
        np.take(self._text_np, ix, axis=0, out=self._stage_tx_np[slot])
        with torch.cuda.stream(self.copy_stream):
            if self._used[slot]:
                self.copy_stream.wait_event(self.consumed[slot])   # hazard 2: device-side edge, the host does not wait
            self._used[slot] = True
            self.dev_u8[slot].copy_(self.stage_u8[slot], non_blocking=True)
            self.dev_tx[slot].copy_(self.stage_tx[slot], non_blocking=True)
            self.ready[slot].record(self.copy_stream)

    def __iter__(self) -> Iterator[Tuple[torch.Tensor, torch.Tensor]]:
        n = len(self.images)
        if self.shuffle:
            g = torch.Generator().manual_seed(self.seed + self.epoch)
            order = torch.randperm(n, generator=g)
        else:
            order = torch.arange(n)
        self.epoch += 1
        nb = len(self)
        if nb == 0:
            return
        self._stage(0, order[0:self.B])
        for b in range(nb):
            slot = b & 1
            if b + 1 < nb:
                self._stage(slot ^ 1, order[(b + 1) * self.B:(b + 2) * self.B])
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(self.ready[slot])
            import ctypes as C
            call("mmvae_u8_to_f32", ptr(self.dev_u8[slot]), self.dev_u8[slot].numel(), 255.0, ptr(self.dev_f32[slot]),
                 C.c_void_p(cur.cuda_stream))
            try:
                yield self.dev_f32[slot], self.dev_tx[slot]
            finally:                                        # also when the consumer abandons the iterator at this batch
                self.consumed[slot].record(cur)


__all__ = ["save_multimnist", "load_multimnist", "synthetic_multimnist", "DeviceBatcher", "FILL"]
