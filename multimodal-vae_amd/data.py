"""Input side of the hot path: the reference's on-disk MultiMNIST format and a device-resident batcher.

The reference stores ``(uint8 tensor (N,50,50), list of digit lists)`` in ``<root>/processed/{training,test}.pt``
(multimnist/datasets.py:180-190) and turns one sample at a time into fp32 with PIL + ``ToTensor`` on the host
(``:57-71``), the labels with ``charlist_tensor`` (multimnist/utils.py:34-37).  Here the uint8 pixels go to the GPU as
they are (4x fewer H2D bytes, from pinned memory, on a copy stream, double-buffered) and ``ToTensor`` is the device
kernel ``mmvae_u8_to_f32``; labels are padded once up front.
"""
from __future__ import annotations

import os
from typing import Iterator, List, Optional, Sequence, Tuple

import numpy as np
import torch

from ._lib import MMVAEError, OwnedStream, call, ptr
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
    (B,102,300) -- the asynchronous H2D image + caption pipeline of the COCO configuration (coco/train.py:117-128).

    Four slots (pinned staging + device buffers each; three measured 7 % slower: the worker may run only as far ahead of the GPU as
    there are free slots), three actors:
      * a worker thread stages batch b+3: it waits ON THE HOST for the step that last read the slot's device buffers
        (``consumed[slot].synchronize()``), gathers the rows into the slot's pinned staging buffers (``np.take`` releases the GIL;
        the COCO batch is 15.7 MB of caption vectors, 1.5-1.9 ms of a single core) and enqueues the slot's H2D copy on the copy
        stream (the library's own, ``_lib.OwnedStream``): uint8 pixels + the second modality;
      * the copy stream moves batches b+1 .. b+3 to the device while step b runs;
      * the enqueue thread only waits (device-side) for the copy event of batch b, converts u8 -> f32 (ToTensor on the device) and
        trains.
    Round 4: the slot reuse used to be a DEVICE-side edge -- ``copy_stream.wait_event(consumed[slot])`` -- and that one call made
    the loader-fed MultiMNIST step 1.5 ms against 0.65 ms with resident inputs (tools/loader_knock.py: without it 0.66): a copy
    stream parked on an event of the compute stream stalls this runtime's queues (DESIGN.md section 5 met the same with the
    overlapped gradient exchange).  Pacing the reuse on the worker's host thread costs the enqueue thread nothing and the worker is
    two batches ahead anyway.  ``MMVAE_LOADER_HOST_PACED=0`` brings the device-side edge back (measurement aid).
    ``drop_last`` because the fused plans are built for a fixed batch size.

    ``pin_dataset=True`` (default for batches up to 2 MB; rows must be multiples of 4 bytes): the whole dataset is pinned once and the
    GPU gathers the B rows over the host link itself on the copy stream (``mmvae_gather_rows_u8_f32``: gather + ToTensor in one kernel,
    ``mmvae_gather_rows`` for the second modality), reading the B indices from a pinned buffer too -- no runtime copy call at all.  Measured on COCO it
    frees the host (0.31 ms per batch) but the gather workgroups sit on CUs for the length of the transfer and the step next to
    them slows down more than that saves (3.8 ms per step): not the default."""

    SLOTS = 4

    def __init__(self, images_u8: torch.Tensor, text: torch.Tensor, batch_size: int, device: torch.device, shuffle: bool = True,
                 seed: int = 0, pin_dataset: Optional[bool] = None, copy_on_worker: bool = True, slots: int = 0):
        assert images_u8.dtype == torch.uint8 and images_u8.dim() in (3, 4) and len(images_u8) == len(text)
        self.images, self.text, self.B, self.device = images_u8, text, int(batch_size), device
        self.shuffle, self.seed, self.epoch = shuffle, seed, 0
        self.copy_on_worker = True          # (round 3's enqueue-thread copy path is gone; the argument is kept for callers)
        self.host_paced = bool(int(os.environ.get("MMVAE_LOADER_HOST_PACED", "1")))
        self._carry, self._next_order, self._gbase = {}, None, 0
        ishape = tuple(images_u8.shape[1:])
        oshape = (1,) + ishape if images_u8.dim() == 3 else ishape
        tshape = tuple(text.shape[1:])
        self.hw = ishape[-2:]
        if slots:
            self.SLOTS = int(slots)
        assert self.SLOTS >= 3
        S = self.SLOTS
        if pin_dataset is None:
            # auto: small batches (MultiMNIST 0.65 MB, MNIST 0.1 MB) are gathered by the device straight from the pinned dataset -- no
            # runtime copy call, no staging, ToTensor in the gather kernel; large ones (CelebA 6 MB, COCO 16 MB of captions) are staged
            # and copied: their gather workgroups would sit on CUs for the length of the transfer
            row_u8 = int(np.prod(ishape)) if ishape else 1
            row_tx = (int(np.prod(tshape)) if tshape else 1) * text.element_size()
            pin_dataset = row_u8 % 4 == 0 and row_tx % 4 == 0 and self.B * (row_u8 + row_tx) <= (2 << 20)
        self.device_gather = bool(pin_dataset)
        if self.device_gather:
            self.row_u8 = int(np.prod(ishape)) if ishape else 1
            self.row_tx = (int(np.prod(tshape)) if tshape else 1) * text.element_size()
            assert self.row_u8 % 4 == 0 and self.row_tx % 4 == 0, "device gather: rows must be multiples of 4 bytes"
            self.images = images_u8 = images_u8.contiguous().pin_memory()
            self.text = text = text.contiguous().pin_memory()
            self.idx_host = [torch.empty(self.B, dtype=torch.int64).pin_memory() for _ in range(S)]
            self.idx_dev = [torch.empty(self.B, dtype=torch.int64, device=device) for _ in range(S)]
        n_stage = 0 if self.device_gather else S
        # ONE pinned staging buffer and ONE device buffer per slot, [uint8 pixels | pad to 256 B | second modality]: the batch crosses
        # in a single hipMemcpyAsync (each asynchronous copy call costs the process ~50 us of runtime lock next to a running step:
        # two of them per batch made the enqueue thread the bottleneck of the loader-fed MultiMNIST step)
        nb_u8 = self.B * (int(np.prod(ishape)) if ishape else 1)
        nb_tx = self.B * (int(np.prod(tshape)) if tshape else 1) * text.element_size()
        off_tx = (nb_u8 + 255) // 256 * 256
        self._slot_bytes = off_tx + nb_tx

        def views(buf):
            return (buf[:nb_u8].view((self.B,) + ishape), buf[off_tx:off_tx + nb_tx].view(text.dtype).view((self.B,) + tshape))
        self._stage_buf = [torch.empty(self._slot_bytes, dtype=torch.uint8).pin_memory() for _ in range(n_stage)]
        self._dev_buf = [torch.empty(self._slot_bytes, dtype=torch.uint8, device=device) for _ in range(S)]
        self.stage_u8 = [views(b)[0] for b in self._stage_buf]
        self.stage_tx = [views(b)[1] for b in self._stage_buf]
        self.dev_u8 = [views(b)[0] for b in self._dev_buf]
        self.dev_tx = [views(b)[1] for b in self._dev_buf]
        self.dev_f32 = [torch.empty((self.B,) + oshape, dtype=torch.float32, device=device) for _ in range(S)]
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
        self._copy_owner = OwnedStream(device)              # (not torch.cuda.Stream(): see _lib.OwnedStream)
        self.copy_stream = self._copy_owner.stream
        if self.host_paced:
            # the library's events: no system-scope fence when they complete.  A default torch.cuda.Event writes the device caches back
            # for the host at every record -- one per step on the compute stream (`consumed`) and one per batch on the copy stream
            # cost the loader-fed MultiMNIST step tens of microseconds
            from ._lib import OwnedEvent
            self.ready = [OwnedEvent() for _ in range(S)]
            self.consumed = [OwnedEvent() for _ in range(S)]
            self._ready_h = [e.handle for e in self.ready]
        else:
            self.ready = [torch.cuda.Event() for _ in range(S)]
            self.consumed = [torch.cuda.Event() for _ in range(S)]
            for e in self.ready:            # (torch creates the hipEvent_t at the first record)
                e.record(self.copy_stream)
            self._ready_h = [int(e.cuda_event) for e in self.ready]
        self._copy_h = int(self.copy_stream.cuda_stream)
        self._bytes_u8 = self.dev_u8[0].numel()
        self._bytes_tx = self.dev_tx[0].numel() * self.dev_tx[0].element_size()
        self._copied = [False] * S          # the slot's staging buffer has an H2D copy recorded in ready[slot]
        self._read = [False] * S            # the slot's device buffers have been handed to a step (consumed[slot] recorded)
        from concurrent.futures import ThreadPoolExecutor
        self._worker = ThreadPoolExecutor(max_workers=1, thread_name_prefix="mmvae-loader")

    def __len__(self) -> int:
        return len(self.images) // self.B

    def _gather(self, slot: int, ix: np.ndarray) -> None:
        """worker thread: batch rows -> the slot's pinned staging buffers"""
        if self.device_gather:
            if self._copied[slot]:
                self.ready[slot].synchronize()              # the index upload of the slot's previous use has left the pinned buffer
            self.idx_host[slot].numpy()[:] = ix
            return
        if self._copied[slot]:
            self.ready[slot].synchronize()                  # this slot's previous H2D copy has left the pinned buffers
        # numpy's single-threaded take, NOT torch.index_select: next to a running HIP process the OpenMP team of a torch
        # CPU op took 8 ms per 640 kB gather on the GPU box (0.03 ms when nothing else runs) -- 8x the training step
        np.take(self._images_np, ix, axis=0, out=self._stage_u8_np[slot])
        np.take(self._text_np, ix, axis=0, out=self._stage_tx_np[slot])

    def _copy(self, slot: int) -> None:
        """enqueue thread: the slot's staged batch -> its device buffers, on the copy stream"""
        if self._read[slot] and self.host_paced:
            self.consumed[slot].synchronize()                   # host-side: the step that read these device buffers is done
        if self.host_paced and not self.device_gather:
            # ONE foreign call for the two copies and the event (round 4: the torch-level calls -- stream context, two copy_, record --
            # held the interpreter lock of the enqueue thread's 49 launches per step: the loader-fed step was HOST-bound, 0.62 ms
            # of enqueue loop per 0.58 ms step)
            import ctypes as C
            call("mmvae_h2d_stage", ptr(self._dev_buf[slot]), ptr(self._stage_buf[slot]), self._slot_bytes,
                 None, None, 0, C.c_void_p(self._ready_h[slot]), C.c_void_p(self._copy_h))
            self._copied[slot] = True
            return
        with torch.cuda.stream(self.copy_stream):
            if self._read[slot] and not self.host_paced:
                self.copy_stream.wait_event(self.consumed[slot])   # device-side edge: the step that read these buffers is done
            if self.device_gather:
                import ctypes as C
                # (the gather kernels read the B indices straight from the slot's PINNED index buffer: no runtime copy call at all)
                st = C.c_void_p(self.copy_stream.cuda_stream)
                # pixels: gather + ToTensor in one kernel, straight into the fp32 batch (nothing but the wait is left on the compute stream)
                call("mmvae_gather_rows_u8_f32", ptr(self.images), ptr(self.idx_host[slot]), self.B, self.row_u8, 255.0, ptr(self.dev_f32[slot]), st)
                call("mmvae_gather_rows", ptr(self.text), ptr(self.idx_host[slot]), self.B, self.row_tx, ptr(self.dev_tx[slot]), st)
            else:
                self.dev_u8[slot].copy_(self.stage_u8[slot], non_blocking=True)
                self.dev_tx[slot].copy_(self.stage_tx[slot], non_blocking=True)
            self.ready[slot].record(self.copy_stream)
            self._copied[slot] = True

    def _epoch_order(self) -> np.ndarray:
        n = len(self.images)
        if self.shuffle:
            g = torch.Generator().manual_seed(self.seed + self.epoch)
            order = torch.randperm(n, generator=g)
        else:
            order = torch.arange(n)
        self.epoch += 1
        return order.numpy()

    def __iter__(self) -> Iterator[Tuple[torch.Tensor, torch.Tensor]]:
        nb = len(self)
        if nb == 0:
            return
        S, B = self.SLOTS, self.B
        AHEAD = S - 1               # the worker stages this many batches ahead of the step being enqueued: every slot but the one in use
        # The ring runs ACROSS epochs: the last steps of an epoch already stage the first batches of the next one (its order is
        # drawn early), so an epoch boundary does not drain the pipeline (an 8-batch epoch lost 12 % to that).
        order = self._next_order if self._next_order is not None else self._epoch_order()
        fut, self._carry, self._next_order = self._carry, {}, None
        g0 = self._gbase            # global index of this epoch's first batch: slots rotate continuously
        for b in range(min(AHEAD, nb)):
            if b not in fut:
                fut[b] = self._worker.submit(self._stage, (g0 + b) % S, order[b * B:(b + 1) * B])
        carry, next_order, b = {}, None, -1
        try:
            for b in range(nb):
                nxt = b + AHEAD
                if nxt < nb:
                    fut[nxt] = self._worker.submit(self._stage, (g0 + nxt) % S, order[nxt * B:(nxt + 1) * B])
                elif nxt - nb < nb:
                    if next_order is None:
                        next_order = self._epoch_order()
                    k = nxt - nb
                    carry[k] = self._worker.submit(self._stage, (g0 + nxt) % S, next_order[k * B:(k + 1) * B])
                fut.pop(b).result()                                 # gathered AND its H2D copy enqueued by the worker, AHEAD steps ago
                slot = (g0 + b) % S
                cur = torch.cuda.current_stream(self.device)
                import ctypes as C
                if self.device_gather:
                    call("mmvae_stream_wait_event", C.c_void_p(cur.cuda_stream), C.c_void_p(self._ready_h[slot]))
                else:
                    call("mmvae_u8_to_f32_after", ptr(self.dev_u8[slot]), self._bytes_u8, 255.0, ptr(self.dev_f32[slot]),
                     C.c_void_p(self._ready_h[slot]), C.c_void_p(cur.cuda_stream))       # wait for the copy's event, then ToTensor
                try:
                    yield self.dev_f32[slot], self.dev_tx[slot]
                finally:                                            # also when the consumer abandons the iterator at this batch
                    self.consumed[slot].record(cur)
                    self._read[slot] = True
        finally:
            if b == nb - 1:                                         # every batch was handed out: the prefetch goes to the next epoch
                self._carry, self._next_order = carry, next_order
                carry = {}
            elif next_order is not None:                            # abandoned: the early draw of the next order is taken back
                self.epoch -= 1
            self._gbase = g0 + b + 1                                # the next epoch's batch k takes slot (gbase + k) % S: the handed-over ones hold theirs
            for f in list(fut.values()) + list(carry.values()):     # an abandoned epoch: let the worker finish what it holds
                f.result()

    def _stage(self, slot: int, ix: np.ndarray) -> None:
        """worker thread (copy_on_worker): gather the batch AND enqueue its H2D copy -- the enqueue thread only waits for the copy's
        event, converts and trains (the copy calls took that thread ~0.3 ms per batch next to a running step)"""
        self._gather(slot, ix)
        self._copy(slot)


__all__ = ["save_multimnist", "load_multimnist", "synthetic_multimnist", "DeviceBatcher", "FILL"]
