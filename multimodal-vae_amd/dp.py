"""Data parallelism for the fused ELBO step: one process per GPU, replicated parameters / Adam state, per-rank
BatchNorm BATCH statistics (each rank normalises with its own shard, i.e. is exactly the reference at batch B on its
shard), and a sum all-reduce of the flat fp32 gradient buffer per step over RCCL/xGMI.  The 1/world_size factor is folded
into the Adam kernel (``grad_scale``), so no extra elementwise pass touches the gradients.

BatchNorm RUNNING statistics (eval / checkpoints) are rank-local between synchronisation points: unlike
torch DDP's ``broadcast_buffers=True`` nothing re-broadcasts them every forward.  ``sync_bn_buffers`` averages
``running_mean`` / ``running_var`` over the ranks (and takes the max of ``num_batches_tracked``); the training drivers call
it before every eval pass and checkpoint, and once at start-up after the parameter broadcast.

``HSA_ENABLE_IPC_MODE_LEGACY=0`` (dmabuf IPC, the only mode this host driver supports) is read by the ROCr runtime at
``hsa_init``: it must be in the environment BEFORE the first GPU call of the process.  ``ensure_ipc_env()`` sets it and
must therefore run at the very top of a launcher script (bench.py, train.py do); ``init_distributed`` refuses to create
an RCCL group when the variable is missing or wrong instead of failing later at the first all-reduce.

The path has exactly one exchange step; everything else is rank-local.  Backend "nccl" is RCCL on ROCm;
the CPU tests drive the same code with "gloo".
"""
from __future__ import annotations

import os
from typing import Optional

import torch
import torch.distributed as dist


def ensure_ipc_env() -> None:
    """Call before anything touches the GPU (``torch.cuda.set_device`` included).

    ``GPU_MAX_HW_QUEUES``: the fused step runs on up to 4 streams and RCCL adds its own; beyond the HIP runtime's default
    of 4 hardware queues the streams share queues and every kernel of the process slows down (measured with a world-1
    RCCL group on MI355X: 0.90 -> 1.20 ms per MultiMNIST step; with 8 queues 0.91)."""
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


def init_distributed(backend: Optional[str] = None, device: Optional[torch.device] = None) -> tuple:
    """Reads RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torch.distributed.run) and creates the default group."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl" and os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY") != "0":
            # setting it here would be a no-op: the HSA runtime read it when the process first touched the GPU
            raise RuntimeError("HSA_ENABLE_IPC_MODE_LEGACY=0 must be exported before the first GPU call "
                               "(dp.ensure_ipc_env() at the top of the launcher, or the torchrun environment)")
        kw = {}
        if backend == "nccl" and device is not None:
            kw["device_id"] = device
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world, local


class GradAllReduce:
    """Callable handed to ``FusedELBOStep(all_reduce=...)``: in-place SUM over ranks of the flat gradient buffer,
    enqueued on the current stream (RCCL) -- the engine scales by 1/world inside Adam."""

    def __init__(self, group=None, bucket_bytes: int = 0, force: bool = False, overlap: bool = False):
        self.group = group
        self.bucket_elems = bucket_bytes // 4
        self.force = force          # issue the collective even in a group of one rank (single-GPU RCCL test)
        self.overlap = overlap      # engines that support it (MultiMNIST) exchange the decoders' gradient ranges on a
                                    # communication stream while the encoders' backward runs (core.FusedELBOStep._call_dp_overlap);
                                    # measured slower than the plain exchange on this runtime: opt-in only

    def __call__(self, flat: torch.Tensor, async_op: bool = False):
        if not dist.is_initialized() or (dist.get_world_size(self.group) == 1 and not self.force):
            return None
        if async_op:        # one message, the caller waits on the returned Work (overlapped exchange)
            return dist.all_reduce(flat, group=self.group, async_op=True)
        if self.bucket_elems and flat.numel() > self.bucket_elems:
            for i in range(0, flat.numel(), self.bucket_elems):            # xGMI ring is per-link bound: a few large
                dist.all_reduce(flat[i:i + self.bucket_elems], group=self.group)   # messages, never many small ones
        else:
            dist.all_reduce(flat, group=self.group)


def broadcast_flat(flat: torch.Tensor, src: int = 0, group=None) -> None:
    """Make every replica start from rank ``src``'s parameters / buffers."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(flat, src=src, group=group)


def sync_bn_buffers(bn_stats: torch.Tensor, bn_nbt: torch.Tensor, group=None) -> None:
    """Average the BatchNorm running statistics over the ranks (num_batches_tracked: max), in place."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(bn_stats, group=group)
        bn_stats.div_(dist.get_world_size(group))
        dist.all_reduce(bn_nbt, op=dist.ReduceOp.MAX, group=group)


def rank_seed(base: int, rank: int) -> int:
    """Per-rank RNG / data-shard seed (shard r uses seed base + r)."""
    return int(base) + int(rank)


def barrier(device: Optional[torch.device] = None) -> None:
    if dist.is_initialized() and dist.get_world_size() > 1:
        if device is not None and device.type == "cuda":
            dist.barrier(device_ids=[device.index])
        else:
            dist.barrier()
