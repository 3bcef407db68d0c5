"""Multi-GPU layer of the hot path: one process per GPU, clips sharded with no
data-path collective, ONE gather of the decoded clips to rank 0 at the end.

The reference shards prompts by rank in exactly this way and never communicates
(scripts/evaluation/ddp_wrapper.py:29-47, inference.py:314-320: contiguous slices, the
remainder silently dropped).  `torch.distributed` backend "nccl" is RCCL on ROCm; the
seven peers each own an xGMI link to rank 0, so the 31.5 MB fp32 clip per rank is a
sub-millisecond point-to-point transfer, not a ring.
"""
from __future__ import annotations

import os
from typing import List, Optional

import torch
import torch.distributed as dist


def shard_indices(n_items: int, world: int, rank: int, drop_remainder: bool = False) -> List[int]:
    """Clip indices of this rank: contiguous slices like the reference; the remainder
    (n_items % world), which the reference drops, is spread over the first ranks unless
    `drop_remainder`."""
    per = n_items // world
    if drop_remainder:
        return list(range(per * rank, per * (rank + 1)))
    rem = n_items % world
    start = per * rank + min(rank, rem)
    return list(range(start, start + per + (1 if rank < rem else 0)))


def init(backend: Optional[str] = None, device: Optional[torch.device] = None) -> tuple:
    """Initialise from the torchrun environment.  Returns (rank, world)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC only on this driver
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
        dist.init_process_group(backend, **kw)
    return rank, world


def gather_clips(video: torch.Tensor, dst: int = 0, out: Optional[List[torch.Tensor]] = None,
                 ragged: bool = False):
    """Gather every rank's decoded clip(s) `(b, 3, T, H, W)` to `dst`.  Returns the list of
    per-rank tensors on `dst` (rank order), None elsewhere.  World size 1: [video].

    `dist.gather` needs the SAME shape -- and the same number of collective calls -- on every rank.  With
    `shard_indices` spreading a remainder, ranks own different clip counts `b`: pass `ragged=True` and every rank
    is padded to the largest count (one extra all_gather of the counts), the padding cut off again on `dst`.
    Every rank must call this exactly once per step, also a rank that owns no clip (b = 0)."""
    if ragged and out is not None:
        raise ValueError("gather_clips(ragged=True) allocates its own padded buffers; `out=` is only for equal shards")
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [video]
    rank, world = dist.get_rank(), dist.get_world_size()
    video = video.contiguous()
    if ragged:
        cnt = torch.tensor([video.shape[0]], dtype=torch.int64, device=video.device)
        cnts = [torch.empty_like(cnt) for _ in range(world)]
        dist.all_gather(cnts, cnt)
        counts = [int(c.item()) for c in cnts]
        mx = max(counts)
        if mx == 0:
            return [video[:0] for _ in range(world)] if rank == dst else None
        if video.shape[0] < mx:
            pad = torch.zeros((mx - video.shape[0], *video.shape[1:]), dtype=video.dtype, device=video.device)
            video = torch.cat([video, pad], 0)
        if rank == dst:
            bufs = [torch.empty_like(video) for _ in range(world)]
            dist.gather(video, bufs, dst=dst)
            return [b[:c] for b, c in zip(bufs, counts)]
        dist.gather(video, None, dst=dst)
        return None
    if rank == dst:
        if out is None:
            out = [torch.empty_like(video) for _ in range(world)]
        dist.gather(video, out, dst=dst)
        return out
    dist.gather(video, None, dst=dst)
    return None
