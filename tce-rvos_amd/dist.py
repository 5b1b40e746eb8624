"""Clip sharding over the GPUs of one node.

Clips are independent units (the reference shards videos over processes with no communication,
inference_ytvos.py:96-113); here each rank (one process per GPU) runs B=1 forwards on its contiguous block of
clips and the per-clip mask logits meet through ONE collective per batch: an all-gather over RCCL/xGMI
(`torch.distributed` backend "nccl" on ROCm).  No other data-path collective exists on this path.
The same code runs on the gloo backend (CPU tensors) for the world_size-2 tests.
"""
import os
from typing import List, Tuple

import torch
import torch.distributed as dist

# TCE_DIST_FORCE=1 (or force=True): run the collective even in a group of ONE rank.  A world of one normally short-cuts
# (nothing to gather); forcing it sends the same tensors through the same RCCL entry points a multi-GPU job uses -- the
# only way to execute the RCCL path on a one-GPU box (tests/test_e2e_gpu.py::test_rccl_world1_*).
FORCE_COLLECTIVE = os.environ.get("TCE_DIST_FORCE") == "1"


def _skip_collective(group, force):
    if not (dist.is_available() and dist.is_initialized()):
        return True
    return dist.get_world_size(group) == 1 and not (force or FORCE_COLLECTIVE)


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of rank `rank`; blocks differ by at most one item, earlier ranks get the extras."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_clip_masks(local: torch.Tensor, n_total: int, group=None, force: bool = False) -> torch.Tensor:
    """local [n_local, ...] (this rank's clips, in clip order) -> [n_total, ...] on every rank, in global clip order.
    Ranks may hold different clip counts (n_total % world != 0): shards are padded to the largest count for the
    collective and trimmed after."""
    if _skip_collective(group, force):
        return local
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    counts = [shard_range(n_total, r, world)[1] - shard_range(n_total, r, world)[0] for r in range(world)]
    nmax = max(counts)
    if local.shape[0] != counts[rank]:
        raise ValueError(f"rank {rank} holds {local.shape[0]} clips, expected {counts[rank]}")
    padded = local
    if local.shape[0] < nmax:
        pad = torch.zeros((nmax - local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        padded = torch.cat([local, pad], 0)
    padded = padded.contiguous()
    out = torch.empty((world * nmax,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    if dist.get_backend(group) == "gloo":
        parts = list(out.view((world, nmax) + tuple(local.shape[1:])).unbind(0))
        dist.all_gather(parts, padded, group=group)
        out = torch.stack(parts, 0).view_as(out)
    else:
        dist.all_gather_into_tensor(out, padded, group=group)
    out = out.view((world, nmax) + tuple(local.shape[1:]))
    return torch.cat([out[r, :counts[r]] for r in range(world)], 0)


class PendingGather:
    """An all-gather of one step's masks in flight on the collective stream; `wait()` returns the gathered tensor.
    Lets the caller launch the next clip's forward before the previous step's masks have met (the collective moves
    1.44 MB per clip over xGMI while the matrix cores work on the next clip)."""

    def __init__(self, work, out, counts, keep):
        self._work, self._out, self._counts, self._keep = work, out, counts, keep

    def wait(self) -> torch.Tensor:
        if self._work is not None:
            self._work.wait()  # stream-ordered for RCCL (no host block), blocking for gloo
            self._work = None
        out, counts = self._out, self._counts
        if all(c == out.shape[1] for c in counts):
            return out.flatten(0, 1)
        return torch.cat([out[r, :counts[r]] for r in range(len(counts))], 0)


def gather_clip_masks_async(local: torch.Tensor, n_total: int, group=None, force: bool = False) -> PendingGather:
    """Non-blocking form of gather_clip_masks for equal or ragged shards (RCCL / gloo)."""
    if _skip_collective(group, force):
        return PendingGather(None, local[None], [local.shape[0]], None)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    counts = [shard_range(n_total, r, world)[1] - shard_range(n_total, r, world)[0] for r in range(world)]
    nmax = max(counts)
    if local.shape[0] != counts[rank]:
        raise ValueError(f"rank {rank} holds {local.shape[0]} clips, expected {counts[rank]}")
    padded = local
    if local.shape[0] < nmax:
        pad = torch.zeros((nmax - local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        padded = torch.cat([local, pad], 0)
    padded = padded.contiguous()
    out = torch.empty((world, nmax) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    if dist.get_backend(group) == "gloo":
        parts = list(out.unbind(0))
        work = dist.all_gather(parts, padded, group=group, async_op=True)
    else:
        work = dist.all_gather_into_tensor(out.view((world * nmax,) + tuple(local.shape[1:])), padded, group=group,
                                           async_op=True)
    return PendingGather(work, out, counts, padded)


def run_sharded(forward_clip, clips: List, n_total: int = None, group=None, like: torch.Tensor = None) -> torch.Tensor:
    """forward_clip(clip) -> per-clip result tensor (mask logits [T,Q,h,w], or the harness's uint8 masks [T,H0,W0]:
    4x fewer bytes on the wire); `clips` is the GLOBAL list (every rank indexes its block).  A rank whose block is
    empty (n_total < world) contributes zero clips: it needs the per-clip shape and dtype, taken from `like` (a tensor
    shaped like one result) or, failing that, learnt from the other ranks through one small object all-gather."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    n_total = len(clips) if n_total is None else n_total
    lo, hi = shard_range(n_total, rank, world)
    outs = [forward_clip(clips[i]) for i in range(lo, hi)]
    meta = (tuple(outs[0].shape), outs[0].dtype, str(outs[0].device)) if outs else None
    if like is not None and meta is None:
        meta = (tuple(like.shape), like.dtype, str(like.device))
    if world > 1 and 0 < n_total < world and like is None:  # same condition on every rank: all of them take part
        metas = [None] * world
        dist.all_gather_object(metas, meta, group=group)
        if meta is None:
            shape, dtype, _ = next(m for m in metas if m is not None)
            dev = "cpu" if dist.get_backend(group) == "gloo" else torch.device("cuda", torch.cuda.current_device())
            meta = (shape, dtype, str(dev))
    if meta is None:
        raise ValueError("run_sharded: this rank has no clips and no `like` tensor to take the result shape from")
    local = torch.stack(outs, 0) if outs else torch.empty((0,) + tuple(meta[0]), dtype=meta[1], device=torch.device(meta[2]))
    return gather_clip_masks(local, n_total, group)
