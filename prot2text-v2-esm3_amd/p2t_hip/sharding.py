"""Rank logic of the data-parallel contrastive step, free of any kernel call.

The reference runs one process per GPU (`mp.spawn`, scripts/train_contrast.py:706-718), shards the batch with a
DistributedSampler (:551-556) and lets DDP average the adapter gradients (:611-614).  Here the same partition plus ONE
exchange step (all-gather of the no-grad text embeddings, so every rank scores its rows against the GLOBAL batch) is
written as plain functions over `torch.distributed`, so that the very code the GPU step runs over RCCL is driven over
gloo with CPU tensors in tests/test_distributed_gloo.py (the arithmetic between the collectives is passed in as callables:
HIP kernels in ContrastiveTrainer, the numpy oracle in the test).
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def world_info(group=None) -> Tuple[int, int]:
    """(rank, world size) of `group`; (0, 1) without an initialised process group."""
    if not (dist.is_available() and dist.is_initialized()):
        return 0, 1
    return dist.get_rank(group), dist.get_world_size(group)


def local_rows(rank: int, world: int, global_batch: int) -> slice:
    """Rows of the global batch rank `rank` encodes: [rank * B_loc, (rank + 1) * B_loc), B_loc = global_batch // world
    (equal shares, tail dropped: DistributedSampler(drop_last=True), scripts/train_contrast.py:551-556,569-575)."""
    if world <= 0 or not 0 <= rank < world:
        raise ValueError(f"rank {rank} outside world of {world}")
    b_loc = global_batch // world
    return slice(rank * b_loc, (rank + 1) * b_loc)


def gather_rows(x_local: torch.Tensor, group=None) -> Tuple[torch.Tensor, int]:
    """All-gather equal [B_loc, D] blocks in rank order -> ([world * B_loc, D], this rank's first row).
    No gradient flows through it (the gathered side is computed under no_grad / is re-differentiated locally)."""
    rank, world = world_info(group)
    if world == 1:
        return x_local, 0
    x_local = x_local.contiguous()
    out = torch.empty((world * x_local.shape[0], x_local.shape[1]), dtype=x_local.dtype, device=x_local.device)
    dist.all_gather_into_tensor(out, x_local, group=group)
    return out, rank * x_local.shape[0]


class PendingGather:
    """An all-gather in flight (`gather_rows_async`): `wait()` makes the CURRENT stream (RCCL) or the host (gloo) wait for it and
    returns (gathered rows, this rank's first row) -- so the kernels enqueued between issue and wait do not queue behind the
    collective, nor behind a slower rank's half of it."""

    def __init__(self, out: torch.Tensor, offset: int, work=None):
        self.out, self.offset, self._work = out, offset, work

    def wait(self) -> Tuple[torch.Tensor, int]:
        if self._work is not None:
            self._work.wait()
            self._work = None
        return self.out, self.offset


def gather_rows_async(x_local: torch.Tensor, group=None) -> PendingGather:
    """`gather_rows` issued with async_op=True: returns at once; the result is valid after `.wait()`."""
    rank, world = world_info(group)
    if world == 1:
        return PendingGather(x_local, 0)
    x_local = x_local.contiguous()
    out = torch.empty((world * x_local.shape[0], x_local.shape[1]), dtype=x_local.dtype, device=x_local.device)
    work = dist.all_gather_into_tensor(out, x_local, group=group, async_op=True)
    return PendingGather(out, rank * x_local.shape[0], work)


def average_loss(loss: torch.Tensor, group=None) -> torch.Tensor:
    """Mean over ranks of a (device) loss scalar -- equal local batches, so this is the loss of the global batch.  The
    reference all-reduces its loss sums once per epoch (scripts/train_contrast.py:468,514); this is the per-step form for
    logging.  Returns a NEW tensor; `loss` is untouched."""
    _, world = world_info(group)
    out = loss.detach().clone()
    if world == 1:
        return out
    dist.all_reduce(out, op=dist.ReduceOp.SUM, group=group)
    out /= world
    return out


def segment_labels(r0: int, r1: int, offset: int, device, dtype=torch.int32) -> torch.Tensor:
    """Target columns of local rows r0..r1-1 in the gathered text matrix: the reference's
    `labels = arange(seg * size, (seg + 1) * size)` (scripts/train_contrast.py:367-371) shifted by the rank's offset
    (rank == segment of the single-process formulation)."""
    return torch.arange(r0 + offset, r1 + offset, device=device, dtype=dtype)


def average_gradients(flat_g: torch.Tensor, group=None) -> torch.Tensor:
    """DDP's gradient averaging (scripts/train_contrast.py:611-614) as ONE all-reduce of the flat adapter-gradient buffer.
    RCCL averages in the collective; gloo has no AVG, so there it is SUM then a division."""
    _, world = world_info(group)
    if world == 1:
        return flat_g
    if dist.get_backend(group) == "nccl":
        dist.all_reduce(flat_g, op=dist.ReduceOp.AVG, group=group)
    else:
        dist.all_reduce(flat_g, op=dist.ReduceOp.SUM, group=group)
        flat_g /= world
    return flat_g


def micro_step_plan(micro: int, ga: int) -> Tuple[bool, float, bool, bool]:
    """Gradient accumulation over `ga` micro-batches (scripts/train_contrast.py:428-465): for micro-batch index `micro`
    (0-based within the window) -> (accumulate into the gradient buffer, factor on this micro-batch's gradients,
    average across ranks after it, run clip + AdamW after it).  The reference lets DDP all-reduce after EVERY backward
    (no `no_sync()`); sums commute with the average, so one exchange per optimizer step gives the same gradients."""
    if ga < 1 or not 0 <= micro < ga:
        raise ValueError(f"micro-batch {micro} outside an accumulation window of {ga}")
    last = micro == ga - 1
    return micro > 0, 1.0 / ga, last, last


Segment = Tuple[int, int, int, float]        # (first row, stop row, padded length, loss weight)


def sharded_forward_backward(*, text_fn: Callable[[], torch.Tensor],
                             segment_fn: Callable[[int, int, int, int, float, torch.Tensor, torch.Tensor, int], None],
                             segments: Sequence[Segment], global_negatives: bool, flat_g: Optional[torch.Tensor],
                             backward: bool = True, reduce: bool = True, group=None,
                             protein_fn: Optional[Callable[[], torch.Tensor]] = None,
                             column_fn: Optional[Callable[[torch.Tensor, torch.Tensor, int], None]] = None,
                             prefetch_fn: Optional[Callable[[int, int, int, int], None]] = None) -> None:
    """One micro-batch of the sharded step on this rank.

        t_local = text_fn()                              normalised text embeddings of the local rows (no grad)
        all-gather over ranks, ISSUED ASYNCHRONOUSLY     (global_negatives; else the local block, offset 0)
        prefetch_fn(0, r0, r1, T)                        optional: enqueue the first segment's encoder (it does not depend on
                                                         the gathered text); the caller keeps the result for segment_fn
        wait for the gather                              -> t_all, offset
        for every protein-side segment (r0, r1, T, w):   segment_fn(s, r0, r1, T, w, t_all, labels, offset)
            with labels = offset + arange(r0, r1)        -> loss rows + adapter gradients of that segment
        average the flat adapter gradients over ranks    (backward and reduce)

    The encoder of the first segment (85 % of the step) thus runs while the text embeddings travel, and a rank that finishes
    its text tower early starts encoding instead of waiting for the slowest rank's text tower.

    Column (text -> protein) term, optional: `protein_fn()` returns the normalised protein embeddings of ALL local rows
    (forward only); they are all-gathered too and `column_fn(p_all, t_all, offset)` prepares the global column
    log-sum-exps before the segments run."""
    t_local = text_fn()
    pending = gather_rows_async(t_local, group) if global_negatives else PendingGather(t_local, 0)
    if column_fn is not None:
        p_local = protein_fn()                          # every segment's forward: runs while the text gather is in flight
        t_all, offset = pending.wait()
        p_all, p_off = gather_rows(p_local, group) if global_negatives else (p_local, 0)
        assert p_off == offset
        column_fn(p_all, t_all, offset)
    else:
        if prefetch_fn is not None and len(segments) > 0:
            r0, r1, T, _ = segments[0]
            prefetch_fn(0, r0, r1, T)
        t_all, offset = pending.wait()
    dev = t_all.device
    for s, (r0, r1, T, weight) in enumerate(segments):
        segment_fn(s, r0, r1, T, weight, t_all, segment_labels(r0, r1, offset, dev), offset)
    if backward and reduce and flat_g is not None:
        average_gradients(flat_g, group)
