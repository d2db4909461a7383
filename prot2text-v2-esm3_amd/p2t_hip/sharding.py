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
                             column_fn: Optional[Callable[[torch.Tensor, torch.Tensor, int], None]] = None) -> None:
    """One micro-batch of the sharded step on this rank.

        t_local = text_fn()                              normalised text embeddings of the local rows (no grad)
        t_all, offset = all-gather over ranks            (global_negatives; else the local block, offset 0)
        for every protein-side segment (r0, r1, T, w):   segment_fn(s, r0, r1, T, w, t_all, labels, offset)
            with labels = offset + arange(r0, r1)        -> loss rows + adapter gradients of that segment
        average the flat adapter gradients over ranks    (backward and reduce)

    Column (text -> protein) term, optional: `protein_fn()` returns the normalised protein embeddings of ALL local rows
    (forward only); they are all-gathered too and `column_fn(p_all, t_all, offset)` prepares the global column
    log-sum-exps before the segments run."""
    t_local = text_fn()
    t_all, offset = gather_rows(t_local, group) if global_negatives else (t_local, 0)
    if column_fn is not None:
        p_local = protein_fn()
        p_all, p_off = gather_rows(p_local, group) if global_negatives else (p_local, 0)
        assert p_off == offset
        column_fn(p_all, t_all, offset)
    dev = t_all.device
    for s, (r0, r1, T, weight) in enumerate(segments):
        segment_fn(s, r0, r1, T, weight, t_all, segment_labels(r0, r1, offset, dev), offset)
    if backward and reduce and flat_g is not None:
        average_gradients(flat_g, group)
