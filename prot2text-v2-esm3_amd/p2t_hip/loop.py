"""Epoch driver of the contrastive stage: host-side mirror of the reference's train_epoch / eval_epoch / epoch loop.

    train_epoch      scripts/train_contrast.py:400-480
    eval_epoch       :483-519
    run_epochs       :650-701   (sampler.set_epoch -> train -> scheduler.step -> barrier -> eval -> rank-0 checkpoint)

What the reference does on the HOST for every batch -- `loss.item()`, the "impossible batch loss" print (:431-434), the
running sums `ddp_loss` / `ddp_gradnorm` (:410-413, :443-444, :461-462) -- stays on the DEVICE here: one launch of
`p2t_epoch_accumulate` per batch adds the loss (and, after an optimizer step, the gradient norm) to four f32 sums and keeps
the guards as four integer flags (include/p2t_hip.h).  The host looks at the flags once every `check_every` batches and at the
epoch end, which is where

  * an impossible batch loss (NaN, inf or <= 0) is reported with the batch index and value of the first one -- the
    reference's message, `check_every` batches late at most;
  * a set fault word (a split-K consumer of an MFMA GEMM timed out, csrc/gemm_w4.hip) RAISES: the product never
    continues on poisoned activations;

and the epoch ends as the reference's does: ONE all-reduce (SUM) of [loss sum, batches] over the ranks (:468, :514), rank 0
prints the reference's summary line, and a NaN epoch loss raises ValueError (:476-480) -- on every rank, since every rank
holds the same reduced sum (the reference raises on rank 0 only and lets `mp.spawn` tear the others down).
"""
from __future__ import annotations

import struct
from typing import Any, Callable, Dict, Iterable, Optional

import torch
import torch.distributed as dist

from . import _lib, sharding
from ._lib import call
from .ops import ptr, stream


class SplitKTimeout(RuntimeError):
    """The GPU's sticky fault word is set: a split-K consumer gave up waiting for its producer, its tiles are NaN."""


class EpochStats:
    """Device-resident bookkeeping of one epoch: sums f32[4] = {loss sum, batches, gradient-norm sum, optimizer steps},
    flags i32[4] = {impossible losses, index of the first, fault word, bits of the first impossible loss}."""

    def __init__(self, device):
        self.sums = torch.zeros((4,), dtype=torch.float32, device=device)
        self.flags = torch.tensor([0, -1, 0, 0], dtype=torch.int32, device=device)
        self._reported = 0

    def add(self, loss: torch.Tensor, grad_norm: Optional[torch.Tensor], batch_idx: int) -> None:
        """One launch, no host synchronisation."""
        call("p2t_epoch_accumulate", ptr(loss), ptr(grad_norm) if grad_norm is not None else None, int(batch_idx),
             ptr(self.sums), ptr(self.flags), stream())

    def check(self, log: Callable[[str], None] = print) -> None:
        """Host look at the flags (synchronises with the stream): report new impossible losses, raise on a fault."""
        n_bad, first, fault, bits = (int(v) for v in self.flags.cpu().tolist())
        if n_bad > self._reported:
            value = struct.unpack("<f", struct.pack("<i", bits))[0]
            more = f" (and {n_bad - 1} more since)" if n_bad > 1 else ""
            log(f"Impossible batch_loss detected at batch {first}: {value}{more}")
            self._reported = n_bad
        if fault:
            raise SplitKTimeout("p2t fault word = %#x: a split-K GEMM consumer timed out waiting for its producer; the step's "
                                "activations were poisoned with NaN.  Clear with p2t_fault_status(clear=1) after fixing the cause "
                                "(two persistent grids that cannot be co-resident?)." % fault)


def _to_device(batch: Dict[str, Any], device) -> Dict[str, Any]:
    """Tensors of a collated batch to `device` (no-op for resident ones); host-side planning keys stay on the host."""
    host_keys = ("protein_lengths", "description_lengths")
    return {k: (v.to(device, non_blocking=True) if torch.is_tensor(v) and k not in host_keys else v) for k, v in batch.items()}


def _reduce_epoch_sums(stats: EpochStats, group=None) -> list:
    """`dist.all_reduce(ddp_loss, op=SUM)` (train_contrast.py:468,514) on [loss sum, batches]; -> the four sums on the host."""
    if sharding.world_info(group)[1] > 1:
        dist.all_reduce(stats.sums[:2], op=dist.ReduceOp.SUM, group=group)
    return [float(v) for v in stats.sums.cpu().tolist()]


def train_epoch(trainer, dataloader: Iterable[Dict[str, Any]], *, rank: int = 0, current_epoch: int = 1, num_epochs: int = 1,
                check_every: int = 50, log: Callable[[str], None] = print, progress: Optional[Callable] = None,
                stats: Optional[EpochStats] = None) -> Dict[str, float]:
    """One epoch of teacher-forced contrastive training (train_contrast.py:400-480) on `trainer` (a ContrastiveTrainer:
    its `step` is forward + backward of one micro-batch and, every `gradient_accumulation_steps` calls, clip + AdamW).

    `progress(batch_idx, batch)` (optional) is called after each enqueue -- a tqdm.update, say; it gets no loss value,
    reading one would put the reference's per-batch host sync back.  `stats`: the bookkeeping object to use (default: a fresh
    EpochStats on the trainer's device).  Dropout follows `trainer.train_mode` (default True = the reference's
    `model.train()`, :409).  Returns {"train_loss", "epoch_lr", "epoch_gradnorm",
    "batches", "optimizer_steps", "impossible_batches"}."""
    if check_every < 1:
        raise ValueError("check_every must be >= 1")
    trainer.model.train(bool(trainer.train_mode))
    trainer._micro = 0                             # optimizer.zero_grad(): accumulated gradients of the last epoch are dropped (:414)
    stats = EpochStats(trainer.dev) if stats is None else stats
    n = 0
    # one batch of look-ahead: the next batch's frozen towers are enqueued beside this step's backward / optimizer tail
    # (ContrastiveTrainer.step(batch, next_batch)); `ready` = an event of this stream after the batch became valid on it
    it = iter(dataloader)
    nxt = next(it, None)
    nxt_dev = _to_device(nxt, trainer.dev) if nxt is not None else None
    batch_idx = 0
    while nxt_dev is not None:
        data_batch, cur = nxt, nxt_dev
        nxt = next(it, None)
        nxt_dev = _to_device(nxt, trainer.dev) if nxt is not None else None       # (the SAME dict object comes back as `cur` next turn)
        ahead, ready = None, None
        if nxt_dev is not None and getattr(trainer, "overlap_streams", False):
            ahead, ready = nxt_dev, torch.cuda.current_stream().record_event()
        steps_before = trainer.step_count
        loss = trainer.step(cur, next_batch=ahead, next_ready=ready) if ahead is not None else trainer.step(cur)
        stats.add(loss, trainer.grad_norm if trainer.step_count > steps_before else None, batch_idx)
        n += 1
        if progress is not None:
            progress(batch_idx, data_batch)
        if n % check_every == 0:
            stats.check(log)
        batch_idx += 1
    stats.check(log)
    loss_sum, batches, gn_sum, steps = _reduce_epoch_sums(stats, trainer.group)
    train_loss = loss_sum / batches if batches else float("nan")
    gradnorm = gn_sum / steps if steps else float("nan")
    lr = trainer.schedule.lr() if trainer.schedule is not None else trainer.hp["lr"]
    if rank == 0:
        log(f"[epoch={current_epoch}/{num_epochs}, train_loss={train_loss}, epoch_lr={lr}, epoch_gradnorm={gradnorm}]")
    if batches and loss_sum != loss_sum:           # NaN detection (:476-480)
        raise ValueError("NaN detected in the training loss of the epoch, training interrupted.")
    return {"train_loss": train_loss, "epoch_lr": lr, "epoch_gradnorm": gradnorm, "batches": batches,
            "optimizer_steps": steps, "impossible_batches": int(stats.flags[0].item())}


@torch.no_grad()
def eval_epoch(trainer, dataloader: Iterable[Dict[str, Any]], *, rank: int = 0, current_epoch: int = 1, num_epochs: int = 1,
               check_every: int = 50, log: Callable[[str], None] = print, progress: Optional[Callable] = None,
               stats: Optional[EpochStats] = None) -> Dict[str, float]:
    """One evaluation pass (train_contrast.py:483-519): forward-only loss of every batch in eval mode, summed on the device,
    one all-reduce at the end, rank 0 prints `eval_loss`."""
    trainer.model.eval()
    stats = EpochStats(trainer.dev) if stats is None else stats
    n = 0
    for batch_idx, data_batch in enumerate(dataloader):
        stats.add(trainer.evaluate(_to_device(data_batch, trainer.dev)), None, batch_idx)
        n += 1
        if progress is not None:
            progress(batch_idx, data_batch)
        if n % check_every == 0:
            stats.check(log)
    stats.check(log)
    loss_sum, batches, _, _ = _reduce_epoch_sums(stats, trainer.group)
    eval_loss = loss_sum / batches if batches else float("nan")
    if rank == 0:
        log(f"[epoch={current_epoch}/{num_epochs}, eval_loss={eval_loss}]")
    return {"eval_loss": eval_loss, "batches": batches}


def run_epochs(trainer, train_loader, eval_loader=None, *, num_epochs: int, rank: int = 0, start_epoch: int = 1,
               checkpoint_dir: Optional[str] = None, train_sampler=None, check_every: int = 50,
               log: Callable[[str], None] = print) -> list:
    """The per-rank epoch loop of train_on_device (train_contrast.py:650-701): for every epoch `sampler.set_epoch`,
    `train_epoch`, `scheduler.step()` (ONCE PER EPOCH, :662 -- `trainer.end_epoch()`), barrier, `eval_epoch`, and on rank 0
    the two checkpoint files of :674-701 (`model_checkpoint_{e}.pt`, `optimizer_scheduler_checkpoint_{e}.pt`)."""
    from . import training_state
    history = []
    world = sharding.world_info(trainer.group)[1]
    for epoch in range(start_epoch, num_epochs + 1):
        if train_sampler is not None and hasattr(train_sampler, "set_epoch"):
            train_sampler.set_epoch(epoch)
        rec = {"epoch": epoch}
        rec.update(train_epoch(trainer, train_loader, rank=rank, current_epoch=epoch, num_epochs=num_epochs,
                               check_every=check_every, log=log))
        trainer.end_epoch()
        if world > 1:
            dist.barrier(group=trainer.group)
        if eval_loader is not None:
            rec.update(eval_epoch(trainer, eval_loader, rank=rank, current_epoch=epoch, num_epochs=num_epochs,
                                  check_every=check_every, log=log))
        if checkpoint_dir is not None and rank == 0:
            training_state.save_checkpoint(trainer, checkpoint_dir, epoch)
        if world > 1:
            dist.barrier(group=trainer.group)
        history.append(rec)
    return history


# ---------------------------------------------------------------------------------------------------------------------------
# inference (scripts/generate_instruct.py:50-147)
# ---------------------------------------------------------------------------------------------------------------------------
def iterative_generation_loop(rank, model, data_batch: Dict[str, Any], max_generation_length: int, num_beams: int = 1, length_penalty: float = 1.0,
                              temperature: float = 1.0, do_sample: bool = False, top_p: float = 1.0, top_k: int = 50,
                              eos_token_id=128009, pad_token_id: int = 128002, **generate_kwargs) -> torch.Tensor:
    """scripts/generate_instruct.py:50-87: the batch's prompt + protein ids to `model.generate`, the new token ids back.  `rank` is the
    device (an index or a torch.device), as upstream's `.to(rank)`; a DistributedDataParallel-style wrapper is unwrapped (`.module`)."""
    model = getattr(model, "module", model)
    dev = torch.device("cuda", rank) if isinstance(rank, int) else torch.device(rank)
    to = lambda k: data_batch[k].to(dev)
    return model.generate(inputs=to("input_ids"), attention_mask=to("attention_mask"), protein_input_ids=to("protein_input_ids"),
                          protein_attention_mask=to("protein_attention_mask"), max_new_tokens=max_generation_length, eos_token_id=eos_token_id,
                          pad_token_id=pad_token_id, return_dict_in_generate=False, num_beams=num_beams, length_penalty=length_penalty,
                          temperature=temperature, do_sample=do_sample, top_p=top_p, top_k=top_k, **generate_kwargs)


def inference_epoch(rank, model, dataloader: Iterable[Dict[str, Any]], llama_tokenizer, args: Dict[str, Any],
                    progress: Optional[Callable] = None) -> Optional[str]:
    """scripts/generate_instruct.py:90-147: generate for every batch of this rank's shard, decode predictions and labels with the
    description tokenizer (`batch_decode(..., skip_special_tokens=True)`) and write
    `generation_{save_generation_postfix_identifier}_rank{rank}.json` = {name: {"true": label, "pred": prediction}} into
    `save_generation_dir` (returned).  Every rank generates on its own (a DistributedSampler shard per rank upstream): no collective."""
    import json
    import os
    model.eval()
    names, preds, labels = [], [], []
    for i, data_batch in enumerate(dataloader):
        with torch.no_grad():
            out = iterative_generation_loop(rank, model, data_batch, args["max_generation_length"], args.get("num_beams", 1),
                                            args.get("length_penalty", 1.0), args.get("temperature", 1.0), args.get("do_sample", False),
                                            args.get("top_p", 1.0), args.get("top_k", 50))
        names.extend(data_batch["name"])
        preds.extend(llama_tokenizer.batch_decode(out.cpu(), skip_special_tokens=True))
        labels.extend(llama_tokenizer.batch_decode(data_batch["description_input_ids"], skip_special_tokens=True))
        if progress is not None:
            progress(i, {"mode": "inference", "batch_maxlen_gen": out.shape[1], "device": f"rank:{rank}"})
    if not args.get("save_generation_dir"):
        return None
    r = rank if isinstance(rank, int) else (torch.device(rank).index or 0)
    path = os.path.join(args["save_generation_dir"], f"generation_{args['save_generation_postfix_identifier']}_rank{r}.json")
    with open(path, "w") as f:
        json.dump({n: {"true": t, "pred": p} for n, t, p in zip(names, labels, preds)}, f, indent=4)
    print(f"Saving {path}")
    return path
