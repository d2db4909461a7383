"""Learning-rate schedule and checkpoint formats of the contrastive stage.

The reference trains with `AdamW(model.parameters(), lr, eps=1e-6, betas=(0.9, 0.999))` and HF
`get_cosine_schedule_with_warmup` (`scripts/train_contrast.py:621-637`) and writes two files per saved epoch
(`:674-701`):

* `model_checkpoint_{epoch}.pt`                 -- `adapter.state_dict()` (fc1 / fc2 / ln1 / ln2 weight + bias);
* `optimizer_scheduler_checkpoint_{epoch}.pt`   -- `{"optimizer_state_dict": AdamW.state_dict(),
                                                    "scheduler_state_dict": LambdaLR.state_dict()}`.

`ContrastiveTrainer` keeps the adapter's fp32 masters and Adam moments in flat device buffers; the functions here
convert between those and the torch formats, so a run can resume from reference checkpoints and the reference can resume
from ours.  Parameter indices in the optimizer state follow `model.parameters()` order, as torch's do.
"""
from __future__ import annotations

import math
import os
from typing import Any, Dict, List, Optional

import torch

ADAPTER_PARAM_NAMES = ("adapter.fc1.weight", "adapter.fc1.bias", "adapter.fc2.weight", "adapter.fc2.bias")


class CosineWarmupSchedule:
    """lr multiplier of HF `get_cosine_schedule_with_warmup` (transformers/optimization.py, half a cosine period):
    step < warmup: step / max(1, warmup); else max(0, 0.5 (1 + cos(pi * 2 * cycles * progress))).
    As with torch's LambdaLR, optimizer step number k (1-based) runs at base_lr * factor(k - 1)."""

    def __init__(self, base_lr: float, num_warmup_steps: int, num_training_steps: int, num_cycles: float = 0.5):
        self.base_lr, self.num_warmup_steps = float(base_lr), int(num_warmup_steps)
        self.num_training_steps, self.num_cycles = int(num_training_steps), float(num_cycles)
        self.last_epoch = 0                      # number of scheduler steps taken (= optimizer steps)

    @classmethod
    def for_run(cls, base_lr: float, num_epochs: int, steps_per_epoch: int, gradient_accumulation_steps: int = 1):
        """total = epochs * (len(loader) // GA), warmup = int(0.06 * total)  (train_contrast.py:626-631)."""
        total = num_epochs * (steps_per_epoch // gradient_accumulation_steps)
        return cls(base_lr, int(0.06 * total), total)

    def factor(self, step: int) -> float:
        if step < self.num_warmup_steps:
            return float(step) / float(max(1, self.num_warmup_steps))
        progress = float(step - self.num_warmup_steps) / float(max(1, self.num_training_steps - self.num_warmup_steps))
        return max(0.0, 0.5 * (1.0 + math.cos(math.pi * self.num_cycles * 2.0 * progress)))

    def lr(self) -> float:
        """Learning rate of the NEXT optimizer step."""
        return self.base_lr * self.factor(self.last_epoch)

    def step(self) -> None:
        self.last_epoch += 1

    def state_dict(self) -> Dict[str, Any]:
        """Keys of `torch.optim.lr_scheduler.LambdaLR.state_dict()` (the lambda itself is not saved by torch either)."""
        return {"base_lrs": [self.base_lr], "last_epoch": self.last_epoch, "_step_count": self.last_epoch + 1,
                "_is_initial": False, "_get_lr_called_within_step": False, "_last_lr": [self.lr()], "lr_lambdas": [None]}

    def load_state_dict(self, sd: Dict[str, Any]) -> None:
        self.base_lr = float(sd["base_lrs"][0])
        self.last_epoch = int(sd["last_epoch"])


def adapter_param_indices(model) -> List[int]:
    """Positions of the four trained tensors in `model.parameters()` (what torch's optimizer state is keyed by)."""
    order = {id(p): i for i, p in enumerate(model.parameters())}
    ad = model.adapter
    return [order[id(p)] for p in (ad.fc1.weight, ad.fc1.bias, ad.fc2.weight, ad.fc2.bias)]


def _adamw_group_template() -> Dict[str, Any]:
    g = dict(torch.optim.AdamW([torch.nn.Parameter(torch.zeros(1))]).param_groups[0])
    g.pop("params")
    return g


def optimizer_state_dict(trainer) -> Dict[str, Any]:
    """`torch.optim.AdamW.state_dict()` of an optimizer built over `model.parameters()` (frozen towers included, as the
    reference does): state only for the four adapter tensors, one param group listing every parameter."""
    hp = trainer.hp
    idx = adapter_param_indices(trainer.model)
    n_params = sum(1 for _ in trainer.model.parameters())
    group = _adamw_group_template()
    lr = trainer.schedule.lr() if trainer.schedule is not None else hp["lr"]
    group.update(lr=lr, betas=tuple(hp["betas"]), eps=hp["eps"], weight_decay=hp["weight_decay"])
    if trainer.schedule is not None:
        group["initial_lr"] = trainer.schedule.base_lr
    group["params"] = list(range(n_params))
    state = {}
    if trainer.step_count > 0:
        for i, m, v in zip(idx, trainer.m, trainer.v):
            state[i] = {"step": torch.tensor(float(trainer.step_count)), "exp_avg": m.detach().cpu().clone(),
                        "exp_avg_sq": v.detach().cpu().clone()}
    return {"state": state, "param_groups": [group]}


def load_optimizer_state_dict(trainer, sd: Dict[str, Any]) -> None:
    idx = adapter_param_indices(trainer.model)
    state, group = sd["state"], sd["param_groups"][0]
    steps = set()
    with torch.no_grad():
        for i, m, v in zip(idx, trainer.m, trainer.v):
            st = state.get(i, state.get(str(i)))
            if st is None:
                m.zero_()
                v.zero_()
                continue
            if tuple(st["exp_avg"].shape) != tuple(m.shape):
                raise ValueError(f"optimizer state of parameter {i}: shape {tuple(st['exp_avg'].shape)} != {tuple(m.shape)}")
            m.copy_(st["exp_avg"].to(m.dtype))
            v.copy_(st["exp_avg_sq"].to(v.dtype))
            steps.add(int(float(st["step"])))
    if len(steps) > 1:
        raise ValueError(f"adapter tensors disagree on the Adam step count: {sorted(steps)}")
    trainer.step_count = steps.pop() if steps else 0
    trainer.hp.update(betas=tuple(group["betas"]), eps=float(group["eps"]), weight_decay=float(group["weight_decay"]))
    trainer.hp["lr"] = float(group.get("initial_lr", group["lr"]))
    if trainer.schedule is not None:
        trainer.schedule.base_lr = trainer.hp["lr"]


def save_checkpoint(trainer, save_checkpoint_dir: str, epoch_idx: int) -> List[str]:
    """The two files of `train_contrast.py:679-698`, same names and contents."""
    trainer.sync_to_module()
    os.makedirs(save_checkpoint_dir, exist_ok=True)
    model_path = os.path.join(save_checkpoint_dir, f"model_checkpoint_{epoch_idx}.pt")
    torch.save({k: v.detach().cpu() for k, v in trainer.model.adapter.state_dict().items()}, model_path)
    opt_path = os.path.join(save_checkpoint_dir, f"optimizer_scheduler_checkpoint_{epoch_idx}.pt")
    sched = trainer.schedule.state_dict() if trainer.schedule is not None else None
    torch.save({"optimizer_state_dict": optimizer_state_dict(trainer), "scheduler_state_dict": sched}, opt_path)
    return [model_path, opt_path]


def load_model_checkpoint(model, path: str, trainer=None) -> None:
    """`model.adapter.load_state_dict(torch.load(path, weights_only=True))` (train_contrast.py:175-183)."""
    sd = torch.load(path, weights_only=True, map_location="cpu")
    model.adapter.load_state_dict(sd)
    if trainer is not None:
        trainer.sync_from_module()


def load_optimizer_scheduler_checkpoint(trainer, path: str) -> None:
    """train_contrast.py:638-647."""
    sd = torch.load(path, weights_only=True, map_location="cpu")
    load_optimizer_state_dict(trainer, sd["optimizer_state_dict"])
    if trainer.schedule is not None and sd.get("scheduler_state_dict") is not None:
        trainer.schedule.load_state_dict(sd["scheduler_state_dict"])
