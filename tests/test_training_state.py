"""Learning-rate schedule and checkpoint formats, CPU part: the schedule against HF
`get_cosine_schedule_with_warmup` (what train_contrast.py:631 calls), the state dicts against torch's own optimizer /
scheduler classes and against the files the reference recipe wrote (tests/golden/train_state_*.pt)."""
import copy
import json
import os
import types

import pytest
import torch
from torch import nn

from p2t_hip import training_state as ts

HERE = os.path.dirname(os.path.abspath(__file__))


class _Adapter(nn.Module):
    def __init__(self):
        super().__init__()
        self.fc1, self.fc2 = nn.Linear(6, 5), nn.Linear(5, 4)
        self.ln1, self.ln2 = nn.LayerNorm(5), nn.LayerNorm(4)


class _Model(nn.Module):
    def __init__(self):
        super().__init__()
        self.esm_encoder = nn.Linear(3, 6)
        self.adapter = _Adapter()
        self.llama_decoder = nn.Linear(4, 4)


def _fake_trainer(schedule=None, step_count=3):
    """Only the attributes training_state touches; the real ContrastiveTrainer needs the GPU library."""
    torch.manual_seed(0)
    model = _Model()
    ad = model.adapter
    shapes = [p.shape for p in (ad.fc1.weight, ad.fc1.bias, ad.fc2.weight, ad.fc2.bias)]
    return types.SimpleNamespace(model=model, m=[torch.randn(s) for s in shapes], v=[torch.rand(s) for s in shapes],
                                 hp=dict(lr=2e-4, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.01), step_count=step_count,
                                 schedule=schedule)


@pytest.mark.parametrize("warmup,total", [(0, 10), (2, 10), (6, 100), (5, 5)])
def test_schedule_matches_hf(warmup, total):
    from transformers import get_cosine_schedule_with_warmup
    p = nn.Parameter(torch.zeros(1))
    opt = torch.optim.AdamW([p], lr=3e-4)
    hf = get_cosine_schedule_with_warmup(opt, num_warmup_steps=warmup, num_training_steps=total)
    mine = ts.CosineWarmupSchedule(3e-4, warmup, total)
    for _ in range(total + 3):
        assert mine.lr() == pytest.approx(opt.param_groups[0]["lr"], rel=1e-12, abs=1e-18)
        opt.step()
        hf.step()
        mine.step()
    assert mine.state_dict()["last_epoch"] == hf.state_dict()["last_epoch"]
    assert mine.state_dict()["_last_lr"][0] == pytest.approx(hf.state_dict()["_last_lr"][0], rel=1e-12, abs=1e-18)


def test_schedule_for_run_uses_upstream_formula():
    s = ts.CosineWarmupSchedule.for_run(2e-4, num_epochs=3, steps_per_epoch=101, gradient_accumulation_steps=4)
    assert s.num_training_steps == 3 * (101 // 4) and s.num_warmup_steps == int(0.06 * s.num_training_steps)


def test_scheduler_state_dict_loads_into_torch():
    from transformers import get_cosine_schedule_with_warmup
    mine = ts.CosineWarmupSchedule(1e-3, 2, 10)
    for _ in range(4):
        mine.step()
    opt = torch.optim.AdamW([nn.Parameter(torch.zeros(1))], lr=1e-3)
    hf = get_cosine_schedule_with_warmup(opt, num_warmup_steps=2, num_training_steps=10)
    hf.load_state_dict(mine.state_dict())
    assert hf.last_epoch == 4 and hf.get_last_lr()[0] == pytest.approx(mine.lr())
    back = ts.CosineWarmupSchedule(5.0, 2, 10)
    back.load_state_dict(hf.state_dict())
    assert back.last_epoch == 4 and back.base_lr == 1e-3


def test_optimizer_state_dict_round_trips_through_torch_adamw():
    tr = _fake_trainer(ts.CosineWarmupSchedule(2e-4, 1, 20))
    for _ in range(3):
        tr.schedule.step()
    sd = ts.optimizer_state_dict(tr)
    idx = ts.adapter_param_indices(tr.model)
    assert idx == [2, 3, 4, 5] and sorted(sd["state"]) == idx
    assert sd["param_groups"][0]["params"] == list(range(12))
    opt = torch.optim.AdamW(tr.model.parameters(), lr=1.0)
    opt.load_state_dict(copy.deepcopy(sd))                       # torch accepts it as its own (it keeps the step tensors)
    g = opt.param_groups[0]
    assert g["lr"] == pytest.approx(tr.schedule.lr()) and g["eps"] == 1e-6 and g["initial_lr"] == 2e-4
    params = list(tr.model.parameters())
    for i, m, v in zip(idx, tr.m, tr.v):
        st = opt.state[params[i]]
        assert float(st["step"]) == 3 and torch.equal(st["exp_avg"], m) and torch.equal(st["exp_avg_sq"], v)
    for p in params:                                             # and can step with it
        p.grad = torch.ones_like(p) if p.requires_grad else None
    opt.step()
    fresh = _fake_trainer(ts.CosineWarmupSchedule(9.0, 1, 20), step_count=0)
    for t in fresh.m + fresh.v:
        t.zero_()
    ts.load_optimizer_state_dict(fresh, sd)
    assert fresh.step_count == 3 and fresh.hp["lr"] == 2e-4 and fresh.schedule.base_lr == 2e-4
    assert all(torch.equal(a, b) for a, b in zip(fresh.m + fresh.v, tr.m + tr.v))


def test_optimizer_state_dict_before_first_step_is_empty():
    sd = ts.optimizer_state_dict(_fake_trainer(step_count=0))
    assert sd["state"] == {} and "initial_lr" not in sd["param_groups"][0]


def test_load_rejects_wrong_shapes():
    tr = _fake_trainer()
    sd = ts.optimizer_state_dict(tr)
    sd["state"][2]["exp_avg"] = torch.zeros(7, 7)
    with pytest.raises(ValueError):
        ts.load_optimizer_state_dict(_fake_trainer(), sd)


def test_reference_checkpoint_files_have_the_documented_layout():
    """The fixture written by the reference recipe (make_golden.py run_train_state): safe loader, torch formats."""
    with open(os.path.join(HERE, "golden", "train_state.json")) as f:
        meta = json.load(f)
    sd = torch.load(os.path.join(HERE, "golden", "train_state_optimizer_scheduler.pt"), weights_only=True)
    names = meta["param_names"]
    want = [names.index(n) for n in ts.ADAPTER_PARAM_NAMES]
    assert sorted(sd["optimizer_state_dict"]["state"]) == want          # only the four trained tensors carry state
    assert sd["optimizer_state_dict"]["param_groups"][0]["params"] == list(range(len(names)))
    # the reference advances its LambdaLR once per EPOCH (train_contrast.py:662): last_epoch counts epochs, the Adam
    # step counts optimizer steps
    n_steps, spe = len(meta["lrs"]), meta["steps_per_epoch"]
    assert sd["scheduler_state_dict"]["last_epoch"] == n_steps // spe == 3
    assert all(float(st["step"]) == n_steps for st in sd["optimizer_state_dict"]["state"].values())
    sched = ts.CosineWarmupSchedule(meta["lr"], meta["warmup"], meta["total_steps"])
    lrs = []
    for i in range(n_steps):
        lrs.append(sched.lr())
        if (i + 1) % spe == 0:
            sched.step()
    assert lrs == pytest.approx(meta["lrs"], rel=1e-12, abs=1e-18) and lrs[0] == 0.0      # lr 0 through epoch 1, as upstream
    assert sched.lr() == pytest.approx(sd["scheduler_state_dict"]["_last_lr"][0], rel=1e-12)
    model_sd = torch.load(os.path.join(HERE, "golden", "train_state_model.pt"), weights_only=True)
    assert sorted(model_sd) == sorted(f"{m}.{p}" for m in ("fc1", "fc2", "ln1", "ln2") for p in ("weight", "bias"))
