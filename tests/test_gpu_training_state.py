"""Training recipe and checkpoints on the GPU: three epochs of two optimizer steps of
ContrastiveTrainer with the cosine schedule stepped once per epoch, as the reference does, against the reference's recipe (tests/golden/train_state*.{json,pt}: AdamW over
model.parameters() + HF warm-up/cosine + the reference loss, make_golden.py run_train_state), the checkpoint files in
both directions, and resume-equals-continue."""
import json
import os

import numpy as np
import pytest
import torch

from gpu_util import build_model, rel, to_dev, to_np
from p2t_hip import specs, synth
from p2t_hip import training_state as ts

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
NAMES = ("fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias")


def _meta():
    with open(os.path.join(HERE, "golden", "train_state.json")) as f:
        return json.load(f)


def _batch(meta, step):
    pid, pmask = synth.protein_batch(100 + step, meta["B"], meta["T_p"], [24, 15, 7, 3])
    tid, tmask = synth.text_batch(100 + step, meta["B"], meta["T_t"], 500, [12, 9, 5, 2], 510, 509)
    return {"protein_input_ids": to_dev(pid), "protein_attention_mask": to_dev(pmask),
            "description_input_ids": to_dev(tid), "description_attention_mask": to_dev(tmask)}


def _trainer(meta, steps=0):
    import p2t_hip as P
    esm, llama, ad = (specs.EsmSpec(**meta["esm"]), specs.LlamaSpec(**meta["llama"]), specs.AdapterSpec(**meta["adapter"]))
    model = build_model(esm, llama, ad, torch.float32, 0)
    model.train()
    sched = ts.CosineWarmupSchedule(meta["lr"], meta["warmup"], meta["total_steps"])
    tr = P.ContrastiveTrainer(model, lr=meta["lr"], schedule=sched, output_llm_layer=meta["layer"], train_mode=True)
    losses, lrs = [], []
    for s in range(steps):
        lrs.append(sched.lr())
        losses.append(float(to_np(tr.step(_batch(meta, s)))[0]))
        if (s + 1) % meta["steps_per_epoch"] == 0:
            tr.end_epoch()                      # scheduler.step() once per epoch (train_contrast.py:662)
    return tr, losses, lrs


def test_parameter_order_matches_reference():
    """torch keys optimizer state by position in model.parameters(): the module tree must enumerate like upstream's."""
    meta = _meta()
    tr, _, _ = _trainer(meta)
    assert [n for n, _ in tr.model.named_parameters()] == meta["param_names"]
    assert ts.adapter_param_indices(tr.model) == [meta["param_names"].index(n) for n in ts.ADAPTER_PARAM_NAMES]


def test_three_epochs_match_reference_recipe(tmp_path):
    meta = _meta()
    tr, losses, lrs = _trainer(meta, steps=6)
    assert lrs[:2] == [0.0, 0.0] and tr.schedule.last_epoch == 3 and tr.step_count == 6
    assert lrs == pytest.approx(meta["lrs"], rel=1e-12, abs=1e-18)
    assert losses == pytest.approx(meta["losses"], rel=2e-4)
    ref_model = torch.load(os.path.join(HERE, "golden", "train_state_model.pt"), weights_only=True)
    for t, n in zip(tr.p, NAMES):
        np.testing.assert_allclose(to_np(t), ref_model[n].numpy(), rtol=2e-4, atol=2e-6)
    ref = torch.load(os.path.join(HERE, "golden", "train_state_optimizer_scheduler.pt"), weights_only=True)
    mine = ts.optimizer_state_dict(tr)
    assert sorted(mine["state"]) == sorted(ref["optimizer_state_dict"]["state"])
    for i in mine["state"]:
        a, b = mine["state"][i], ref["optimizer_state_dict"]["state"][i]
        assert float(a["step"]) == float(b["step"]) == 6
        assert rel(a["exp_avg"].numpy(), b["exp_avg"].numpy()) < 1e-3
        assert rel(a["exp_avg_sq"].numpy(), b["exp_avg_sq"].numpy()) < 1e-3
    ga, gb = mine["param_groups"][0], ref["optimizer_state_dict"]["param_groups"][0]
    assert set(ga) == set(gb) and ga["params"] == gb["params"]
    for k in ("lr", "eps", "weight_decay", "initial_lr"):
        assert ga[k] == pytest.approx(gb[k], rel=1e-12)
    assert tuple(ga["betas"]) == tuple(gb["betas"])
    sa, sb = tr.schedule.state_dict(), ref["scheduler_state_dict"]
    assert set(sa) == set(sb) and sa["last_epoch"] == sb["last_epoch"] and sa["_step_count"] == sb["_step_count"]
    assert sa["_last_lr"][0] == pytest.approx(sb["_last_lr"][0], rel=1e-12)
    # our files: the upstream names, loadable with the safe loader, adapter file has upstream's eight keys
    paths = ts.save_checkpoint(tr, str(tmp_path), 7)
    assert [os.path.basename(p) for p in paths] == ["model_checkpoint_7.pt", "optimizer_scheduler_checkpoint_7.pt"]
    sd = torch.load(paths[0], weights_only=True)
    assert sorted(sd) == sorted(ref_model)
    for n in NAMES:
        np.testing.assert_allclose(sd[n].numpy(), ref_model[n].numpy(), rtol=2e-4, atol=2e-6)


def test_resume_from_reference_checkpoint_equals_continuing(tmp_path):
    """A fresh trainer that loads the REFERENCE's files (scheduler `last_epoch` = epochs completed, Adam `step` = optimizer
    steps) continues exactly like the trainer that ran the three epochs itself would from the same state; and resuming from
    our own files is bit-identical to not stopping."""
    meta = _meta()
    cont, _, _ = _trainer(meta, steps=6)
    own = ts.save_checkpoint(cont, str(tmp_path), 1)
    loss_cont = float(to_np(cont.step(_batch(meta, 4)))[0])
    res, _, _ = _trainer(meta)
    ts.load_model_checkpoint(res.model, own[0], trainer=res)
    ts.load_optimizer_scheduler_checkpoint(res, own[1])
    assert res.step_count == 6 and res.schedule.last_epoch == 3
    loss_res = float(to_np(res.step(_batch(meta, 4)))[0])
    assert loss_res == loss_cont
    for a, b in zip(res.p + res.m + res.v, cont.p + cont.m + cont.v):
        assert torch.equal(a, b)
    ref_tr, _, _ = _trainer(meta)
    ts.load_model_checkpoint(ref_tr.model, os.path.join(HERE, "golden", "train_state_model.pt"), trainer=ref_tr)
    ts.load_optimizer_scheduler_checkpoint(ref_tr, os.path.join(HERE, "golden", "train_state_optimizer_scheduler.pt"))
    ref_sched = torch.load(os.path.join(HERE, "golden", "train_state_optimizer_scheduler.pt"), weights_only=True)["scheduler_state_dict"]
    assert ref_tr.step_count == 6 and ref_tr.schedule.last_epoch == 3 and ref_tr.schedule.lr() == pytest.approx(ref_sched["_last_lr"][0], rel=1e-12)
    loss_ref = float(to_np(ref_tr.step(_batch(meta, 4)))[0])
    assert loss_ref == pytest.approx(loss_cont, rel=2e-4)
    for a, b in zip(ref_tr.p, cont.p):
        np.testing.assert_allclose(to_np(a), to_np(b), rtol=2e-4, atol=2e-6)


def test_device_prefetcher_on_gpu_feeds_the_trainer():
    """Batches staged through pinned memory on the copy stream arrive intact, in order, on the device, and the fused step
    consumes them while the next copy is in flight (same losses as feeding the batches directly)."""
    import p2t_hip as P
    meta = _meta()
    host = []
    for s in range(4):
        pid, pmask = synth.protein_batch(100 + s, meta["B"], meta["T_p"], [24, 15, 7, 3])
        tid, tmask = synth.text_batch(100 + s, meta["B"], meta["T_t"], 500, [12, 9, 5, 2], 510, 509)
        host.append({"protein_input_ids": torch.from_numpy(pid), "protein_attention_mask": torch.from_numpy(pmask),
                     "description_input_ids": torch.from_numpy(tid), "description_attention_mask": torch.from_numpy(tmask),
                     "name": [f"b{s}"]})
    tr, _, _ = _trainer(meta)
    got, names = [], []
    for b in P.DevicePrefetcher(host, "cuda:0"):
        assert b["protein_input_ids"].is_cuda and torch.equal(b["protein_input_ids"].cpu(), host[len(got)]["protein_input_ids"])
        names.append(b["name"][0])
        got.append(float(to_np(tr.step(b))[0]))
    assert names == ["b0", "b1", "b2", "b3"]
    assert got[:3] == pytest.approx(meta["losses"][:3], rel=2e-4)       # lr is 0 through epoch 1 of the recipe: the first three losses see the initial parameters


def test_gradient_accumulation_matches_reference_loop_semantics():
    """train_contrast.py:428-465: each micro-batch back-propagates loss / GA into the same gradient buffers; clip + AdamW +
    scheduler run once per GA micro-batches; the logged loss is the micro-batch's own."""
    import p2t_hip as P
    meta = _meta()
    esm, llama, ad = (specs.EsmSpec(**meta["esm"]), specs.LlamaSpec(**meta["llama"]), specs.AdapterSpec(**meta["adapter"]))
    model = build_model(esm, llama, ad, torch.float32, 0)
    b0, b1 = _batch(meta, 0), _batch(meta, 1)
    one = P.ContrastiveTrainer(model, lr=meta["lr"], output_llm_layer=meta["layer"], train_mode=False)
    l0 = float(to_np(one.forward_backward(b0))[0]); g0 = to_np(one.flat_g).copy()
    l1 = float(to_np(one.forward_backward(b1))[0]); g1 = to_np(one.flat_g).copy()
    p_before = to_np(one.flat_p).copy()

    sched = ts.CosineWarmupSchedule(meta["lr"], 0, 10)
    ga = P.ContrastiveTrainer(model, lr=meta["lr"], output_llm_layer=meta["layer"], train_mode=False, schedule=sched,
                              gradient_accumulation_steps=2)
    assert float(to_np(ga.step(b0))[0]) == pytest.approx(l0, rel=1e-6)
    assert ga.step_count == 0 and sched.last_epoch == 0 and np.array_equal(to_np(ga.flat_p), p_before)    # no update yet
    assert rel(to_np(ga.flat_g), 0.5 * g0) < 1e-6
    assert float(to_np(ga.step(b1))[0]) == pytest.approx(l1, rel=1e-6)                                   # own loss, not a sum
    assert ga.step_count == 1 and sched.last_epoch == 0           # the schedule moves at end_epoch(), as upstream (:662)
    ga.end_epoch()
    assert sched.last_epoch == 1
    per_step = P.ContrastiveTrainer(model, schedule=ts.CosineWarmupSchedule(meta["lr"], 0, 10), schedule_step="step")
    per_step.optimizer_step()
    assert per_step.schedule.last_epoch == 1                      # schedule_step="step": the conventional per-step reading
    assert rel(to_np(ga.flat_g), 0.5 * (g0 + g1)) < 1e-5
    # the update equals one AdamW step (torch) on the averaged gradients
    ref_p = torch.nn.Parameter(torch.from_numpy(p_before.copy()))
    opt = torch.optim.AdamW([ref_p], lr=meta["lr"], eps=1e-6, betas=(0.9, 0.999))
    ref_p.grad = torch.from_numpy(0.5 * (g0 + g1))
    opt.step()
    assert rel(to_np(ga.flat_p), ref_p.detach().numpy()) < 1e-6
    # third micro-batch starts a fresh accumulation
    g_acc = to_np(ga.flat_g).copy()
    ga.step(b0)
    assert ga.step_count == 1 and ga._micro == 1
    assert rel(to_np(ga.flat_g), g_acc) > 0.1 and np.linalg.norm(to_np(ga.flat_g)) < 0.8 * np.linalg.norm(g0)   # overwritten, half weight
    with pytest.raises(ValueError):
        P.ContrastiveTrainer(model, gradient_accumulation_steps=0)
