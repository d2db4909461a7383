"""The epoch driver (p2t_hip/loop.py = scripts/train_contrast.py:400-519, 650-701) on the GPU with the real trainer: device-side
epoch sums (p2t_epoch_accumulate) against per-batch host reads of a twin trainer, the reference's two guards (impossible
batch loss, epoch NaN abort), the split-K time-out word (sticky, NaN-poisons consumers, raises at the next check), the epoch
loop with scheduler / checkpoints, and two ranks of it on one GPU (gloo) against the single-process values."""
import json
import math
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

from gpu_util import build_model, dev, to_dev, to_np
from p2t_hip import specs, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _specs():
    esm = specs.EsmSpec(num_hidden_layers=2, hidden_size=128, intermediate_size=256, num_attention_heads=2)
    llama = specs.LlamaSpec(num_hidden_layers=2, hidden_size=128, intermediate_size=256, num_attention_heads=2, num_key_value_heads=1, vocab_size=512)
    return esm, llama, specs.AdapterSpec(esm.hidden_size, 64, llama.hidden_size, 0.0)


def _host_batches(n, B=4, Tp=48, Tt=16):
    out = []
    for i in range(n):
        pid, pmask = synth.protein_batch(300 + i, B, Tp, [48, 31, 12, 5])
        tid, tmask = synth.text_batch(300 + i, B, Tt, 500, [16, 9, 4, 2], 511, 510)
        out.append({k: torch.from_numpy(v) for k, v in dict(protein_input_ids=pid, protein_attention_mask=pmask,
                                                            description_input_ids=tid, description_attention_mask=tmask).items()})
    return out


def _trainer(ga=1, seed=3, **kw):
    import p2t_hip as P
    esm, llama, ad = _specs()
    model = build_model(esm, llama, ad, torch.float32, seed)
    return P.ContrastiveTrainer(model, num_segments=2, output_llm_layer=2, train_mode=False, lr=1e-3, gradient_accumulation_steps=ga, **kw)


def test_epoch_sums_on_the_device_equal_per_batch_host_reads():
    from p2t_hip import loop
    batches = _host_batches(5)
    tr, twin = _trainer(ga=2), _trainer(ga=2)
    losses, norms = [], []
    for b in batches:                                   # the reference's way: .item() after every batch
        before = twin.step_count
        losses.append(float(to_np(twin.step({k: v.cuda() for k, v in b.items()}))[0]))
        if twin.step_count > before:
            norms.append(float(to_np(twin.grad_norm)[0]))
    lines = []
    rec = loop.train_epoch(tr, batches, rank=0, current_epoch=2, num_epochs=5, check_every=2, log=lines.append)
    assert rec["batches"] == 5 and rec["optimizer_steps"] == 2 and rec["impossible_batches"] == 0
    assert abs(rec["train_loss"] - np.mean(losses)) < 1e-5 * max(1.0, abs(np.mean(losses)))
    assert abs(rec["epoch_gradnorm"] - np.mean(norms)) < 1e-5 * max(1.0, abs(np.mean(norms)))
    assert lines == [f"[epoch=2/5, train_loss={rec['train_loss']}, epoch_lr=0.001, epoch_gradnorm={rec['epoch_gradnorm']}]"]
    # eval: forward-only, no dropout, parameters untouched
    p_before = tr.flat_p.clone()
    ev = loop.eval_epoch(tr, batches[:3], rank=0, current_epoch=2, num_epochs=5, log=lines.append)
    want = np.mean([float(to_np(twin.evaluate({k: v.cuda() for k, v in b.items()}))[0]) for b in batches[:3]])
    assert abs(ev["eval_loss"] - want) < 1e-5 * max(1.0, abs(want)) and torch.equal(tr.flat_p, p_before)
    assert lines[-1] == f"[epoch=2/5, eval_loss={ev['eval_loss']}]"


def test_impossible_loss_is_reported_and_a_nan_epoch_aborts():
    from p2t_hip import loop
    tr = _trainer()
    batches = _host_batches(3)
    lines = []
    # a NaN in the adapter weights from batch 1 on: every later loss is NaN (the epoch loop fetches one batch ahead -- batch 2 is
    # drawn from the loader before step 1 is enqueued -- so that is where the generator plants it)
    def poisoned():
        for i, b in enumerate(batches):
            if i == 2:
                tr.w1.fill_(float("nan"))
            yield b
    with pytest.raises(ValueError, match="NaN detected in the training loss of the epoch, training interrupted."):
        loop.train_epoch(tr, poisoned(), log=lines.append, check_every=1)
    assert lines[0].startswith("Impossible batch_loss detected at batch 1: nan")
    assert any(l.startswith("[epoch=1/1, train_loss=nan") for l in lines)                 # the summary is printed before the abort, as upstream
    # the device kernel on its own: <= 0 and inf count as impossible too (train_contrast.py:433), the first one is remembered
    st = loop.EpochStats(dev())
    for i, v in enumerate([1.5, 0.0, float("inf"), -2.0, 0.25]):
        st.add(torch.tensor([v], device=dev()), torch.tensor([2.0], device=dev()) if i % 2 else None, i)
    assert st.flags.cpu().tolist()[:2] == [3, 1] and st.sums.cpu().tolist()[1:] == [5.0, 4.0, 2.0]
    assert math.isinf(st.sums.cpu().tolist()[0])
    out = []
    st.check(out.append)
    assert out == ["Impossible batch_loss detected at batch 1: 0.0 (and 2 more since)"]


def test_split_k_time_out_word_is_sticky_poisons_consumers_and_raises():
    from p2t_hip import _lib, loop, ops
    from p2t_hip.ops import stream
    assert _lib.fault_status() == 0
    # a shape the default policy runs with in-kernel split-K pairs (1.25 rounds of tiles, K = 6144)
    M, N, K = 8192, 2560, 6144
    a = torch.empty((M, K), dtype=torch.bfloat16, device=dev())
    w = torch.empty((N, K), dtype=torch.bfloat16, device=dev())
    ops.fill_hash_(a, 5, "flt.a", 1.0)
    ops.fill_hash_(w, 5, "flt.w", 0.5)
    ws = ops.gemm_fix_workspace(dev())
    clean = ops.gemm_nt(a, w, None, out_dtype=torch.float32, use_mfma=1, fix_ws=ws, fix_epoch=1)
    assert bool(torch.isfinite(clean).all())
    try:
        _lib.call("p2t_fault_inject", 1, stream())
        poisoned = ops.gemm_nt(a, w, None, out_dtype=torch.float32, use_mfma=1, fix_ws=ws, fix_epoch=2)
        bad = int(torch.isnan(poisoned[:, :N]).sum())
        assert bad > 0 and bad % (256 * 256) == 0                    # whole tiles of the split-K pairs are NaN, the others are not
        assert _lib.fault_status() == 1                              # ... and nothing re-zeroed the word (the towers' forward used to)
        tr = _trainer()
        with pytest.raises(loop.SplitKTimeout):
            loop.train_epoch(tr, _host_batches(2), log=lambda s: None, check_every=1)
    finally:
        assert _lib.fault_status(clear=True) == 1
    assert _lib.fault_status() == 0
    again = ops.gemm_nt(a, w, None, out_dtype=torch.float32, use_mfma=1, fix_ws=ws, fix_epoch=3)
    assert torch.equal(again, clean)


def test_run_epochs_steps_the_schedule_once_per_epoch_and_writes_the_reference_checkpoints(tmp_path):
    from p2t_hip import loop
    from p2t_hip import training_state as ts
    sched = ts.CosineWarmupSchedule(1e-3, 1, 3)
    tr = _trainer(schedule=sched)
    hist = loop.run_epochs(tr, _host_batches(2), _host_batches(1), num_epochs=3, checkpoint_dir=str(tmp_path), log=lambda s: None)
    assert [h["epoch"] for h in hist] == [1, 2, 3] and sched.last_epoch == 3 and tr.step_count == 6
    assert hist[0]["epoch_lr"] == 0.0 and hist[1]["epoch_lr"] == pytest.approx(1e-3)      # factor(0) = 0 through epoch 1, as upstream
    assert all(np.isfinite(h["train_loss"]) and np.isfinite(h["eval_loss"]) for h in hist)
    for e in (1, 2, 3):
        assert os.path.exists(tmp_path / f"model_checkpoint_{e}.pt") and os.path.exists(tmp_path / f"optimizer_scheduler_checkpoint_{e}.pt")
    sd = torch.load(tmp_path / "optimizer_scheduler_checkpoint_3.pt", weights_only=True)
    assert sd["scheduler_state_dict"]["last_epoch"] == 3


WORKER = r'''
import json, os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.path.join(ROOT, "prot2text-v2-esm3_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import p2t_hip as P
from p2t_hip import loop, specs, synth
from gpu_util import build_model
from test_gpu_loop import _specs
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
if world > 1:
    dist.init_process_group("gloo", rank=rank, world_size=world)
torch.cuda.set_device(0)
esm, llama, ad = _specs()
model = build_model(esm, llama, ad, torch.float32, 3)
B, Tp, Tt = 8, 48, 16
tr = P.ContrastiveTrainer(model, num_segments=1 if world > 1 else 2, output_llm_layer=2, train_mode=False, lr=1e-3,
                          overlap_streams=os.environ.get("P2T_TEST_OVERLAP") == "1")    # (True: towers on side streams, epoch loop pipelined)
sl = slice(rank * (B // world), (rank + 1) * (B // world))
batches = []
for i in range(3):
    pid, pmask = synth.protein_batch(400 + i, B, Tp, [48, 40, 33, 21, 17, 9, 5, 48])
    tid, tmask = synth.text_batch(400 + i, B, Tt, 500, [16, 12, 9, 7, 5, 3, 2, 16], 511, 510)
    batches.append({k: torch.from_numpy(np.ascontiguousarray(v[sl])) for k, v in dict(protein_input_ids=pid, protein_attention_mask=pmask,
                                                                                   description_input_ids=tid, description_attention_mask=tmask).items()})
lines = []
rec = loop.train_epoch(tr, batches, rank=rank, log=lines.append, check_every=2)
ev = loop.eval_epoch(tr, batches, rank=rank, log=lines.append)
gl = float(tr.global_loss().cpu()[0])               # last step's loss, averaged over ranks
params = tr.flat_p.detach().cpu().numpy()[::53].astype(float).tolist()
# second epoch with a NaN on rank (world - 1) only: the epoch must abort on EVERY rank
if rank == world - 1:
    tr.w1.fill_(float("nan"))
try:
    loop.train_epoch(tr, batches[:1], rank=rank, log=lines.append)
    aborted = False
except ValueError:
    aborted = True
json.dump({"train_loss": rec["train_loss"], "eval_loss": ev["eval_loss"], "batches": rec["batches"], "global_last": gl, "aborted": aborted,
           "lines": lines, "p": params}, open(os.environ["P2T_TEST_OUT"] + str(rank), "w"))
if world > 1:
    dist.barrier()
    dist.destroy_process_group()
'''


def _run(world, tmp_path, overlap=False):
    out = str(tmp_path / f"loop_w{world}_o{int(overlap)}_r")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   P2T_TEST_OUT=out, HSA_ENABLE_IPC_MODE_LEGACY="0", P2T_TEST_OVERLAP="1" if overlap else "0")
        procs.append(subprocess.Popen([sys.executable, "-c", f"ROOT = {ROOT!r}\n" + WORKER], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = [p.communicate(timeout=300)[0].decode(errors="replace") for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)[-3000:]
    return [json.load(open(out + str(r))) for r in range(world)]


@pytest.mark.parametrize("overlap", [False, True])
def test_two_ranks_epoch_on_one_gpu_equals_the_single_process_epoch(tmp_path, overlap):
    """Two ranks of train_epoch / eval_epoch with the real trainer on the one GPU (gloo): rank r encodes rows [4 r, 4 r + 4) of every
    batch; the all-reduced epoch loss, the parameters after the epoch and the rank-averaged last loss equal the single-process run on
    the whole batches with two segments; a NaN on one rank aborts the epoch on both.  overlap: the towers on side streams and the
    epoch loop pipelined through them (next batch's towers beside this step's backward, all-reduce and optimizer)."""
    two, one = _run(2, tmp_path, overlap), _run(1, tmp_path, overlap)[0]
    for r in two:
        assert r["batches"] == 6.0 and abs(r["train_loss"] - one["train_loss"]) < 2e-5 and abs(r["eval_loss"] - one["eval_loss"]) < 2e-5
        assert abs(r["global_last"] - one["global_last"]) < 2e-5 and r["aborted"]
        assert np.linalg.norm(np.array(r["p"]) - np.array(one["p"])) <= 1e-5 * np.linalg.norm(one["p"])
    assert one["batches"] == 3.0 and one["aborted"]
    assert sum(l.startswith("[epoch=1/1, train_loss=") for l in two[0]["lines"]) == 2 and not any(l.startswith("[epoch") for l in two[1]["lines"])
    assert any(l.startswith("Impossible batch_loss") for l in two[1]["lines"]) and not any(l.startswith("Impossible") for l in two[0]["lines"])


def test_steps_pipelined_through_the_frozen_towers_equal_the_plain_sequence():
    """`step(batch, next_batch)`: the encoder / text tower of the next batch are enqueued on the side streams beside this step's
    backward + AdamW tail (they depend on nothing the optimizer writes).  Same kernels, same inputs: the losses and the parameters
    after four steps equal the plain sequence bit for bit -- also when the announced batch is NOT the one that follows (the
    prefetched towers are dropped), with gradient accumulation, and with the towers' default stream layout off (a no-op)."""
    to_cuda = lambda b: {k: v.to(dev()) for k, v in b.items()}
    batches = [to_cuda(b) for b in _host_batches(4)]
    ref = _trainer(overlap_streams=True)
    want = [float(to_np(ref.step(b))[0]) for b in batches]
    want_p = to_np(ref.flat_p).copy()
    for ga, kw in ((1, dict(overlap_streams=True)), (2, dict(overlap_streams=True)), (1, dict(overlap_streams=False))):
        plain, piped = _trainer(ga=ga, **kw), _trainer(ga=ga, **kw)
        got_plain = [float(to_np(plain.step(b))[0]) for b in batches]
        got = []
        for i, b in enumerate(batches):
            nxt = batches[i + 1] if i + 1 < len(batches) else None
            if i == 1:
                nxt = batches[3]                        # announced, but batches[2] comes next: the prefetch must be discarded
            got.append(float(to_np(piped.step(b, next_batch=nxt))[0]))
        assert got == got_plain and np.array_equal(to_np(piped.flat_p), to_np(plain.flat_p))
        if ga == 1:
            assert got == want and np.array_equal(to_np(piped.flat_p), want_p)
        assert piped._pf is None
    # the epoch driver looks one batch ahead by itself when the trainer runs its towers on side streams
    import p2t_hip as P
    ep = _trainer(overlap_streams=True)
    rec = P.train_epoch(ep, _host_batches(4), log=lambda *_: None)
    assert rec["batches"] == 4 and np.array_equal(to_np(ep.flat_p), want_p)
    assert abs(rec["train_loss"] - float(np.mean(np.float32(want)))) < 1e-6
