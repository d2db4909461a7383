"""CPU, world_size 2 over gloo: the rank logic of the multi-GPU step, driven through the PRODUCT's
own code -- p2t_hip.sharding.{local_rows, sharded_forward_backward, gather_rows, segment_labels, average_gradients,
micro_step_plan}, the functions ContrastiveTrainer.forward_backward / .step run over RCCL on the GPUs.  Only the arithmetic
between the collectives is swapped: the numpy oracle stands where the HIP kernels are (there is no GPU here), exactly as
the trainer passes kernel closures.

Checked against the reference formulation run in ONE process on the concatenated global batch with
contrastive_num_segments = world (rank == segment, scripts/train_contrast.py:356-379):
  * global_negatives=True : N-rank mean loss / averaged gradients == single-process segmented loss / gradients;
  * global_negatives=False: each rank's loss == the reference loss on that rank's own slice (the reference's per-rank
    negatives), gradients == mean of the per-slice gradients;
  * gradient accumulation (GA = 2): gradients of loss / GA summed over the window, ONE cross-rank average at its end;
  * column term (column_weight = 0.5): protein embeddings all-gathered too, == single-process symmetric loss / gradients.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import case_setup, model_weights
from conftest import load_golden

NAMES = ("adapter.fc1.weight", "adapter.fc1.bias", "adapter.fc2.weight", "adapter.fc2.bias")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _micro_batch(meta, micro):
    """Micro-batch 0 is the golden batch; micro-batch 1 the same pairs in another order (a different batch for GA)."""
    esm, llama, ad, pid, pmask, tid, tmask = case_setup(meta)
    if micro:
        perm = np.array([2, 0, 3, 1])
        pid, pmask, tid, tmask = pid[perm], pmask[perm], tid[perm], tmask[perm]
    return esm, llama, ad, pid, pmask, tid, tmask


class OracleRank:
    """What ContrastiveTrainer is on a GPU, with the oracle in place of the kernels: owns the flat gradient buffer and runs
    sharding.sharded_forward_backward with closures over its LOCAL rows."""

    def __init__(self, meta, rank, world, global_negatives, column_weight=0.0, num_segments=1):
        from oracle import p2t_oracle as O
        self.O, self.meta, self.rank, self.world = O, meta, rank, world
        self.global_negatives, self.cw, self.nseg = global_negatives, column_weight, num_segments
        esm, llama, ad, *_ = case_setup(meta)
        self.W = model_weights(esm, llama, ad, meta["seed_w"])
        n = sum(int(np.prod(self.W[k].shape)) for k in NAMES)
        self.flat_g = torch.zeros((n,), dtype=torch.float32)
        self.loss = 0.0
        self.prefetched = 0

    def forward_backward(self, micro, accumulate, grad_scale, reduce):
        from p2t_hip import sharding
        O, meta, cw = self.O, self.meta, self.cw
        esm, llama, ad, pid, pmask, tid, tmask = _micro_batch(meta, micro)
        rows = sharding.local_rows(self.rank, self.world, pid.shape[0])
        pid, pmask, tid, tmask = pid[rows], pmask[rows], tid[rows], tmask[rows]
        B = pid.shape[0]
        k = meta["layers"][-1]
        bs = B // self.nseg
        segments = [(s * bs, (s + 1) * bs, pid.shape[1], 1.0 / self.nseg) for s in range(self.nseg)]
        keeps, state = {}, {"loss": 0.0, "col_sm": None}
        if not accumulate:
            self.flat_g.zero_()

        def text_fn():
            return torch.from_numpy(O.text_embeddings(llama, self.W, tid, tmask, k, "mix"))

        def seg_forward(s, r0, r1):
            keep = {}
            p = O.protein_embeddings(esm, self.W, pid[r0:r1], pmask[r0:r1], "mix", keep=keep)
            keeps[s] = (keep, p)
            return p

        def protein_fn():
            return torch.from_numpy(np.concatenate([seg_forward(s, r0, r1) for s, (r0, r1, _, _) in enumerate(segments)], 0))

        def column_fn(p_all, t_all, offset):
            cols = np.arange(offset, offset + B)
            loss_c, g_all = O.infonce_columns(p_all.numpy(), t_all.numpy(), cols, return_grad=True)
            state["loss"] += cw * float(loss_c)
            state["col_grad"] = g_all          # d sum_j(col_j) / dP over ALL columns, [N, D]

        def segment_fn(s, r0, r1, T, weight, t_all, labels, offset):
            if s not in keeps:
                seg_forward(s, r0, r1)
            keep, p = keeps.pop(s)
            loss, dp, _ = O.infonce_segmented(p, t_all.numpy(), labels.numpy(), return_grad=True)
            state["loss"] += weight * (1.0 - cw) * float(loss)
            dp = dp * np.float32(weight * (1.0 - cw) * grad_scale)
            if cw > 0:                       # per row: cw * weight / n_s = cw / B_loc (contrastive.py, segment_fn)
                dp = dp + state["col_grad"][offset + r0:offset + r1] * np.float32(cw * weight * grad_scale / (r1 - r0))
            dpooled = O.l2_normalize_backward(keep["pooled"], dp)
            dad = O.readout_backward(keep["adapter_out"], keep["rmask"], "mix", dpooled)
            grads = O.adapter_backward(self.W, keep, dad, prefix="adapter.")
            self.flat_g += torch.from_numpy(np.concatenate([grads[n].ravel() for n in NAMES]))

        def prefetch_fn(s, r0, r1, T):          # the trainer's: first segment's encoder before the wait for the gathered text
            self.prefetched += 1
            seg_forward(s, r0, r1)

        sharding.sharded_forward_backward(text_fn=text_fn, segment_fn=segment_fn, segments=segments,
                                          global_negatives=self.global_negatives, flat_g=self.flat_g, backward=True,
                                          reduce=reduce, protein_fn=protein_fn if cw > 0 else None,
                                          column_fn=column_fn if cw > 0 else None,
                                          prefetch_fn=prefetch_fn if (cw == 0 and self.global_negatives) else None)
        self.loss = state["loss"]
        return self.loss

    def step(self, micro, ga):
        """ContrastiveTrainer.step: the accumulation window of sharding.micro_step_plan."""
        from p2t_hip import sharding
        accumulate, grad_scale, reduce, do_step = sharding.micro_step_plan(micro, ga)
        return self.forward_backward(micro, accumulate, grad_scale, reduce), do_step


def _worker(rank, world, port, q, mode):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from p2t_hip import sharding
        meta = load_golden("tiny")["meta"]
        assert sharding.world_info() == (rank, world)
        if mode == "ga2":
            r = OracleRank(meta, rank, world, True)
            losses = []
            for micro in range(2):
                loss, do_step = r.step(micro, 2)
                losses.append(loss)
                assert do_step == (micro == 1)
                if micro == 0:                  # no exchange yet: the buffer holds this rank's own half-weighted gradients
                    local = r.flat_g.clone()
            out = (losses, r.flat_g.numpy().copy(), local.numpy())
        elif mode.startswith("epoch"):
            out = _epoch_worker(meta, rank, world, mode)
        else:
            r = OracleRank(meta, rank, world, global_negatives=(mode != "local"), column_weight=0.5 if mode == "column" else 0.0,
                           num_segments=2 if mode == "column" else 1)
            loss = r.forward_backward(0, False, 1.0, True)
            assert r.prefetched == (1 if mode == "global" else 0)      # the encoder went out before the wait for the gather
            out = ([loss], r.flat_g.numpy().copy(), None)
        gathered = [None] * world
        dist.all_gather_object(gathered, out)
        if rank == 0:
            q.put(gathered)
    finally:
        dist.destroy_process_group()


class _HostStats:
    """p2t_epoch_accumulate's arithmetic on host tensors (csrc/misc.hip epoch_accumulate_kernel): stands where the kernel is,
    like the oracle stands where the tower kernels are -- the epoch logic around it is the product's loop.train_epoch."""

    def __new__(cls):
        from p2t_hip import loop

        class HostStats(loop.EpochStats):
            def __init__(self):
                self.sums = torch.zeros(4)
                self.flags = torch.tensor([0, -1, 0, 0], dtype=torch.int32)
                self._reported = 0
                self.fault = 0

            def add(self, loss, grad_norm, batch_idx):
                l = float(loss)
                self.sums[0] += l
                self.sums[1] += 1
                if grad_norm is not None:
                    self.sums[2] += float(grad_norm)
                    self.sums[3] += 1
                if not (l > 0.0) or l == float("inf"):
                    if int(self.flags[0]) == 0:
                        self.flags[1] = batch_idx
                        self.flags[3] = int(np.float32(l).view(np.int32))
                    self.flags[0] += 1
                self.flags[2] |= self.fault
        return HostStats()


class _EpochTrainer:
    """The attributes loop.train_epoch / eval_epoch use of a ContrastiveTrainer, over an OracleRank."""

    def __init__(self, meta, rank, world, ga, poison_at=None):
        self.r = OracleRank(meta, rank, world, True)
        self.gradient_accumulation_steps, self._micro, self.step_count = ga, 0, 0
        self.dev, self.group, self.schedule, self.hp, self.train_mode = "cpu", None, None, {"lr": 2e-4}, False
        self.grad_norm = torch.zeros(1)
        self.model = torch.nn.Identity()
        self.poison_at, self.seen = poison_at, 0

    def step(self, batch):
        from p2t_hip import sharding
        accumulate, grad_scale, reduce, do_step = sharding.micro_step_plan(self._micro, self.gradient_accumulation_steps)
        loss = self.r.forward_backward(batch["micro"], accumulate, grad_scale, reduce)
        if self.poison_at is not None and self.seen == self.poison_at:
            loss = float("nan")
        self.seen += 1
        self._micro += 1
        if do_step:
            self.step_count += 1
            self.grad_norm = self.r.flat_g.norm().reshape(1)
            self._micro = 0
        return torch.tensor([loss], dtype=torch.float32)

    def evaluate(self, batch):
        return torch.tensor([self.r.forward_backward(batch["micro"], False, 1.0, False)], dtype=torch.float32)


def _epoch_worker(meta, rank, world, mode):
    """loop.train_epoch / eval_epoch on two ranks: device sums -> ONE all-reduce -> the reference's epoch summary."""
    from p2t_hip import loop
    lines = []
    batches = [{"micro": 0}, {"micro": 1}]
    if mode == "epoch":
        tr = _EpochTrainer(meta, rank, world, ga=2)
        rec = loop.train_epoch(tr, batches, rank=rank, current_epoch=1, num_epochs=3, check_every=1, log=lines.append, stats=_HostStats())
        ev = loop.eval_epoch(tr, batches, rank=rank, current_epoch=1, num_epochs=3, log=lines.append, stats=_HostStats())
        return ([rec["train_loss"], ev["eval_loss"], rec["epoch_gradnorm"], rec["batches"], rec["optimizer_steps"]], np.zeros(1), lines)
    if mode == "epoch_nan":             # rank 1's second batch is NaN: reported there, and the epoch aborts on EVERY rank
        tr = _EpochTrainer(meta, rank, world, ga=1, poison_at=1 if rank == 1 else None)
        try:
            loop.train_epoch(tr, batches, rank=rank, check_every=1, log=lines.append, stats=_HostStats())
            raised = False
        except ValueError as e:
            raised = "NaN detected in the training loss of the epoch" in str(e)
        return ([float(raised)], np.zeros(1), lines)
    if mode == "epoch_fault":           # the fault word set (here on both GPUs' stand-ins): the rank raises at its next check,
        tr = _EpochTrainer(meta, rank, world, ga=1)        # BEFORE the epoch's all-reduce -- a faulted rank never reports a loss
        st = _HostStats()
        st.fault = 1
        try:
            loop.eval_epoch(tr, batches[:1], rank=rank, check_every=1, log=lines.append, stats=st)
            raised = False
        except loop.SplitKTimeout:
            raised = True
        return ([float(raised)], np.zeros(1), lines)
    raise ValueError(mode)


def _run(mode, world=2):
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, mode)) for r in range(world)]
    for p in procs:
        p.start()
    res = q.get()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    return res


def _flat(grads):
    return np.concatenate([grads[n].ravel() for n in NAMES])


def _single_process(meta, micro=0, num_segments=2, column_weight=0.0, rows=None):
    from oracle import p2t_oracle as O
    esm, llama, ad, pid, pmask, tid, tmask = _micro_batch(meta, micro)
    if rows is not None:
        pid, pmask, tid, tmask = pid[rows], pmask[rows], tid[rows], tmask[rows]
    W = model_weights(esm, llama, ad, meta["seed_w"])
    return O.contrastive_step(esm, llama, W, pid, pmask, tid, tmask, layer=meta["layers"][-1], num_segments=num_segments,
                              with_grads=True, column_weight=column_weight)


def _close(a, b, tol=1e-5):
    # fp32 BLAS sums differ in order between a 2-row and a 4-row batch: compare in relative L2
    return np.linalg.norm(np.asarray(a, np.float64) - np.asarray(b, np.float64)) <= tol * max(np.linalg.norm(b), 1e-30)


@pytest.mark.timeout(300)
def test_two_ranks_global_negatives_equal_single_process_segments():
    g = load_golden("tiny")
    meta = g["meta"]
    res = _run("global")
    ref = _single_process(meta, num_segments=2)
    mean_loss = np.mean([r[0][0] for r in res])
    assert abs(mean_loss - float(ref["loss"])) < 2e-5
    for r in res:                                   # after the average every rank holds the same gradients
        assert _close(r[1], _flat(ref["grads"]))
    np.testing.assert_allclose(res[0][1], _flat(ref["grads"]), rtol=1e-3, atol=1e-6)
    # and the single-process value is the reference's own (golden from scripts/train_contrast.py)
    k = meta["layers"][-1]
    assert abs(mean_loss - float(g[f"loss_seg2_mix_L{k}"])) < 2e-5


@pytest.mark.timeout(300)
def test_two_ranks_local_negatives_reproduce_the_reference_per_rank_loss():
    meta = load_golden("tiny")["meta"]
    res = _run("local")
    per_rank = [_single_process(meta, num_segments=1, rows=slice(2 * r, 2 * r + 2)) for r in range(2)]
    for r in range(2):
        assert abs(res[r][0][0] - float(per_rank[r]["loss"])) < 2e-5          # each rank: the reference loss on its own slice
    want = 0.5 * (_flat(per_rank[0]["grads"]) + _flat(per_rank[1]["grads"]))     # DDP average
    assert _close(res[0][1], want) and _close(res[1][1], want)


@pytest.mark.timeout(300)
def test_two_ranks_gradient_accumulation_one_exchange_per_window():
    meta = load_golden("tiny")["meta"]
    res = _run("ga2")
    refs = [_single_process(meta, micro=m, num_segments=2) for m in range(2)]
    for m in range(2):
        assert abs(np.mean([r[0][m] for r in res]) - float(refs[m]["loss"])) < 2e-5      # each micro-batch logs its own loss
    want = 0.5 * (_flat(refs[0]["grads"]) + _flat(refs[1]["grads"]))                    # sum of loss / GA gradients
    assert _close(res[0][1], want) and _close(res[1][1], want)
    # after micro-batch 0 nothing had been exchanged: the two ranks' buffers differ and average to half the global gradient
    assert not _close(res[0][2], res[1][2], 1e-3)
    assert _close(0.5 * (res[0][2] + res[1][2]), 0.5 * _flat(refs[0]["grads"]))


@pytest.mark.timeout(300)
def test_two_ranks_column_term_equals_single_process_symmetric_loss():
    meta = load_golden("tiny")["meta"]
    res = _run("column")
    ref = _single_process(meta, num_segments=1, column_weight=0.5)
    assert abs(np.mean([r[0][0] for r in res]) - float(ref["loss"])) < 2e-5
    assert _close(res[0][1], _flat(ref["grads"]), 2e-5) and _close(res[1][1], _flat(ref["grads"]), 2e-5)


def test_sharding_helpers_single_process():
    from p2t_hip import sharding
    assert sharding.world_info() == (0, 1)
    assert sharding.local_rows(3, 8, 256) == slice(96, 128) and sharding.local_rows(0, 3, 10) == slice(0, 3)
    with pytest.raises(ValueError):
        sharding.local_rows(2, 2, 8)
    x = torch.arange(6.0).view(3, 2)
    y, off = sharding.gather_rows(x)
    assert y is x and off == 0
    assert sharding.segment_labels(2, 5, 8, "cpu").tolist() == [10, 11, 12]
    assert sharding.micro_step_plan(0, 1) == (False, 1.0, True, True)
    assert sharding.micro_step_plan(0, 4) == (False, 0.25, False, False) and sharding.micro_step_plan(3, 4) == (True, 0.25, True, True)
    with pytest.raises(ValueError):
        sharding.micro_step_plan(2, 2)
    g = torch.ones(4)
    assert sharding.average_gradients(g) is g and g.tolist() == [1.0] * 4


@pytest.mark.timeout(300)
def test_two_ranks_epoch_sums_are_all_reduced_once_and_guards_fire():
    """loop.train_epoch / eval_epoch (scripts/train_contrast.py:400-519) over two ranks: the epoch loss is the all-reduced sum
    over ranks and batches / the all-reduced batch count (= the mean of the single-process segmented losses of the two
    micro-batches), rank 0 prints the reference's summary lines; a NaN batch on ONE rank is reported there with its index and
    aborts the epoch on BOTH (the reduced sum is NaN everywhere); a set fault word raises SplitKTimeout."""
    meta = load_golden("tiny")["meta"]
    refs = [_single_process(meta, micro=m, num_segments=2) for m in range(2)]
    want = float(np.mean([float(r["loss"]) for r in refs]))
    res = _run("epoch")
    for r in res:
        assert abs(r[0][0] - want) < 2e-5 and abs(r[0][1] - want) < 2e-5           # train and eval loss, identical on both ranks
        assert r[0][3] == 4.0 and r[0][4] == 1.0                                    # 2 ranks x 2 batches; one optimizer step (GA = 2, local count)
    assert any(line.startswith("[epoch=1/3, train_loss=") and "epoch_gradnorm=" in line for line in res[0][2])
    assert any(line.startswith("[epoch=1/3, eval_loss=") for line in res[0][2])
    assert not res[1][2]                                                            # only rank 0 prints the summaries
    res = _run("epoch_nan")
    assert res[0][0] == [1.0] and res[1][0] == [1.0]
    assert not any("Impossible" in line for line in res[0][2])
    assert any(line.startswith("Impossible batch_loss detected at batch 1: nan") for line in res[1][2])
    res = _run("epoch_fault")
    assert res[0][0] == [1.0] and res[1][0] == [1.0]
