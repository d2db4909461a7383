"""CPU, world_size 2 over gloo: the sharding logic of the multi-GPU step.

Each rank encodes its own slice with the numpy oracle, the text embeddings go through the product's
all-gather helper (p2t_hip.contrastive._gather_text), each rank scores only its rows with labels
offset by rank * B_loc, and the mean over ranks of loss / gradients must equal the reference
formulation run in ONE process on the concatenated global batch with contrastive_num_segments = 2
(rank == segment, scripts/train_contrast.py:356-379)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import case_setup, model_weights
from conftest import load_golden


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import p2t_oracle as O
        from p2t_hip.contrastive import _gather_text
        meta = load_golden("tiny")["meta"]
        esm, llama, ad, pid, pmask, tid, tmask = case_setup(meta)
        W = model_weights(esm, llama, ad, meta["seed_w"])
        B = pid.shape[0]
        bl = B // world
        sl = slice(rank * bl, (rank + 1) * bl)
        k = meta["layers"][-1]
        t_loc = O.text_embeddings(llama, W, tid[sl], tmask[sl], k, "mix")
        t_all, offset = _gather_text(torch.from_numpy(t_loc))
        assert offset == rank * bl and tuple(t_all.shape) == (B, t_loc.shape[1])
        keep = {}
        p_loc = O.protein_embeddings(esm, W, pid[sl], pmask[sl], "mix", keep=keep)
        labels = np.arange(bl) + offset
        loss, dp, _ = O.infonce_segmented(p_loc, t_all.numpy(), labels, return_grad=True)
        dpooled = O.l2_normalize_backward(keep["pooled"], dp)
        dad = O.readout_backward(keep["adapter_out"], keep["rmask"], "mix", dpooled)
        grads = O.adapter_backward(W, keep, dad, prefix="adapter.")
        flat = torch.from_numpy(np.concatenate([grads[n].ravel() for n in sorted(grads)] + [np.array([loss], np.float32)]))
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)        # one flat exchange per step, then average
        flat /= world
        if rank == 0:
            q.put((flat.numpy(), t_all.numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_loss_and_grads_equal_single_process_segments():
    from oracle import p2t_oracle as O
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    flat, t_all = q.get()
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    g = load_golden("tiny")
    meta = g["meta"]
    esm, llama, ad, pid, pmask, tid, tmask = case_setup(meta)
    W = model_weights(esm, llama, ad, meta["seed_w"])
    k = meta["layers"][-1]
    ref = O.contrastive_step(esm, llama, W, pid, pmask, tid, tmask, layer=k, num_segments=2, with_grads=True)
    np.testing.assert_allclose(t_all, ref["text"], rtol=1e-6, atol=1e-7)            # gather order = rank order
    names = sorted(ref["grads"])
    ref_flat = np.concatenate([ref["grads"][n].ravel() for n in names] + [np.array([ref["loss"]], np.float32)])
    # fp32 BLAS sums differ in order between a 2-row and a 4-row batch: compare in relative L2
    assert np.linalg.norm(flat - ref_flat) < 1e-5 * np.linalg.norm(ref_flat)
    np.testing.assert_allclose(flat, ref_flat, rtol=1e-3, atol=1e-6)
    # and the single-process value is the reference's own (golden from scripts/train_contrast.py)
    assert abs(float(flat[-1]) - float(g[f"loss_seg2_mix_L{k}"])) < 2e-5
