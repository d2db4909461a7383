"""Two and four ranks of the REAL trainer on one GPU (all processes on cuda:0, torch.distributed over gloo -- RCCL refuses two ranks
on one device, and the pool gives one GPU; its process guard allows at most 6 processes on the card, so the 8-rank case of
BASELINE.json's target is covered with the oracle in place of the kernels by tests/test_distributed_gloo.py, world 8, and here by
world 4: ranks r >= 2 already exercise the label offsets r * B_loc, the gather order and the AVG all-reduce beyond a swap): ContrastiveTrainer.forward_backward / .step with its HIP kernels AND its
collectives (text all-gather, label offsets, flat-gradient average, one exchange per accumulation window), against the
single-process formulation on the concatenated global batch with contrastive_num_segments = 2 (rank == segment,
scripts/train_contrast.py:356-379).  What the CPU gloo test (tests/test_distributed_gloo.py) drives with the oracle in
place of the kernels, this drives end to end; only the transport (gloo instead of RCCL over xGMI) differs from an 8-GPU run.
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import json, os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.path.join(ROOT, "prot2text-v2-esm3_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import p2t_hip as P
from p2t_hip import specs, synth
from gpu_util import build_model
rank, world, mode = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), os.environ["P2T_TEST_MODE"]
if world > 1:
    dist.init_process_group("gloo", rank=rank, world_size=world)
torch.cuda.set_device(0)
esm = specs.EsmSpec(num_hidden_layers=2, hidden_size=128, intermediate_size=256, num_attention_heads=2)        # head_dim 64: fused QKV + RoPE
llama = specs.LlamaSpec(num_hidden_layers=2, hidden_size=128, intermediate_size=256, num_attention_heads=2, num_key_value_heads=1, vocab_size=512)
ad = specs.AdapterSpec(esm.hidden_size, 64, llama.hidden_size, 0.0)
B, Tp, Tt = 8, 96, 24
pid, pmask = synth.protein_batch(7, B, Tp, [96, 90, 64, 33, 80, 17, 50, 96])
tid, tmask = synth.text_batch(7, B, Tt, 500, [24, 20, 9, 3, 16, 24, 5, 11], 511, 510)
model = build_model(esm, llama, ad, torch.float32, 3)
ga = 2 if mode == "ga2" else 1
cw = 0.5 if mode == "column" else 0.0
if world > 1:
    sl = slice(rank * (B // world), (rank + 1) * (B // world))
    tr = P.ContrastiveTrainer(model, num_segments=1, output_llm_layer=2, train_mode=False, lr=1e-3, global_negatives=(mode != "local"),
                              gradient_accumulation_steps=ga, column_weight=cw)
else:
    sl = slice(0, B)
    tr = P.ContrastiveTrainer(model, num_segments=int(os.environ.get("P2T_TEST_SEGMENTS", "2")), output_llm_layer=2, train_mode=False, lr=1e-3,
                              gradient_accumulation_steps=ga, column_weight=cw)
def batch(perm=None):
    idx = np.arange(B) if perm is None else perm
    f = lambda a: torch.from_numpy(np.ascontiguousarray(a[idx][sl] if world == 1 else a[idx][sl])).cuda()
    return dict(protein_input_ids=f(pid), protein_attention_mask=f(pmask), description_input_ids=f(tid), description_attention_mask=f(tmask))
losses = []
if ga == 1:
    losses.append(float(tr.forward_backward(batch()).cpu()[0]))
else:
    # window of two micro-batches: rows permuted WITHIN each rank's share so both formulations see the same shares
    nsh = max(world, int(os.environ.get("P2T_TEST_SEGMENTS", "2")))
    perm = np.concatenate([s0 * (B // nsh) + np.arange(B // nsh)[::-1] for s0 in range(nsh)])
    losses.append(float(tr.step(batch()).cpu()[0]))
    losses.append(float(tr.step(batch(perm)).cpu()[0]))
g = tr.flat_g.detach().cpu().numpy().astype(np.float64)
p = tr.flat_p.detach().cpu().numpy().astype(np.float64) if hasattr(tr, "flat_p") else np.zeros(1)
if world > 1:
    t = torch.tensor(losses, dtype=torch.float64)
    dist.all_reduce(t)                       # mean over ranks = the global loss (equal shares)
    losses = (t / world).tolist()
if rank == 0:
    json.dump({"loss": losses, "g_norm": float(np.linalg.norm(g)), "g": g[::97].tolist(), "p": p[::97].tolist()}, open(os.environ["P2T_TEST_OUT"], "w"))
if world > 1:
    dist.barrier()
    dist.destroy_process_group()
'''


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(world, mode, tmp_path, segments=2):
    out = str(tmp_path / f"w{world}_{mode}_{segments}.json")
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   P2T_TEST_MODE=mode, P2T_TEST_OUT=out, P2T_TEST_SEGMENTS=str(segments), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, "-c", f"ROOT = {ROOT!r}\n" + WORKER], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = [p.communicate(timeout=300)[0].decode(errors="replace") for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)[-3000:]
    return json.load(open(out))


@pytest.mark.parametrize("mode", ["global", "ga2", "column"])
def test_two_ranks_on_one_gpu_equal_the_single_process_segmented_step(mode, tmp_path):
    two = _run(2, mode, tmp_path)
    one = _run(1, mode, tmp_path)
    assert np.allclose(two["loss"], one["loss"], rtol=2e-5, atol=2e-6), (two["loss"], one["loss"])
    g2, g1 = np.array(two["g"]), np.array(one["g"])
    assert np.linalg.norm(g2 - g1) <= 2e-4 * max(np.linalg.norm(g1), 1e-12), (np.linalg.norm(g2 - g1), np.linalg.norm(g1))
    if mode == "ga2":                          # the optimizer ran once at the end of the window: same parameters
        p2, p1 = np.array(two["p"]), np.array(one["p"])
        assert np.linalg.norm(p2 - p1) <= 1e-5 * max(np.linalg.norm(p1), 1e-12)


@pytest.mark.parametrize("mode", ["global", "ga2"])
def test_four_ranks_on_one_gpu_equal_the_single_process_four_segment_step(mode, tmp_path):
    """World 4 (two pairs per rank): rank r's labels start at r * B_loc, the gathered text rows are in rank order, the flat gradient
    is averaged over four ranks -- against ONE process with contrastive_num_segments = 4 on the same batch."""
    four = _run(4, mode, tmp_path)
    one = _run(1, mode, tmp_path, segments=4)
    assert np.allclose(four["loss"], one["loss"], rtol=2e-5, atol=2e-6), (four["loss"], one["loss"])
    g4, g1 = np.array(four["g"]), np.array(one["g"])
    assert np.linalg.norm(g4 - g1) <= 2e-4 * max(np.linalg.norm(g1), 1e-12), (np.linalg.norm(g4 - g1), np.linalg.norm(g1))
    if mode == "ga2":
        p4, p1 = np.array(four["p"]), np.array(one["p"])
        assert np.linalg.norm(p4 - p1) <= 1e-5 * max(np.linalg.norm(p1), 1e-12)


def test_bench_py_four_ranks_rehearsal_on_one_gpu():
    """`python bench.py --gpus 4 --rehearse-shared-gpu --config cfg1`: bench.py's own launcher and rank logic at four ranks (the most
    this box's process guard leaves room for): one JSON line, n_gpus 4, global batch 4 x 4, a finite loss."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--rehearse-shared-gpu", "--config", "cfg1", "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline", "--no-batch64-check"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 4 and d["scaling"] == "weak" and d["value"] > 0 and "REHEARSAL" in d["data"]
    assert d["config"]["global_batch"] == 4 * 4 and "dp4 over gloo" in d["config"]["parallelism"]
    assert np.isfinite(d["config"]["loss"])


def test_bench_py_two_ranks_rehearsal_on_one_gpu():
    """`python bench.py --gpus 2 --rehearse-shared-gpu`: bench.py's OWN multi-rank path (self-launch of the ranks, rank setup, barriers,
    max-over-ranks timing, rank 0's JSON line, the trainer's collectives) with both ranks on the one GPU over gloo."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse-shared-gpu", "--config", "cfg2", "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline", "--no-batch64-check"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0 and "REHEARSAL" in d["data"]
    assert d["config"]["global_batch"] == 2 * 32 and "gloo" in d["config"]["parallelism"]
    assert np.isfinite(d["config"]["loss"])


RCCL_WORKER = r'''
import os, sys
import torch, torch.distributed as dist
sys.path.insert(0, os.path.join(ROOT, "prot2text-v2-esm3_amd"))
from p2t_hip import sharding
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)          # bench.py's call for N > 1
assert dist.get_backend() == "nccl"
x = torch.arange(8 * 16, dtype=torch.float32, device=dev).view(8, 16)
out = torch.empty_like(x)
w = dist.all_gather_into_tensor(out, x, async_op=True)        # sharding.gather_rows_async's collective
w.wait()
g = torch.full((1 << 20,), 3.0, device=dev)
dist.all_reduce(g, op=dist.ReduceOp.AVG)                      # sharding.average_gradients' collective (RCCL averages in the collective)
t = torch.tensor([1.5, 2.0], dtype=torch.float64, device=dev)
dist.all_reduce(t)                                            # the epoch loop's [loss sum, batches] reduction
dist.barrier()
torch.cuda.synchronize()
assert torch.equal(out, x) and float(g[0]) == 3.0 and t.tolist() == [1.5, 2.0]
# the product's own wrappers on the initialised group (world 1: they must short-circuit, not call into RCCL with one rank)
rows, first = sharding.gather_rows_async(x).wait()
assert first == 0 and rows.data_ptr() == x.data_ptr()
assert sharding.average_gradients(g).data_ptr() == g.data_ptr()
dist.destroy_process_group()
print("rccl world-1 ok")
'''


def test_rccl_backend_initialises_and_runs_the_trainers_collectives_at_world_1():
    """The one RCCL configuration a one-GPU box allows: `init_process_group("nccl", device_id=...)` exactly as bench.py calls it for N > 1,
    then the three collectives the trainer issues (async all_gather_into_tensor, all_reduce AVG, all_reduce SUM of float64) and a barrier on
    GPU tensors.  It cannot show scaling or a multi-rank exchange; it shows that this image's RCCL accepts the calls the N > 1 path makes."""
    port = _free_port()
    env = dict({k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", f"ROOT = {ROOT!r}\n" + RCCL_WORKER], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0 and "rccl world-1 ok" in r.stdout, r.stdout[-3000:]
