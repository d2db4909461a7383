#!/usr/bin/env python3
"""Headline benchmark: contrastive-step samples/s (protein-text pairs) on N MI355X.

    python bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU over RCCL.  Under a launcher (torch.distributed.run sets RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_*) this process IS one rank; without one, this process only starts the N ranks as fresh child processes -- it never
touches the GPU itself -- relays rank 0's JSON line and exits non-zero if any rank does (the reference does the same with
mp.spawn, scripts/train_contrast.py:706-718).

A step = one pass of the hot path over one synthetic batch, inputs resident in HBM: frozen text
tower (Llama layers 1..16) -> frozen ESM2 encoder -> adapter forward -> readout / normalise ->
all-gather of text embeddings (N > 1) -> InfoNCE -> adapter backward -> all-reduce of adapter grads
(N > 1) -> clip + AdamW.  Default workload = BASELINE.json configs[2]: esm2_t36_3B +
Llama-3.1-8B-Instruct shapes, bf16, 16 x 1024 residues / 16 x 128 text tokens per GPU, random-init
weights and random residues/tokens from the repo's counter-hash generator (no checkpoints or
datasets exist offline).  Weak scaling: per-GPU batch fixed, value = N * B * K / max-over-ranks time.

Prints ONE JSON line (rank 0) with the contract fields plus
  "roofline":     dominant kernel (bf16 MFMA GEMM): algorithmic FLOPs per launch / its mean launch duration,
                  measured live with HIP events on the launch stream, vs the 2.5 PFLOP/s dense bf16 peak
  "cpu_baseline": the numpy oracle (a port of the reference algorithm, fp32) timed on the host cores on a
                  bounded sample of the same workload (rank 0, N = 1 only)
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "prot2text-v2-esm3_amd"))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0          # MI355X dense bf16 MFMA (MI355X_MICROARCH.md, chip-level parameters)
PEAK_FP8_TFLOPS = 5000.0           # dense fp8 (block-scaled v_mfma_scale_f32_16x16x128_f8f6f4: 2x the bf16 rate, same guide)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="cfg3", help="cfg2 | cfg3 | cfg4 | cfg5 = fp8 tower GEMMs (SURVEY.md section 8 table)")
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch override")
    ap.add_argument("--segments", type=int, default=1, help="contrastive_num_segments")
    ap.add_argument("--eval-mode", action="store_true", help="no adapter dropout (default: train mode, p=0.3)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--rehearse-shared-gpu", action="store_true",
                    help="REHEARSAL, not a measurement: all --gpus N ranks share GPU 0 and talk over gloo (RCCL refuses two ranks on one "
                         "device) -- runs this script's own multi-rank path (launcher, rank setup, barriers, max-over-ranks timing, the "
                         "trainer's collectives) on a one-GPU box; the JSON line says so in `data`")
    ap.add_argument("--gemm-policy", type=int, default=0, help="p2t_set_gemm_policy for A/B runs (0 = the library's default; include/p2t_hip.h)")
    ap.add_argument("--event-steps", type=int, default=3,
                    help="timed steps whose MFMA launches are bracketed by HIP events for the roofline block "
                         "(default: the last 3 of the timed steps -- the event records cost ~1.5 %% of the step when on every launch; "
                         "-1 = all, 0 = none)")
    ap.add_argument("--no-overlap", action="store_true",
                    help="timed region on ONE HIP stream.  Default: the text tower and the encoder segments on separate HIP streams "
                         "(ContrastiveTrainer(overlap_streams=True): co-scheduled kernels fill each other's partial rounds, +2-3 %%); the "
                         "roofline block's per-kernel event times always come from a separate single-stream pass after the timed region")
    ap.add_argument("--overlap", action="store_true", help="(default since round 3; accepted for old command lines)")
    ap.add_argument("--no-generate-check", action="store_true",
                    help="skip the extra (untimed-for-`value`) measurement of `generate`: ms per KV-cache decode step at 8 prompts and its HBM roofline")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="do not enqueue the next step's frozen towers beside this step's backward / optimizer tail (the timed default does)")
    ap.add_argument("--cpu-sample", type=int, default=1, help="pairs of the benchmarked config in the CPU-baseline sample")
    ap.add_argument("--cpu-cfg2", action="store_true", help="also time config 2 on the CPU (1 warm-up + 1 repetition, ~1.5 min)")
    ap.add_argument("--no-batch64-check", action="store_true",
                    help="skip the extra (untimed-for-`value`) measurement at the north-star batch 64 x 1024")
    return ap.parse_args()


class GpuWeights:
    """dict-like view of the GPU model's parameters as fp32 numpy (fetched per access, never all at once)."""

    def __init__(self, model, cache=False):
        self.p = dict(model.named_parameters())
        self.fetch_s = 0.0            # time spent downloading weights (excluded from the CPU-work figure)
        self.cache = {} if cache else None      # keep what was fetched (tests that run the oracle several times: ~40 GB of host
                                                # memory for the cfg3 models)

    def __getitem__(self, k):
        if self.cache is not None and k in self.cache:
            return self.cache[k]
        t0 = time.perf_counter()
        t = self.p[k]
        if k.endswith("embed_tokens.weight") or k.endswith("word_embeddings.weight"):
            return _Rows(t, self)
        out = t.detach().float().cpu().numpy()
        self.fetch_s += time.perf_counter() - t0
        if self.cache is not None:
            self.cache[k] = out
        return out


class _Rows:
    def __init__(self, t, owner):
        self.t, self.owner = t, owner

    def __getitem__(self, ids):
        import torch
        t0 = time.perf_counter()
        idx = torch.as_tensor(ids).to(self.t.device)
        out = self.t.detach()[idx].float().cpu().numpy()
        self.owner.fetch_s += time.perf_counter() - t0
        return out


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


def gpu_vs_oracle_parity(model, gpu_batch, ref, n_pairs, layer):
    """The timed configuration checked against the oracle on the pair(s) the CPU leg just computed (rows 0..n_pairs-1 of the
    timed batch; same weights: the oracle read the GPU model's): relative L2 error of the pooled, normalised embeddings, and
    the InfoNCE loss of those rows against the timed batch's text embeddings once with the GPU's rows and once with the
    oracle's rows in their place (with one pair alone the loss would be log 1 = 0 on both sides).  Eval mode (no dropout),
    the benchmark's dtype (bf16 / fp8 GEMMs) on the GPU against the fp32 oracle = the reference's CPU arithmetic."""
    import torch
    import p2t_hip as P
    was_training = model.training
    model.eval()
    try:
        with torch.no_grad():
            p = P.l2_normalize(P.get_sequence_embeddings(model, gpu_batch["protein_input_ids"][:n_pairs], gpu_batch["protein_attention_mask"][:n_pairs]))
            t = P.l2_normalize(P.get_description_embeddings(model, gpu_batch["description_input_ids"], gpu_batch["description_attention_mask"], layer))
            labels = torch.arange(n_pairs, device=p.device)
            loss_fn = P.SegmentedBatchInfoNCELoss()
            loss_gpu = float(loss_fn(p, t, labels))
            t_mixed = t.clone()
            t_mixed[:n_pairs] = torch.from_numpy(np.ascontiguousarray(ref["text"], dtype=np.float32)).to(t.device)
            p_ref = torch.from_numpy(np.ascontiguousarray(ref["protein"], dtype=np.float32)).to(p.device)
            loss_ref = float(loss_fn(p_ref, t_mixed, labels))
    finally:
        model.train(was_training)
    rel = lambda a, b: float(np.linalg.norm(a.astype(np.float64) - b) / max(np.linalg.norm(b), 1e-30))
    return {"protein_rel": round(rel(p.cpu().numpy(), ref["protein"]), 6), "text_rel": round(rel(t[:n_pairs].cpu().numpy(), ref["text"]), 6),
            "loss_abs": round(abs(loss_gpu - loss_ref), 6), "loss_gpu": round(loss_gpu, 5), "loss_oracle_rows": round(loss_ref, 5),
            "pairs": n_pairs, "negatives": int(t.shape[0]), "oracle": "fp32 (the reference's CPU arithmetic), same weights"}


def cpu_baseline(model, esm, llama, cfg_name, Tp, Tt, n_pairs, gpu_batch, with_cfg2=False):
    """The oracle (a numpy port of the reference algorithm, fp32) timed on the host cores, BASELINE.md section 3 protocol:
    forward + InfoNCE only; config 1 (the reference's own CPU-runnable case) with 1 warm-up + 3 timed repetitions, mean
    and min; the benchmarked config on `n_pairs` pair(s), scaled per pair ("extrapolated": the CPU time is linear in
    pairs); config 2 the same way on request (--cpu-cfg2, ~1.5 min).  `value` is the figure for the benchmarked workload."""
    import numpy as np
    from oracle import p2t_oracle as O
    from oracle.weights import model_weights
    from p2t_hip import specs, synth
    from threadpoolctl import threadpool_limits
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, 16))          # the GPU box gives one GPU a 16-core share
    extra = {"cpu_model": _cpu_model()}

    def timed_config(name, reps):
        e_name, l_name, _, B, tp, tt = specs.CONFIGS[name]
        e, l = specs.esm_spec(e_name), specs.llama_spec(l_name)
        Wc = model_weights(e, l, specs.adapter_spec(e, l), 0, cache=True)
        pid, pmask = synth.protein_batch(1234, B, tp)
        tid, tmask = synth.text_batch(1234, B, tt)
        times = []
        for r in range(reps + 1):                               # repetition 0 is the warm-up (also materialises the weights)
            t0 = time.perf_counter()
            out = O.contrastive_step(e, l, Wc, pid, pmask, tid, tmask, layer=min(16, l.num_hidden_layers), num_segments=1)
            if r:
                times.append(time.perf_counter() - t0)
        assert np.isfinite(out["loss"])
        return {"samples_per_s_mean": round(B / float(np.mean(times)), 3), "samples_per_s_best": round(B / min(times), 3),
                "s_per_step_mean": round(float(np.mean(times)), 3), "reps": reps, "batch": B, "T_p": tp, "T_t": tt}

    with threadpool_limits(limits=cores):
        extra["cfg1"] = timed_config("cfg1", 3)
        if with_cfg2:
            extra["cfg2"] = timed_config("cfg2", 1)
        pid, pmask = synth.protein_batch(1234, n_pairs, Tp)
        tid, tmask = synth.text_batch(1234, n_pairs, Tt)
        W = GpuWeights(model)
        t0 = time.perf_counter()
        out = O.contrastive_step(esm, llama, W, pid, pmask, tid, tmask, layer=min(16, llama.num_hidden_layers), num_segments=1)
        dt = time.perf_counter() - t0
    assert np.isfinite(out["loss"])
    work = dt - W.fetch_s
    return {"value": round(n_pairs / work, 5), "unit": "samples/s", "cores": int(cores), "kind": "port",
            "parity": gpu_vs_oracle_parity(model, gpu_batch, out, n_pairs, min(16, llama.num_hidden_layers)),
            "sample": f"{n_pairs} pair(s) of {cfg_name} (T_p={Tp}, T_t={Tt}), extrapolated per pair: fp32 numpy/OpenBLAS oracle, forward + InfoNCE, "
                      f"{cores} BLAS threads, one un-warmed repetition, {work:.1f} s of CPU work (+{W.fetch_s:.1f} s downloading the GPU model's "
                      f"weights, excluded); cfg1 (B=4, 128/64 tokens) timed in full with 1 warm-up + 3 repetitions: see `protocol`",
            "protocol": extra}


def kernel_src_sha16():
    """Same stamp as tools/pmc_traffic.py: sha256 over csrc/*.hip, *.h and *.inc (printed in the roofline block of the JSON line)."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "prot2text-v2-esm3_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h", ".inc")):
            with open(os.path.join(d, name), "rb") as f:
                h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def pmc_traffic(cfg_name, batch, family="gemm_nt_mfma*"):
    """HBM-side bytes per launch of the dominant kernel from rocprofv3 PMC passes (FETCH_SIZE x2 on gfx950 + WRITE_SIZE,
    MI355X_MICROARCH.md HBM section).  Counters cannot be read from inside the benchmark process, so the figure comes from
    the committed passes -- and only if they were taken on THESE kernel sources (kernel_src_sha16 stamp) and this workload;
    otherwise null."""
    import glob
    stamp = kernel_src_sha16()
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic*.json")), reverse=True):
        try:
            with open(path) as f:
                d = json.load(f)
            if d.get("kernel_src_sha16") != stamp or d.get("workload", "cfg3/16") != f"{cfg_name}/{batch}":
                continue
            return round(d["kernels"][family]["hbm_bytes_per_launch"])
        except (KeyError, ValueError, OSError):
            continue
    return None


def generate_check(model, llama, Tp, dev, B=8, n_prompt=64, n_new=33):
    """ms per KV-cache decode step of `model.generate` (greedy) and the step's HBM roofline fraction."""
    import torch
    from p2t_hip import synth
    rs = np.random.RandomState(0)
    T = Tp + n_prompt
    ids = rs.randint(0, 128000, size=(B, T)).astype(np.int64)
    ids[:, 16:16 + Tp] = model.config.placeholder_id
    pid, pmask = synth.protein_batch(5, B, Tp)
    t = lambda a: torch.from_numpy(a).to(dev)
    kw = dict(inputs=t(ids), attention_mask=torch.ones((B, T), dtype=torch.int64, device=dev), protein_input_ids=t(pid),
              protein_attention_mask=t(pmask), eos_token_id=None, pad_token_id=128002, do_sample=False)
    model.eval()
    model.generate(**kw, max_new_tokens=3)                       # engines, stream copies of the weights, lazy initialisation
    torch.cuda.synchronize()
    times = []
    for n in (1, n_new):
        t0 = time.perf_counter()
        model.generate(**kw, max_new_tokens=n)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    dt = (times[1] - times[0]) / (n_new - 1)
    H, F, L, V = llama.hidden_size, llama.intermediate_size, llama.num_hidden_layers, llama.vocab_size
    nh, nkv, d = llama.num_attention_heads, llama.num_key_value_heads, llama.head_dim
    w_bytes = 2 * (L * ((nh + 2 * nkv) * d * H + nh * d * H + 3 * H * F) + V * H)
    kv_bytes = 2 * 2 * L * nkv * d * B * (T + n_new / 2)
    # drop the 15 GB of stream copies again: the legs after this one run the CPU oracle beside the GPU model
    model.llama_decoder.model._stream_engine = None
    torch.cuda.empty_cache()
    return {"workload": f"{B} prompts x {T} tokens ({Tp} protein placeholders), greedy, all {L} decoder layers + LM head per token",
            "ms_per_decode_step": round(dt * 1e3, 3), "tokens_per_s": round(B / dt, 1), "prompt_phase_ms": round(times[0] * 1e3, 1),
            "step_bytes_gb": round((w_bytes + kv_bytes) / 1e9, 2), "hbm_tb_per_s": round((w_bytes + kv_bytes) / dt / 1e12, 2),
            "frac_of_hbm_peak": round((w_bytes + kv_bytes) / dt / 8e12, 3)}


def spawn_ranks(n: int) -> int:
    """Start `n` ranks of this script (same argv) as child processes with the torch.distributed environment set, before
    anything in THIS process has touched the GPU (no torch import here).  Rank 0's stdout is relayed line by line; the
    first failing rank ends the run (the others are terminated) and its exit code is returned."""
    import socket
    import subprocess
    import threading
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), P2T_BENCH_CHILD="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")         # dmabuf IPC: RCCL peer access on this pool
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))

    def relay():
        for line in procs[0].stdout:
            sys.stdout.write(line)
            sys.stdout.flush()
    t = threading.Thread(target=relay, daemon=True)
    t.start()
    rc = 0
    alive = set(range(n))
    while alive and rc == 0:
        for r in list(alive):
            code = procs[r].poll()
            if code is not None:
                alive.discard(r)
                if code != 0:
                    rc = code
                    print(f"bench.py: rank {r} exited with code {code}; stopping the other ranks", file=sys.stderr)
        time.sleep(0.2)
    for r in alive:                                   # a rank failed: the others would wait in a collective forever
        procs[r].terminate()
    for pr in procs:
        try:
            pr.wait(timeout=30)
        except subprocess.TimeoutExpired:
            pr.kill()
    t.join(timeout=5)
    return rc


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus))
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py rank {rank}: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    n_dev = torch.cuda.device_count()                 # counting devices does not initialise the GPU
    if args.rehearse_shared_gpu:
        local = 0
    if local >= n_dev:
        raise SystemExit(f"bench.py rank {rank}: --gpus {args.gpus} needs {args.gpus} GPUs on this node, it exposes {n_dev} "
                         f"(one process per GPU; there is no CPU or shared-GPU fallback)")
    import p2t_hip as P
    from p2t_hip import _lib, specs, synth
    if args.gemm_policy:
        _lib.call("p2t_set_gemm_policy", int(args.gemm_policy))
    torch.cuda.set_device(local)
    dev = torch.device(f"cuda:{local}")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.rehearse_shared_gpu:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)

    esm_name, llama_name, dtype_name, B, Tp, Tt = specs.CONFIGS[args.config]
    B = args.batch or B
    esm, llama = specs.esm_spec(esm_name), specs.llama_spec(llama_name)
    ad = specs.adapter_spec(esm, llama)
    fp8 = dtype_name == "fp8"
    dtype = torch.float32 if dtype_name == "f32" else torch.bfloat16
    model = P.Esm2LlamaInstructForCausalLM.from_specs(esm, llama, ad, dtype=dtype, device=dev, seed=0,
                                                      gemm_dtype="fp8" if fp8 else "model")
    model.esm_encoder.requires_grad_(False)
    model.llama_decoder.requires_grad_(False)
    overlap = not args.no_overlap
    trainer = P.ContrastiveTrainer(model, num_segments=args.segments, train_mode=not args.eval_mode, global_negatives=True,
                                  overlap_streams=overlap)

    pid, pmask = synth.protein_batch(1234 + rank, B, Tp)
    tid, tmask = synth.text_batch(1234 + rank, B, Tt)
    batch = {k: torch.from_numpy(v).to(dev) for k, v in dict(protein_input_ids=pid, protein_attention_mask=pmask,
                                                              description_input_ids=tid, description_attention_mask=tmask).items()}

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        trainer.step(batch)
    barrier()
    t0 = time.perf_counter()
    # Steps are pipelined through the frozen towers (ContrastiveTrainer.step(batch, next_batch)): the encoder / text tower of step
    # i + 1 depend on nothing step i's optimizer writes, so they are enqueued on the side streams beside step i's backward + clip +
    # AdamW tail.  The timed region starts and ends clean: the warm-up's last step prefetches nothing and neither does the last timed
    # step, so exactly `steps` tower passes and `steps` tails run between the two barriers.
    pipeline = overlap and not args.no_pipeline
    for i in range(args.steps):
        loss = trainer.step(batch, next_batch=batch if (pipeline and i + 1 < args.steps) else None)
    barrier()
    elapsed = time.perf_counter() - t0
    loss = trainer.global_loss(loss)            # mean over ranks (a copy; outside the timed region)
    # Per-kernel event times for the roofline block: a SEPARATE pass on one stream after the timed region (with two streams an
    # event pair around a launch also spans whatever the other stream runs meanwhile).  One un-timed step to settle, then
    # `ev_steps` steps with every MFMA launch bracketed by HIP events on its launch stream; also timed as a whole = the
    # single-stream rate of the same step.
    ev_steps = args.steps if args.event_steps < 0 else min(args.event_steps, args.steps)
    ms = (ctypes.c_double * 3)()
    cnt = (ctypes.c_int64 * 3)()
    fl = (ctypes.c_double * 3)()
    one_stream_rate = None
    if ev_steps > 0:
        trainer.overlap_streams = False
        trainer.step(batch)
        barrier()
        _lib.call("p2t_prof_enable", 1)
        t1 = time.perf_counter()
        for i in range(ev_steps):
            trainer.step(batch)
        barrier()
        one_stream_rate = world * B * ev_steps / (time.perf_counter() - t1)
        _lib.call("p2t_prof_collect", ms, cnt, fl, 3)          # synchronises the recorded events
        _lib.call("p2t_prof_enable", 0)
        trainer.overlap_streams = overlap
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    loss_val = float(loss.cpu()[0])

    if rank == 0:
        f = specs.flops_per_sample(esm, llama, ad, Tp, Tt, 16, backward=True)
        value = world * B * args.steps / elapsed
        step_tflops = f["total"] * B * args.steps / elapsed / 1e12          # per GPU
        # dominant kernel: the bf16 MFMA GEMM family (gemm_nt_w4_kernel, gemm_nt_mfma*: one algorithm, several launch forms; a HIP-event record = one kernel launch).  algorithmic GEMM FLOPs per sample = linear layers of both towers
        # + adapter fwd/bwd (everything except the attention score/value products)
        He, Le = esm.hidden_size, esm.num_hidden_layers
        attn_flops = Le * 4 * Tp * Tp * He + min(16, llama.num_hidden_layers) * 2 * (Tt + 1) * Tt * llama.hidden_size
        gemm_flops_step = (f["total"] - attn_flops) * B
        if fp8:
            # dominant kernel: the fp8 MFMA GEMM (the tower projections); the adapter's bf16 GEMMs are a separate, small family.
            # Its algorithmic FLOPs are the towers' linear layers = HIP-event class 2's own tally (2 M N K per launch).
            dom = 2
            gemm_flops_step = fl[2] / max(ev_steps, 1)
        else:
            dom = 0
        launches_step = cnt[dom] / max(ev_steps, 1)
        avg_ms = ms[dom] / max(cnt[dom], 1)
        achieved = (gemm_flops_step / max(launches_step, 1)) / (avg_ms * 1e-3) / 1e12 if cnt[dom] else 0.0
        out = {
            "metric": "contrastive-step samples/sec (protein-text pairs)", "value": round(value, 3), "unit": "samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "fp8-e4m3 weights and GEMM operands (per-row E8M0 scales, fp8 MFMA), bf16 activations" if fp8 else dtype_name,
            "data": "synthetic" if not args.rehearse_shared_gpu else f"synthetic (REHEARSAL: {world} ranks share one GPU over gloo -- not a scaling measurement)",
            "config": {"workload": f"{args.config}: {esm_name} + {llama_name} (text layers 1-16), per-GPU batch {B} x {Tp} residues / "
                                   f"{B} x {Tt} text tokens, readout mix, InfoNCE tau=0.05, adapter fwd+bwd + clip + AdamW, "
                                   f"{'train mode (dropout 0.3)' if not args.eval_mode else 'eval mode'}, segments {args.segments}",
                       "global_batch": world * B,
                       "parallelism": (f"dp{dist.get_world_size()} over {dist.get_backend()}{' (RCCL)' if dist.get_backend() == 'nccl' else ''}: {dist.get_world_size()} processes, "
                                       f"{'one GPU each' if not args.rehearse_shared_gpu else 'ALL ON ONE GPU (rehearsal)'}; "
                                       "text-embedding all-gather + one adapter-gradient all-reduce per step") if world > 1
                                      else "dp1 (single process, no collective)",
                       "algorithmic_tflop_per_sample": round(f["total"] / 1e12, 4),
                       "step_tflops_per_gpu": round(step_tflops, 1), "step_frac_of_bf16_peak": round(step_tflops / PEAK_BF16_TFLOPS, 4),
                       "loss": round(loss_val, 5)},
            "roofline": {"bound": "mfma", "kernel_src_sha16": kernel_src_sha16(),
                         "kernel": ("gemm_nt_fp8_w4_kernel + gemm_nt_fp8_kernel (e4m3 16x16x128 block-scaled MFMA GEMM: four-wave persistent form, eight-wave per-tile form, all epilogues)" if fp8 else
                                    "gemm_nt_w4_kernel + gemm_nt_mfma*_kernel (bf16 16x16x32 MFMA GEMM: four-wave persistent form, eight-wave persistent / per-tile / split-K-tail forms, all epilogues)"),
                         "achieved": round(achieved, 1), "peak": PEAK_FP8_TFLOPS if fp8 else PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / (PEAK_FP8_TFLOPS if fp8 else PEAK_BF16_TFLOPS), 4),
                         "frac_of_bf16_peak": round(achieved / PEAK_BF16_TFLOPS, 4),
                         "traffic": pmc_traffic(args.config, B, "gemm_nt_fp8*" if fp8 else "gemm_nt_mfma*"),
                         "launches_per_step": round(launches_step, 1), "avg_launch_ms": round(avg_ms, 4),
                         "gemm_ms_per_step": round(ms[dom] / max(ev_steps, 1), 3), "attention_ms_per_step": round(ms[1] / max(ev_steps, 1), 3),
                         "bf16_gemm_ms_per_step": round(ms[0] / max(ev_steps, 1), 3),
                         "event_steps": ev_steps,
                         "event_pass": f"{ev_steps} single-stream steps after the timed region (HIP events on the launch stream; the rocprofv3 "
                                       "summary under profiles/ is `bench.py --no-overlap`)",
                         "attention_tflops": round(fl[1] / max(ms[1], 1e-9) / 1e9, 1)},
        }
        if fp8:
            out["config"]["step_frac_of_fp8_peak"] = round(step_tflops / PEAK_FP8_TFLOPS, 4)
        if world == 1 and not args.no_batch64_check and B != 64 and args.config in ("cfg3", "cfg4"):
            # BASELINE.json north_star quotes its >= 40 % target "at batch 64 x 1024 residues": measured here, after
            # and outside the timed region that produces `value`, on the same model.
            pid64, pm64 = synth.protein_batch(99, 64, Tp)
            tid64, tm64 = synth.text_batch(99, 64, Tt)
            b64 = {k: torch.from_numpy(v).to(dev) for k, v in dict(protein_input_ids=pid64, protein_attention_mask=pm64,
                                                                     description_input_ids=tid64, description_attention_mask=tm64).items()}
            trainer.step(b64)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(3):
                trainer.step(b64)
            torch.cuda.synchronize()
            dt64 = (time.perf_counter() - t1) / 3
            tf64 = f["total"] * 64 / dt64 / 1e12
            out["config"]["batch64_check"] = {"samples_per_s": round(64 / dt64, 2), "ms_per_step": round(dt64 * 1e3, 2),
                                              "step_tflops": round(tf64, 1), "frac_of_bf16_peak": round(tf64 / PEAK_BF16_TFLOPS, 4)}
        if world == 1 and not args.no_batch64_check and args.config in ("cfg3", "cfg4"):
            # BASELINE.json configs[4]'s one-GPU share (the same models with fp8-e4m3 tower GEMMs, batch 64 x 1024 residues) on the
            # model already built: driver-visible beside the bf16 headline.  Outside the timed region;
            # 1 settling step, then 3 timed on two streams, then 2 single-stream steps with every fp8 GEMM launch bracketed by HIP
            # events (class 2: the same accounting as the roofline block of a `--config cfg5` run).
            try:
                model.set_gemm_dtype("fp8")
                t5 = P.ContrastiveTrainer(model, num_segments=args.segments, train_mode=not args.eval_mode, global_negatives=True, overlap_streams=overlap)
                if "b64" not in locals():
                    pid64, pm64 = synth.protein_batch(99, 64, Tp)
                    tid64, tm64 = synth.text_batch(99, 64, Tt)
                    b64 = {k: torch.from_numpy(v).to(dev) for k, v in dict(protein_input_ids=pid64, protein_attention_mask=pm64,
                                                                             description_input_ids=tid64, description_attention_mask=tm64).items()}
                t5.step(b64)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(3):
                    l5 = t5.step(b64)
                torch.cuda.synchronize()
                dt5 = (time.perf_counter() - t1) / 3
                t5.overlap_streams = False
                t5.step(b64)
                torch.cuda.synchronize()
                ms5, cnt5, fl5 = (ctypes.c_double * 3)(), (ctypes.c_int64 * 3)(), (ctypes.c_double * 3)()
                _lib.call("p2t_prof_enable", 1)
                for _ in range(2):
                    t5.step(b64)
                torch.cuda.synchronize()
                _lib.call("p2t_prof_collect", ms5, cnt5, fl5, 3)
                _lib.call("p2t_prof_enable", 0)
                tf5 = f["total"] * 64 / dt5 / 1e12
                g5 = fl5[2] / max(ms5[2], 1e-9) / 1e9
                out["config"]["cfg5_check"] = {"workload": "cfg5 share of one GPU: the same models, fp8-e4m3 tower GEMMs (per-row E8M0 scales), batch 64 x 1024 residues",
                                               "samples_per_s": round(64 / dt5, 2), "ms_per_step": round(dt5 * 1e3, 2), "step_tflops": round(tf5, 1),
                                               "step_frac_of_fp8_peak": round(tf5 / PEAK_FP8_TFLOPS, 4), "fp8_gemm_tflops": round(g5, 1),
                                               "fp8_gemm_frac_of_fp8_peak": round(g5 / PEAK_FP8_TFLOPS, 4), "fp8_gemm_launches_per_step": cnt5[2] // 2,
                                               "fp8_gemm_avg_launch_ms": round(ms5[2] / max(cnt5[2], 1), 4), "attention_ms_per_step": round(ms5[1] / 2, 3),
                                               "loss": round(float(l5.cpu()[0]), 5)}
                del t5
            except Exception as e:                                      # noqa: BLE001
                out["config"]["cfg5_check"] = {"error": f"{type(e).__name__}: {e}"[:300]}
            finally:
                model.set_gemm_dtype("model")
        if world == 1 and not args.no_batch64_check and args.config in ("cfg3", "cfg4", "cfg5"):
            # Ragged workload: 64 pairs with protein lengths from a clipped log-normal (median ~315
            # residues, the shape of UniProt lengths; crop at 1024), once as the padded 64 x T_max step and once
            # length-sorted (data.sort_batch_by_length) and cut into segments that each run at their own longest length
            # (ContrastiveTrainer(trim_padding=True), contrastive.plan_length_segments).  Same loss by construction (tests/test_gpu_ragged.py).
            from p2t_hip.data import sort_batch_by_length
            rs = np.random.RandomState(0)
            lens = np.clip(np.round(rs.lognormal(5.75, 0.6, 64)), 16, Tp).astype(int).tolist()
            tl = np.clip(np.round(rs.lognormal(4.0, 0.5, 64)), 4, Tt).astype(int).tolist()
            pidr, pmr = synth.protein_batch(77, 64, Tp, lens)
            tidr, tmr = synth.text_batch(77, 64, Tt, lengths=tl)
            Tmax = int(max(lens))
            host = dict(protein_input_ids=torch.from_numpy(pidr[:, :Tmax].copy()), protein_attention_mask=torch.from_numpy(pmr[:, :Tmax].copy()),
                        description_input_ids=torch.from_numpy(tidr), description_attention_mask=torch.from_numpy(tmr))
            to_dev = lambda b: {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in b.items()}
            padded, srt = to_dev(host), to_dev(sort_batch_by_length(host))
            t_trim = P.ContrastiveTrainer(model, train_mode=not args.eval_mode, trim_padding=True)

            def rate(tr, b):
                tr.step(b)
                torch.cuda.synchronize()
                t = time.perf_counter()
                for _ in range(2):
                    tr.step(b)
                torch.cuda.synchronize()
                return 64 * 2 / (time.perf_counter() - t)
            r_pad, r_trim = rate(trainer, padded), rate(t_trim, srt)       # t_trim: segments on two streams (its default)
            t_trim.overlap_streams = False
            r_trim1 = rate(t_trim, srt)
            out["config"]["ragged_check"] = {"lengths": "64 pairs, protein lengths lognormal(5.75, 0.6) clipped to [16, T_p]",
                                             "mean_length": round(float(np.mean(lens)), 1), "longest": Tmax,
                                             "padded_samples_per_s": round(r_pad, 2), "sorted_trimmed_samples_per_s": round(r_trim, 2),
                                             "sorted_trimmed_one_stream_samples_per_s": round(r_trim1, 2),
                                             "segments_rows_x_length": [[b - a, t] for a, b, t, _ in t_trim._segments(srt, 64, Tmax)]}
            del t_trim
        out["config"]["streams"] = ("text tower and encoder segments on separate HIP streams" if overlap else "one HIP stream")
        out["config"]["pipeline"] = ("frozen towers of step i + 1 enqueued beside the backward / optimizer tail of step i (steps - 1 of the timed steps; "
                                     "the region starts and ends with nothing in flight)" if pipeline else "none")
        if one_stream_rate is not None:
            # the same step on ONE stream, taken from the event pass above (includes the ~1.5 % cost of the event records)
            out["config"]["one_stream_check"] = {"samples_per_s": round(one_stream_rate, 2)}
        if world == 1 and not args.no_generate_check and not args.no_batch64_check and args.config in ("cfg3", "cfg4"):
            # `generate` of the same model (reference models/modeling_esm2llama_instruct.py:217-251): 8 prompts of T_p placeholders + 64
            # tokens, greedy; per decode step = (time of 33 new tokens - time of 1) / 32.  HBM-bound: every decoder weight + the KV cache
            # once per step.  Not part of `value`; a failure here is reported, never fatal to the headline.
            try:
                out["config"]["generate_check"] = generate_check(model, llama, Tp, dev)
            except Exception as e:                                      # noqa: BLE001
                out["config"]["generate_check"] = {"error": f"{type(e).__name__}: {e}"[:300]}
            finally:
                model.train(not args.eval_mode)
        if world == 1 and not args.no_cpu_baseline:
            # forward + loss only (eval mode): the quantity the CPU baseline below measures
            trainer.evaluate(batch)
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            for _ in range(3):
                trainer.evaluate(batch)
            torch.cuda.synchronize()
            out["config"]["forward_only_samples_per_s"] = round(B * 3 / (time.perf_counter() - t2), 2)
            out["cpu_baseline"] = cpu_baseline(model, esm, llama, args.config, Tp, Tt, args.cpu_sample, batch, args.cpu_cfg2)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
