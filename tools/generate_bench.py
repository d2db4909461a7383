"""`generate` at cfg3 sizes (reference scripts/generate_instruct.py: ESM2-3B -> adapter -> placeholder scatter -> Llama-3.1-8B, greedy
decoding): B prompts of 1024 residues + 64 prompt tokens, N new tokens.  Prints the prefill time, the time per decode step and the
HBM roofline of the step: every decoder weight is read once per step (bf16: 2 bytes x (32 layers x (qkv + o + gate/up + down) + LM
head)) plus the keys / values of the cache -- the algorithmic bytes of a step; frac = (bytes / step time) / 8 TB/s.
python tools/generate_bench.py [B] [N] [beams] [fp8] > gpurun_out/generate_bench.log      (fp8: set_gemm_dtype("fp8"): e4m3 decoder weights,
half the projection bytes of a step; the LM head stays bf16)"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "prot2text-v2-esm3_amd"))
import p2t_hip as P                                             # noqa: E402
from p2t_hip import generation, specs, synth                    # noqa: E402


def main():
    dev = torch.device("cuda:0")
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    beams = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    fp8 = len(sys.argv) > 4 and sys.argv[4] == "fp8"
    esm_name, llama_name, _, _, Tp, _ = specs.CONFIGS["cfg3"]
    esm, llama = specs.esm_spec(esm_name), specs.llama_spec(llama_name)
    ad = specs.adapter_spec(esm, llama)
    model = P.Esm2LlamaInstructForCausalLM.from_specs(esm, llama, ad, dtype=torch.bfloat16, device=dev, seed=0)
    model.eval()
    if fp8:
        model.set_gemm_dtype("fp8")
    n_prompt = 64
    T = Tp + n_prompt
    rs = np.random.RandomState(0)
    ids = rs.randint(0, 128000, size=(B, T)).astype(np.int64)
    ids[:, 16:16 + Tp] = model.config.placeholder_id
    pid, pmask = synth.protein_batch(5, B, Tp)
    t = lambda a: torch.from_numpy(a).to(dev)
    kw = dict(inputs=t(ids), attention_mask=torch.ones((B, T), dtype=torch.int64, device=dev), protein_input_ids=t(pid),
              protein_attention_mask=t(pmask), max_new_tokens=N, eos_token_id=None, pad_token_id=128002, do_sample=False, num_beams=beams)
    out = model.generate(**{**kw, "max_new_tokens": 4})          # warm-up: engines, workspaces, lazy initialisation
    torch.cuda.synchronize()
    # prompt phase alone
    t0 = time.perf_counter()
    out1 = model.generate(**{**kw, "max_new_tokens": 1})
    torch.cuda.synchronize()
    t_prompt = time.perf_counter() - t0
    res = {}
    for graph in ((True, False) if beams == 1 else (False,)):
        t0 = time.perf_counter()
        out = model.generate(**kw, use_graph=graph)
        torch.cuda.synchronize()
        res[graph] = (time.perf_counter() - t0 - t_prompt) / (N - 1)
    H, F, L, V = llama.hidden_size, llama.intermediate_size, llama.num_hidden_layers, llama.vocab_size
    nh, nkv, d = llama.num_attention_heads, llama.num_key_value_heads, llama.head_dim
    w_bytes = (1 if fp8 else 2) * L * ((nh + 2 * nkv) * d * H + nh * d * H + 3 * H * F) + 2 * V * H
    BB = B * beams
    kv_bytes = 2 * 2 * L * nkv * d * (B * T + BB * N / 2)       # keys + values, bf16, at the mean generated length
    step_bytes = w_bytes + kv_bytes
    for graph, dt in res.items():
        print(f"generate cfg3{' fp8 GEMMs' if fp8 else ''}: B={B} beams={beams} prompt {T} tokens, {N} new tokens, {'HIP graph' if graph else 'eager launches'}: "
              f"{dt * 1e3:.3f} ms/step = {BB / dt:.0f} tokens/s ({1 / dt:.1f} steps/s); step bytes {step_bytes / 1e9:.2f} GB (weights {w_bytes / 1e9:.2f}, "
              f"cache {kv_bytes / 1e9:.2f}) -> {step_bytes / dt / 1e12:.2f} TB/s = {step_bytes / dt / 8e12:.3f} of 8 TB/s", flush=True)
    print(f"prompt phase (ESM2-3B + adapter + scatter + compaction + prefill of {B} x {T} tokens + first token): {t_prompt * 1e3:.1f} ms; "
          f"first ids {out[0, :6].tolist()}", flush=True)


if __name__ == "__main__":
    main()
