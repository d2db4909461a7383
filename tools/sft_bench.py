"""Stage-2 (SFT) step of Esm2LlamaInstructForCausalLM at cfg3 sizes: ESM2-3B encode -> adapter -> placeholder
scatter -> 32-layer Llama-3.1-8B -> LM head -> shifted CE, once forward only and once as the training step `loss.backward()`
(frozen towers, adapter trainable: training forward with the activation tape + the dX chain of csrc/llama_train.hip + adapter
backward), with a per-kernel-family breakdown of the backward from the torch profiler-free HIP-event brackets below.
python tools/sft_bench.py [B] > gpurun_out/sft_bench.log"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "prot2text-v2-esm3_amd"))
import p2t_hip as P                                             # noqa: E402
from p2t_hip import specs, synth                                # noqa: E402


def main():
    dev = torch.device("cuda:0")
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    esm_name, llama_name, _, _, Tp, _ = specs.CONFIGS["cfg3"]
    esm, llama = specs.esm_spec(esm_name), specs.llama_spec(llama_name)
    ad = specs.adapter_spec(esm, llama)
    model = P.Esm2LlamaInstructForCausalLM.from_specs(esm, llama, ad, dtype=torch.bfloat16, device=dev, seed=0)
    model.eval()
    n_prompt, n_desc = 64, 128
    T = Tp + n_prompt + n_desc                                      # placeholders (one per residue token) + prompt + description
    ph = model.config.placeholder_id
    rs = np.random.RandomState(0)
    ids = rs.randint(0, 128000, size=(B, T)).astype(np.int64)
    ids[:, 16:16 + Tp] = ph                                         # chat template: system text, <protein placeholders>, question, answer
    labels = ids.copy()
    labels[:, :Tp + n_prompt] = -100
    pid, pmask = synth.protein_batch(5, B, Tp)
    t = lambda a: torch.from_numpy(a).to(dev)
    kw = dict(input_ids=t(ids), attention_mask=torch.ones((B, T), dtype=torch.int64, device=dev), labels=t(labels),
              protein_input_ids=t(pid), protein_attention_mask=t(pmask))
    with torch.no_grad():
        out = model(**kw)
        torch.cuda.synchronize()
        n = 3
        t0 = time.perf_counter()
        for _ in range(n):
            out = model(**kw)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
    He, Fe, Le = esm.hidden_size, esm.intermediate_size, esm.num_hidden_layers
    Hl, Fl, Ll = llama.hidden_size, llama.intermediate_size, llama.num_hidden_layers
    kv = llama.num_key_value_heads * llama.head_dim
    f_esm = Le * Tp * (2 * (4 * He * He + 2 * He * Fe) + 4 * Tp * He)
    f_ad = 2 * Tp * (He * ad.intermediate_dim + ad.intermediate_dim * Hl)
    f_llama = Ll * T * (2 * (2 * Hl * Hl + 2 * Hl * kv + 3 * Hl * Fl) + 2 * (T + 1) * Hl) + 2 * T * Hl * llama.vocab_size
    f = f_esm + f_ad + f_llama
    # ---- training step: frozen towers, adapter trainable
    model.requires_grad_(False)
    model.adapter.requires_grad_(True)

    def step():
        model.zero_grad(set_to_none=True)
        o = model(**kw)
        o.loss.backward()
        return o
    step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2):
        o2 = step()
    torch.cuda.synchronize()
    dt_train = (time.perf_counter() - t0) / 2
    gn = float(model.adapter.fc2.weight.grad.float().norm())
    # the dX chain costs the decoder's linear FLOPs once more (no weight gradients) + attention backward (2.5 x its forward)
    f_bwd = Ll * T * (2 * (2 * Hl * Hl + 2 * Hl * kv + 3 * Hl * Fl) + 2.5 * 2 * (T + 1) * Hl) + 2 * T * Hl * llama.vocab_size + 2 * f_ad
    print(f"sft train step cfg3: B={B}: {dt_train * 1e3:.1f} ms/batch = {B / dt_train:.2f} samples/s, {(f + f_bwd) * B / dt_train / 1e12:.0f} TFLOP/s algorithmic "
          f"(forward {f / 1e12:.2f} + backward {f_bwd / 1e12:.2f} TF/sample); backward alone ~{(dt_train - dt) * 1e3:.1f} ms; loss {float(o2.loss):.4f}, "
          f"|grad fc2.weight| {gn:.3e}", flush=True)
    print(f"sft forward cfg3: B={B}, {Tp} residues, {T} decoder tokens ({n_desc} supervised): {dt * 1e3:.1f} ms/batch = {B / dt:.2f} samples/s, "
          f"{B * T / dt:.0f} decoder tokens/s, {f * B / dt / 1e12:.0f} TFLOP/s algorithmic ({f / 1e12:.2f} TF/sample: ESM {f_esm / 1e12:.2f}, "
          f"decoder + LM head {f_llama / 1e12:.2f}); loss {float(out.loss):.4f}", flush=True)
    if "lora" in sys.argv:
        # the reference's stage-2 recipe: LoRA (r = 16, alpha = 32, dropout 0.1) on the seven decoder projections + the adapter trainable
        # (scripts/train_instruct.py:146-183); per-layer path of p2t_hip/decoder_train.py (correctness first, not tuned)
        del o2, out
        torch.cuda.empty_cache()
        lora = model.add_lora(r=16, lora_alpha=32, lora_dropout=0.1)
        model.train()

        def lstep():
            model.zero_grad(set_to_none=True)
            lora.zero_grad(set_to_none=True)
            o = model(**kw)
            o.loss.backward()
            return o
        o3 = lstep()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(2):
            o3 = lstep()
        torch.cuda.synchronize()
        dt_l = (time.perf_counter() - t0) / 2
        ga = [q.grad for q in lora.parameters() if q.grad is not None]
        n_l = sum(q.numel() for q in lora.parameters())
        print(f"sft LoRA train step cfg3: B={B}, r=16 on {len(ga)} of {len(list(lora.parameters()))} matrices with a gradient ({n_l / 1e6:.1f} M LoRA parameters): "
              f"{dt_l * 1e3:.1f} ms/batch = {B / dt_l:.2f} samples/s; loss {float(o3.loss):.4f}; |grad| of the first B matrix "
              f"{float([q.grad for n, q in lora.named_parameters() if n.endswith('B')][0].float().norm()):.3e}; "
              f"peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)


if __name__ == "__main__":
    main()
