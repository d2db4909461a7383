"""Lab experiment (needs tools/ab/libp2t_cub.so: gemm_mfma.hip with a settable `cu_count()`): the text tower's persistent GEMMs
launched with a grid of `t` blocks and the encoder's with `e` blocks, t + e = 256, on their two streams -- each persistent block
takes a CU whole, so the grids partition the chip by themselves and both towers run at their own round granularity instead of the
text tower only filling the tails of the encoder's kernels.  P2T_HIP_LIB=tools/ab/libp2t_cub.so python tools/lab/cu_partition.py"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "prot2text-v2-esm3_amd"))
import p2t_hip as P                                             # noqa: E402
from p2t_hip import _lib, contrastive, specs, synth             # noqa: E402

dev = torch.device("cuda:0")
esm_name, llama_name, _, B, Tp, Tt = specs.CONFIGS["cfg3"]
esm, llama = specs.esm_spec(esm_name), specs.llama_spec(llama_name)
model = P.Esm2LlamaInstructForCausalLM.from_specs(esm, llama, specs.adapter_spec(esm, llama), dtype=torch.bfloat16, device=dev, seed=0)
tr = P.ContrastiveTrainer(model, output_llm_layer=16, overlap_streams=True)
pid, pmask = synth.protein_batch(1, B, Tp)
tid, tmask = synth.text_batch(1, B, Tt)
batch = {k: torch.from_numpy(v).to(dev) for k, v in dict(protein_input_ids=pid, protein_attention_mask=pmask, description_input_ids=tid,
                                                         description_attention_mask=tmask).items()}
budget = {"text": 0, "enc": 0}
set_budget = _lib.lib.p2t_lab_set_cu_budget
orig_text, orig_encode = tr.text_embeddings, model.esm_encoder.encode


def text_embeddings(*a, **k):
    set_budget(budget["text"])
    try:
        return orig_text(*a, **k)
    finally:
        set_budget(0)


def encode(*a, **k):
    set_budget(budget["enc"])
    try:
        return orig_encode(*a, **k)
    finally:
        set_budget(0)


tr.text_embeddings, model.esm_encoder.encode = text_embeddings, encode


def run(t, e, steps=8):
    budget["text"], budget["enc"] = t, e
    for _ in range(2):
        tr.step(batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        loss = tr.step(batch, next_batch=batch if i + 1 < steps else None)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f"text grid {t or 256:3d} / encoder grid {e or 256:3d}: {B / dt:7.2f} samples/s ({dt * 1e3:.2f} ms/step), loss {float(loss.cpu()[0]):.5f}", flush=True)


for t, e in ((0, 0), (32, 224), (48, 208), (64, 192), (24, 232), (0, 224), (0, 0)):
    run(t, e)
