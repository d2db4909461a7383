"""Where does the bf16 text tower's error against the fp32 oracle come from (text 3.6e-3 vs protein
1.3e-3 at cfg3)?  Per layer: relative error of hidden_states[k] (bf16 HIP tower vs fp32 oracle on the same bf16-rounded weights)
for a full 128-token and a 33-token description; then the pooled readout split into its mean and std halves, before and after
L2 normalisation.   python3 tools/text_error_probe.py [cfg3]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "prot2text-v2-esm3_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import p2t_hip as P  # noqa: E402
from p2t_hip import specs, synth  # noqa: E402
from oracle import p2t_oracle as O  # noqa: E402
from bench import GpuWeights  # noqa: E402
from gpu_util import build_model  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
esm_name, llama_name, _, _, Tp, Tt = specs.CONFIGS[cfg]
esm, llama = specs.esm_spec(esm_name), specs.llama_spec(llama_name)
ad = specs.adapter_spec(esm, llama)
model = build_model(esm, llama, ad, torch.bfloat16, 0).eval()
W = GpuWeights(model, cache=True)
dev = torch.device("cuda:0")
rel = lambda a, b: float(np.linalg.norm(a.astype(np.float64) - b.astype(np.float64)) / max(np.linalg.norm(b.astype(np.float64)), 1e-30))
K = min(16, llama.num_hidden_layers)
tid, tmask = synth.text_batch(51, 2, Tt, 128000, [Tt, 33], 128002, 128009)
for i in range(2):
    n = int(tmask[i].sum())
    ids, mask = tid[i:i + 1, :n], tmask[i:i + 1, :n]
    d_ids, d_mask = torch.from_numpy(ids).to(dev), torch.from_numpy(mask).to(dev)
    with torch.no_grad():
        hs = model.llama_decoder.model(d_ids, d_mask, output_hidden_states=True).hidden_states
    # oracle, layer by layer (LLAMA:381-417 as oracle/p2t_oracle.py llama_hidden_state restates it)
    x = W["llama_decoder.model.embed_tokens.weight"][ids].astype(np.float32)
    allowed = np.tril(np.ones((n, n), dtype=bool))[None, None] & (mask[:, None, None, :] != 0)
    bias = np.where(allowed, np.float32(0.0), O.NEG).astype(np.float32)
    cos, sin = O.rope_cos_sin(O.llama_inv_freq(llama), np.arange(n))
    print(f"description of {n} tokens: relative error of hidden_states[k], bf16 HIP vs fp32 oracle (|x| rms of the oracle beside it)")
    for k in range(K + 1):
        if k:
            x = O.llama_layer(llama, W, k - 1, x, bias, cos, sin, O.FP32, "llama_decoder.")
        g = hs[k].float().cpu().numpy()
        tok = np.linalg.norm(g[0] - x[0], axis=1) / np.maximum(np.linalg.norm(x[0], axis=1), 1e-30)
        print(f"  k={k:2d}  rel {rel(g, x):.2e}   rms {np.sqrt((x ** 2).mean()):9.3f}   worst token {int(tok.argmax()):3d} at {tok.max():.2e}   median token {np.median(tok):.2e}", flush=True)
    g = hs[K].float().cpu().numpy()
    m = mask
    po, pg = O.readout_embeddings(x, m, "mix"), O.readout_embeddings(g, m, "mix")
    H = x.shape[-1]
    print(f"  readout of hidden_states[{K}]: mean half {rel(pg[:, :H], po[:, :H]):.2e} (norm {np.linalg.norm(po[:, :H]):.2f})   std half {rel(pg[:, H:], po[:, H:]):.2e} "
          f"(norm {np.linalg.norm(po[:, H:]):.2f})   mix {rel(pg, po):.2e}   after L2 normalisation {rel(O.l2_normalize(pg), O.l2_normalize(po)):.2e}")
    # the product's own readout + normalisation kernels on its bf16 hidden state
    with torch.no_grad():
        t = P.l2_normalize(P.get_description_embeddings(model, d_ids, d_mask, K)).float().cpu().numpy()
    print(f"  product pooled + normalised vs oracle: {rel(t, O.l2_normalize(po)):.2e}", flush=True)
