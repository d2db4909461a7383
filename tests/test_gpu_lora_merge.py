"""LoRA adapters at inference: `load_and_merge_adapter` = the reference's `PeftModel.from_pretrained(...).merge_and_unload()`
(scripts/generate_instruct.py:187-191) on a checkpoint directory in peft's layout (adapter_config.json + adapter_model.safetensors with
base_model.model.<module>.lora_A / lora_B.weight and full copies of the `modules_to_save` modules).  peft itself is not installed: the
layout is restated, the ARITHMETIC is pinned here -- generation with the merged model equals the oracle run on W + (alpha / r) B A."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import p2t_oracle as O
from gpu_util import build_model, dev, to_dev, to_np
from helpers import model_weights
from p2t_hip import specs

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


class _Overlay:
    """the synthetic weights with a few tensors replaced"""

    def __init__(self, base):
        self.base, self.over = base, {}

    def __contains__(self, k):
        return k in self.over or k in self.base

    def __getitem__(self, k):
        return self.over[k] if k in self.over else self.base[k]

    def __setitem__(self, k, v):
        self.over[k] = v


TARGETS = ("self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj", "self_attn.o_proj", "mlp.gate_proj", "mlp.up_proj", "mlp.down_proj")


@pytest.mark.parametrize("case,rslora", [("d64", False), ("d128", True)])
def test_merged_adapter_generates_like_the_oracle_on_the_merged_weights(tmp_path, case, rslora):
    from safetensors.torch import save_file
    from p2t_hip.lora import load_and_merge_adapter
    z = np.load(os.path.join(HERE, "golden", "generate_tiny.npz"))
    meta = json.loads(bytes(z["meta_json"]).decode())
    m = meta["cases"][case]
    esm, llama, ad = specs.EsmSpec(**m["esm"]), specs.LlamaSpec(**m["llama"]), specs.AdapterSpec(**m["adapter"])
    model = build_model(esm, llama, ad, torch.float32, m["weight_seed"]).eval()
    model.config.placeholder_id = meta["placeholder_id"]
    W = _Overlay(model_weights(esm, llama, ad, m["weight_seed"], lm_head=True))
    r, alpha = 4, 8.0
    scaling = alpha / (np.sqrt(r) if rslora else r)
    rs = np.random.RandomState(3)
    tensors = {}
    shapes = {n: tuple(p.shape) for n, p in model.named_parameters()}
    for i in range(llama.num_hidden_layers):
        for t in TARGETS:
            name = f"llama_decoder.model.layers.{i}.{t}"
            out_f, in_f = shapes[name + ".weight"]
            A = (rs.randn(r, in_f) * 0.05).astype(np.float32)
            B = (rs.randn(out_f, r) * 0.05).astype(np.float32)
            tensors[f"base_model.model.{name}.lora_A.weight"] = torch.from_numpy(A)
            tensors[f"base_model.model.{name}.lora_B.weight"] = torch.from_numpy(B)
            W[name + ".weight"] = (np.asarray(W[name + ".weight"], dtype=np.float32) + np.float32(scaling) * (B @ A)).astype(np.float32)
    for n in ("adapter.fc1.weight", "adapter.fc1.bias", "adapter.fc2.weight", "adapter.fc2.bias"):          # modules_to_save: full copies
        v = (np.asarray(W[n], dtype=np.float32) * np.float32(0.9) + np.float32(0.01)).astype(np.float32)
        tensors["base_model.model." + n] = torch.from_numpy(v)
        W[n] = v
    os.makedirs(tmp_path / "ad")
    save_file(tensors, str(tmp_path / "ad" / "adapter_model.safetensors"))
    json.dump({"peft_type": "LORA", "r": r, "lora_alpha": alpha, "use_rslora": rslora, "fan_in_fan_out": False, "target_modules": list(TARGETS),
               "modules_to_save": ["adapter.fc1", "adapter.fc2"]}, open(tmp_path / "ad" / "adapter_config.json", "w"))
    before = model.generate(inputs=to_dev(z["input_ids"]), attention_mask=to_dev(z["attention_mask"]), protein_input_ids=to_dev(z["protein_input_ids"]),
                            protein_attention_mask=to_dev(z["protein_attention_mask"]), max_new_tokens=3, pad_token_id=meta["pad_id"])
    rec = load_and_merge_adapter(model, str(tmp_path / "ad"))
    assert rec == {"merged": 7 * llama.num_hidden_layers, "replaced": 4}
    kw = dict(inputs=to_dev(z["input_ids"]), attention_mask=to_dev(z["attention_mask"]), protein_input_ids=to_dev(z["protein_input_ids"]),
              protein_attention_mask=to_dev(z["protein_attention_mask"]))
    n = 6
    out = model.generate(**kw, max_new_tokens=n, eos_token_id=None, pad_token_id=meta["pad_id"], do_sample=False, return_dict_in_generate=True, output_logits=True)
    toks, lg = to_np(out.sequences), to_np(torch.stack(out.logits, 0))
    enc = O.esm2_forward(esm, W, z["protein_input_ids"], z["protein_attention_mask"], O.FP32, prefix="esm_encoder.")
    emb = O.sft_decoder_inputs(llama, W, z["input_ids"], O.adapter_forward(W, enc, O.FP32, prefix="adapter."), z["protein_attention_mask"],
                               meta["placeholder_id"])
    _, ref = O.generate_greedy(llama, W, emb, z["attention_mask"], n, (), meta["pad_id"], forced=toks)
    assert np.abs(lg - ref).max() < 3e-4
    assert not np.array_equal(lg[0], to_np(torch.zeros(1))) and before.shape == (3, 3)
    # a tensor for a module the model does not have is refused (strict) or skipped
    tensors["base_model.model.llama_decoder.model.layers.0.ffn.3.lora_A.weight"] = torch.zeros((r, 8))
    tensors["base_model.model.llama_decoder.model.layers.0.ffn.3.lora_B.weight"] = torch.zeros((8, r))
    save_file(tensors, str(tmp_path / "ad" / "adapter_model.safetensors"))
    with pytest.raises(KeyError):
        load_and_merge_adapter(model, str(tmp_path / "ad"))
