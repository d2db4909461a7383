"""LoRA adapters of stage 2 at INFERENCE time: what `PeftModel.from_pretrained(model, dir).merge_and_unload()` does in the reference's
generation scripts (scripts/generate_instruct.py:187-191 after scripts/train_instruct.py:146-183: LoraConfig(r, lora_alpha = 2 r,
target_modules = the seven decoder projections [+ ESM-C names that match nothing in ESM2], modules_to_save = adapter.fc1 / fc2)).

`peft` is not installed in this image, so this is a restatement of its published checkpoint layout (peft 0.10, README.md:49), not a
call into it -- parity unpinned against the library itself; the arithmetic is pinned by tests/test_gpu_lora_merge.py:
    adapter_config.json          {"r", "lora_alpha", "use_rslora", "fan_in_fan_out", "target_modules", "modules_to_save", ...}
    adapter_model.safetensors    base_model.model.<module>.lora_A.weight [r, in], base_model.model.<module>.lora_B.weight [out, r],
                                 base_model.model.<module>.weight / .bias for every module in modules_to_save
merged weight = W + (lora_alpha / r) * B @ A      (lora_alpha / sqrt(r) with use_rslora; A, B swapped roles with fan_in_fan_out).
The merge is a one-time weight transformation at load (torch matmul in f32 on the device, like the packing of the engines); the
model then runs the unchanged kernels.  Training the LoRA matrices themselves is not built."""
from __future__ import annotations

import json
import math
import os
from typing import Dict, Optional

import torch

_PREFIX = "base_model.model."


def _load_adapter_tensors(adapter_dir: str) -> Dict[str, torch.Tensor]:
    st = os.path.join(adapter_dir, "adapter_model.safetensors")
    if os.path.exists(st):
        from safetensors.torch import load_file
        return load_file(st)
    pt = os.path.join(adapter_dir, "adapter_model.bin")
    if os.path.exists(pt):
        return torch.load(pt, map_location="cpu", weights_only=True)
    raise FileNotFoundError(f"no adapter_model.safetensors / adapter_model.bin in {adapter_dir}")


def merge_lora_state_dict(model, tensors: Dict[str, torch.Tensor], r: int, lora_alpha: float, *, use_rslora: bool = False,
                          fan_in_fan_out: bool = False, strict: bool = True) -> Dict[str, int]:
    """Fold LoRA pairs (and full copies of `modules_to_save` modules) into `model`'s parameters in place.
    -> {"merged": pairs folded, "replaced": tensors overwritten}."""
    if r <= 0:
        raise ValueError("LoRA rank r must be positive")
    scaling = float(lora_alpha) / (math.sqrt(r) if use_rslora else r)
    params = dict(model.named_parameters())
    strip = lambda k: k[len(_PREFIX):] if k.startswith(_PREFIX) else k
    pairs, full = {}, {}
    for key, t in tensors.items():
        k = strip(key).replace(".default.", ".")                      # in-memory names carry the adapter name, saved ones do not
        if ".lora_A.weight" in k:
            pairs.setdefault(k.replace(".lora_A.weight", ""), {})["A"] = t
        elif ".lora_B.weight" in k:
            pairs.setdefault(k.replace(".lora_B.weight", ""), {})["B"] = t
        elif "lora_" in k:
            raise NotImplementedError(f"adapter tensor {key}: only plain LoRA (lora_A / lora_B weights) is supported")
        else:
            full[k.replace(".modules_to_save.", ".").replace(".original_module.", ".")] = t
    merged = replaced = 0
    with torch.no_grad():
        for mod, ab in pairs.items():
            if "A" not in ab or "B" not in ab:
                raise ValueError(f"{mod}: lora_A and lora_B come in pairs")
            name = mod + ".weight"
            if name not in params:
                if strict:
                    raise KeyError(f"LoRA target {mod} is not a module of this model")
                continue
            w = params[name]
            A, B = ab["A"].to(device=w.device, dtype=torch.float32), ab["B"].to(device=w.device, dtype=torch.float32)
            if A.shape[0] != r or B.shape[1] != r:
                raise ValueError(f"{mod}: rank {A.shape[0]} / {B.shape[1]} does not match r = {r}")
            delta = (B @ A) * scaling                                     # [out, in]
            if fan_in_fan_out:
                delta = delta.t()
            if tuple(delta.shape) != tuple(w.shape):
                raise ValueError(f"{mod}: LoRA delta {tuple(delta.shape)} does not fit the weight {tuple(w.shape)}")
            w.copy_((w.float() + delta).to(w.dtype))
            merged += 1
        for name, t in full.items():
            if name not in params:
                if strict:
                    raise KeyError(f"adapter tensor {name} has no parameter in this model")
                continue
            params[name].copy_(t.to(device=params[name].device, dtype=params[name].dtype))
            replaced += 1
    for m in model.modules():                                             # packed weight copies of the engines are stale now
        inv = getattr(m, "invalidate_engine", None)
        if callable(inv):
            inv()
    return {"merged": merged, "replaced": replaced}


def load_and_merge_adapter(model, adapter_dir: str, *, strict: bool = True, config: Optional[dict] = None) -> Dict[str, int]:
    """`PeftModel.from_pretrained(model, adapter_dir).merge_and_unload()` for a LoRA checkpoint directory."""
    if config is None:
        with open(os.path.join(adapter_dir, "adapter_config.json")) as f:
            config = json.load(f)
    if str(config.get("peft_type", "LORA")).upper() != "LORA":
        raise NotImplementedError(f"peft_type {config.get('peft_type')}: only LORA adapters can be merged")
    return merge_lora_state_dict(model, _load_adapter_tensors(adapter_dir), int(config["r"]), float(config["lora_alpha"]),
                                 use_rslora=bool(config.get("use_rslora", False)), fan_in_fan_out=bool(config.get("fan_in_fan_out", False)),
                                 strict=strict)
