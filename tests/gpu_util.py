"""Helpers for the -m gpu tests: numpy <-> device tensors, bf16 rounding, model construction."""
import numpy as np
import torch

from p2t_hip import specs, synth


def dev():
    return torch.device("cuda:0")


def to_dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(dev())
    return t.to(dtype) if dtype is not None else t


def to_np(t):
    return t.detach().float().cpu().numpy() if t.dtype in (torch.bfloat16, torch.float16) else t.detach().cpu().numpy()


def bf16r(a):
    return synth.bf16_round(np.asarray(a, dtype=np.float32))


def rel(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def maxabs(a, b):
    return float(np.max(np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64))))


def rnd(seed, name, shape, scale=1.0, offset=0.0):
    return synth.uniform_f32(seed, name, shape, scale, offset)


def build_model(esm, llama, ad, dtype, seed=0, adapter_dtype=None):
    from p2t_hip import Esm2LlamaInstructForCausalLM
    return Esm2LlamaInstructForCausalLM.from_specs(esm, llama, ad, dtype=dtype, device=dev(), seed=seed,
                                                   adapter_dtype=adapter_dtype)
