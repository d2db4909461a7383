"""Tower shapes and the synthetic-weight recipe shared by the numpy oracle,
the golden-vector generator (which loads the same tensors into the reference's HF modules) and the
GPU fill kernel.  Names are HuggingFace state-dict keys, i.e. the keys the reference's checkpoints
use (`esm_encoder.*`, `adapter.*`, `llama_decoder.*`, models/modeling_esm2llama_instruct.py:88-106).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field, asdict
from typing import Iterator

import numpy as np

from . import synth


@dataclass
class EsmSpec:
    num_hidden_layers: int
    hidden_size: int
    intermediate_size: int
    num_attention_heads: int
    vocab_size: int = 33
    pad_token_id: int = 1
    mask_token_id: int = 32
    layer_norm_eps: float = 1e-5
    token_dropout: bool = True
    emb_layer_norm_before: bool = False
    position_embedding_type: str = "rotary"
    rope_theta: float = 10000.0
    max_position_embeddings: int = 1026

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_attention_heads


@dataclass
class LlamaSpec:
    num_hidden_layers: int
    hidden_size: int
    intermediate_size: int
    num_attention_heads: int
    num_key_value_heads: int
    vocab_size: int = 128256
    rms_norm_eps: float = 1e-5
    rope_theta: float = 500000.0
    rope_type: str = "llama3"           # "default" | "llama3"
    rope_factor: float = 8.0
    rope_low_freq_factor: float = 1.0
    rope_high_freq_factor: float = 4.0
    rope_original_max_position_embeddings: int = 8192
    max_position_embeddings: int = 131072
    tie_word_embeddings: bool = False
    head_dim: int = 0
    qk_norm: bool = False               # Qwen3: RMSNorm over head_dim on every query / key head before the rotation

    def __post_init__(self):
        if not self.head_dim:
            self.head_dim = self.hidden_size // self.num_attention_heads


@dataclass
class AdapterSpec:
    input_dim: int
    intermediate_dim: int
    output_dim: int
    dropout_rate: float = 0.3


ESM2 = {
    "esm2_t6_8M": dict(num_hidden_layers=6, hidden_size=320, intermediate_size=1280, num_attention_heads=20),
    "esm2_t12_35M": dict(num_hidden_layers=12, hidden_size=480, intermediate_size=1920, num_attention_heads=20),
    "esm2_t30_150M": dict(num_hidden_layers=30, hidden_size=640, intermediate_size=2560, num_attention_heads=20),
    "esm2_t33_650M": dict(num_hidden_layers=33, hidden_size=1280, intermediate_size=5120, num_attention_heads=20),
    "esm2_t36_3B": dict(num_hidden_layers=36, hidden_size=2560, intermediate_size=10240, num_attention_heads=40),
}
LLAMA = {
    "Llama-3.2-1B": dict(num_hidden_layers=16, hidden_size=2048, intermediate_size=8192, num_attention_heads=32,
                         num_key_value_heads=8, rope_factor=32.0, tie_word_embeddings=True),
    "Llama-3.1-8B-Instruct": dict(num_hidden_layers=32, hidden_size=4096, intermediate_size=14336,
                                  num_attention_heads=32, num_key_value_heads=8, rope_factor=8.0),
}


# Qwen3 = the Llama block + per-head q / k RMSNorm, explicit head_dim 128, no llama3 rope scaling (the text tower the
# fork's scripts instantiate: models/esmc_config.py:9 "Qwen/Qwen3-14B"; public HF config values, restated not fetched)
LLAMA["Qwen3-14B"] = dict(num_hidden_layers=40, hidden_size=5120, intermediate_size=17408, num_attention_heads=40,
                          num_key_value_heads=8, head_dim=128, vocab_size=151936, rms_norm_eps=1e-6, rope_theta=1e6,
                          rope_type="default", rope_factor=1.0, max_position_embeddings=40960, qk_norm=True)
LLAMA["Qwen3-0.6B"] = dict(num_hidden_layers=28, hidden_size=1024, intermediate_size=3072, num_attention_heads=16,
                           num_key_value_heads=8, head_dim=128, vocab_size=151936, rms_norm_eps=1e-6, rope_theta=1e6,
                           rope_type="default", rope_factor=1.0, max_position_embeddings=40960, qk_norm=True,
                           tie_word_embeddings=True)


def esm_spec(name: str, **over) -> EsmSpec:
    return EsmSpec(**{**ESM2[name], **over})


def llama_spec(name: str, **over) -> LlamaSpec:
    return LlamaSpec(**{**LLAMA[name], **over})


def adapter_spec(esm: EsmSpec, llama: LlamaSpec, intermediate_dim: int = 2048, dropout_rate: float = 0.3):
    """scripts/train_contrast.py:153-157: input_dim = encoder width, intermediate 2048, output = LLM width."""
    return AdapterSpec(esm.hidden_size, intermediate_dim, llama.hidden_size, dropout_rate)


# BASELINE.json configs -> (esm, llama, dtype, B per GPU, T_p, T_t)   
CONFIGS = {
    "cfg1": ("esm2_t6_8M", "Llama-3.2-1B", "f32", 4, 128, 64),
    "cfg2": ("esm2_t12_35M", "Llama-3.2-1B", "bf16", 32, 512, 128),
    "cfg3": ("esm2_t36_3B", "Llama-3.1-8B-Instruct", "bf16", 16, 1024, 128),
    "cfg4": ("esm2_t36_3B", "Llama-3.1-8B-Instruct", "bf16", 32, 1024, 128),
    # configs[4]: fp8-e4m3 tower weights + GEMM operands (fp8 MFMA), bf16 activations; global batch 512 over 8 GPUs
    "cfg5": ("esm2_t36_3B", "Llama-3.1-8B-Instruct", "fp8", 64, 1024, 128),
}


# ---------------------------------------------------------------------------------------------
# synthetic weight recipe: (state-dict key, shape, uniform half-width, offset)
# ---------------------------------------------------------------------------------------------
def _lin(fan_in: int) -> float:
    return math.sqrt(3.0 / fan_in)          # uniform(-a, a) with std 1/sqrt(fan_in)


def esm_tensors(s: EsmSpec, prefix: str = "") -> Iterator[tuple[str, tuple, float, float]]:
    H, F = s.hidden_size, s.intermediate_size
    yield prefix + "embeddings.word_embeddings.weight", (s.vocab_size, H), 1.0, 0.0
    for i in range(s.num_hidden_layers):
        p = f"{prefix}encoder.layer.{i}."
        for n in ("query", "key", "value"):
            yield p + f"attention.self.{n}.weight", (H, H), _lin(H), 0.0
            yield p + f"attention.self.{n}.bias", (H,), 0.1, 0.0
        yield p + "attention.output.dense.weight", (H, H), _lin(H), 0.0
        yield p + "attention.output.dense.bias", (H,), 0.1, 0.0
        yield p + "attention.LayerNorm.weight", (H,), 0.1, 1.0
        yield p + "attention.LayerNorm.bias", (H,), 0.1, 0.0
        yield p + "intermediate.dense.weight", (F, H), _lin(H), 0.0
        yield p + "intermediate.dense.bias", (F,), 0.1, 0.0
        yield p + "output.dense.weight", (H, F), _lin(F), 0.0
        yield p + "output.dense.bias", (H,), 0.1, 0.0
        yield p + "LayerNorm.weight", (H,), 0.1, 1.0
        yield p + "LayerNorm.bias", (H,), 0.1, 0.0
    yield prefix + "encoder.emb_layer_norm_after.weight", (H,), 0.1, 1.0
    yield prefix + "encoder.emb_layer_norm_after.bias", (H,), 0.1, 0.0


def llama_tensors(s: LlamaSpec, prefix: str = "", layers: int | None = None,
                  lm_head: bool = True) -> Iterator[tuple[str, tuple, float, float]]:
    H, F, d = s.hidden_size, s.intermediate_size, s.head_dim
    nq, nkv = s.num_attention_heads * d, s.num_key_value_heads * d
    yield prefix + "model.embed_tokens.weight", (s.vocab_size, H), 1.0, 0.0
    for i in range(s.num_hidden_layers if layers is None else layers):
        p = f"{prefix}model.layers.{i}."
        yield p + "self_attn.q_proj.weight", (nq, H), _lin(H), 0.0
        yield p + "self_attn.k_proj.weight", (nkv, H), _lin(H), 0.0
        yield p + "self_attn.v_proj.weight", (nkv, H), _lin(H), 0.0
        yield p + "self_attn.o_proj.weight", (H, nq), _lin(nq), 0.0
        if s.qk_norm:
            yield p + "self_attn.q_norm.weight", (d,), 0.2, 1.0
            yield p + "self_attn.k_norm.weight", (d,), 0.2, 1.0
        yield p + "mlp.gate_proj.weight", (F, H), _lin(H), 0.0
        yield p + "mlp.up_proj.weight", (F, H), _lin(H), 0.0
        yield p + "mlp.down_proj.weight", (H, F), _lin(F), 0.0
        yield p + "input_layernorm.weight", (H,), 0.1, 1.0
        yield p + "post_attention_layernorm.weight", (H,), 0.1, 1.0
    yield prefix + "model.norm.weight", (H,), 0.1, 1.0
    if lm_head and not s.tie_word_embeddings:
        yield prefix + "lm_head.weight", (s.vocab_size, H), _lin(H), 0.0


def adapter_tensors(s: AdapterSpec, prefix: str = "") -> Iterator[tuple[str, tuple, float, float]]:
    yield prefix + "fc1.weight", (s.intermediate_dim, s.input_dim), _lin(s.input_dim), 0.0
    yield prefix + "fc1.bias", (s.intermediate_dim,), 0.1, 0.0
    yield prefix + "fc2.weight", (s.output_dim, s.intermediate_dim), _lin(s.intermediate_dim), 0.0
    yield prefix + "fc2.bias", (s.output_dim,), 0.1, 0.0
    # ln1/ln2 are constructed but never used by forward (modeling_esm2llama_instruct.py:56-57)
    yield prefix + "ln1.weight", (s.intermediate_dim,), 0.0, 1.0
    yield prefix + "ln1.bias", (s.intermediate_dim,), 0.0, 0.0
    yield prefix + "ln2.weight", (s.output_dim,), 0.0, 1.0
    yield prefix + "ln2.bias", (s.output_dim,), 0.0, 0.0


def materialize(tensors, seed: int = 0, skip=()) -> dict[str, np.ndarray]:
    """numpy fp32 state dict for an iterator of (name, shape, scale, offset)."""
    out = {}
    for name, shape, scale, offset in tensors:
        if any(name.endswith(s) for s in skip):
            continue
        if scale == 0.0:
            out[name] = np.full(shape, offset, dtype=np.float32)
        else:
            out[name] = synth.uniform_f32(seed, name, shape, scale, offset)
    return out


# ---------------------------------------------------------------------------------------------
# algorithmic FLOPs per sample -- the roofline numerator
# ---------------------------------------------------------------------------------------------
def flops_per_sample(esm: EsmSpec, llama: LlamaSpec, ad: AdapterSpec, t_p: int, t_t: int,
                     llama_layers: int = 16, backward: bool = True) -> dict[str, float]:
    He, Fe, Le = esm.hidden_size, esm.intermediate_size, esm.num_hidden_layers
    f_esm = Le * t_p * (2 * (4 * He * He + 2 * He * Fe) + 4 * t_p * He)
    Hl, Fl, d = llama.hidden_size, llama.intermediate_size, llama.head_dim
    kv = llama.num_key_value_heads
    n = min(llama_layers, llama.num_hidden_layers)
    f_llama = n * t_t * (2 * (2 * Hl * Hl + 2 * Hl * kv * d + 3 * Hl * Fl) + 2 * (t_t + 1) * Hl)
    I = ad.intermediate_dim
    f_ad = 2 * t_p * (He * I + I * Hl)
    if backward:
        f_ad += 2 * t_p * He * I + 4 * t_p * I * Hl
    return {"esm": float(f_esm), "llama": float(f_llama), "adapter": float(f_ad),
            "total": float(f_esm + f_llama + f_ad)}


def spec_dict(x) -> dict:
    return asdict(x)
