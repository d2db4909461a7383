"""Config objects of the drop-in surface.

Same names, constructor arguments, defaults and `model_type` strings as the reference's
models/modality_config.py:2-18 (ModalityAdapterConfig) and
models/configuration_esm2llama_instruct.py:12-33 (Esm2LlamaInstructConfig), so configs and
checkpoints written by either side load in the other.  Sub-configs are the plain HuggingFace
`EsmConfig` / `LlamaConfig` containers (no modelling code is taken from transformers).
"""
from __future__ import annotations

from transformers import EsmConfig, LlamaConfig, PretrainedConfig

from . import specs


class ModalityAdapterConfig(PretrainedConfig):
    model_type = "modality_adapter"

    def __init__(self, input_dim: int = 0, intermediate_dim: int = 0, output_dim: int = 0,
                 dropout_rate: float = 0.3, **kwargs):
        super().__init__(**kwargs)
        self.input_dim = int(input_dim)
        self.intermediate_dim = int(intermediate_dim)
        self.output_dim = int(output_dim)
        self.dropout_rate = float(dropout_rate)

    def to_spec(self) -> specs.AdapterSpec:
        return specs.AdapterSpec(self.input_dim, self.intermediate_dim, self.output_dim, self.dropout_rate)


class Esm2LlamaInstructConfig(PretrainedConfig):
    """esm_config + adapter_config + llama_config and the protein placeholder token id
    (128003 = <|reserved_special_token_1|>, reference configuration_esm2llama_instruct.py:26)."""
    model_type = "esm2llama_instruct"

    def __init__(self, esm_config: EsmConfig = None, adapter_config: ModalityAdapterConfig = None,
                 llama_config: LlamaConfig = None, placeholder_id: int = 128003, **kwargs):
        super().__init__(**kwargs)
        # `save_pretrained` writes the sub-configs as nested dicts (PretrainedConfig.to_dict); `from_pretrained` hands
        # them back as dicts -- rebuilt into config objects here so a saved config round-trips
        if isinstance(esm_config, dict):
            esm_config = EsmConfig(**{k: v for k, v in esm_config.items() if k != "model_type"})
        if isinstance(adapter_config, dict):
            adapter_config = ModalityAdapterConfig(**{k: v for k, v in adapter_config.items() if k != "model_type"})
        if isinstance(llama_config, dict):
            if llama_config.get("model_type") == "qwen3":
                from transformers import Qwen3Config
                llama_config = Qwen3Config(**{k: v for k, v in llama_config.items() if k != "model_type"})
            else:
                llama_config = LlamaConfig(**{k: v for k, v in llama_config.items() if k != "model_type"})
        self.esm_config = esm_config
        self.adapter_config = adapter_config
        self.llama_config = llama_config
        self.placeholder_id = placeholder_id


# ---------------------------------------------------------------------------------------------
# HF config <-> tower spec (the spec is what the packer and the C structs are filled from)
# ---------------------------------------------------------------------------------------------
def esm_spec_from_config(c: EsmConfig) -> specs.EsmSpec:
    pe = getattr(c, "position_embedding_type", "absolute")
    if pe != "rotary":
        raise ValueError(f"ESM position_embedding_type={pe!r}: only rotary ESM2 checkpoints are supported")
    if getattr(c, "is_decoder", False) or getattr(c, "add_cross_attention", False):
        raise ValueError("ESM decoder / cross-attention configurations are not on this path")
    return specs.EsmSpec(num_hidden_layers=c.num_hidden_layers, hidden_size=c.hidden_size,
                         intermediate_size=c.intermediate_size, num_attention_heads=c.num_attention_heads,
                         vocab_size=c.vocab_size, pad_token_id=c.pad_token_id, mask_token_id=c.mask_token_id,
                         layer_norm_eps=c.layer_norm_eps, token_dropout=bool(c.token_dropout),
                         emb_layer_norm_before=bool(c.emb_layer_norm_before), position_embedding_type=pe,
                         rope_theta=float(getattr(c, "rope_theta", None) or 10000.0),
                         max_position_embeddings=c.max_position_embeddings)


def esm_config_from_spec(s: specs.EsmSpec) -> EsmConfig:
    return EsmConfig(vocab_size=s.vocab_size, hidden_size=s.hidden_size, num_hidden_layers=s.num_hidden_layers,
                     num_attention_heads=s.num_attention_heads, intermediate_size=s.intermediate_size,
                     hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0,
                     max_position_embeddings=s.max_position_embeddings, layer_norm_eps=s.layer_norm_eps,
                     position_embedding_type=s.position_embedding_type, token_dropout=s.token_dropout,
                     emb_layer_norm_before=s.emb_layer_norm_before, pad_token_id=s.pad_token_id,
                     mask_token_id=s.mask_token_id)


def _rope_dict(c: LlamaConfig) -> dict:
    rp = getattr(c, "rope_parameters", None)
    if rp is None:                                  # transformers 4.x spelling
        rp = dict(getattr(c, "rope_scaling", None) or {})
        rp.setdefault("rope_theta", getattr(c, "rope_theta", 10000.0))
    rp = dict(rp)
    rp.setdefault("rope_type", rp.pop("type", "default") if "type" in rp else "default")
    return rp


def llama_spec_from_config(c: LlamaConfig) -> specs.LlamaSpec:
    """LlamaConfig, or Qwen3Config (model_type "qwen3": the same block plus q_norm / k_norm, models/esmc_qwen_arc.py's LLM)."""
    rp = _rope_dict(c)
    mt = getattr(c, "model_type", "llama")
    if mt not in ("llama", "qwen3"):
        raise ValueError(f"decoder model_type={mt!r} is not supported (llama, qwen3)")
    if mt == "qwen3" and (getattr(c, "use_sliding_window", False) or any(t != "full_attention" for t in (getattr(c, "layer_types", None) or []))):
        raise ValueError("Qwen3 sliding-window layers are not supported")
    if rp["rope_type"] not in ("default", "llama3"):
        raise ValueError(f"Llama rope_type={rp['rope_type']!r} is not supported (default, llama3)")
    if getattr(c, "attention_bias", False) or getattr(c, "mlp_bias", False):
        raise ValueError("Llama attention_bias / mlp_bias are not supported")
    return specs.LlamaSpec(num_hidden_layers=c.num_hidden_layers, hidden_size=c.hidden_size,
                           intermediate_size=c.intermediate_size, num_attention_heads=c.num_attention_heads,
                           num_key_value_heads=c.num_key_value_heads or c.num_attention_heads,
                           vocab_size=c.vocab_size, rms_norm_eps=c.rms_norm_eps, rope_theta=float(rp["rope_theta"]),
                           rope_type=rp["rope_type"], rope_factor=float(rp.get("factor", 1.0)),
                           rope_low_freq_factor=float(rp.get("low_freq_factor", 1.0)),
                           rope_high_freq_factor=float(rp.get("high_freq_factor", 4.0)),
                           rope_original_max_position_embeddings=int(rp.get("original_max_position_embeddings", 8192)),
                           max_position_embeddings=c.max_position_embeddings,
                           tie_word_embeddings=bool(getattr(c, "tie_word_embeddings", False)),
                           head_dim=getattr(c, "head_dim", None) or c.hidden_size // c.num_attention_heads,
                           qk_norm=mt == "qwen3")


def llama_config_from_spec(s: specs.LlamaSpec) -> LlamaConfig:
    rope = {"rope_type": s.rope_type, "rope_theta": s.rope_theta}
    if s.qk_norm:
        from transformers import Qwen3Config
        return Qwen3Config(vocab_size=s.vocab_size, hidden_size=s.hidden_size, intermediate_size=s.intermediate_size,
                           num_hidden_layers=s.num_hidden_layers, num_attention_heads=s.num_attention_heads,
                           num_key_value_heads=s.num_key_value_heads, head_dim=s.head_dim, rms_norm_eps=s.rms_norm_eps,
                           max_position_embeddings=s.max_position_embeddings, rope_parameters=rope,
                           tie_word_embeddings=s.tie_word_embeddings, attention_bias=False, attention_dropout=0.0,
                           use_sliding_window=False, pad_token_id=None, bos_token_id=None, eos_token_id=None)
    if s.rope_type == "llama3":
        rope.update(factor=s.rope_factor, low_freq_factor=s.rope_low_freq_factor,
                    high_freq_factor=s.rope_high_freq_factor,
                    original_max_position_embeddings=s.rope_original_max_position_embeddings)
    return LlamaConfig(vocab_size=s.vocab_size, hidden_size=s.hidden_size, intermediate_size=s.intermediate_size,
                       num_hidden_layers=s.num_hidden_layers, num_attention_heads=s.num_attention_heads,
                       num_key_value_heads=s.num_key_value_heads, rms_norm_eps=s.rms_norm_eps,
                       max_position_embeddings=s.max_position_embeddings, rope_parameters=rope,
                       tie_word_embeddings=s.tie_word_embeddings, attention_bias=False, mlp_bias=False,
                       attention_dropout=0.0, pad_token_id=None, bos_token_id=None, eos_token_id=None)
