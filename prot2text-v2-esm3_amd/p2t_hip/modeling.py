"""Host-side mirror of the reference's model classes for the contrastive path.

Same class names, constructor conventions, attribute names (`esm_encoder`, `adapter`,
`llama_decoder`), forward kwargs / early-exit flags and state-dict keys as the reference's
models/modeling_esm2llama_instruct.py:45-268, with the arithmetic delegated to libp2t_hip.so:

    Esm2LlamaInstructForCausalLM.forward(protein_input_ids, protein_attention_mask,
        return_encoder_outputs=True)   -> p2t_esm2_forward            (reference :175-189)
        return_adapter_outputs=True    -> + p2t_adapter_forward       (reference :191-193)
    model.llama_decoder.model(input_ids, attention_mask, output_hidden_states=True).hidden_states[k]
                                       -> p2t_llama_hidden_forward    (scripts/train_contrast.py:292-304)

The modules below are PARAMETER CONTAINERS with HuggingFace key names (so reference checkpoints
load with load_state_dict) plus an "engine": packed GEMM-layout copies of the weights and the C
structs that point at them.  No torch op computes anything on this path.  After the adapter exit the
forward continues as upstream (placeholder scatter -> full decoder -> LM head -> shifted cross-entropy,
SURVEY.md section 8f row 3) as a forward-only path; generation and the decoder backward are out of
scope and raise NotImplementedError.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Optional

import torch
from torch import nn
from transformers import PreTrainedModel
from transformers.modeling_outputs import BaseModelOutputWithPoolingAndCrossAttentions

from . import _lib, ops, specs
from ._lib import call
from .configuration import (Esm2LlamaInstructConfig, ModalityAdapterConfig, esm_config_from_spec,
                            esm_spec_from_config, llama_config_from_spec, llama_spec_from_config)
from .ops import ptr, round_up, stream


# ---------------------------------------------------------------------------------------------
# parameter trees with HuggingFace key names
# ---------------------------------------------------------------------------------------------
def _build_tree(root: nn.Module, tensors, dtype, device, requires_grad=False):
    for name, shape, _scale, _offset in tensors:
        *path, leaf = name.split(".")
        mod = root
        for p in path:
            if p not in mod._modules:
                mod.add_module(p, nn.Module())
            mod = mod._modules[p]
        mod.register_parameter(leaf, nn.Parameter(torch.empty(shape, dtype=dtype, device=device),
                                                  requires_grad=requires_grad))


def _init_tree(root: nn.Module, std: float = 0.02):
    """Random init in the spirit of HF `_init_weights`: matrices of standard deviation std, zero biases, unit norms.  On the
    GPU the matrices come from the library's own counter-hash fill (one pass at HBM rate, no ATen RNG kernels: a 3B + 8B model
    initialises in a fraction of a second and a rocprof trace of a run holds this library's kernels only)."""
    with torch.no_grad():
        for name, p in root.named_parameters():
            if p.dim() == 2 and p.is_cuda:
                ops.fill_hash_(p.data, 0x1217, "init." + name, std * 3.0 ** 0.5)        # uniform(-a, a) has std a / sqrt(3)
            elif p.dim() == 2:
                p.normal_(0.0, std)
            elif name.endswith("bias"):
                p.zero_()
            else:
                p.fill_(1.0)


def _fill_synthetic(root: nn.Module, tensors, seed: int, prefix: str = ""):
    params = dict(root.named_parameters())
    with torch.no_grad():
        for name, shape, scale, offset in tensors:
            p = params[name]
            assert tuple(p.shape) == tuple(shape), (name, p.shape, shape)
            ops.fill_hash_(p.data, seed, prefix + name, scale, offset)


def _pad_cols(w: torch.Tensor, ld: int, dtype: torch.dtype) -> torch.Tensor:
    """[N, K] -> [N, ld] (zero padded) in `dtype`; aliases the parameter when nothing changes."""
    w = w.detach()
    if w.shape[1] == ld and w.dtype == dtype and w.is_contiguous():
        return w
    out = torch.zeros((w.shape[0], ld), dtype=dtype, device=w.device)
    out[:, : w.shape[1]].copy_(w)
    return out


def _quant_fp8(w: torch.Tensor, k: int):
    """Packed [N, ld] weight matrix (model dtype) -> (e4m3 bytes [N, round_up(k, 128)], E8M0 scale byte per output channel)."""
    if w.dtype not in (torch.bfloat16, torch.float32):
        raise ValueError("fp8 GEMM weights are quantised from bf16 / f32 masters")
    return ops.quant_rows_fp8(w.contiguous(), cols=k, ld_q=round_up(k, 128))


def _f32(v: torch.Tensor) -> torch.Tensor:
    v = v.detach()
    return v if v.dtype == torch.float32 and v.is_contiguous() else v.float().contiguous()


class _Workspace:
    """uint8 workspace tensors cached per key (the HIP stream), grow-only, reused across calls and shapes."""

    def __init__(self):
        self._cache = {}

    def get(self, key, nbytes: int, device) -> torch.Tensor:
        t = self._cache.get(key)
        if t is None or t.numel() < nbytes or t.device != device:
            t = torch.empty((nbytes,), dtype=torch.uint8, device=device)
            self._cache[key] = t
        return t


# ---------------------------------------------------------------------------------------------
# ESM2 encoder
# ---------------------------------------------------------------------------------------------
class EsmEncoder(nn.Module):
    """Stands where the reference puts HF `EsmModel(config, add_pooling_layer=False)`
    (models/modeling_esm2llama_instruct.py:90-93).  forward = HF EsmModel.forward
    (transformers/models/esm/modeling_esm.py:685-755) on the HIP path."""

    def __init__(self, config, dtype=torch.float32, device="cuda"):
        super().__init__()
        self.config = config
        self.spec = esm_spec_from_config(config)
        s = self.spec
        if s.hidden_size % s.num_attention_heads:
            raise ValueError(f"The hidden size ({s.hidden_size}) is not a multiple of the number of attention heads "
                             f"({s.num_attention_heads})")
        _build_tree(self, specs.esm_tensors(s), dtype, device)
        # HF EsmModel also owns a contact head; kept so state dicts line up (unused on this path)
        self.add_module("contact_head", nn.Module())
        self.contact_head.add_module("regression", nn.Module())
        reg = self.contact_head.regression
        reg.register_parameter("weight", nn.Parameter(torch.zeros((1, s.num_hidden_layers * s.num_attention_heads),
                                                                  dtype=dtype, device=device), requires_grad=False))
        reg.register_parameter("bias", nn.Parameter(torch.zeros((1,), dtype=dtype, device=device), requires_grad=False))
        self.add_module("rotary_embeddings", nn.Module())
        d = s.head_dim
        inv = 1.0 / (s.rope_theta ** (torch.arange(0, d, 2, dtype=torch.float) / d))     # modeling_esm.py:127-138
        self.rotary_embeddings.register_buffer("inv_freq", inv.to(device))
        _init_tree(self)
        self._engine = None
        self.gemm_fp8 = False            # True: the four projections of every layer run on the fp8 MFMA kernel
        self.fp8_fused_gelu = True       # fp8: FFN-up writes e4m3 directly under the per-token bound scale (False: bf16 + quantise pass)
        self._ws = _Workspace()
        self._register_load_state_dict_pre_hook(self._remap_legacy_inv_freq)
        self.register_load_state_dict_post_hook(lambda module, _incompatible: module.invalidate_engine())

    @staticmethod
    def _remap_legacy_inv_freq(state_dict, prefix, *args):
        """Old ESM2 checkpoints store inv_freq per attention layer; keep the checkpoint's values (they may
        have been saved in fp16) under the model-level key, as HF does (modeling_esm.py:654-674)."""
        new_key = f"{prefix}rotary_embeddings.inv_freq"
        old = sorted(k for k in list(state_dict) if k.startswith(prefix) and k.endswith(".attention.self.rotary_embeddings.inv_freq"))
        if new_key not in state_dict and old:
            state_dict[new_key] = state_dict[old[0]]
        for k in old:
            del state_dict[k]

    @property
    def dtype(self):
        return self.embeddings.word_embeddings.weight.dtype

    def invalidate_engine(self):
        self._engine = None

    def _build_engine(self):
        s, dt = self.spec, self.dtype
        H, F = s.hidden_size, s.intermediate_size
        Hp, Fp = round_up(H, 64), round_up(F, 64)
        keep, layers = [], (_lib.EsmLayerC * s.num_hidden_layers)()
        P = dict(self.named_parameters())
        for i in range(s.num_hidden_layers):
            p = f"encoder.layer.{i}."
            qkv_w = _pad_cols(torch.cat([P[p + f"attention.self.{n}.weight"].detach() for n in ("query", "key", "value")], 0), Hp, dt)
            qkv_b = torch.cat([P[p + f"attention.self.{n}.bias"].detach().float() for n in ("query", "key", "value")], 0).contiguous()
            t = dict(qkv_w=qkv_w, qkv_b=qkv_b,
                     o_w=_pad_cols(P[p + "attention.output.dense.weight"], Hp, dt), o_b=_f32(P[p + "attention.output.dense.bias"]),
                     ln1_w=_f32(P[p + "attention.LayerNorm.weight"]), ln1_b=_f32(P[p + "attention.LayerNorm.bias"]),
                     fc1_w=_pad_cols(P[p + "intermediate.dense.weight"], Hp, dt), fc1_b=_f32(P[p + "intermediate.dense.bias"]),
                     fc2_w=_pad_cols(P[p + "output.dense.weight"], Fp, dt), fc2_b=_f32(P[p + "output.dense.bias"]),
                     ln2_w=_f32(P[p + "LayerNorm.weight"]), ln2_b=_f32(P[p + "LayerNorm.bias"]))
            bounds = None
            if self.gemm_fp8:           # e4m3 bytes + one E8M0 scale per output channel (include/p2t_hip.h, p2t_esm2_layer)
                # bound of the FFN-up pre-activation (Cauchy-Schwarz): max_n ||W_n||_2 -- inflated by (1 + 2^-4)^2 because the
                # GEMM multiplies e4m3-rounded operands, each element up to 2^-4 larger than its source -- and max_n |b_n|
                bounds = (float(t["fc1_w"][:, :H].float().norm(dim=1).max()) * (1.0 + 2.0 ** -4) ** 2, float(t["fc1_b"].abs().max()))
                for name, kdim in (("qkv", H), ("o", H), ("fc1", H), ("fc2", F)):
                    t[name + "_w"], t[name + "_ws"] = _quant_fp8(t[name + "_w"], kdim)
            keep.append(t)
            for k, v in t.items():
                setattr(layers[i], k, v.data_ptr())
            if bounds is not None and self.fp8_fused_gelu:
                layers[i].fc1_wnorm_bound, layers[i].fc1_babs_bound = bounds
        fw, fb = _f32(P["encoder.emb_layer_norm_after.weight"]), _f32(P["encoder.emb_layer_norm_after.bias"])
        emb = P["embeddings.word_embeddings.weight"].detach().contiguous()
        inv = self.rotary_embeddings.inv_freq.detach().float().contiguous()
        keep += [fw, fb, emb, inv]
        w = _lib.EsmWeightsC(word_emb=emb.data_ptr(), emb_ln_w=None, emb_ln_b=None,
                             layers=C.cast(layers, C.POINTER(_lib.EsmLayerC)),
                             final_ln_w=fw.data_ptr(), final_ln_b=fb.data_ptr(), inv_freq=inv.data_ptr())
        cfg = _lib.EsmConfigC(n_layers=s.num_hidden_layers, hidden=H, ffn=F, heads=s.num_attention_heads,
                              head_dim=s.head_dim, vocab=s.vocab_size, pad_id=s.pad_token_id, mask_id=s.mask_token_id,
                              token_dropout=int(s.token_dropout), emb_layer_norm_before=int(s.emb_layer_norm_before),
                              layer_norm_eps=s.layer_norm_eps, rope_theta=s.rope_theta, dtype=ops.dt_of(dt),
                              gemm_fp8=int(self.gemm_fp8))
        if self.gemm_fp8 and dt != torch.bfloat16:
            raise ValueError("fp8 GEMMs need a bf16 model (activations and attention stay bf16)")
        self._engine = dict(cfg=cfg, w=w, layers=layers, keep=keep, Hp=Hp)
        return self._engine

    def ensure_engine(self):
        """Build the packed weight copies now, on the current stream (callers that fan out to side streams do this first)."""
        return self._engine or self._build_engine()

    def encode(self, input_ids: torch.Tensor, attention_mask: Optional[torch.Tensor]) -> torch.Tensor:
        """-> padded last_hidden_state [B, T, Hp] (columns >= hidden are zero; the adapter reads it as is)."""
        if input_ids is None or input_ids.dim() != 2:
            raise ValueError("protein_input_ids must be a [batch, seq_len] tensor of token ids")
        B, T = input_ids.shape
        if attention_mask is None:
            attention_mask = torch.ones_like(input_ids)
        if tuple(attention_mask.shape) != (B, T):
            raise ValueError(f"protein_attention_mask shape {tuple(attention_mask.shape)} != input ids {(B, T)}")
        e = self._engine or self._build_engine()
        dev = self.embeddings.word_embeddings.weight.device
        ids = input_ids.to(device=dev, dtype=torch.int64).contiguous()
        mask = attention_mask.to(device=dev, dtype=torch.int64).contiguous()
        nbytes = call("p2t_esm2_workspace_bytes", C.byref(e["cfg"]), B, T)
        ws = self._ws.get(torch.cuda.current_stream().cuda_stream, nbytes, dev)     # one grow-only workspace per stream
        out = torch.empty((B, T, e["Hp"]), dtype=self.dtype, device=dev)
        call("p2t_esm2_forward", C.byref(e["cfg"]), C.byref(e["w"]), ptr(ids), ptr(mask), B, T, ptr(out), e["Hp"],
             ptr(ws), ws.numel(), stream())
        return out

    def forward(self, input_ids=None, attention_mask=None, position_ids=None, head_mask=None, inputs_embeds=None,
                use_cache=None, output_attentions=None, output_hidden_states=None, return_dict=None, **kwargs):
        if inputs_embeds is not None or position_ids is not None or head_mask is not None:
            raise NotImplementedError("protein_inputs_embeds / protein_position_ids / protein_head_mask are not "
                                      "supported on the HIP path (the contrastive stage never passes them)")
        if output_attentions or output_hidden_states:
            raise NotImplementedError("output_attentions / output_hidden_states are not available from the fused encoder")
        h = self.encode(input_ids, attention_mask)[:, :, : self.spec.hidden_size]
        if return_dict is False:
            return (h,)
        return BaseModelOutputWithPoolingAndCrossAttentions(last_hidden_state=h, pooler_output=None)


# ---------------------------------------------------------------------------------------------
# ModalityAdapter
# ---------------------------------------------------------------------------------------------
class _AdapterFn(torch.autograd.Function):
    """y = normalize(drop(gelu(fc2(drop(gelu(fc1(x))))))) with the hand-written backward
    (p2t_adapter_forward / p2t_adapter_backward).  Gradients: fc1/fc2 weight and bias only -- the
    encoder is frozen on this path (scripts/train_contrast.py:186)."""

    @staticmethod
    def forward(ctx, x_pad, w1, b1, w2, b2, adapter, use_dropout, need_grad):
        M = x_pad.shape[0]
        c = adapter.config
        dt = x_pad.dtype
        I, O = c.intermediate_dim, c.output_dim
        ld1, ld2 = round_up(I, 64), round_up(O, 64)
        dev = x_pad.device
        p = float(adapter.dropout.p) if use_dropout else 0.0
        seed = adapter._next_seed() if p > 0.0 else 0
        cfg = _lib.AdapterConfigC(input_dim=c.input_dim, intermediate_dim=I, output_dim=O, dropout_p=p,
                                  dropout_seed=seed, dtype=ops.dt_of(dt))
        K1 = round_up(c.input_dim, 64)
        w1p, w2p = _pad_cols(w1, K1, dt), _pad_cols(w2, ld1, dt)
        b1f, b2f = _f32(b1), _f32(b2)
        wts = _lib.AdapterWeightsC(fc1_w=w1p.data_ptr(), fc1_b=b1f.data_ptr(), fc2_w=w2p.data_ptr(), fc2_b=b2f.data_ptr())
        h1 = torch.empty((M, ld1), dtype=dt, device=dev)
        g2 = torch.empty((M, ld2), dtype=dt, device=dev)
        z1 = torch.empty((M, ld1), dtype=dt, device=dev) if need_grad else None
        z2 = torch.empty((M, ld2), dtype=dt, device=dev) if need_grad else None
        inv = torch.empty((M,), dtype=torch.float32, device=dev) if need_grad else None
        saved = _lib.AdapterSavedC(z1=ops.ptr(z1), h1=h1.data_ptr(), z2=ops.ptr(z2), g2=g2.data_ptr(), inv_norm=ops.ptr(inv))
        y = torch.empty((M, ld2), dtype=dt, device=dev)
        call("p2t_adapter_forward", C.byref(cfg), C.byref(wts), ptr(x_pad), x_pad.stride(0), M, ptr(y), C.byref(saved), stream())
        if need_grad:
            ctx.cfg, ctx.keep = cfg, (x_pad, w1p, b1f, w2p, b2f, z1, h1, z2, g2, inv)
            ctx.shapes = (w1.shape, b1.shape, w2.shape, b2.shape, w1.dtype)
        ctx.need_grad = need_grad
        return y[:, :O]

    @staticmethod
    def backward(ctx, dy):
        if not ctx.need_grad:
            return (None,) * 8
        x_pad, w1p, b1f, w2p, b2f, z1, h1, z2, g2, inv = ctx.keep
        cfg = ctx.cfg
        M, dev = x_pad.shape[0], x_pad.device
        wts = _lib.AdapterWeightsC(fc1_w=w1p.data_ptr(), fc1_b=b1f.data_ptr(), fc2_w=w2p.data_ptr(), fc2_b=b2f.data_ptr())
        saved = _lib.AdapterSavedC(z1=z1.data_ptr(), h1=h1.data_ptr(), z2=z2.data_ptr(), g2=g2.data_ptr(), inv_norm=inv.data_ptr())
        s1, sb1, s2, sb2, pdt = ctx.shapes
        g = [torch.empty(s, dtype=torch.float32, device=dev) for s in (s1, sb1, s2, sb2)]
        nbytes = call("p2t_adapter_backward_workspace_bytes", C.byref(cfg), M)
        ws = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
        dyf = dy.float().contiguous() if (dy.dtype != torch.float32 or not dy.is_contiguous()) else dy
        call("p2t_adapter_backward", C.byref(cfg), C.byref(wts), ptr(x_pad), x_pad.stride(0), M, C.byref(saved), ptr(dyf),
             ptr(g[0]), ptr(g[1]), ptr(g[2]), ptr(g[3]), 0, ptr(ws), ws.numel(), stream())
        g = [t if pdt == torch.float32 else ops.cast(t, pdt) for t in g]
        return None, g[0], g[1], g[2], g[3], None, None, None


class ModalityAdapter(nn.Module):
    """2-layer adapter, reference models/modeling_esm2llama_instruct.py:45-68.  `ln1`/`ln2` are
    constructed (and saved) but unused, exactly as upstream (":56-57 DEPRECATED")."""
    config_class = ModalityAdapterConfig

    def __init__(self, config: ModalityAdapterConfig, dtype=torch.float32, device="cuda"):
        super().__init__()
        self.config = config
        kw = dict(dtype=dtype, device=device)
        self.fc1 = nn.Linear(config.input_dim, config.intermediate_dim, **kw)
        self.fc2 = nn.Linear(config.intermediate_dim, config.output_dim, **kw)
        self.activation = nn.GELU()
        self.dropout = nn.Dropout(p=config.dropout_rate)
        self.ln1 = nn.LayerNorm(config.intermediate_dim, **kw)
        self.ln2 = nn.LayerNorm(config.output_dim, **kw)
        with torch.no_grad():
            for lin in (self.fc1, self.fc2):
                lin.weight.normal_(0.0, 0.02)
                lin.bias.zero_()
        self._seed_state = 0x5DEECE66D
        self._calls = 0

    def _next_seed(self) -> int:
        self._calls += 1
        return (self._seed_state + 0x9E3779B97F4A7C15 * self._calls) & ((1 << 64) - 1)

    def manual_seed(self, seed: int):
        self._seed_state, self._calls = int(seed) & ((1 << 64) - 1), 0

    def forward_padded(self, x_pad: torch.Tensor) -> torch.Tensor:
        """x_pad: [M, ld >= round_up(input_dim, 64)] with zero padding -> [M, output_dim]."""
        params = (self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias)
        need_grad = torch.is_grad_enabled() and any(p.requires_grad for p in params)
        return _AdapterFn.apply(x_pad, *params, self, bool(self.training), need_grad)

    def forward(self, hidden_states: torch.Tensor) -> torch.Tensor:
        c = self.config
        if hidden_states.shape[-1] != c.input_dim:
            raise ValueError(f"adapter input has {hidden_states.shape[-1]} features, config.input_dim={c.input_dim}")
        lead = hidden_states.shape[:-1]
        x = hidden_states.reshape(-1, c.input_dim)
        K1 = round_up(c.input_dim, 64)
        wd = self.fc1.weight.dtype
        if x.dtype != wd or x.shape[1] != K1 or x.stride(1) != 1 or x.stride(0) < K1:
            xp = torch.zeros((x.shape[0], K1), dtype=wd, device=self.fc1.weight.device)
            xp[:, : c.input_dim].copy_(x)
            x = xp
        return self.forward_padded(x).reshape(*lead, c.output_dim)


# ---------------------------------------------------------------------------------------------
# Llama text tower
# ---------------------------------------------------------------------------------------------
class LazyHiddenStates:
    """`outputs.hidden_states` of the text tower: indexing [k] runs the decoder up to layer k only
    (the reference materialises all L+1 states and reads index 16, train_contrast.py:294-304)."""

    def __init__(self, model: "LlamaTextModel", ids, mask, embeds=None):
        self._m, self._ids, self._mask, self._embeds, self._cache = model, ids, mask, embeds, {}

    def __len__(self):
        return self._m.spec.num_hidden_layers + 1

    def __getitem__(self, k):
        if isinstance(k, slice):
            return tuple(self[i] for i in range(*k.indices(len(self))))
        if k < 0:
            k += len(self)
        if not 0 <= k < len(self):
            raise IndexError(f"hidden_states index {k} out of range for {len(self) - 1} layers")
        if k not in self._cache:
            self._cache[k] = (self._m.hidden_state(self._ids, self._mask, k) if self._embeds is None
                              else self._m.hidden_state_from_embeds(self._embeds, self._mask, k))
        return self._cache[k]

    def __iter__(self):
        return (self[i] for i in range(len(self)))


class LlamaTextOutput:
    def __init__(self, hidden_states: LazyHiddenStates):
        self.hidden_states = hidden_states
        self.past_key_values = None

    @property
    def last_hidden_state(self):
        return self.hidden_states[-1]

    def __getitem__(self, i):
        return (self.last_hidden_state, self.hidden_states)[i]


class LlamaTextModel(nn.Module):
    """Stands where HF `LlamaModel` sits (`llama_decoder.model`); forward restates
    transformers/models/llama/modeling_llama.py:367-417 up to the requested hidden state."""

    def __init__(self, spec: specs.LlamaSpec, dtype, device):
        super().__init__()
        self.spec = spec
        self._engine, self._ws = None, _Workspace()
        self.gemm_fp8 = False
        self.register_load_state_dict_post_hook(lambda module, _incompatible: module.invalidate_engine())

    @property
    def dtype(self):
        return self.embed_tokens.weight.dtype

    def invalidate_engine(self):
        self._engine = None
        self._train_engine = None
        self._stream_engine = None          # generation.stream_weights

    def _build_engine(self, n_layers: int):
        s, dt = self.spec, self.dtype
        H, F, d = s.hidden_size, s.intermediate_size, s.head_dim
        if F % 32:
            raise ValueError("Llama intermediate_size must be a multiple of 32")
        Hp, Fp, QO = round_up(H, 64), round_up(F, 64), round_up(s.num_attention_heads * d, 64)
        keep, layers = [], (_lib.LlamaLayerC * max(n_layers, 1))()
        P = dict(self.named_parameters())
        for i in range(n_layers):
            p = f"layers.{i}."
            qkv = torch.cat([P[p + f"self_attn.{n}_proj.weight"].detach() for n in ("q", "k", "v")], 0)
            if d == 128 and not s.qk_norm:   # per head: rows 0..31, 64..95, 32..63, 96..127 (include/p2t_hip.h, p2t_llama_layer)
                qkv = qkv.view(-1, 2, 2, 32, H).transpose(1, 2).reshape(-1, H)
            gate, up = P[p + "mlp.gate_proj.weight"].detach(), P[p + "mlp.up_proj.weight"].detach()
            gu = torch.stack([gate.view(F // 32, 32, H), up.view(F // 32, 32, H)], 1).reshape(2 * F, H)   # 32-row gate/up blocks
            t = dict(qkv_w=_pad_cols(qkv, Hp, dt), o_w=_pad_cols(P[p + "self_attn.o_proj.weight"], QO, dt),
                     gu_w=_pad_cols(gu, Hp, dt), down_w=_pad_cols(P[p + "mlp.down_proj.weight"], Fp, dt),
                     ln1_w=_f32(P[p + "input_layernorm.weight"]), ln2_w=_f32(P[p + "post_attention_layernorm.weight"]))
            if s.qk_norm:               # Qwen3: natural row order, the per-head norm runs between projection and rotation
                t["q_norm_w"], t["k_norm_w"] = _f32(P[p + "self_attn.q_norm.weight"]), _f32(P[p + "self_attn.k_norm.weight"])
            if self.gemm_fp8:
                for name, kdim in (("qkv", H), ("o", s.num_attention_heads * d), ("gu", H), ("down", F)):
                    t[name + "_w"], t[name + "_ws"] = _quant_fp8(t[name + "_w"], kdim)
            keep.append(t)
            for k, v in t.items():
                setattr(layers[i], k, v.data_ptr())
        emb, fn = P["embed_tokens.weight"].detach().contiguous(), _f32(P["norm.weight"])
        inv = self._inv_freq().to(emb.device)
        keep += [emb, fn, inv]
        w = _lib.LlamaWeightsC(embed=emb.data_ptr(), layers=C.cast(layers, C.POINTER(_lib.LlamaLayerC)),
                               final_norm_w=fn.data_ptr(), inv_freq=inv.data_ptr())
        cfg = _lib.LlamaConfigC(n_layers=s.num_hidden_layers, hidden=H, ffn=F, heads=s.num_attention_heads,
                                kv_heads=s.num_key_value_heads, head_dim=d, vocab=s.vocab_size,
                                rms_norm_eps=s.rms_norm_eps, rope_theta=s.rope_theta,
                                rope_llama3=int(s.rope_type == "llama3"), rope_factor=s.rope_factor,
                                rope_low_freq_factor=s.rope_low_freq_factor, rope_high_freq_factor=s.rope_high_freq_factor,
                                rope_original_max_pos=s.rope_original_max_position_embeddings, dtype=ops.dt_of(dt),
                                gemm_fp8=int(self.gemm_fp8))
        if self.gemm_fp8 and dt != torch.bfloat16:
            raise ValueError("fp8 GEMMs need a bf16 model (activations and attention stay bf16)")
        self._engine = dict(cfg=cfg, w=w, layers=layers, keep=keep, n=n_layers)
        return self._engine

    def _inv_freq(self) -> torch.Tensor:
        """fp32 inv_freq exactly as transformers computes it (modeling_rope_utils.py:580-662)."""
        s = self.spec
        d = s.head_dim
        inv = 1.0 / (s.rope_theta ** (torch.arange(0, d, 2, dtype=torch.int64).to(dtype=torch.float) / d))
        if s.rope_type == "llama3":
            old = s.rope_original_max_position_embeddings
            low_wl, high_wl = old / s.rope_low_freq_factor, old / s.rope_high_freq_factor
            wavelen = 2 * math.pi / inv
            inv_l = torch.where(wavelen > low_wl, inv / s.rope_factor, inv)
            smooth = (old / wavelen - s.rope_low_freq_factor) / (s.rope_high_freq_factor - s.rope_low_freq_factor)
            smoothed = (1 - smooth) * inv_l / s.rope_factor + smooth * inv_l
            medium = ~(wavelen < high_wl) * ~(wavelen > low_wl)
            inv = torch.where(medium, smoothed, inv_l)
        return inv.float().contiguous()

    def _build_train_engine(self):
        """Transposed weights of every layer for the dX GEMMs of the stage-2 backward (include/p2t_hip.h, p2t_llama_layer_t):
        [forward K][forward N rounded up to 64], built once -- the decoder is frozen on that path."""
        s, dt = self.spec, self.dtype
        if self.gemm_fp8:
            raise ValueError("stage-2 training runs the decoder GEMMs in the model dtype (set_gemm_dtype('model'))")
        if s.qk_norm:             # (LlamaDecoder.forward routes Qwen3 decoders to the per-layer step of p2t_hip/decoder_train.py)
            raise NotImplementedError("the fused frozen-decoder chain has no per-head q/k RMSNorm backward: use LlamaDecoder.forward(labels=...)")
        H, F, L = s.hidden_size, s.intermediate_size, s.num_hidden_layers
        keep, layers = [], (_lib.LlamaLayerTC * L)()
        P = dict(self.named_parameters())
        tr = lambda w: _pad_cols(w.detach().t().contiguous(), round_up(w.shape[0], 64), dt)      # [N, K] -> [K, N padded]
        for i in range(L):
            p = f"layers.{i}."
            qkv = torch.cat([P[p + f"self_attn.{n}_proj.weight"].detach() for n in ("q", "k", "v")], 0)     # NATURAL row order
            gate, up = P[p + "mlp.gate_proj.weight"].detach(), P[p + "mlp.up_proj.weight"].detach()
            gu = torch.stack([gate.view(F // 32, 32, H), up.view(F // 32, 32, H)], 1).reshape(2 * F, H)
            t = dict(qkv_wT=tr(qkv), o_wT=tr(P[p + "self_attn.o_proj.weight"]), gu_wT=tr(gu), down_wT=tr(P[p + "mlp.down_proj.weight"]))
            keep.append(t)
            for k, v in t.items():
                setattr(layers[i], k, v.data_ptr())
        self._train_engine = dict(layers=layers, keep=keep)
        return self._train_engine

    def train_forward(self, inputs_embeds: torch.Tensor, attention_mask: torch.Tensor):
        """All layers + final RMSNorm from f32 [B, T, hidden] inputs, keeping the activation tape for `train_backward`
        (p2t_llama_train_forward).  -> (post-norm hidden states f32 [B, T, hidden], tape handle)."""
        B, T, H = inputs_embeds.shape
        L = self.spec.num_hidden_layers
        e = self.ensure_engine(L)
        dev = self.embed_tokens.weight.device
        emb = inputs_embeds.to(device=dev, dtype=torch.float32).contiguous()
        mask = attention_mask.to(device=dev, dtype=torch.int64).contiguous()
        tape = torch.empty((call("p2t_llama_tape_bytes", C.byref(e["cfg"]), B, T),), dtype=torch.uint8, device=dev)
        ws = self._ws.get(torch.cuda.current_stream().cuda_stream, call("p2t_llama_train_workspace_bytes", C.byref(e["cfg"]), B, T), dev)
        out = torch.empty((B, T, H), dtype=torch.float32, device=dev)
        call("p2t_llama_train_forward", C.byref(e["cfg"]), C.byref(e["w"]), ptr(emb), ptr(mask), B, T, ptr(out), ptr(tape), tape.numel(),
             ptr(ws), ws.numel(), stream())
        return out, (tape, mask, B, T)

    def train_backward(self, d_out: torch.Tensor, handle) -> torch.Tensor:
        """d loss / d inputs_embeds (f32 [B, T, hidden]) from d loss / d (post-norm hidden states) and the tape."""
        tape, mask, B, T = handle
        L = self.spec.num_hidden_layers
        e = self.ensure_engine(L)
        te = getattr(self, "_train_engine", None) or self._build_train_engine()
        dev = tape.device
        ws = self._ws.get(torch.cuda.current_stream().cuda_stream, call("p2t_llama_train_workspace_bytes", C.byref(e["cfg"]), B, T), dev)
        d_in = torch.empty((B, T, self.spec.hidden_size), dtype=torch.float32, device=dev)
        call("p2t_llama_train_backward", C.byref(e["cfg"]), C.byref(e["w"]), C.cast(te["layers"], C.POINTER(_lib.LlamaLayerTC)), ptr(mask), B, T,
             ptr(d_out.float().contiguous()), ptr(tape), tape.numel(), ptr(d_in), ptr(ws), ws.numel(), stream())
        return d_in

    def ensure_engine(self, k: int):
        """Packed weights of the first k layers, built now on the current stream if missing."""
        e = self._engine
        return e if (e is not None and e["n"] >= k) else self._build_engine(k)

    def hidden_state(self, input_ids, attention_mask, k: int) -> torch.Tensor:
        """hidden_states[k] as f32 [B, T, hidden] (k = n_layers -> post final RMSNorm)."""
        if input_ids is None or input_ids.dim() != 2:
            raise ValueError("input_ids must be a [batch, seq_len] tensor of token ids")
        B, T = input_ids.shape
        if attention_mask is None:
            attention_mask = torch.ones_like(input_ids)
        if tuple(attention_mask.shape) != (B, T):
            raise ValueError(f"attention_mask shape {tuple(attention_mask.shape)} != input ids {(B, T)}")
        dev = self.embed_tokens.weight.device
        ids = input_ids.to(device=dev, dtype=torch.int64).contiguous()
        return self._run(ids, None, attention_mask, B, T, k)

    def _run(self, ids, embeds, attention_mask, B, T, k):
        e = self._engine
        if e is None or e["n"] < k:
            e = self._build_engine(k)
        dev = self.embed_tokens.weight.device
        mask = attention_mask.to(device=dev, dtype=torch.int64).contiguous()
        nbytes = call("p2t_llama_workspace_bytes", C.byref(e["cfg"]), B, T)
        ws = self._ws.get(torch.cuda.current_stream().cuda_stream, nbytes, dev)
        out = torch.empty((B, T, self.spec.hidden_size), dtype=torch.float32, device=dev)
        if embeds is None:
            call("p2t_llama_hidden_forward", C.byref(e["cfg"]), C.byref(e["w"]), ptr(ids), ptr(mask), B, T, int(k), ptr(out),
                 ptr(ws), ws.numel(), stream())
        else:
            call("p2t_llama_hidden_forward_embeds", C.byref(e["cfg"]), C.byref(e["w"]), ptr(embeds), ptr(mask), B, T, int(k),
                 ptr(out), ptr(ws), ws.numel(), stream())
        return out

    def embed(self, input_ids: torch.Tensor) -> torch.Tensor:
        """`get_input_embeddings()(input_ids)` as the f32 residual-stream input [B, T, hidden]."""
        if input_ids is None or input_ids.dim() != 2:
            raise ValueError("input_ids must be a [batch, seq_len] tensor of token ids")
        e = self._engine or self._build_engine(0)
        dev = self.embed_tokens.weight.device
        ids = input_ids.to(device=dev, dtype=torch.int64).contiguous()
        out = torch.empty((*ids.shape, self.spec.hidden_size), dtype=torch.float32, device=dev)
        call("p2t_llama_embed_tokens", C.byref(e["cfg"]), C.byref(e["w"]), ptr(ids), ids.numel(), ptr(out), stream())
        return out

    def hidden_state_from_embeds(self, inputs_embeds: torch.Tensor, attention_mask, k: int) -> torch.Tensor:
        """hidden_states[k] from caller-supplied layer-0 inputs (f32 [B, T, hidden])."""
        if inputs_embeds is None or inputs_embeds.dim() != 3 or inputs_embeds.shape[2] != self.spec.hidden_size:
            raise ValueError(f"inputs_embeds must be [batch, seq_len, {self.spec.hidden_size}]")
        B, T, _ = inputs_embeds.shape
        if attention_mask is None:
            attention_mask = torch.ones((B, T), dtype=torch.int64, device=inputs_embeds.device)
        if tuple(attention_mask.shape) != (B, T):
            raise ValueError(f"attention_mask shape {tuple(attention_mask.shape)} != inputs {(B, T)}")
        emb = inputs_embeds.to(device=self.embed_tokens.weight.device, dtype=torch.float32).contiguous()
        return self._run(None, emb, attention_mask, B, T, k)

    def forward(self, input_ids=None, attention_mask=None, position_ids=None, past_key_values=None, inputs_embeds=None,
                use_cache=None, output_attentions=None, output_hidden_states=None, return_dict=None, **kwargs):
        if position_ids is not None or past_key_values is not None or use_cache:
            raise NotImplementedError("position_ids / past_key_values are not arguments of this forward: the KV cache lives inside `generate` (p2t_hip/generation.py)")
        if (input_ids is None) == (inputs_embeds is None):
            raise ValueError("You must specify exactly one of input_ids or inputs_embeds")
        if output_attentions:
            raise NotImplementedError("output_attentions is not available from the fused decoder")
        return LlamaTextOutput(LazyHiddenStates(self, input_ids, attention_mask, inputs_embeds))


class _ScatterRowsFn(torch.autograd.Function):
    """`inputs_embeds[placeholder_mask] = encoder_hidden_states[encoder_mask]` (reference :138) with its backward: the gradient of
    the scattered rows flows back to the encoder (adapter) states, every other encoder row gets zero."""

    @staticmethod
    def forward(ctx, src2d, holder, dst_pos, n_dst, src_pos, n_src, H):
        embeds2d = holder[0]                                    # the freshly embedded tokens (no graph of their own), filled in place
        ops.scatter_rows(embeds2d, dst_pos, n_dst, src2d, src_pos, n_src, H)
        ctx.save_for_backward(dst_pos, n_dst, src_pos, n_src)
        ctx.H, ctx.src_shape, ctx.src_dtype = H, tuple(src2d.shape), src2d.dtype
        return embeds2d

    @staticmethod
    def backward(ctx, g):
        dst_pos, n_dst, src_pos, n_src = ctx.saved_tensors
        g = g.float().contiguous()
        d_src = torch.zeros(ctx.src_shape, dtype=torch.float32, device=g.device)
        call("p2t_gather_rows_f32", ptr(d_src), d_src.stride(0), ptr(src_pos), ptr(g), g.stride(0), ptr(dst_pos), ptr(n_src), ptr(n_dst),
             min(dst_pos.numel(), src_pos.numel()), int(ctx.H), stream())
        return (d_src if ctx.src_dtype == torch.float32 else ops.cast(d_src, ctx.src_dtype)), None, None, None, None, None, None


class _DecoderLossFn(torch.autograd.Function):
    """LM loss of the FROZEN decoder as a function of `inputs_embeds` (reference scripts/train_instruct.py:192-213 with the
    decoder's parameters frozen): forward = p2t_llama_train_forward -> LM head -> shifted cross-entropy, backward =
    cross-entropy backward -> LM-head dX GEMM -> p2t_llama_train_backward.  All kernels hand-written; torch only links them."""

    @staticmethod
    def forward(ctx, inputs_embeds, decoder, attention_mask, labels):
        s, m = decoder.spec, decoder.model
        B, T, H = inputs_embeds.shape
        dt = m.dtype
        h, handle = m.train_forward(inputs_embeds, attention_mask)
        a = h.view(B * T, H) if dt == torch.float32 else ops.cast(h.view(B * T, H), dt)
        logits = ops.gemm_nt(a, decoder._lm_head_padded(), None, n=s.vocab_size, k=H, out_dtype=dt).view(B, T, -1)
        lab = labels.to(logits.device).to(torch.int64).contiguous()
        loss, count = ops.cross_entropy_shifted(logits, lab, s.vocab_size)
        ctx.decoder, ctx.handle, ctx.logits, ctx.labels, ctx.count = decoder, handle, logits, lab, count
        ctx.mark_non_differentiable(logits)
        return loss[0], logits

    @staticmethod
    def backward(ctx, g_loss, _g_logits):
        dec, logits = ctx.decoder, ctx.logits
        s = dec.spec
        B, T, ld = logits.shape
        H, V = s.hidden_size, s.vocab_size
        d_logits = torch.empty_like(logits)
        call("p2t_cross_entropy_shifted_backward", ptr(logits), ld, ops.dt_of(logits), ptr(ctx.labels), B, T, V, -100, ptr(ctx.count), ptr(d_logits), ld,
             stream())
        d_h = ops.gemm_nt(d_logits.view(B * T, ld), dec._lm_head_transposed(), None, n=H, k=round_up(V, 64), epilogue=_lib.EPI_STORE_F32)   # [B*T, H] f32
        d_in = dec.model.train_backward(d_h.view(B, T, H), ctx.handle)
        call("p2t_scale_by_device_scalar", ptr(d_in), d_in.numel(), ptr(g_loss.float().reshape(1).contiguous()), stream())
        ctx.handle = ctx.logits = None                          # free the tape
        return d_in, None, None, None


class CausalLMOutput:
    """The fields of HF `CausalLMOutputWithPast` this path fills; indexable like a ModelOutput ((loss,) logits)."""

    def __init__(self, loss, logits):
        self.loss, self.logits, self.past_key_values, self.hidden_states, self.attentions = loss, logits, None, None, None

    def to_tuple(self):
        return tuple(v for v in (self.loss, self.logits) if v is not None)

    def __getitem__(self, i):
        return getattr(self, i) if isinstance(i, str) else self.to_tuple()[i]

    def __iter__(self):
        return iter(self.to_tuple())


class LlamaDecoder(nn.Module):
    """Stands where the reference puts HF `LlamaForCausalLM` (attribute `llama_decoder`): owns
    `.model` (text tower) and `.lm_head`; `forward` is the cache-less causal-LM forward (logits + optional loss)."""

    def __init__(self, config, dtype=torch.float32, device="cuda"):
        super().__init__()
        self.config = config
        self.spec = llama_spec_from_config(config)
        s = self.spec
        self.model = LlamaTextModel(s, dtype, device)
        tens = [(n[len("model."):], sh, sc, of) for n, sh, sc, of in specs.llama_tensors(s, lm_head=False)]
        _build_tree(self.model, tens, dtype, device)
        self.add_module("lm_head", nn.Module())
        if s.tie_word_embeddings:
            self.lm_head.weight = self.model.embed_tokens.weight
        else:
            self.lm_head.register_parameter("weight", nn.Parameter(
                torch.empty((s.vocab_size, s.hidden_size), dtype=dtype, device=device), requires_grad=False))
        _init_tree(self)

    def get_input_embeddings(self):
        return self.model.embed_tokens

    def _lm_head_padded(self) -> torch.Tensor:
        w = self.lm_head.weight
        key = (w.data_ptr(), w._version)
        if getattr(self, "_lm_key", None) != key:
            self._lm_w, self._lm_key = _pad_cols(w.detach(), round_up(self.spec.hidden_size, 64), w.dtype), key
        return self._lm_w

    def _lm_head_transposed(self) -> torch.Tensor:
        """lm_head.weight^T as the operand of the LM-head dX GEMM: [hidden, vocab rounded up to 64]."""
        w = self.lm_head.weight
        key = (w.data_ptr(), w._version)
        if getattr(self, "_lmT_key", None) != key:
            self._lmT_w, self._lmT_key = _pad_cols(w.detach().t().contiguous(), round_up(w.shape[0], 64), w.dtype), key
        return self._lmT_w

    def forward(self, input_ids=None, attention_mask=None, position_ids=None, past_key_values=None, inputs_embeds=None,
                labels=None, use_cache=None, output_attentions=None, output_hidden_states=None, return_dict=None,
                cache_position=None, **kwargs):
        """LlamaForCausalLM.forward without a KV cache: all layers -> final RMSNorm -> LM head -> (shifted cross-entropy).
        Restates transformers/models/llama/modeling_llama.py (LlamaForCausalLM.forward) and loss_utils.ForCausalLMLoss.
        With `labels`, gradients enabled and `inputs_embeds` requiring grad (stage 2: the adapter's rows sit in it), the loss
        carries an autograd node whose backward is the hand-written chain of llama_train.hip -- the decoder's own parameters
        are frozen on this path (no weight gradients; LoRA matrices are not built yet)."""
        if (input_ids is None) == (inputs_embeds is None):
            raise ValueError("You must specify exactly one of input_ids or inputs_embeds")
        if position_ids is not None or past_key_values is not None or use_cache or cache_position is not None:
            raise NotImplementedError("position_ids / past_key_values / cache_position are not arguments of this forward: the KV cache lives inside `generate` (p2t_hip/generation.py)")
        if output_attentions or output_hidden_states:
            raise NotImplementedError("output_attentions / output_hidden_states are not available from the fused decoder")
        s, m = self.spec, self.model
        if s.hidden_size % 64:
            raise ValueError("the LM head path needs hidden_size % 64 == 0")
        L = s.num_hidden_layers
        lora = getattr(self, "lora", None)
        lora_live = lora is not None and any(q.requires_grad for q in lora.parameters())
        if lora is not None and (labels is None or inputs_embeds is None):
            raise NotImplementedError("this decoder carries LoRA adapters: only the teacher-forced LM loss (inputs_embeds + labels) runs with the "
                                      "branches in place; merge them for inference (p2t_hip.lora.load_and_merge_adapter on peft_state_dict())")
        if labels is not None and inputs_embeds is not None and (lora is not None or
                                                                 (torch.is_grad_enabled() and (inputs_embeds.requires_grad or lora_live))):
            B, T, _ = inputs_embeds.shape
            if tuple(labels.shape) != (B, T):
                raise ValueError(f"labels shape {tuple(labels.shape)} != {(B, T)}")
            if attention_mask is None:
                attention_mask = torch.ones((B, T), dtype=torch.int64, device=inputs_embeds.device)
            if getattr(self, "gradient_checkpointing", False) and not getattr(self, "_warned_checkpointing", False):
                # the caller asked for activation checkpointing (reference :253-268 forwards the flag to the decoder): the stage-2
                # step keeps its whole activation tape (p2t_llama_tape_bytes: ~26 GB per 4 x 1216 tokens of Llama-3.1-8B) and
                # recomputes nothing -- say so once instead of silently returning no memory
                import warnings
                warnings.warn("gradient checkpointing was requested, but the stage-2 step of this decoder keeps the full activation tape "
                              f"({call('p2t_llama_tape_bytes', C.byref(m.ensure_engine(L)['cfg']), B, T) / 2 ** 30:.1f} GiB for this batch) and recomputes "
                              "nothing: lower the micro-batch if memory is the limit", RuntimeWarning, stacklevel=2)
                self._warned_checkpointing = True
            if lora is not None or s.qk_norm:
                # LoRA branches (scripts/train_instruct.py:146-183) or Qwen3's per-head q / k norm sit between the fused blocks of
                # p2t_llama_train_forward: the per-layer form of the same step (p2t_hip/decoder_train.py)
                if m.gemm_fp8:
                    raise ValueError("stage-2 training runs the decoder GEMMs in the model dtype (set_gemm_dtype('model'))")
                from .decoder_train import lora_lm_loss
                loss, logits = lora_lm_loss(self, lora, inputs_embeds, attention_mask, labels)
                return CausalLMOutput(loss=loss, logits=logits)
            loss, logits = _DecoderLossFn.apply(inputs_embeds, self, attention_mask, labels)
            return CausalLMOutput(loss=loss, logits=logits[..., : s.vocab_size])
        h = m.hidden_state(input_ids, attention_mask, L) if inputs_embeds is None else m.hidden_state_from_embeds(inputs_embeds, attention_mask, L)
        B, T, H = h.shape
        dt = m.dtype
        a = h.view(B * T, H) if dt == torch.float32 else ops.cast(h.view(B * T, H), dt)      # post-norm states in model dtype
        logits = ops.gemm_nt(a, self._lm_head_padded(), None, n=s.vocab_size, k=H, out_dtype=dt)   # [B*T, ld(V)]
        logits = logits.view(B, T, -1)
        loss = None
        if labels is not None:
            if tuple(labels.shape) != (B, T):
                raise ValueError(f"labels shape {tuple(labels.shape)} != {(B, T)}")
            loss = ops.cross_entropy_shifted(logits, labels.to(logits.device), s.vocab_size)[0][0]
        return CausalLMOutput(loss=loss, logits=logits[..., : s.vocab_size])

    def generate(self, inputs=None, attention_mask=None, inputs_embeds=None, **kwargs):
        """`LlamaForCausalLM.generate` over the KV-cache decode path (p2t_hip/generation.py): prompts as `inputs_embeds` (what
        Esm2LlamaInstructForCausalLM.generate passes, reference :246-250) or as ids (`inputs` / `input_ids`); greedy, sampling
        (temperature / top_k / top_p) and beam search (num_beams, length_penalty).  The result holds the NEW tokens only (HF does
        the same for `inputs_embeds`; for id prompts HF prepends the prompt, this path does not)."""
        from . import generation
        ids = kwargs.pop("input_ids", inputs)
        return generation.generate(self, inputs_embeds=inputs_embeds, attention_mask=attention_mask, input_ids=ids, **kwargs)


# ---------------------------------------------------------------------------------------------
# the assembled model
# ---------------------------------------------------------------------------------------------
class Esm2LlamaInstructForCausalLM(PreTrainedModel):
    """Esm2LlamaInstructForCausalLM = ESM2 encoder + ModalityAdapter + Llama decoder
    (reference models/modeling_esm2llama_instruct.py:71-268).  Initialise with either a
    configuration OR all three components; `kwargs` override standalone config attributes.

    A `transformers.PreTrainedModel` like the reference's class (:71-78): `isinstance` checks, `config_class`,
    `save_pretrained(dir, state_dict=..., safe_serialization=...)` as `transformers.Trainer.save_model` calls it,
    `from_pretrained(dir, torch_dtype=...)`, `gradient_checkpointing_enable(gradient_checkpointing_kwargs=...)`.  The modelling
    code is this package's (HIP kernels behind the three sub-modules), none of it is transformers'."""
    config_class = Esm2LlamaInstructConfig
    base_model_prefix = "model"
    main_input_name = "input_ids"
    supports_gradient_checkpointing = True          # accepted; a no-op (the frozen towers run without autograd, see below)
    _no_split_modules = ["EsmEncoder", "ModalityAdapter", "LlamaDecoder"]
    _supports_sdpa = False
    _supports_flash_attn = False

    def __init__(self, config: Optional[Esm2LlamaInstructConfig] = None, esm_encoder: Optional[EsmEncoder] = None,
                 adapter: Optional[ModalityAdapter] = None, llama_decoder: Optional[LlamaDecoder] = None,
                 dtype=torch.float32, device="cuda", **kwargs):
        if config is not None:                    # components ignored if config is provided (reference :88-95)
            super().__init__(config)
            self.esm_encoder = EsmEncoder(config.esm_config, dtype, device)
            self.adapter = ModalityAdapter(config.adapter_config, dtype, device)
            self.llama_decoder = LlamaDecoder(config.llama_config, dtype, device)
        else:
            if esm_encoder is None or adapter is None or llama_decoder is None:
                raise ValueError("pass either `config` or all of esm_encoder, adapter and llama_decoder")
            super().__init__(Esm2LlamaInstructConfig(esm_config=esm_encoder.config, adapter_config=adapter.config,
                                                     llama_config=llama_decoder.config, **kwargs))      # reference :96-106
            self.esm_encoder, self.adapter, self.llama_decoder = esm_encoder, adapter, llama_decoder

    def _init_weights(self, module):
        """Parameters are created by the sub-modules (torch.empty) and then filled from a checkpoint or from the synthetic
        generator (`fill_synthetic`); there is no random initialisation scheme to apply (the reference class has none either)."""

    # ---- synthetic construction (bench / tests: no checkpoints exist offline) ----
    @classmethod
    def from_specs(cls, esm: specs.EsmSpec, llama: specs.LlamaSpec, adapter: specs.AdapterSpec, dtype=torch.bfloat16,
                   device="cuda", seed: Optional[int] = 0, adapter_dtype=None, gemm_dtype: str = "model"):
        model = cls(esm_encoder=EsmEncoder(esm_config_from_spec(esm), dtype, device),
                    adapter=ModalityAdapter(ModalityAdapterConfig(adapter.input_dim, adapter.intermediate_dim,
                                                                  adapter.output_dim, adapter.dropout_rate),
                                            adapter_dtype or dtype, device),
                    llama_decoder=LlamaDecoder(llama_config_from_spec(llama), dtype, device))
        if seed is not None:
            model.fill_synthetic(seed)
        return model.set_gemm_dtype(gemm_dtype)

    # ---- HF-style persistence (the reference class is a PreTrainedModel: models/modeling_esm2llama_instruct.py:71-106) ----
    def save_pretrained(self, save_directory: str, state_dict=None, safe_serialization: bool = True, is_main_process: bool = True,
                        **kwargs):
        """`config.json` (Esm2LlamaInstructConfig.save_pretrained) + the HF-named state dict (`esm_encoder.*`, `adapter.*`,
        `llama_decoder.*`) as `model.safetensors` (or `pytorch_model.bin`), as `PreTrainedModel.save_pretrained` lays them out.
        `state_dict` / `is_main_process` / further keywords as `transformers.Trainer.save_model` passes them (sharding,
        `push_to_hub` and friends are not supported: one file, local directory)."""
        import os
        if kwargs.get("push_to_hub"):
            raise ValueError("there is no hub access on this path")
        if not is_main_process:
            return
        os.makedirs(save_directory, exist_ok=True)
        self.config.save_pretrained(save_directory)
        sd = {k: v.detach().cpu().contiguous() for k, v in (self.state_dict() if state_dict is None else state_dict).items()}
        if self.llama_decoder.spec.tie_word_embeddings:
            sd.pop("llama_decoder.lm_head.weight", None)             # tied to embed_tokens: stored once, as HF does
        if safe_serialization:
            from safetensors.torch import save_file
            save_file(sd, os.path.join(save_directory, "model.safetensors"), metadata={"format": "pt"})
        else:
            torch.save(sd, os.path.join(save_directory, "pytorch_model.bin"))

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path: str, *model_args, dtype=None, torch_dtype=None, device="cuda", config=None,
                        **kwargs):
        """Local directory written by `save_pretrained` (of this class or of the reference's): config -> modules -> weights.
        Weight files are read with loaders that execute nothing (safetensors, or torch.load(weights_only=True)).
        `torch_dtype=` (the keyword of the reference's transformers 4.40) and `dtype=` (transformers 5) both select the
        parameter dtype; default: the dtype the checkpoint was written in."""
        import os
        d = pretrained_model_name_or_path
        if not os.path.isdir(d):
            raise ValueError(f"{d!r} is not a local directory (there is no hub access on this path)")
        if config is None:
            config = Esm2LlamaInstructConfig.from_pretrained(d)
        st = os.path.join(d, "model.safetensors")
        if os.path.exists(st):
            from safetensors.torch import load_file
            sd = load_file(st)
        else:
            sd = torch.load(os.path.join(d, "pytorch_model.bin"), weights_only=True, map_location="cpu")
        dtype = dtype if dtype is not None else torch_dtype
        if isinstance(dtype, str):
            dtype = None if dtype == "auto" else getattr(torch, dtype)
        if dtype is None:
            dtype = next(iter(sd.values())).dtype
        for k in ("low_cpu_mem_usage", "device_map", "attn_implementation", "trust_remote_code", "use_safetensors", "local_files_only"):
            kwargs.pop(k, None)                    # loader hints of HF's implementation: nothing to do here
        model = cls(config=config, dtype=dtype, device=device, **kwargs)
        res = model.load_state_dict(sd, strict=False)
        bad = [k for k in res.missing_keys if not ("contact_head" in k or "inv_freq" in k or k.endswith("lm_head.weight"))]
        if bad or res.unexpected_keys:
            raise RuntimeError(f"checkpoint does not match the configuration: missing {bad}, unexpected {res.unexpected_keys}")
        model.esm_encoder.invalidate_engine()
        model.llama_decoder.model.invalidate_engine()
        return model.eval()                        # as PreTrainedModel.from_pretrained returns its models

    def set_gemm_dtype(self, gemm_dtype: str = "model"):
        """"fp8": the eight projections of both frozen towers run on the fp8 MFMA kernel -- weights quantised once to e4m3
        with one power-of-two (E8M0) scale per output channel, activations per token on the fly (BASELINE.json configs[4];
        DESIGN.md section 9).  "model": GEMMs in the model dtype.  The HF-named parameters stay the masters either way
        (checkpoints are unaffected); the adapter, the only trained block, always stays in the model dtype."""
        if gemm_dtype not in ("model", "fp8"):
            raise ValueError("gemm_dtype must be 'model' or 'fp8'")
        on = gemm_dtype == "fp8"
        if on and self.esm_encoder.dtype != torch.bfloat16:
            raise ValueError("fp8 GEMMs need a bf16 model")
        for tower in (self.esm_encoder, self.llama_decoder.model):
            if tower.gemm_fp8 != on:
                tower.gemm_fp8 = on
                tower.invalidate_engine()
        return self

    def fill_synthetic(self, seed: int = 0):
        """Weights from the counter-hash generator (p2t_hip.synth), identical to what the oracle and
        the golden-vector generator materialise for the same seed."""
        _fill_synthetic(self.esm_encoder, specs.esm_tensors(self.esm_encoder.spec), seed, "esm_encoder.")
        _fill_synthetic(self.adapter, specs.adapter_tensors(self.adapter.config.to_spec()), seed, "adapter.")
        ls = self.llama_decoder.spec
        _fill_synthetic(self.llama_decoder, specs.llama_tensors(ls), seed, "llama_decoder.")
        self.esm_encoder.invalidate_engine()
        self.llama_decoder.model.invalidate_engine()
        return self

    def prepare_decoder_inputs(self, input_ids, encoder_hidden_states, attention_mask=None, encoder_attention_mask=None):
        """Embed `input_ids` and replace the placeholder positions by the encoder (adapter) states, in row-major order on
        both sides -- `inputs_embeds[input_ids == placeholder_id] = encoder_hidden_states[encoder_attention_mask.bool()]`
        (reference :108-139).  Returns (inputs_embeds f32 [B, T, H], attention_mask)."""
        if input_ids is None or input_ids.dim() != 2:
            raise ValueError("input_ids must be passed to locate the placeholders")
        B, T = input_ids.shape
        dev = self.llama_decoder.model.embed_tokens.weight.device
        enc = encoder_hidden_states
        if enc.dim() != 3 or enc.shape[0] != B:
            raise ValueError(f"encoder_hidden_states must be [batch={B}, encoder_seq_len, hidden]")
        H = self.llama_decoder.spec.hidden_size
        if enc.shape[2] != H:
            raise ValueError(f"encoder states have {enc.shape[2]} features, the decoder embeds {H}")
        if attention_mask is None:
            attention_mask = torch.ones((B, T), dtype=torch.long, device=dev)
        if encoder_attention_mask is None:
            encoder_attention_mask = torch.ones(enc.shape[:2], dtype=torch.long, device=dev)
        ids = input_ids.to(dev)
        embeds = self.llama_decoder.model.embed(ids)
        dst_pos, n_dst = ops.positions_where(ids, int(self.config.placeholder_id))
        src_pos, n_src = ops.positions_where(encoder_attention_mask.to(dev))
        src = enc.reshape(B * enc.shape[1], H)
        if src.dtype not in (torch.float32, torch.bfloat16) or src.stride(1) != 1:
            src = src.float().contiguous()
        if torch.is_grad_enabled() and src.requires_grad:       # stage 2: the gradient of the placeholder rows reaches the adapter
            embeds = _ScatterRowsFn.apply(src, [embeds.view(B * T, H)], dst_pos, n_dst, src_pos, n_src, H).view(B, T, H)
        else:
            ops.scatter_rows(embeds.view(B * T, H), dst_pos, n_dst, src, src_pos, n_src, H)
        nd, ns = int(n_dst.item()), int(n_src.item())          # torch's boolean-mask assignment raises on a count mismatch
        if nd != ns:
            raise RuntimeError(f"shape mismatch: {ns} encoder states cannot be assigned to {nd} placeholder positions")
        return embeds, attention_mask

    def forward(self, input_ids=None, attention_mask=None, position_ids=None, past_key_values=None, labels=None,
                protein_input_ids=None, protein_attention_mask=None, protein_position_ids=None, protein_head_mask=None,
                protein_inputs_embeds=None, use_cache=None, output_attentions=None, output_hidden_states=None,
                return_dict=None, return_encoder_outputs: bool = False, return_adapter_outputs: bool = False,
                return_decoder_inputs: bool = False, cache_position=None):
        if protein_position_ids is not None or protein_head_mask is not None or protein_inputs_embeds is not None:
            raise NotImplementedError("protein_position_ids / protein_head_mask / protein_inputs_embeds are not supported")
        if return_encoder_outputs:                 # reference :175-189
            return self.esm_encoder(input_ids=protein_input_ids, attention_mask=protein_attention_mask,
                                    output_attentions=output_attentions, output_hidden_states=output_hidden_states,
                                    return_dict=return_dict)
        if output_attentions or output_hidden_states:
            raise NotImplementedError("output_attentions / output_hidden_states are not available from the fused encoder")
        enc = self.esm_encoder.encode(protein_input_ids, protein_attention_mask)      # [B, T, Hp], zero padded
        B, T, Hp = enc.shape
        adapter_output = self.adapter.forward_padded(enc.view(B * T, Hp)).reshape(B, T, -1)   # reference :191
        if return_adapter_outputs:                 # reference :192-193
            return adapter_output, protein_attention_mask
        inputs_embeds, attention_mask = self.prepare_decoder_inputs(input_ids=input_ids, encoder_hidden_states=adapter_output,
                                                                    attention_mask=attention_mask,
                                                                    encoder_attention_mask=protein_attention_mask)   # :195-201
        if return_decoder_inputs:                  # :202-203
            return inputs_embeds, attention_mask
        return self.llama_decoder.forward(input_ids=None, attention_mask=attention_mask, position_ids=position_ids,
                                          past_key_values=past_key_values, inputs_embeds=inputs_embeds, labels=labels,
                                          use_cache=use_cache, output_attentions=output_attentions, return_dict=return_dict,
                                          cache_position=cache_position)                                              # :204-215

    def generate(self, inputs: torch.Tensor, attention_mask: Optional[torch.Tensor] = None, protein_input_ids: Optional[torch.Tensor] = None,
                 protein_attention_mask: Optional[torch.Tensor] = None, protein_inputs_embeds: Optional[torch.Tensor] = None, **kwargs):
        """Reference :217-251: `inputs` is the [prompt] only; the prompt embeddings (placeholders replaced by the adapter rows) come
        from `forward(return_decoder_inputs=True)` and go to `llama_decoder.generate(inputs_embeds=, attention_mask=, **kwargs)`.
        The output does not repeat the prompt (it went in as embeddings)."""
        with torch.no_grad():
            prompt_embeds, prompt_mask = self(input_ids=inputs, attention_mask=attention_mask, protein_input_ids=protein_input_ids,
                                              protein_attention_mask=protein_attention_mask, protein_inputs_embeds=protein_inputs_embeds,
                                              use_cache=False, output_attentions=False, output_hidden_states=False, return_dict=False,
                                              return_decoder_inputs=True)
        return self.llama_decoder.generate(inputs_embeds=prompt_embeds, attention_mask=prompt_mask, **kwargs)

    def add_lora(self, r: int, lora_alpha: Optional[float] = None, lora_dropout: float = 0.1, target_modules=None, seed: int = 0):
        """`get_peft_model(model, LoraConfig(r, lora_alpha = 2 r, lora_dropout = 0.1, target_modules = [...decoder projections...],
        modules_to_save = adapter.fc1 / fc2))` of scripts/train_instruct.py:155-183, for the decoder targets (the script's ESM-C
        target names match no module of the ESM2 encoder): trainable A / B pairs on the seven projections of every decoder layer
        (p2t_hip/decoder_train.py), the base weights frozen, the modality adapter left trainable.  Returns the DecoderLora module
        (its parameters are what the optimizer takes next to the adapter's)."""
        from .decoder_train import TARGETS, DecoderLora
        self.llama_decoder.lora = DecoderLora(self.llama_decoder, r, lora_alpha, lora_dropout, TARGETS if target_modules is None else target_modules, seed)
        self.esm_encoder.requires_grad_(False)
        for n, q in self.llama_decoder.named_parameters():
            if not n.startswith("lora."):
                q.requires_grad_(False)
        return self.llama_decoder.lora

    def gradient_checkpointing_enable(self, gradient_checkpointing_kwargs=None):
        """Accepted for loop compatibility (reference :253-261; `transformers.Trainer(gradient_checkpointing=True)` passes
        `gradient_checkpointing_kwargs`), and a no-op by construction: activation checkpointing trades recomputation for the
        memory autograd holds, and on this path the frozen towers run WITHOUT autograd (nothing is kept), while the adapter keeps
        one set of activations per segment (z1, h1, z2: ContrastiveTrainer._buffers) -- `contrastive_num_segments` bounds that,
        as it does upstream.  The stage-2 step (LM loss through the frozen decoder) is different: it keeps its whole activation
        tape and recomputes nothing, and warns once when it runs under this flag."""
        self._gradient_checkpointing_requested = True
        self.llama_decoder.gradient_checkpointing = True      # the stage-2 step warns once that its tape is kept whole (LlamaDecoder.forward)

    def gradient_checkpointing_disable(self):
        """No-op (reference :263-268)."""
        self._gradient_checkpointing_requested = False
        self.llama_decoder.gradient_checkpointing = False

    @property
    def is_gradient_checkpointing(self) -> bool:
        return bool(getattr(self, "_gradient_checkpointing_requested", False))
