"""CPU oracle for the contrastive-alignment hot path -- TEST INFRASTRUCTURE, NOT PRODUCT.

A plain numpy restatement (fp32 storage, BLAS matmuls) of the reference algorithm for the path
named by BASELINE.json `north_star`.  Only tests/, __graft_entry__.smoke() and bench.py's
`cpu_baseline` leg may import this file; the product (prot2text-v2-esm3_amd/) never does and fails
loudly when its HIP library is missing.

Parity pinning: the reference ships no tests and no golden vectors, so this
oracle is pinned against outputs of the reference itself, generated in the build container by
tests/golden/make_golden.py (which imports /root/reference and HuggingFace transformers 5.15.0)
and committed as tests/golden/*.npz; tests/test_oracle_golden.py checks every function here
against them.

Each function cites what it restates.  "HF" = site-packages/transformers (third-party code the
reference delegates the tower arithmetic to; reference pins transformers==4.40.2 in README.md:42-49,
the container has 5.15.0 -- the math on this path is unchanged).
    REF   = /root/reference
    ESM   = transformers/models/esm/modeling_esm.py
    LLAMA = transformers/models/llama/modeling_llama.py
    ROPE  = transformers/modeling_rope_utils.py
"""
from __future__ import annotations

import math
from typing import Callable, Optional

import numpy as np
from scipy.special import erf as _erf

F32 = np.float32


# ---------------------------------------------------------------------------------------------
# precision model: identity (fp32 oracle) or bf16 rounding at the points where the HIP bf16 path
# stores bf16 (weights, GEMM inputs, attention probabilities).  Accumulation is fp32 either way.
# ---------------------------------------------------------------------------------------------
def _bf16(x: np.ndarray) -> np.ndarray:
    """Round-to-nearest-even to bf16, widened back to f32 (finite inputs; uint32 arithmetic, three in-place passes)."""
    x = np.ascontiguousarray(x, dtype=F32)
    u = x.view(np.uint32)
    t = u >> np.uint32(16)
    t &= np.uint32(1)
    t += np.uint32(0x7FFF)
    t += u                                               # a mantissa carry into the exponent is the correct rounding
    t &= np.uint32(0xFFFF0000)
    return t.view(F32).reshape(x.shape)


def e4m3_round(x: np.ndarray) -> np.ndarray:
    """OCP e4m3fn rounding of f32 values (4 exponent bits, bias 7, 3 mantissa bits, subnormals of 2^-9, largest finite
    448, no infinities), round-to-nearest-even -- what v_cvt_pk_fp8_f32 does on gfx950 and torch.float8_e4m3fn on the
    CPU (tests/test_fp8_host.py pins this function on the latter).  |x| <= 448 is the caller's business (scaled rows)."""
    x = np.ascontiguousarray(x, dtype=F32)
    u = x.view(np.uint32)
    a = u & np.uint32(0x7FFFFFFF)                        # |x| as bits
    # normal range (|x| >= 2^-6): keep 3 of the 23 mantissa bits, ties to even -- the bf16 trick 20 bits lower
    t = a >> np.uint32(20)
    t &= np.uint32(1)
    t += np.uint32(0x7FFFF)
    t += a
    t &= np.uint32(0xFFF00000)
    r = t.view(F32)
    small = a < np.uint32(0x3C800000)                    # below 2^-6 the spacing stays that of the subnormals, 2^-9
    if small.any():
        r = np.where(small, np.rint(a.view(F32) * F32(512.0)) * F32(1.0 / 512.0), r)     # rint: ties to even; both exact in f32
    r = np.minimum(r, F32(448.0))
    return np.copysign(r, x).astype(F32, copy=False)


def e8m0_of_amax(amax: np.ndarray) -> np.ndarray:
    """Biased E8M0 exponent of a row's power-of-two scale: the smallest 2^e with amax / 2^e <= 448, by the same integer
    rule on the f32 bits as csrc/quant.hip (e = exponent(amax) - 8 + [mantissa > 0.75]); 127 for an all-zero row."""
    u = np.ascontiguousarray(amax, dtype=F32).view(np.uint32)
    ea = ((u >> np.uint32(23)) & np.uint32(0xFF)).astype(np.int32) - 127
    e = ea - 8 + ((u & np.uint32(0x7FFFFF)) > np.uint32(0x600000)).astype(np.int32)
    E = np.clip(e + 127, 1, 254)
    return np.where(amax > 0, E, 127).astype(np.uint8)


def quant_rows_e4m3(x: np.ndarray):
    """Row-wise fp8 quantisation of a GEMM operand (BASELINE.json configs[4]): -> (dequantised values f32 [rows, cols],
    E8M0 scale byte per row, the e4m3 values before rescaling).  x / 2^e is exact, so e4m3_round is the only rounding."""
    x = np.ascontiguousarray(x, dtype=F32)
    E = e8m0_of_amax(np.abs(x).max(axis=-1))
    scale = np.ldexp(F32(1.0), E.astype(np.int32) - 127).astype(F32)[..., None]
    q = e4m3_round(x / scale)
    return (q * scale).astype(F32), E, q


FP8_OPERAND_GROWTH = (1.0 + 2.0 ** -4) ** 2      # e4m3 rounding enlarges an element by at most 2^-4, relative: two operands


def gelu_bound_scale(h: np.ndarray, w: np.ndarray, b: np.ndarray) -> np.ndarray:
    """E8M0 byte per row of the scale under which gelu(h W^T + b) is stored as e4m3 when the FFN-up GEMM writes fp8 directly
    (csrc/quant.hip norm_fp8_kernel `bound_scale`, csrc/epilogue.h EpiGeluFp8): |gelu(z)| <= |z| <= ||h_row||_2 max_n ||W_n||_2
    + max_n |b_n| (Cauchy-Schwarz), the weight norm inflated by FP8_OPERAND_GROWTH for the rounded operands."""
    bw = F32(np.sqrt((w.astype(F32) ** 2).sum(-1, dtype=F32)).max() * FP8_OPERAND_GROWTH)
    bb = F32(np.abs(b).max())
    hn = np.sqrt((h.astype(F32) ** 2).sum(-1, dtype=F32)).astype(F32)
    return e8m0_of_amax((hn.astype(np.float64) * np.float64(bw) + np.float64(bb)).astype(F32))


def quant_rows_e4m3_scaled(x: np.ndarray, E: np.ndarray) -> np.ndarray:
    """e4m3 storage of x under GIVEN row scales (E8M0 bytes) -> dequantised f32 values."""
    scale = np.ldexp(F32(1.0), E.astype(np.int32) - 127).astype(F32)[..., None]
    return (e4m3_round(np.ascontiguousarray(x, dtype=F32) / scale) * scale).astype(F32)


class Precision:
    """identity (fp32 oracle); "bf16": bf16 rounding where the HIP bf16 path stores bf16; "fp8": the same, plus e4m3
    row-quantised operands (weights per output channel, activations per token) in the tower GEMMs -- `g` is applied to
    the two operands of every tower projection, `q` to everything the bf16 path rounds."""

    def __init__(self, kind: str = "f32", fused_gelu: bool = True):
        assert kind in ("f32", "bf16", "fp8")
        self.kind = kind
        self.fused_gelu = fused_gelu and kind == "fp8"    # ESM FFN-up output stored as e4m3 under the bound scale (no bf16 copy)

    def q(self, x):           # "quantise a stored activation / weight"
        return x if self.kind == "f32" else _bf16(x)

    def g(self, x):           # "quantise a tower-GEMM operand" (rows of a 2-D array)
        return quant_rows_e4m3(x)[0] if self.kind == "fp8" else x

    def qs(self, x, c):       # the query operand of attention: the HIP bf16 path stores q * c (c = softmax scale * log2 e folded
        if self.kind == "f32":    # into q where it is written, csrc/towers.hip) and rounds THAT to bf16; same value, other rounding points
            return x
        c = F32(c)
        return (_bf16(x * c) / c).astype(F32)

    def a(self, x):           # a norm output feeding a projection: bf16 storage, or straight to fp8 (no bf16 in between)
        return x if self.kind == "fp8" else self.q(x)


LOG2E = 1.4426950408889634
FP32 = Precision("f32")
BF16 = Precision("bf16")
FP8 = Precision("fp8")
FP8_UNFUSED = Precision("fp8", fused_gelu=False)


# ---------------------------------------------------------------------------------------------
# small ops
# ---------------------------------------------------------------------------------------------
def gelu_erf(x):
    """ESM:82-86 `x * 0.5 * (1 + erf(x / sqrt(2)))`; also torch.nn.GELU() (exact) used by the
    adapter, REF models/modeling_esm2llama_instruct.py:54."""
    x = x.astype(F32, copy=False)
    return (x * F32(0.5) * (F32(1.0) + _erf(x / F32(math.sqrt(2.0))).astype(F32))).astype(F32)


def gelu_erf_grad(x):
    x = x.astype(F32, copy=False)
    cdf = F32(0.5) * (F32(1.0) + _erf(x / F32(math.sqrt(2.0))).astype(F32))
    pdf = np.exp(-0.5 * x * x).astype(F32) * F32(1.0 / math.sqrt(2.0 * math.pi))
    return (cdf + x * pdf).astype(F32)


def layer_norm(x, w, b, eps):
    """torch.nn.LayerNorm (ESM:418,429,480,518,529,553): biased variance, eps inside the sqrt."""
    x = x.astype(F32, copy=False)
    mu = x.mean(-1, keepdims=True, dtype=F32)
    xc = x - mu
    var = (xc * xc).mean(-1, keepdims=True, dtype=F32)
    return (xc / np.sqrt(var + F32(eps)) * w + b).astype(F32)


def rms_norm(x, w, eps):
    """LLAMA:62-67: fp32 variance, x * rsqrt(var + eps), weight multiplied last."""
    x = x.astype(F32, copy=False)
    var = (x * x).mean(-1, keepdims=True, dtype=F32)
    return (w * (x / np.sqrt(var + F32(eps)))).astype(F32)


def rotate_half(x):
    """ESM:48-52 / LLAMA:130-134."""
    h = x.shape[-1] // 2
    return np.concatenate([-x[..., h:], x[..., :h]], axis=-1)


def rope_cos_sin(inv_freq, positions):
    """ESM:146-160 / LLAMA:108-127: freqs = pos x inv_freq, emb = cat(freqs, freqs), fp32."""
    freqs = positions.astype(F32)[:, None] * inv_freq.astype(F32)[None, :]
    emb = np.concatenate([freqs, freqs], axis=-1)
    return np.cos(emb).astype(F32), np.sin(emb).astype(F32)


def default_inv_freq(theta: float, dim: int):
    """ESM:127-138 / LLAMA:87-106: 1 / theta^(2i/dim)."""
    return (1.0 / (F32(theta) ** (np.arange(0, dim, 2, dtype=F32) / F32(dim)))).astype(F32)


def llama3_inv_freq(theta, dim, factor, low_freq_factor, high_freq_factor, original_max_pos):
    """ROPE:580-662 `_compute_llama3_parameters` (fp32 throughout, as torch computes it)."""
    inv = default_inv_freq(theta, dim)
    low_wl = F32(original_max_pos / low_freq_factor)
    high_wl = F32(original_max_pos / high_freq_factor)
    wavelen = (F32(2.0 * math.pi) / inv).astype(F32)
    inv_l = np.where(wavelen > low_wl, inv / F32(factor), inv).astype(F32)
    smooth = ((F32(original_max_pos) / wavelen - F32(low_freq_factor))
              / F32(high_freq_factor - low_freq_factor)).astype(F32)
    smoothed = ((F32(1.0) - smooth) * inv_l / F32(factor) + smooth * inv_l).astype(F32)
    medium = (~(wavelen < high_wl)) & (~(wavelen > low_wl))
    return np.where(medium, smoothed, inv_l).astype(F32)


def softmax_rows(s):
    m = s.max(-1, keepdims=True)
    e = np.exp(s - m, dtype=F32)
    return (e / e.sum(-1, keepdims=True, dtype=F32)).astype(F32)


NEG = F32(np.finfo(np.float32).min)


def attention_heads(q, k, v, bias, scale, prec):
    """softmax(scale * q k^T + bias) v per (batch, head) -- the eager attention of ESM:292-317 and
    LLAMA:191-213 -- evaluated head by head so the [T, T] score block stays cache-sized.
    q [B,nh,T,d]; k, v [B,nkv,T,d] (GQA: query head h reads kv head h // (nh/nkv), i.e. repeat_kv);
    bias broadcastable to [B, 1, T, T].  Returns [B, T, nh*d]."""
    B, nh, T, d = q.shape
    rep = nh // k.shape[1]
    out = np.empty((B, T, nh * d), dtype=F32)
    bias = np.broadcast_to(bias, (B, 1, T, T))
    for b in range(B):
        bb = bias[b, 0]
        for h in range(nh):
            s = q[b, h] @ k[b, h // rep].T
            if scale != 1.0:
                s *= F32(scale)
            s += bb
            pr = prec.q(softmax_rows(s))
            out[b, :, h * d:(h + 1) * d] = pr @ v[b, h // rep]
    return out


# ---------------------------------------------------------------------------------------------
# ESM2 encoder  (REF models/modeling_esm2llama_instruct.py:175-185 -> HF EsmModel.forward ESM:685-755)
# ---------------------------------------------------------------------------------------------
def esm2_embeddings(spec, W, ids, mask, prefix=""):
    """ESM:224-271 with position_embedding_type='rotary' (no absolute table): gather, zero <mask>
    tokens, scale by (1-0.12)/(1-observed mask ratio), multiply by attention_mask."""
    emb = W[prefix + "embeddings.word_embeddings.weight"][ids]            # (B,T,H)
    if spec.token_dropout:
        is_mask = (ids == spec.mask_token_id)
        emb = np.where(is_mask[..., None], F32(0.0), emb)
        src_len = mask.sum(-1).astype(F32)
        ratio = is_mask.sum(-1).astype(F32) / src_len
        emb = emb * F32(1.0 - 0.15 * 0.8) / (F32(1.0) - ratio)[:, None, None]
    if spec.emb_layer_norm_before:
        emb = layer_norm(emb, W[prefix + "embeddings.layer_norm.weight"],
                         W[prefix + "embeddings.layer_norm.bias"], spec.layer_norm_eps)
    emb = emb * mask[..., None].astype(F32)
    return emb.astype(F32)


def esm2_layer(spec, W, i, x, key_bias, cos, sin, prec: Precision = FP32, prefix=""):
    """One pre-LN block, ESM:420-439 (attention) + ESM:517-521 (FFN).
    q is scaled by d^-1/2 BEFORE rotary and SDPA runs with scale 1.0 (ESM:345,374)."""
    B, T, H = x.shape
    nh, d = spec.num_attention_heads, spec.head_dim
    p = f"{prefix}encoder.layer.{i}."
    q_, g_ = prec.q, prec.g
    h = prec.a(layer_norm(x, W[p + "attention.LayerNorm.weight"], W[p + "attention.LayerNorm.bias"],
                          spec.layer_norm_eps))
    h2 = g_(h.reshape(B * T, H))

    def lin(name, inp):           # inp: already a GEMM operand (g_ applied by the caller, once per activation)
        return inp @ g_(q_(W[p + name + ".weight"])).T + W[p + name + ".bias"]

    q = lin("attention.self.query", h2).reshape(B, T, nh, d).transpose(0, 2, 1, 3)
    k = lin("attention.self.key", h2).reshape(B, T, nh, d).transpose(0, 2, 1, 3)
    v = lin("attention.self.value", h2).reshape(B, T, nh, d).transpose(0, 2, 1, 3)
    q = q * F32(d ** -0.5)
    q = q * cos + rotate_half(q) * sin                                    # ESM:74-79 (fp32)
    k = k * cos + rotate_half(k) * sin
    q, k, v = prec.qs(q, LOG2E), q_(k), q_(v)
    o = g_(q_(attention_heads(q, k, v, key_bias, 1.0, prec).reshape(B * T, H)))     # ESM:310-317, scale 1.0
    x = x + lin("attention.output.dense", o).reshape(B, T, H)               # ESM:399-409
    h = prec.a(layer_norm(x, W[p + "LayerNorm.weight"], W[p + "LayerNorm.bias"], spec.layer_norm_eps))
    z = lin("intermediate.dense", g_(h.reshape(B * T, H)))                    # ESM:442-450
    if prec.fused_gelu:
        E = gelu_bound_scale(h.reshape(B * T, H), q_(W[p + "intermediate.dense.weight"]), W[p + "intermediate.dense.bias"])
        f = quant_rows_e4m3_scaled(gelu_erf(z), E)
    else:
        f = g_(q_(gelu_erf(z)))
    x = x + lin("output.dense", f).reshape(B, T, H)                          # ESM:453-463
    return x.astype(F32)


def esm2_forward(spec, W, ids, mask, prec: Precision = FP32, prefix="", layers=None):
    """HF EsmModel(add_pooling_layer=False).forward -> last_hidden_state (B,T,H)."""
    if spec.position_embedding_type != "rotary":
        raise ValueError("only rotary ESM2 checkpoints are on this path")
    ids = np.asarray(ids); mask = np.asarray(mask)
    B, T = ids.shape
    x = esm2_embeddings(spec, W, ids, mask, prefix)
    # bidirectional key-padding mask (ESM:758-788): additive dtype-min on padded KEYS only
    key_bias = np.where(mask[:, None, None, :] != 0, F32(0.0), NEG).astype(F32)
    inv = default_inv_freq(spec.rope_theta, spec.head_dim)
    cos, sin = rope_cos_sin(inv, np.arange(T))                            # ESM:731-735
    n = spec.num_hidden_layers if layers is None else layers
    for i in range(n):
        x = esm2_layer(spec, W, i, x, key_bias, cos, sin, prec, prefix)
    if layers is None:
        x = layer_norm(x, W[prefix + "encoder.emb_layer_norm_after.weight"],
                       W[prefix + "encoder.emb_layer_norm_after.bias"], spec.layer_norm_eps)  # ESM:552-553
    return x


# ---------------------------------------------------------------------------------------------
# ModalityAdapter  (REF models/modeling_esm2llama_instruct.py:45-68), eval mode / dropout p = 0
# ---------------------------------------------------------------------------------------------
def l2_normalize(x, eps=1e-12):
    """torch.nn.functional.normalize(p=2, dim=-1): x / max(||x||, eps)."""
    x = x.astype(F32, copy=False)
    n = np.sqrt((x * x).sum(-1, keepdims=True, dtype=F32))
    return (x / np.maximum(n, F32(eps))).astype(F32)


def adapter_forward(W, x, prec: Precision = FP32, prefix="", keep=None):
    """normalize(gelu(fc2(gelu(fc1(x)))))  -- ln1/ln2 exist but are unused (REF :56-57)."""
    q_ = prec.q
    shp = x.shape
    x2 = q_(x.reshape(-1, shp[-1]).astype(F32))
    z1 = x2 @ q_(W[prefix + "fc1.weight"]).T + W[prefix + "fc1.bias"]
    h1 = q_(gelu_erf(z1))
    z2 = h1 @ q_(W[prefix + "fc2.weight"]).T + W[prefix + "fc2.bias"]
    g2 = gelu_erf(z2)
    y = l2_normalize(g2)
    if keep is not None:
        keep.update(x2=x2, z1=z1, h1=h1, z2=z2, g2=g2)
    return y.reshape(*shp[:-1], -1)


def adapter_backward(W, keep, dy, prec: Precision = FP32, prefix=""):
    """Manual backward of adapter_forward w.r.t. fc1/fc2 weight and bias (the only tensors that
    receive gradients in the reference's contrastive stage, train_contrast.py:186-187)."""
    q_ = prec.q
    g2, z2, h1, z1, x2 = keep["g2"], keep["z2"], keep["h1"], keep["z1"], keep["x2"]
    dy = dy.reshape(g2.shape).astype(F32)
    n = np.maximum(np.sqrt((g2 * g2).sum(-1, keepdims=True, dtype=F32)), F32(1e-12))
    y = g2 / n
    dg2 = (dy - y * (dy * y).sum(-1, keepdims=True, dtype=F32)) / n
    dz2 = q_(dg2 * gelu_erf_grad(z2))
    grads = {prefix + "fc2.weight": dz2.T @ h1, prefix + "fc2.bias": dz2.sum(0, dtype=F32)}
    dh1 = dz2 @ q_(W[prefix + "fc2.weight"])
    dz1 = q_(dh1 * gelu_erf_grad(z1))
    grads[prefix + "fc1.weight"] = dz1.T @ x2
    grads[prefix + "fc1.bias"] = dz1.sum(0, dtype=F32)
    return {k: v.astype(F32) for k, v in grads.items()}


# ---------------------------------------------------------------------------------------------
# Llama text tower up to hidden_states[k]  (REF scripts/train_contrast.py:284-304 -> LLAMA:367-417)
# ---------------------------------------------------------------------------------------------
def llama_inv_freq(spec):
    if spec.rope_type == "llama3":
        return llama3_inv_freq(spec.rope_theta, spec.head_dim, spec.rope_factor, spec.rope_low_freq_factor,
                               spec.rope_high_freq_factor, spec.rope_original_max_position_embeddings)
    return default_inv_freq(spec.rope_theta, spec.head_dim)


def llama_layer(spec, W, i, x, bias, cos, sin, prec: Precision = FP32, prefix=""):
    """LLAMA:296-324 decoder layer; attention LLAMA:191-213,254-281 (GQA by repeat_kv, fp32 softmax)."""
    B, T, H = x.shape
    nh, nkv, d = spec.num_attention_heads, spec.num_key_value_heads, spec.head_dim
    p = f"{prefix}model.layers.{i}."
    q_, g_ = prec.q, prec.g

    def wq(name):                 # weights: model-dtype storage, then (fp8 mode) one e4m3 scale per output channel
        return g_(q_(W[p + name + ".weight"]))
    h = g_(prec.a(rms_norm(x, W[p + "input_layernorm.weight"], spec.rms_norm_eps)).reshape(B * T, H))
    q = (h @ wq("self_attn.q_proj").T).reshape(B, T, nh, d).transpose(0, 2, 1, 3)
    k = (h @ wq("self_attn.k_proj").T).reshape(B, T, nkv, d).transpose(0, 2, 1, 3)
    v = (h @ wq("self_attn.v_proj").T).reshape(B, T, nkv, d).transpose(0, 2, 1, 3)
    if getattr(spec, "qk_norm", False):
        # Qwen3 (transformers/models/qwen3/modeling_qwen3.py, Qwen3Attention.forward): q_norm / k_norm = RMSNorm over
        # head_dim of every head after the projection (stored in the model dtype here), before the rotation
        q = rms_norm(q_(q), W[p + "self_attn.q_norm.weight"], spec.rms_norm_eps)
        k = rms_norm(q_(k), W[p + "self_attn.k_norm.weight"], spec.rms_norm_eps)
        v = q_(v)
    q = q * cos + rotate_half(q) * sin
    k = k * cos + rotate_half(k) * sin
    q, k, v = prec.qs(q, LOG2E * d ** -0.5), q_(k), q_(v)
    o = g_(q_(attention_heads(q, k, v, bias, d ** -0.5, prec).reshape(B * T, nh * d)))
    x = x + (o @ wq("self_attn.o_proj").T).reshape(B, T, H)
    h = g_(prec.a(rms_norm(x, W[p + "post_attention_layernorm.weight"], spec.rms_norm_eps)).reshape(B * T, H))
    g = h @ wq("mlp.gate_proj").T
    u = h @ wq("mlp.up_proj").T
    a = g_(q_((g / (F32(1.0) + np.exp(-g, dtype=F32))) * u))              # SiLU(g) * u, LLAMA:174-176
    x = x + (a @ wq("mlp.down_proj").T).reshape(B, T, H)
    return x.astype(F32)


def llama_hidden_state(spec, W, ids, mask, k: int = 16, prec: Precision = FP32, prefix=""):
    """`outputs.hidden_states[k]` of LlamaModel(..., output_hidden_states=True)
    (train_contrast.py:294-304).  Index k < L is the residual stream after k layers; index L is the
    post-final-RMSNorm last_hidden_state (transformers/utils/output_capturing.py:269-277)."""
    ids = np.asarray(ids); mask = np.asarray(mask)
    B, T = ids.shape
    L = spec.num_hidden_layers
    if not 0 <= k <= L:
        raise IndexError(f"hidden_states[{k}] out of range for {L} layers")
    x = W[prefix + "model.embed_tokens.weight"][ids].astype(F32)             # LLAMA:381
    # causal AND key-padding (LLAMA:391-397 create_causal_mask): allowed(i,j) = j <= i and mask[j]
    allowed = np.tril(np.ones((T, T), dtype=bool))[None, None] & (mask[:, None, None, :] != 0)
    bias = np.where(allowed, F32(0.0), NEG).astype(F32)
    cos, sin = rope_cos_sin(llama_inv_freq(spec), np.arange(T))
    for i in range(k):
        x = llama_layer(spec, W, i, x, bias, cos, sin, prec, prefix)
    if k == L:
        x = rms_norm(x, W[prefix + "model.norm.weight"], spec.rms_norm_eps)
    return x


# ---------------------------------------------------------------------------------------------
# readout / loss  (REF scripts/train_contrast.py)
# ---------------------------------------------------------------------------------------------
def readout_embeddings(emb, mask, readout_fn: str):
    """REF scripts/train_contrast.py:198-248 ("last" :207-215, "mean" :217-221, "std" :223-235
    population std with no eps, "mix" :237-248 = cat(mean, std))."""
    emb = emb.astype(F32, copy=False)
    if readout_fn == "last":
        idx = mask.sum(1).astype(np.int64) - 1
        return emb[np.arange(emb.shape[0]), idx, :]
    m = mask.astype(F32)[..., None]
    cnt = mask.sum(1, keepdims=True).astype(F32)
    mean = (emb * m).sum(1, dtype=F32) / cnt
    if readout_fn == "mean":
        return mean.astype(F32)
    diff = emb - mean[:, None, :]
    std = np.sqrt((diff * diff * m).sum(1, dtype=F32) / cnt).astype(F32)
    if readout_fn == "std":
        return std
    if readout_fn == "mix":
        return np.concatenate([mean, std], axis=1).astype(F32)
    raise ValueError(readout_fn)


def readout_backward(emb, mask, readout_fn: str, dout):
    """d(readout)/d(emb) applied to dout -- used for the adapter gradients."""
    emb = emb.astype(F32, copy=False)
    B, T, D = emb.shape
    m = mask.astype(F32)[..., None]
    cnt = mask.sum(1, keepdims=True).astype(F32)[..., None]
    if readout_fn == "last":
        g = np.zeros_like(emb)
        idx = mask.sum(1).astype(np.int64) - 1
        g[np.arange(B), idx, :] = dout
        return g
    if readout_fn == "mean":
        return (m * dout[:, None, :] / cnt).astype(F32)
    mean = (emb * m).sum(1, keepdims=True, dtype=F32) / cnt
    diff = emb - mean
    var = (diff * diff * m).sum(1, keepdims=True, dtype=F32) / cnt
    std = np.sqrt(var)
    if readout_fn == "std":
        dmean, dstd = None, dout[:, None, :]
    else:
        dmean, dstd = dout[:, None, :D], dout[:, None, D:]
    # std = sqrt(sum m (x-mean)^2 / cnt): d/dx_t = m_t (x_t-mean)/(cnt std) (the mean-path term
    # -(sum m diff)/cnt vanishes because sum_t m_t diff_t = 0)
    g = dstd * m * diff / (cnt * std)
    if dmean is not None:
        g = g + m * dmean / cnt
    return g.astype(F32)


def infonce_segmented(seg_out1, batch_out2, labels, temperature: float = 0.05, return_grad=False):
    """SegmentedBatchInfoNCELoss.forward, REF scripts/train_contrast.py:100-114:
    logits = P T^T / tau;  loss = -mean_i log(exp(l[i, y_i]) / sum_j exp(l[i, j]))  (rows only)."""
    p = seg_out1.astype(F32); t = batch_out2.astype(F32)
    logits = (p @ t.T) / F32(temperature)
    e = np.exp(logits, dtype=F32)
    den = e.sum(1, dtype=F32)
    num = e[np.arange(p.shape[0]), labels]
    loss = F32(-np.log(num / den, dtype=F32).mean(dtype=F32))
    if not return_grad:
        return loss
    sm = e / den[:, None]
    sm[np.arange(p.shape[0]), labels] -= F32(1.0)
    dlogits = sm / F32(p.shape[0])
    dp = (dlogits @ t) / F32(temperature)
    return loss, dp.astype(F32), logits


def infonce_batch(out1, out2, temperature: float = 0.05):
    """BatchInfoNCELoss.forward, REF scripts/train_contrast.py:86-91 (labels = diagonal)."""
    return infonce_segmented(out1, out2, np.arange(out1.shape[0]), temperature)


def infonce_columns(p_all, t_all, cols=None, temperature: float = 0.05, return_grad=False):
    """Column (text -> protein) term: the reference's BatchInfoNCELoss with its two arguments swapped
    (REF scripts/train_contrast.py:72-91 called as loss(text, protein)), i.e. F.cross_entropy(logits.T, arange):
    mean over the columns j in `cols` (default all) of logsumexp_i(l_ij) - l_jj, l = P T^T / tau, positives on the diagonal.
    return_grad: also d(sum_j (lse_j - l_jj)) / dP [N, D] over ALL columns (the caller scales by 1 / N), the gradient the
    protein side receives from the column term of the whole batch (the text side is frozen, :348-354)."""
    p = p_all.astype(F32); t = t_all.astype(F32)
    logits = (p @ t.T) / F32(temperature)                      # [i, j]
    e = np.exp(logits, dtype=F32)
    den = e.sum(0, dtype=F32)                                   # per column j, over proteins i
    col = (np.log(den, dtype=F32) - np.diag(logits)).astype(F32)
    cols = np.arange(p.shape[0]) if cols is None else np.asarray(cols)
    loss = F32(col[cols].mean(dtype=F32))
    if not return_grad:
        return loss
    sm = e / den[None, :]
    sm[np.arange(p.shape[0]), np.arange(p.shape[0])] -= F32(1.0)
    return loss, ((sm @ t) / F32(temperature)).astype(F32)


def l2_normalize_backward(x, dy, eps=1e-12):
    n = np.maximum(np.sqrt((x * x).sum(-1, keepdims=True, dtype=F32)), F32(eps))
    y = x / n
    return ((dy - y * (dy * y).sum(-1, keepdims=True, dtype=F32)) / n).astype(F32)


# ---------------------------------------------------------------------------------------------
# the contrastive step  (REF scripts/train_contrast.py:313-379 teacher_forcing_forward_pass)
# ---------------------------------------------------------------------------------------------
def protein_embeddings(esm_spec, W, prot_ids, prot_mask, readout="mix", ones_mask=False,
                       prec: Precision = FP32, keep=None):
    """Esm2LlamaInstructForCausalLM.forward(return_adapter_outputs=True) (REF models/...:175-193)
    -> readout -> F.normalize (train_contrast.py:361-365).  `ones_mask=True` reproduces the fork's
    all-ones readout mask (train_contrast.py:269-275)."""
    enc = esm2_forward(esm_spec, W, prot_ids, prot_mask, prec, prefix="esm_encoder.")
    ad = adapter_forward(W, enc, prec, prefix="adapter.", keep=keep)
    rmask = np.ones_like(prot_mask) if ones_mask else prot_mask
    pooled = readout_embeddings(ad, rmask, readout)
    if keep is not None:
        keep.update(enc=enc, adapter_out=ad, rmask=rmask, pooled=pooled)
    return l2_normalize(pooled)


def text_embeddings(llama_spec, W, text_ids, text_mask, layer=16, readout="mix", prec: Precision = FP32):
    """get_description_embeddings (train_contrast.py:284-310) + F.normalize (:354)."""
    hs = llama_hidden_state(llama_spec, W, text_ids, text_mask, layer, prec, prefix="llama_decoder.")
    return l2_normalize(readout_embeddings(hs, text_mask, readout))


def contrastive_loss(p_norm, t_norm, num_segments: int = 1, temperature: float = 0.05,
                     label_offset: int = 0):
    """Loop of train_contrast.py:356-379: mean over segments of the segmented InfoNCE.  Rows of
    p_norm beyond segment_size*num_segments are dropped (as the reference does, :337-343)."""
    B = p_norm.shape[0]
    seg = B // num_segments
    acc = F32(0.0)
    for s in range(num_segments):
        labels = np.arange(s * seg, (s + 1) * seg) + label_offset
        acc = acc + infonce_segmented(p_norm[s * seg:(s + 1) * seg], t_norm, labels, temperature)
    return F32(acc / F32(num_segments))


def contrastive_step(esm_spec, llama_spec, W, prot_ids, prot_mask, text_ids, text_mask, *,
                     layer=16, readout="mix", ones_mask=False, num_segments=1, temperature=0.05,
                     prec: Precision = FP32, with_grads=False, column_weight=0.0):
    """Forward (+ adapter gradients) of one contrastive step.  Returns a dict with the normalised
    pooled embeddings, the loss and, if requested, d loss / d adapter.{fc1,fc2}.{weight,bias}.
    column_weight = cw > 0 (not in the reference loop): loss = (1 - cw) * row term + cw * column term."""
    keep = {} if with_grads else None
    cw = F32(column_weight)
    t = text_embeddings(llama_spec, W, text_ids, text_mask, layer, readout, prec)
    p = protein_embeddings(esm_spec, W, prot_ids, prot_mask, readout, ones_mask, prec, keep)
    out = {"protein": p, "text": t, "loss": contrastive_loss(p, t, num_segments, temperature)}
    if column_weight > 0:
        out["loss"] = F32((F32(1.0) - cw) * out["loss"] + cw * infonce_columns(p, t, None, temperature))
    if with_grads:
        B = p.shape[0]
        seg = B // num_segments
        dp = np.zeros_like(p)
        for s in range(num_segments):
            labels = np.arange(s * seg, (s + 1) * seg)
            _, g, _ = infonce_segmented(p[s * seg:(s + 1) * seg], t, labels, temperature, return_grad=True)
            dp[s * seg:(s + 1) * seg] = g / F32(num_segments)
        if column_weight > 0:
            _, gcol = infonce_columns(p, t, None, temperature, return_grad=True)
            dp = (F32(1.0) - cw) * dp + cw * gcol / F32(B)
        dpooled = l2_normalize_backward(keep["pooled"], dp)
        dad = readout_backward(keep["adapter_out"], keep["rmask"], readout, dpooled)
        out["grads"] = adapter_backward(W, keep, dad, prec, prefix="adapter.")
    return out


# ---------------------------------------------------------------------------------------------
# stage 2 (SFT): LM loss through the frozen decoder and its gradient back to the adapter
# (REF models/modeling_esm2llama_instruct.py:108-139,195-215; scripts/train_instruct.py:192-213 `loss.backward()`).
# fp32 only; manual backward (what torch autograd does through HF LlamaForCausalLM, restated op by op).
# ---------------------------------------------------------------------------------------------
def rms_norm_backward(x, w, eps, dy):
    """y = x * rsqrt(mean(x^2) + eps) * w (LLAMA:62-67)  ->  dx."""
    x = x.astype(F32, copy=False)
    r = (F32(1.0) / np.sqrt((x * x).mean(-1, keepdims=True, dtype=F32) + F32(eps))).astype(F32)
    gy = (dy * w).astype(F32)
    return (r * gy - x * (r * r * r) * (gy * x).mean(-1, keepdims=True, dtype=F32)).astype(F32)


def rotate_half_transposed(y):
    """Transpose of rotate_half: rotate_half(x) = [-x2, x1]  ->  R^T y = [y2, -y1]."""
    h = y.shape[-1] // 2
    return np.concatenate([y[..., h:], -y[..., :h]], axis=-1)


def llama_lm_loss_and_grad(spec, W, inputs_embeds, mask, labels, prefix="llama_decoder."):
    """LlamaForCausalLM(inputs_embeds, attention_mask, labels): all layers -> final RMSNorm -> LM head -> shifted cross-entropy
    (ForCausalLMLoss: logits[:, :-1] against labels[:, 1:], ignore_index -100, mean over the counted positions), and the gradient
    of that loss with respect to inputs_embeds.  -> (loss, d_inputs_embeds [B, T, H], logits [B, T, V])."""
    x = np.ascontiguousarray(inputs_embeds, dtype=F32)
    mask = np.asarray(mask); labels = np.asarray(labels)
    B, T, H = x.shape
    L, nh, nkv, d = spec.num_hidden_layers, spec.num_attention_heads, spec.num_key_value_heads, spec.head_dim
    rep, scale = nh // nkv, F32(d ** -0.5)
    allowed = np.tril(np.ones((T, T), dtype=bool))[None, None] & (mask[:, None, None, :] != 0)
    bias = np.where(allowed, F32(0.0), NEG).astype(F32)
    cos, sin = rope_cos_sin(llama_inv_freq(spec), np.arange(T))
    saved = []
    for i in range(L):
        p = f"{prefix}model.layers.{i}."
        w1, w2 = W[p + "input_layernorm.weight"], W[p + "post_attention_layernorm.weight"]
        Wq, Wk, Wv, Wo = (W[p + f"self_attn.{n}_proj.weight"] for n in "qkvo")
        Wg, Wu, Wd = (W[p + f"mlp.{n}_proj.weight"] for n in ("gate", "up", "down"))
        h = rms_norm(x, w1, spec.rms_norm_eps).reshape(B * T, H)
        q = (h @ Wq.T).reshape(B, T, nh, d).transpose(0, 2, 1, 3)
        k = (h @ Wk.T).reshape(B, T, nkv, d).transpose(0, 2, 1, 3)
        v = (h @ Wv.T).reshape(B, T, nkv, d).transpose(0, 2, 1, 3)
        q = q * cos + rotate_half(q) * sin
        k = k * cos + rotate_half(k) * sin
        kr, vr = np.repeat(k, rep, axis=1), np.repeat(v, rep, axis=1)               # repeat_kv (LLAMA:181-188)
        P = softmax_rows(np.einsum("bhid,bhjd->bhij", q, kr) * scale + bias)
        o = np.einsum("bhij,bhjd->bhid", P, vr).transpose(0, 2, 1, 3).reshape(B * T, nh * d)
        x1 = x + (o @ Wo.T).reshape(B, T, H)
        h2 = rms_norm(x1, w2, spec.rms_norm_eps).reshape(B * T, H)
        g, u = h2 @ Wg.T, h2 @ Wu.T
        sg = (F32(1.0) / (F32(1.0) + np.exp(-g, dtype=F32))).astype(F32)
        x2 = x1 + ((g * sg * u) @ Wd.T).reshape(B, T, H)
        saved.append((x, q, kr, vr, P, x1, g, u, sg))
        x = x2.astype(F32)
    wn = W[prefix + "model.norm.weight"]
    Wlm = W[prefix + ("model.embed_tokens.weight" if spec.tie_word_embeddings else "lm_head.weight")]
    if not isinstance(Wlm, np.ndarray):                   # a row-gather table (oracle/weights.py): the tied LM head needs every row
        Wlm = Wlm[np.arange(spec.vocab_size)]
    logits = (rms_norm(x, wn, spec.rms_norm_eps).reshape(B * T, H) @ Wlm.T).reshape(B, T, -1).astype(F32)
    # ForCausalLMLoss (transformers/loss/loss_utils.py): shift, flatten, F.cross_entropy(ignore_index=-100, reduction mean)
    tgt = np.full((B, T), -100, dtype=np.int64)
    tgt[:, :-1] = labels[:, 1:]
    counted = tgt != -100
    n = max(int(counted.sum()), 1)
    z = logits.astype(np.float64)
    z = z - z.max(-1, keepdims=True)
    logp = z - np.log(np.exp(z).sum(-1, keepdims=True))
    bi, ti = np.nonzero(counted)
    loss = F32(-logp[bi, ti, tgt[bi, ti]].sum() / n)
    dlogits = np.exp(logp)
    dlogits[bi, ti, tgt[bi, ti]] -= 1.0
    dlogits = (dlogits * counted[..., None] / n).astype(F32)
    # ---- backward
    dx = rms_norm_backward(x, wn, spec.rms_norm_eps, (dlogits.reshape(B * T, -1) @ Wlm).reshape(B, T, H))
    for i in reversed(range(L)):
        p = f"{prefix}model.layers.{i}."
        w1, w2 = W[p + "input_layernorm.weight"], W[p + "post_attention_layernorm.weight"]
        Wq, Wk, Wv, Wo = (W[p + f"self_attn.{n}_proj.weight"] for n in "qkvo")
        Wg, Wu, Wd = (W[p + f"mlp.{n}_proj.weight"] for n in ("gate", "up", "down"))
        x0, q, kr, vr, P, x1, g, u, sg = saved[i]
        da = dx.reshape(B * T, H) @ Wd
        dg = da * u * (sg * (F32(1.0) + g * (F32(1.0) - sg)))
        du = da * (g * sg)
        dx1 = dx + rms_norm_backward(x1, w2, spec.rms_norm_eps, (dg @ Wg + du @ Wu).reshape(B, T, H))
        do = (dx1.reshape(B * T, H) @ Wo).reshape(B, T, nh, d).transpose(0, 2, 1, 3)
        dvr = np.einsum("bhij,bhid->bhjd", P, do)
        dP = np.einsum("bhid,bhjd->bhij", do, vr)
        dS = P * (dP - (dP * P).sum(-1, keepdims=True, dtype=F32))              # softmax backward; (dP * P).sum = rowsum(dO * O)
        dq = np.einsum("bhij,bhjd->bhid", dS, kr) * scale
        dkr = np.einsum("bhij,bhid->bhjd", dS, q) * scale
        dk = dkr.reshape(B, nkv, rep, T, d).sum(2)                                # the repeated heads share one key / value head
        dv = dvr.reshape(B, nkv, rep, T, d).sum(2)
        dq = dq * cos + rotate_half_transposed(dq * sin)
        dk = dk * cos + rotate_half_transposed(dk * sin)
        flat = lambda t, heads: t.transpose(0, 2, 1, 3).reshape(B * T, heads * d)
        dh = flat(dq, nh) @ Wq + flat(dk, nkv) @ Wk + flat(dv, nkv) @ Wv
        dx = (dx1 + rms_norm_backward(x0, w1, spec.rms_norm_eps, dh.reshape(B, T, H))).astype(F32)
    return loss, dx, logits


def sft_decoder_inputs(llama_spec, W, input_ids, adapter_out, prot_mask, placeholder_id, prefix="llama_decoder."):
    """prepare_decoder_inputs (REF models/modeling_esm2llama_instruct.py:108-139): token embeddings with the placeholder positions
    replaced by the adapter rows under the protein mask, both in row-major order."""
    emb = W[prefix + "model.embed_tokens.weight"][np.asarray(input_ids)].astype(F32)
    emb[np.asarray(input_ids) == placeholder_id] = adapter_out[np.asarray(prot_mask) != 0]
    return emb


def sft_step(esm_spec, llama_spec, W, prot_ids, prot_mask, input_ids, attention_mask, labels, placeholder_id):
    """One stage-2 step with the towers frozen and the adapter trainable (REF scripts/train_instruct.py:192-213 without LoRA):
    -> dict(loss, logits, inputs_embeds, d_inputs_embeds, grads of adapter.{fc1,fc2}.{weight,bias})."""
    keep = {}
    enc = esm2_forward(esm_spec, W, prot_ids, prot_mask, FP32, prefix="esm_encoder.")
    ad = adapter_forward(W, enc, FP32, prefix="adapter.", keep=keep)
    emb = sft_decoder_inputs(llama_spec, W, input_ids, ad, prot_mask, placeholder_id)
    loss, demb, logits = llama_lm_loss_and_grad(llama_spec, W, emb, attention_mask, labels)
    dad = np.zeros_like(ad)
    dad[np.asarray(prot_mask) != 0] = demb[np.asarray(input_ids) == placeholder_id]      # backward of the boolean-mask assignment
    return {"loss": loss, "logits": logits, "inputs_embeds": emb, "d_inputs_embeds": demb, "d_adapter_out": dad,
            "grads": adapter_backward(W, keep, dad, FP32, prefix="adapter.")}


# ---------------------------------------------------------------------------------------------
# generation  (REF models/modeling_esm2llama_instruct.py:217-251 -> HF GenerationMixin.generate with `inputs_embeds`;
# GEN = transformers/generation/utils.py).  No KV cache here: every step re-runs the decoder over the whole row, which is the
# same function of (prompt, tokens so far) -- and so an independent check of the cached HIP path.
# ---------------------------------------------------------------------------------------------
def _lm_head_weight(spec, W, prefix):
    Wlm = W[prefix + ("model.embed_tokens.weight" if spec.tie_word_embeddings else "lm_head.weight")]
    return Wlm if isinstance(Wlm, np.ndarray) else Wlm[np.arange(spec.vocab_size)]


def llama_next_token_logits(spec, W, rows, prec: Precision = FP32, prefix="llama_decoder."):
    """rows: one f32 [n_b, H] array per sequence = the embeddings of its tokens UNDER the attention mask, in order.  HF gives the
    token under the mask with rank r the position r (`position_ids = cumsum(attention_mask) - 1`, GEN prepare_inputs_for_generation)
    and hides the masked ones from every query, so dropping them is exact.  -> logits of the last position, f32 [B, V] (rounded to
    the model dtype first, as HF keeps `outputs.logits` in it before the f32 up-cast of `_sample`)."""
    Wlm = prec.q(_lm_head_weight(spec, W, prefix))            # the LM head stays in the model dtype under fp8 tower GEMMs too
    out = []
    for x in rows:
        n = x.shape[0]
        bias = np.where(np.tril(np.ones((n, n), dtype=bool))[None, None], F32(0.0), NEG).astype(F32)
        cos, sin = rope_cos_sin(llama_inv_freq(spec), np.arange(n))
        h = np.ascontiguousarray(x[None], dtype=F32)
        for i in range(spec.num_hidden_layers):
            h = llama_layer(spec, W, i, h, bias, cos, sin, prec, prefix)
        last = prec.q(rms_norm(h[0, -1:], W[prefix + "model.norm.weight"], spec.rms_norm_eps))
        out.append(prec.q((last @ Wlm.T).astype(F32))[0])
    return np.stack(out).astype(F32)


def _prompt_rows(inputs_embeds, mask):
    return [np.ascontiguousarray(inputs_embeds[b][np.asarray(mask[b]) != 0], dtype=F32) for b in range(len(inputs_embeds))]


def generate_greedy(spec, W, inputs_embeds, mask, max_new_tokens, eos_ids=(), pad_id=0, prec: Precision = FP32,
                    prefix="llama_decoder.", forced=None):
    """GEN `_sample` with do_sample=False: argmax of the last logits (first index on ties), finished rows emit pad_id, a row finishes
    when it emits an eos id, the loop stops once every row has.  `forced` (i64 [B, n]): feed these tokens instead of the argmax
    (teacher forcing, to compare per-step logits of a lower-precision path on the same prefix).
    -> (tokens i64 [B, n], per-step logits f32 [n, B, V])."""
    rows = _prompt_rows(inputs_embeds, mask)
    emb = W[prefix + "model.embed_tokens.weight"]
    B = len(rows)
    finished = np.zeros(B, dtype=bool)
    toks, logs = [], []
    for step in range(max_new_tokens):
        lg = llama_next_token_logits(spec, W, rows, prec, prefix)
        nxt = lg.argmax(-1).astype(np.int64) if forced is None else np.asarray(forced)[:, step].astype(np.int64)
        if forced is None:
            nxt[finished] = pad_id
        toks.append(nxt); logs.append(lg)
        finished |= np.isin(nxt, list(eos_ids))
        if forced is None and eos_ids and finished.all():
            break
        rows = [np.concatenate([r, emb[np.asarray([t])].astype(F32)], 0) for r, t in zip(rows, nxt)]
    return np.stack(toks, 1), np.stack(logs, 0)


def _log_softmax(x):
    m = x.max(-1, keepdims=True)
    return (x - m - np.log(np.exp(x - m, dtype=F32).sum(-1, keepdims=True, dtype=F32))).astype(F32)


def _topk(x, k):
    """torch.topk along the last axis: values descending, lowest index first among equal values."""
    idx = np.argsort(-x, axis=-1, kind="stable")[..., :k]
    return np.take_along_axis(x, idx, -1), idx


def generate_beam(spec, W, inputs_embeds, mask, max_new_tokens, num_beams, eos_ids=(), pad_id=0, length_penalty=1.0,
                  early_stopping=False, prec: Precision = FP32, prefix="llama_decoder.", num_return_sequences=1):
    """GEN `_beam_search` (5.x vectorised form) with an empty id prompt (decoder_prompt_len 0, max_length = max_new_tokens): per
    step the top max(2, 1 + n_eos) * num_beams continuations of every prompt by accumulated log-probability; those that hit a
    stopping criterion (eos, max length) compete -- length-normalised by (cur_len + 1) ** length_penalty -- for the num_beams
    finished slots if they were among the top num_beams, the best num_beams others go on; the loop ends when no running beam can
    still beat the worst finished one (the early_stopping=False heuristic), or nothing is left to continue.
    -> (sequences i64 [B * num_return_sequences, n]: the best hypotheses of every prompt in order, their scores f32)."""
    rows0 = _prompt_rows(inputs_embeds, mask)
    emb = W[prefix + "model.embed_tokens.weight"]
    B, nb, V, L = len(rows0), num_beams, spec.vocab_size, max_new_tokens
    keep = max(2, 1 + len(eos_ids)) * nb
    top_mask = np.arange(keep) < nb
    running = np.full((B, nb, L), pad_id, dtype=np.int64)
    sequences = running.copy()
    run_sc = np.zeros((B, nb), dtype=F32); run_sc[:, 1:] = -1e9
    beam_sc = np.full((B, nb), -1e9, dtype=F32)
    done = np.zeros((B, nb), dtype=bool)
    lens_done = np.zeros((B, nb), dtype=np.int64)
    open_ = np.ones((B, 1), dtype=bool)
    cur = 0
    gather = lambda t, i: np.take_along_axis(t, i.reshape(i.shape + (1,) * (t.ndim - 2)), 1)
    while True:
        rows = [np.concatenate([rows0[b], emb[running[b, j, :cur]].astype(F32)], 0) for b in range(B) for j in range(nb)]
        lp = _log_softmax(llama_next_token_logits(spec, W, rows, prec, prefix)).reshape(B, nb, V)
        acc = (lp + run_sc[:, :, None]).reshape(B, nb * V)
        top_lp, top_i = _topk(acc, keep)
        src = top_i // V
        top_seq = gather(running, src).copy()
        top_seq[:, :, cur] = top_i % V
        hits = np.full((B, keep), cur + 1 >= L) | np.isin(top_seq[:, :, cur], list(eos_ids))
        live = (top_lp + hits.astype(F32) * F32(-1e9)).astype(F32)
        _, nxt = _topk(live, nb)
        running, run_sc = gather(top_seq, nxt), gather(live, nxt)
        just = hits & top_mask[None]
        fin = (top_lp / F32((cur + 1) ** length_penalty)).astype(F32)
        full = done.all(-1, keepdims=True) & (early_stopping is True)
        fin = fin + full.astype(F32) * F32(-1e9) + (~open_).astype(F32) * F32(-1e9) + (~just).astype(F32) * F32(-1e9)
        m_seq, m_sc = np.concatenate([sequences, top_seq], 1), np.concatenate([beam_sc, fin], 1)
        m_done = np.concatenate([done, just], 1)
        m_len = np.concatenate([lens_done, np.full((B, keep), cur + 1)], 1)
        _, sel = _topk(m_sc, nb)
        sequences, beam_sc, done, lens_done = gather(m_seq, sel), gather(m_sc, sel), gather(m_done, sel), gather(m_len, sel)
        cur += 1
        best_len = L if (early_stopping == "never" and length_penalty > 0.0) else cur
        best_running = run_sc[:, :1] / F32(best_len ** length_penalty)
        worst_done = np.where(done, beam_sc.min(1, keepdims=True), F32(-1e9))
        open_ = open_ & (best_running > worst_done).any(-1, keepdims=True)
        if not (open_.any() and not (done.all() and early_stopping is True) and not hits.all()):
            break
    R = num_return_sequences
    n = int(lens_done[:, :R].max())
    return sequences[:, :R, :n].reshape(B * R, n), beam_sc[:, :R].reshape(B * R)


# ---------------------------------------------------------------------------------------------
# optimizer tail  (REF scripts/train_contrast.py:453-465,621-626: clip_grad_norm_ then AdamW)
# ---------------------------------------------------------------------------------------------
def clip_and_adamw(params, grads, m, v, step, lr=2e-4, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.01,
                   max_norm=float("inf")):
    """torch.nn.utils.clip_grad_norm_ (total L2 norm, coef = max_norm/(norm+1e-6) clamped to 1)
    followed by torch.optim.AdamW (decoupled decay, bias correction).  In place; returns grad norm."""
    tot = F32(math.sqrt(sum(float((g.astype(np.float64) ** 2).sum()) for g in grads.values())))
    coef = min(1.0, max_norm / (float(tot) + 1e-6)) if math.isfinite(max_norm) else 1.0
    b1, b2 = betas
    for k in params:
        g = grads[k] * F32(coef)
        params[k] *= F32(1.0 - lr * weight_decay)
        m[k][...] = F32(b1) * m[k] + F32(1.0 - b1) * g
        v[k][...] = F32(b2) * v[k] + F32(1.0 - b2) * g * g
        bc1 = 1.0 - b1 ** step
        bc2 = 1.0 - b2 ** step
        denom = np.sqrt(v[k]) / F32(math.sqrt(bc2)) + F32(eps)
        params[k] -= F32(lr / bc1) * (m[k] / denom)
    return tot
