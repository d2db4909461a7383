"""Stage-2 step of the decoder with LoRA adapters and / or per-head q / k RMSNorm:

    reference  scripts/train_instruct.py:146-183   LoraConfig(r, lora_alpha = 2 r, lora_dropout = 0.1, bias = "none") on the seven
                                                   decoder projections, modules_to_save = adapter.fc1 / fc2
               scripts/train_instruct.py:192-213   loss = model(**batch).loss; loss.backward()
               models/esmc_config.py:9             the fork's text tower is a Qwen3 (per-head q_norm / k_norm)

`p2t_llama_train_forward / _backward` (csrc/llama_train.hip) run the FROZEN decoder as fused blocks and hand back only the
gradient at its inputs.  A LoRA branch sits between those blocks (y = W x + (alpha / r) B (A drop(x)) ahead of the rotation / the
SwiGLU), and so does Qwen3's q / k norm, so this module drives the same HIP kernels one by one through the C ABI, per layer:
p2t_rmsnorm, p2t_gemm_nt (frozen weights: MFMA where K allows; low-rank products: the same entry point), p2t_qkv_post,
p2t_attention (+ log-sum-exps), p2t_swiglu_gu, and backwards p2t_gemm_nt on transposed weights, p2t_attention_backward,
p2t_rope_backward_pack, p2t_rmsnorm_backward, p2t_transpose (the token axis made contiguous for dA / dB), p2t_dropout_rows (the
branch's input dropout: a counter-hash mask regenerated in the backward, never stored).  torch allocates, slices, concatenates and
wires autograd; no torch op computes on the path.

Parity: tests/golden/sft_lora_tiny.npz = torch autograd through the REFERENCE class with every target wrapped by a hand-written
LoRA linear (tests/golden/make_golden.py run_sft_lora) and, for Qwen3, through HF Qwen3ForCausalLM.  `peft` is not importable in
this image: the arithmetic is LoRA's published one, parity against peft's own code is UNPINNED (as is the checkpoint key layout
`peft_state_dict` writes; p2t_hip/lora.py reads the same restated layout).
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence, Tuple

import torch
from torch import nn

from . import _lib, ops
from ._lib import call
from .ops import ptr, round_up, stream

TARGETS = ("self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj", "self_attn.o_proj", "mlp.gate_proj", "mlp.up_proj", "mlp.down_proj")


class DecoderLora(nn.Module):
    """The trainable A [r, in] / B [out, r] pairs of `LoraConfig(r, lora_alpha, lora_dropout, target_modules)` on a LlamaDecoder.
    Initialised as peft does (`init_lora_weights=True`: A ~ kaiming-uniform(a = sqrt 5), B = 0), kept in fp32."""

    def __init__(self, decoder, r: int, lora_alpha: Optional[float] = None, lora_dropout: float = 0.1, target_modules: Sequence[str] = TARGETS, seed: int = 0):
        super().__init__()
        if r < 1:
            raise ValueError("LoRA rank must be >= 1")
        bad = [t for t in target_modules if t not in TARGETS]
        if bad:
            raise ValueError(f"unsupported LoRA targets {bad}; decoder targets are {TARGETS}")
        self.r, self.alpha, self.p, self.targets = int(r), float(2 * r if lora_alpha is None else lora_alpha), float(lora_dropout), tuple(target_modules)
        self.seed, self.step_count = int(seed), 0
        s = decoder.spec
        dev = decoder.model.embed_tokens.weight.device
        P = dict(decoder.model.named_parameters())
        gen = torch.Generator(device="cpu").manual_seed(seed)
        for i in range(s.num_hidden_layers):
            for t in self.targets:
                w = P[f"layers.{i}.{t}.weight"]
                a = torch.empty((r, w.shape[1]), dtype=torch.float32)
                bound = 1.0 / math.sqrt(w.shape[1])                    # kaiming_uniform_(a = sqrt(5)) on [r, in]
                a.uniform_(-bound, bound, generator=gen)
                self.register_parameter(self._name(i, t, "A"), nn.Parameter(a.to(dev)))
                self.register_parameter(self._name(i, t, "B"), nn.Parameter(torch.zeros((w.shape[0], r), dtype=torch.float32, device=dev)))

    @staticmethod
    def _name(i: int, target: str, which: str) -> str:
        return f"l{i}_{target.replace('.', '_')}_{which}"

    @property
    def scale(self) -> float:
        return self.alpha / self.r

    def get(self, i: int, target: str) -> Optional[Tuple[nn.Parameter, nn.Parameter]]:
        if target not in self.targets:
            return None
        return getattr(self, self._name(i, target, "A")), getattr(self, self._name(i, target, "B"))

    def peft_state_dict(self, prefix: str = "base_model.model.llama_decoder.model.") -> Dict[str, torch.Tensor]:
        """The adapter in the key layout p2t_hip/lora.py reads (peft's, restated without the library: unverified against it)."""
        out = {}
        for name, p in self.named_parameters():
            li, rest = name[1:].split("_", 1)
            tgt, which = rest.rsplit("_", 1)
            tgt = tgt.replace("self_attn_", "self_attn.").replace("mlp_", "mlp.")
            out[f"{prefix}layers.{li}.{tgt}.lora_{which}.weight"] = p.detach().clone()
        return out


def _transposed(decoder, name: str) -> torch.Tensor:
    """[in, out padded to 64] copy of a frozen projection for the dX GEMMs, built once per weight version."""
    cache = decoder.model.__dict__.setdefault("_dt_wT", {})
    w = dict(decoder.model.named_parameters())[name]
    key = (w.data_ptr(), w._version)
    hit = cache.get(name)
    if hit is None or hit[0] != key:
        cache[name] = (key, ops.transpose(w.detach(), round_up(w.shape[0], 64)))
    return cache[name][1]


class _Lin:
    """One projection of one layer: frozen W [N, K] (+ LoRA A, B), forward y = W x + s B A drop(x) and the pieces of its backward."""

    def __init__(self, decoder, lora: Optional[DecoderLora], i: int, target: str, dt):
        self.name = f"layers.{i}.{target}.weight"
        self.w = dict(decoder.model.named_parameters())[self.name].detach()
        self.decoder, self.dt = decoder, dt
        self.N, self.K = self.w.shape
        ab = lora.get(i, target) if lora is not None else None
        self.lora = None
        if ab is not None:
            a, b = ab
            s = lora.scale
            r = a.shape[0]
            rp = round_up(r, 16)                        # p2t_gemm_nt wants N % 16 == 0: the rank axis is zero-padded to rp everywhere
            # operands of the low-rank products in the model dtype (the masters stay fp32): A [rp, K], s B [N, 64 k], zero padded
            a16 = torch.zeros((rp, round_up(self.K, 8)), dtype=dt, device=a.device)
            a16[:r, :self.K] = a.detach().to(dt)
            self.a16 = a16
            bs = torch.zeros((self.N, round_up(rp, 64)), dtype=dt, device=a.device)
            bs[:, :r] = (b.detach() * s).to(dt)
            self.bs16 = bs
            self.rp = rp
            self.lora = (a, b, r, s, lora.p, (lora.seed * 1000003 + lora.step_count * 7919 + i * 131 + TARGETS.index(target)) & 0x7FFFFFFFFFFFFFFF)

    # -- forward: f32 [M, N] (or accumulated into the fp32 residual stream `resid`)
    def forward(self, x: torch.Tensor, resid: Optional[torch.Tensor] = None):
        epi = _lib.EPI_RESID if resid is not None else _lib.EPI_STORE_F32
        y = ops.gemm_nt(x, self.w, None, n=self.N, k=self.K, epilogue=epi, out=resid)
        u = None
        if self.lora is not None:
            _, _, r, _, p, seed = self.lora
            xd = self.dropped(x)
            u = ops.gemm_nt(xd, self.a16, None, n=self.rp, k=self.K, epilogue=_lib.EPI_STORE, out_dtype=self.dt)    # [M, 64]: u = drop(x) A^T
            if resid is not None:
                ops.gemm_nt(u, self.bs16, None, n=self.N, k=self.rp, epilogue=_lib.EPI_RESID, out=resid)
            else:
                ops.gemm_nt(u, self.bs16, None, n=self.N, k=self.rp, epilogue=_lib.EPI_STORE_F32, out=y, accumulate=True)
        return y, u

    def dropped(self, x: torch.Tensor) -> torch.Tensor:
        _, _, _, _, p, seed = self.lora
        if p <= 0.0:
            return x
        xd = torch.empty((x.shape[0], round_up(self.K, 64)), dtype=self.dt, device=x.device)
        if xd.shape[1] != self.K:
            xd.zero_()
        call("p2t_dropout_rows", ptr(x), ops.dt_of(x), x.stride(0), ptr(xd), ops.dt_of(xd), xd.stride(0), x.shape[0], self.K, float(p), int(seed), 0, stream())
        return xd

    # -- backward: dy `dt` [M, >= N] -> dX; LoRA gradients into `grads`
    def backward(self, dy: torch.Tensor, x: torch.Tensor, u: Optional[torch.Tensor], out: Optional[torch.Tensor], out_f32: bool, accumulate: bool, grads: dict):
        """dX (+)= dy W  (+ the branch's share); x: the projection's input as the forward saw it (before the dropout)."""
        wT = _transposed(self.decoder, self.name)                                                   # [K, N padded]
        if out_f32:
            dx = ops.gemm_nt(dy, wT, None, n=self.K, k=self.N, epilogue=_lib.EPI_STORE_F32, out=out, accumulate=accumulate)
        else:
            assert not accumulate
            dx = ops.gemm_nt(dy, wT, None, n=self.K, k=self.N, epilogue=_lib.EPI_STORE, out=out, out_dtype=self.dt)
        if self.lora is None:
            return dx
        a, b, r, s, p, seed = self.lora
        rp = self.rp
        M = dy.shape[0]
        xd = self.dropped(x)
        # du = dy (s B)  [M, rp]; dB = s dy^T u; dA = du^T drop(x); dX += drop'(du A)
        bsT = ops.transpose(self.bs16[:, :rp].contiguous(), round_up(self.N, 64))                   # [rp, N]
        du = ops.gemm_nt(dy, bsT, None, n=rp, k=self.N, epilogue=_lib.EPI_STORE, out_dtype=self.dt) # [M, 64]
        dyT, uT = ops.transpose(dy[:, :self.N]), ops.transpose(u[:, :rp])                           # [N, Mp], [rp, Mp] (token axis contiguous, zero padded)
        _zero_tail(dyT, M), _zero_tail(uT, M)
        dB = torch.zeros((self.N, rp), dtype=torch.float32, device=dy.device)
        ops.gemm_nt(dyT, uT, None, n=rp, k=round_up(M, 64), epilogue=_lib.EPI_STORE_F32, out=dB)    # [N, rp] = dy^T u
        duT, xdT = ops.transpose(du[:, :rp]), ops.transpose(xd[:, :self.K])
        _zero_tail(duT, M), _zero_tail(xdT, M)
        dA = ops.gemm_nt(duT, xdT, None, n=self.K, k=round_up(M, 64), epilogue=_lib.EPI_STORE_F32)  # [rp, K] = du^T drop(x)
        grads[id(a)] = (dA, r, self.K, 1.0)           # (buffer, rows, columns, factor on top of the upstream gradient)
        grads[id(b)] = (dB, self.N, r, s)
        aT = ops.transpose(self.a16[:, :self.K], round_up(rp, 8))                                   # [K, rp] = A^T
        t = ops.gemm_nt(du, aT, None, n=self.K, k=rp, epilogue=_lib.EPI_STORE_F32)                  # [M, K] f32
        call("p2t_dropout_rows", ptr(t), _lib.F32, t.stride(0), ptr(dx), ops.dt_of(dx), dx.stride(0), M, self.K, float(p), int(seed), 1, stream())
        return dx


def _zero_tail(t: torch.Tensor, m: int):
    if t.shape[1] > m:
        t[:, m:].zero_()


def _interleave(g: torch.Tensor, u: torch.Tensor, F: int) -> torch.Tensor:
    """[M, F] gate, up -> [M, 2F] in the 32-column gate / up blocks of p2t_llama_layer.gu_w (a copy, no arithmetic)."""
    M = g.shape[0]
    return torch.stack([g[:, :F].reshape(M, F // 32, 32), u[:, :F].reshape(M, F // 32, 32)], 2).reshape(M, 2 * F).contiguous()


def _deinterleave(d_gu: torch.Tensor, F: int):
    M = d_gu.shape[0]
    v = d_gu[:, :2 * F].reshape(M, F // 32, 2, 32)
    return v[:, :, 0].reshape(M, F).contiguous(), v[:, :, 1].reshape(M, F).contiguous()


class DecoderLoraLossFn(torch.autograd.Function):
    """LM loss of the decoder as a function of `inputs_embeds` and the LoRA parameters (frozen base weights)."""

    @staticmethod
    def forward(ctx, inputs_embeds, decoder, lora, attention_mask, labels, *params):
        s, m = decoder.spec, decoder.model
        dt = m.dtype
        B, T, H = inputs_embeds.shape
        M = B * T
        nh, nkv, d, F, L = s.num_attention_heads, s.num_key_value_heads, s.head_dim, s.intermediate_size, s.num_hidden_layers
        if F % 32:
            raise ValueError("Llama intermediate_size must be a multiple of 32")
        dev = m.embed_tokens.weight.device
        P = dict(m.named_parameters())
        mask = attention_mask.to(device=dev, dtype=torch.int64).contiguous()
        key_mask, kv_info, _ = ops.mask_prepare(mask)
        inv_freq = m._inv_freq().to(dev)
        l2s = dt == torch.bfloat16                       # as the towers: q carries scale * log2 e, the logits are ln 2 * q.k
        scale = float(d) ** -0.5
        q_fold = scale * 1.4426950408889634 if l2s else 1.0
        x = inputs_embeds.detach().to(device=dev, dtype=torch.float32).reshape(M, H).contiguous().clone()
        tape = []
        f32v = lambda n: P[n].detach().float().contiguous()
        for i in range(L):
            p = f"layers.{i}."
            lin = {t: _Lin(decoder, lora, i, t, dt) for t in TARGETS}
            rec = dict(lin=lin, x_in=x.clone())
            h = ops.rmsnorm(x, f32v(p + "input_layernorm.weight"), s.rms_norm_eps, out_dtype=dt)
            parts = []
            for t in ("self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj"):
                y, u = lin[t].forward(h)
                rec["u_" + t] = u
                parts.append(y)
            if s.qk_norm:                               # Qwen3Attention.forward: RMSNorm over head_dim of every head, before the rotation
                rec["q_raw"], rec["k_raw"] = parts[0], parts[1]
                parts[0] = ops.rmsnorm(parts[0].view(M * nh, d), f32v(p + "self_attn.q_norm.weight"), s.rms_norm_eps, ld_out=d).view(M, nh * d)
                parts[1] = ops.rmsnorm(parts[1].view(M * nkv, d), f32v(p + "self_attn.k_norm.weight"), s.rms_norm_eps, ld_out=d).view(M, nkv * d)
            qkv = ops.cast(torch.cat(parts, 1), dt) if dt != torch.float32 else torch.cat(parts, 1)
            if qkv.shape[1] % 8:
                qkv = torch.nn.functional.pad(qkv, (0, 8 - qkv.shape[1] % 8))
            q4, k4, v4 = ops.qkv_post(qkv.contiguous(), inv_freq, B, T, nh, nkv, d, q_fold)
            lse = torch.empty((B, nh, T), dtype=torch.float32, device=dev)
            ao = ops.attention(q4, k4, v4, key_mask, kv_info, d, 1.0 if l2s else scale, True, log2_scores=l2s, lse=lse)      # [M, QO]
            rec.update(q=q4, k=k4, v=v4, lse=lse, ao=ao)
            _, rec["u_self_attn.o_proj"] = lin["self_attn.o_proj"].forward(ao, resid=x)
            rec["x_mid"] = x.clone()
            h2 = ops.rmsnorm(x, f32v(p + "post_attention_layernorm.weight"), s.rms_norm_eps, out_dtype=dt)
            g, rec["u_mlp.gate_proj"] = lin["mlp.gate_proj"].forward(h2)
            up, rec["u_mlp.up_proj"] = lin["mlp.up_proj"].forward(h2)
            gu = _interleave(g, up, F)
            gu = ops.cast(gu, dt) if dt != torch.float32 else gu
            rec["gu"] = gu
            act = torch.empty((M, round_up(F, 64)), dtype=dt, device=dev)
            call("p2t_swiglu_gu", ptr(gu), gu.stride(0), None, 0, ptr(act), act.stride(0), M, F, ops.dt_of(dt), stream())
            _, rec["u_mlp.down_proj"] = lin["mlp.down_proj"].forward(act, resid=x)
            tape.append(rec)
        x_last = x
        hN = ops.rmsnorm(x_last, f32v("norm.weight"), s.rms_norm_eps, out_dtype=dt)
        logits = ops.gemm_nt(hN, decoder._lm_head_padded(), None, n=s.vocab_size, k=H, out_dtype=dt).view(B, T, -1)
        lab = labels.to(dev).to(torch.int64).contiguous()
        loss, count = ops.cross_entropy_shifted(logits, lab, s.vocab_size)
        ctx.state = dict(decoder=decoder, tape=tape, x_last=x_last, logits=logits, labels=lab, count=count, key_mask=key_mask, kv_info=kv_info,
                         inv_freq=inv_freq, l2s=l2s, scale=scale, q_fold=q_fold, shape=(B, T, H), params=params, in_dtype=inputs_embeds.dtype)
        ctx.mark_non_differentiable(logits)
        return loss[0], logits

    @staticmethod
    def backward(ctx, g_loss, _g_logits):
        st = ctx.state
        dec = st["decoder"]
        s, m = dec.spec, dec.model
        dt = m.dtype
        B, T, H = st["shape"]
        M = B * T
        nh, nkv, d, F = s.num_attention_heads, s.num_key_value_heads, s.head_dim, s.intermediate_size
        dp = ops.head_dim_padded(d)
        V = s.vocab_size
        dev = st["x_last"].device
        P = dict(m.named_parameters())
        f32v = lambda n: P[n].detach().float().contiguous()
        logits = st["logits"]
        ld = logits.shape[2]
        d_logits = torch.empty_like(logits)
        call("p2t_cross_entropy_shifted_backward", ptr(logits), ld, ops.dt_of(logits), ptr(st["labels"]), B, T, V, -100, ptr(st["count"]), ptr(d_logits), ld, stream())
        d_h = ops.gemm_nt(d_logits.view(M, ld), dec._lm_head_transposed(), None, n=H, k=round_up(V, 64), epilogue=_lib.EPI_STORE_F32)      # [M, H] f32
        g = torch.empty((M, H), dtype=torch.float32, device=dev)

        def rms_bwd(x, w, dy, out, acc, rows=M, cols=H):
            call("p2t_rmsnorm_backward", ptr(x), x.stride(0), ptr(w), float(s.rms_norm_eps), ptr(dy), dy.stride(0), 0 if dy.dtype == torch.float32 else 1,
                 ptr(out), out.stride(0), rows, cols, int(acc), stream())
        rms_bwd(st["x_last"], f32v("norm.weight"), d_h, g, 0)
        grads: dict = {}
        c_s = 0.6931471805599453 if st["l2s"] else st["scale"]
        to_dt = lambda t: ops.cast(t, dt) if t.dtype != dt else t
        for i in range(len(st["tape"]) - 1, -1, -1):
            rec = st["tape"][i]
            lin = rec["lin"]
            p = f"layers.{i}."
            # ---- MLP branch: x2 = x1 + down(silu(g) u)
            g16 = to_dt(g)
            act = torch.empty((M, round_up(F, 64)), dtype=dt, device=dev)
            call("p2t_swiglu_gu", ptr(rec["gu"]), rec["gu"].stride(0), None, 0, ptr(act), act.stride(0), M, F, ops.dt_of(dt), stream())
            d_act = lin["mlp.down_proj"].backward(g16, act, rec["u_mlp.down_proj"], None, False, False, grads)                  # [M, Fp]
            d_gu = torch.empty((M, 2 * F), dtype=dt, device=dev)
            call("p2t_swiglu_gu", ptr(rec["gu"]), rec["gu"].stride(0), ptr(d_act), d_act.stride(0), ptr(d_gu), d_gu.stride(0), M, F, ops.dt_of(dt), stream())
            d_gate, d_up = _deinterleave(d_gu, F)
            h2 = ops.rmsnorm(rec["x_mid"], f32v(p + "post_attention_layernorm.weight"), s.rms_norm_eps, out_dtype=dt)
            d_h2 = lin["mlp.gate_proj"].backward(d_gate, h2, rec["u_mlp.gate_proj"], None, True, False, grads)
            lin["mlp.up_proj"].backward(d_up, h2, rec["u_mlp.up_proj"], d_h2, True, True, grads)
            rms_bwd(rec["x_mid"], f32v(p + "post_attention_layernorm.weight"), d_h2, g, 1)
            # ---- attention branch: x1 = x + o(attn(...))
            g16 = to_dt(g)
            d_ao = lin["self_attn.o_proj"].backward(g16, rec["ao"], rec["u_self_attn.o_proj"], None, False, False, grads)       # [M, QO]
            dq, dk, dv = ops.attention_backward(rec["q"], rec["k"], rec["v"], rec["ao"], d_ao, rec["lse"], st["key_mask"], st["kv_info"], d, c_s, True,
                                                log2_scores=st["l2s"])
            NQ = (nh + 2 * nkv) * d
            d_qkv = torch.zeros((M, round_up(NQ, 64)), dtype=dt, device=dev)
            cs = torch.empty((T, d), dtype=torch.float32, device=dev)
            call("p2t_rope_backward_pack", ptr(dq), ptr(dk), ptr(dv), ptr(st["inv_freq"]), ptr(cs), ptr(d_qkv), d_qkv.stride(0), B, T, nh, nkv, d, dp,
                 float(st["q_fold"]), ops.dt_of(dt), stream())
            d_q, d_k, d_v = d_qkv[:, :nh * d], d_qkv[:, nh * d:(nh + nkv) * d], d_qkv[:, (nh + nkv) * d:NQ]
            if s.qk_norm:
                def norm_bwd(raw, w, dy, heads):
                    dyc = dy.contiguous().view(M * heads, d)
                    out = torch.empty((M * heads, d), dtype=torch.float32, device=dev)
                    rms_bwd(raw.view(M * heads, d), f32v(w), dyc, out, 0, rows=M * heads, cols=d)
                    return to_dt(out.view(M, heads * d))
                d_q = norm_bwd(rec["q_raw"], p + "self_attn.q_norm.weight", d_q, nh)
                d_k = norm_bwd(rec["k_raw"], p + "self_attn.k_norm.weight", d_k, nkv)
            h1 = ops.rmsnorm(rec["x_in"], f32v(p + "input_layernorm.weight"), s.rms_norm_eps, out_dtype=dt)
            pad8 = lambda t: t if (t.stride(0) % 8 == 0 and t.stride(1) == 1) else torch.nn.functional.pad(t, (0, (-t.shape[1]) % 8)).contiguous()
            d_h1 = lin["self_attn.q_proj"].backward(pad8(d_q), h1, rec["u_self_attn.q_proj"], None, True, False, grads)
            lin["self_attn.k_proj"].backward(pad8(d_k), h1, rec["u_self_attn.k_proj"], d_h1, True, True, grads)
            lin["self_attn.v_proj"].backward(pad8(d_v), h1, rec["u_self_attn.v_proj"], d_h1, True, True, grads)
            rms_bwd(rec["x_in"], f32v(p + "input_layernorm.weight"), d_h1, g, 1)
            st["tape"][i] = None                        # free the layer's activations
        gl = g_loss.float().reshape(1).contiguous()
        call("p2t_scale_by_device_scalar", ptr(g), g.numel(), ptr(gl), stream())
        out_params = []
        for prm in st["params"]:
            gp = grads.get(id(prm))
            if gp is None:
                out_params.append(None)
                continue
            buf, rows, cols, factor = gp            # dB carries alpha / r on top of the upstream gradient
            call("p2t_scale_by_device_scalar", ptr(buf), buf.numel(), ptr(gl if factor == 1.0 else (gl * factor).contiguous()), stream())
            out_params.append(buf[:rows, :cols].to(prm.dtype))
        ctx.state = None
        return (g.view(B, T, H).to(st["in_dtype"]), None, None, None, None, *out_params)


def lora_lm_loss(decoder, lora: Optional[DecoderLora], inputs_embeds: torch.Tensor, attention_mask: torch.Tensor, labels: torch.Tensor):
    """(loss, logits) of `llama_decoder(inputs_embeds=..., attention_mask=..., labels=...)` with the LoRA branches in the graph."""
    params = tuple(lora.parameters()) if lora is not None else ()
    if lora is not None and lora.training:
        lora.step_count += 1                            # a fresh dropout mask per step
    loss, logits = DecoderLoraLossFn.apply(inputs_embeds, decoder, lora, attention_mask, labels, *params)
    return loss, logits[..., : decoder.spec.vocab_size]
