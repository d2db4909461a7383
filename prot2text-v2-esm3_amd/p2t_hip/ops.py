"""Tensor-level wrappers over the C ABI (torch is only the allocator and the stream provider).

Each function checks shapes/dtypes on the host (ValueError, mirroring the reference's own argument
checks, e.g. models/esmc_qwen_arc.py:137-141) before a kernel sees a pointer, then enqueues on the
current HIP stream.  Nothing here computes with torch.
"""
from __future__ import annotations

import ctypes as C
import math

import torch

from . import _lib, synth
from ._lib import BF16, F32, call

_DT = {torch.float32: F32, torch.bfloat16: BF16}


def dt_of(t: torch.Tensor | torch.dtype) -> int:
    d = t if isinstance(t, torch.dtype) else t.dtype
    if d not in _DT:
        raise ValueError(f"unsupported dtype {d}: the HIP path stores float32 or bfloat16")
    return _DT[d]


def ptr(t: torch.Tensor | None):
    if t is None:
        return None
    if not t.is_cuda:
        raise ValueError("tensor must live on the GPU (there is no CPU path)")
    return C.c_void_p(t.data_ptr())


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


def _chk(cond, msg):
    if not cond:
        raise ValueError(msg)


# ---------------------------------------------------------------------------------------------
def fill_hash_(t: torch.Tensor, seed: int, name: str, scale: float, offset: float = 0.0) -> torch.Tensor:
    """In-place synthetic fill, bit-identical to synth.uniform_f32(seed, name, t.shape, scale, offset)."""
    _chk(t.is_contiguous(), "fill_hash_: tensor must be contiguous")
    if scale == 0.0:
        t.fill_(offset)
        return t
    add, xor = synth.stream_key(seed, name)
    call("p2t_fill_hash", ptr(t), t.numel(), add, xor, float(synth.scale_f32(scale)), float(offset), dt_of(t), stream())
    return t


def cast(src: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    out = torch.empty_like(src, dtype=dtype)
    call("p2t_cast", ptr(src.contiguous()), dt_of(src), ptr(out), dt_of(out), src.numel(), stream())
    return out


def transpose(src: torch.Tensor, ld_dst: int | None = None) -> torch.Tensor:
    _chk(src.dim() == 2 and src.stride(1) == 1, "transpose: 2-D row-major input")
    rows, cols = src.shape
    ld_dst = round_up(rows, 64) if ld_dst is None else ld_dst
    out = torch.empty((cols, ld_dst), dtype=src.dtype, device=src.device)
    call("p2t_transpose", ptr(src), rows, cols, src.stride(0), ptr(out), ld_dst, dt_of(src), stream())
    return out


def gemm_nt(a: torch.Tensor, w: torch.Tensor, bias: torch.Tensor | None = None, *, n: int | None = None,
            k: int | None = None, epilogue: int = _lib.EPI_STORE, out: torch.Tensor | None = None,
            out_dtype: torch.dtype | None = None, z: torch.Tensor | None = None, accumulate: bool = False,
            use_mfma: int = -1, fix_ws: torch.Tensor | None = None, fix_epoch: int = 0) -> torch.Tensor:
    """out = epilogue(a[:, :k] @ w[:n, :k].T).  a: [M, lda], w: [>=n, ldw] (row strides = storage width)."""
    _chk(a.dim() == 2 and w.dim() == 2 and a.stride(1) == 1 and w.stride(1) == 1, "gemm_nt: 2-D row-major operands")
    _chk(a.dtype == w.dtype, "gemm_nt: operand dtypes differ")
    M = a.shape[0]
    n = w.shape[0] if n is None else n
    k = min(a.shape[1], w.shape[1]) if k is None else k
    n_out = n // 2 if epilogue == _lib.EPI_SWIGLU else n
    if out is None:
        od = torch.float32 if epilogue in (_lib.EPI_RESID, _lib.EPI_STORE_F32) else (out_dtype or a.dtype)
        ldc = n_out if epilogue in (_lib.EPI_RESID, _lib.EPI_STORE_F32) else round_up(n_out, 64)
        out = torch.empty((M, ldc), dtype=od, device=a.device)
    call("p2t_gemm_nt", ptr(a), a.stride(0), ptr(w), w.stride(0), ptr(bias), ptr(out), out.stride(0), ptr(z), M, n, k,
         dt_of(a), dt_of(out), epilogue, int(accumulate), use_mfma, ptr(fix_ws), fix_ws.numel() if fix_ws is not None else 0,
         int(fix_epoch), stream())
    return out


def gemm_qkv_rope(a: torch.Tensor, w: torch.Tensor, bias: torch.Tensor | None, inv_freq: torch.Tensor, seq: int, nh: int,
                  nkv: int, head_dim: int, q_scale: float, *, k: int | None = None, use_mfma: int = -1,
                  fix_ws: torch.Tensor | None = None, fix_epoch: int = 0):
    """The towers' fused QKV projection (bias + query scale + rotary + head split) on its own:
    a [B * seq, lda] @ w [(nh + 2 nkv) * head_dim, ldw]^T -> q [B, nh, seq, d], k, v [B, nkv, seq, d]."""
    _chk(a.dim() == 2 and w.dim() == 2 and a.stride(1) == 1 and w.stride(1) == 1 and a.dtype == w.dtype, "gemm_qkv_rope: operands")
    _chk(head_dim in (64, 128) and w.shape[0] == (nh + 2 * nkv) * head_dim and a.shape[0] % seq == 0, "gemm_qkv_rope: shapes")
    M, B = a.shape[0], a.shape[0] // seq
    k = min(a.shape[1], w.shape[1]) if k is None else k
    e = lambda h: torch.empty((B, h, seq, head_dim), dtype=a.dtype, device=a.device)
    q, kk, v = e(nh), e(nkv), e(nkv)
    cs = torch.empty((seq, head_dim), dtype=torch.float32, device=a.device)
    call("p2t_gemm_qkv_rope", ptr(a), a.stride(0), ptr(w), w.stride(0), ptr(bias), M, k, dt_of(a), ptr(inv_freq.float().contiguous()),
         ptr(cs), ptr(q), ptr(kk), ptr(v), seq, nh, nkv, head_dim, float(q_scale), use_mfma, ptr(fix_ws),
         fix_ws.numel() if fix_ws is not None else 0, int(fix_epoch), stream())
    return q, kk, v


def quant_rows_fp8(x: torch.Tensor, cols: int | None = None, ld_q: int | None = None):
    """Row-wise e4m3 quantisation with one E8M0 (power-of-two) scale byte per row -> (q uint8 [rows, ld_q], scale uint8 [rows])."""
    _chk(x.dim() == 2 and x.stride(1) == 1, "quant_rows_fp8: 2-D row-major input")
    rows = x.shape[0]
    cols = x.shape[1] if cols is None else cols
    ld_q = round_up(cols, 128) if ld_q is None else ld_q
    q = torch.empty((rows, ld_q), dtype=torch.uint8, device=x.device)
    sc = torch.empty((rows,), dtype=torch.uint8, device=x.device)
    call("p2t_quant_rows_fp8", ptr(x), dt_of(x), x.stride(0), rows, cols, ptr(q), ld_q, ptr(sc), stream())
    return q, sc


def norm_fp8(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor | None, eps: float, ld_q: int | None = None,
             bound: tuple[float, float] | None = None):
    """LayerNorm (b given) or RMSNorm (b None) of the f32 stream, written as e4m3 + E8M0 row scales.  bound = (w_norm_max,
    b_abs_max) of the projection that follows (LayerNorm only): also returns the per-row bound scale bytes for its GELU output."""
    _chk(x.dtype == torch.float32 and x.dim() == 2, "norm_fp8: x must be f32 [rows, cols]")
    rows, cols = x.shape
    ld_q = round_up(cols, 128) if ld_q is None else ld_q
    q = torch.empty((rows, ld_q), dtype=torch.uint8, device=x.device)
    sc = torch.empty((rows,), dtype=torch.uint8, device=x.device)
    if b is not None:
        bs = torch.empty((rows,), dtype=torch.uint8, device=x.device) if bound is not None else None
        call("p2t_layernorm_fp8", ptr(x), x.stride(0), ptr(w), ptr(b), float(eps), ptr(q), ld_q, ptr(sc), rows, cols,
             float(bound[0]) if bound else 0.0, float(bound[1]) if bound else 0.0, ptr(bs), stream())
        return (q, sc, bs) if bound is not None else (q, sc)
    _chk(bound is None, "norm_fp8: the bound scale is a LayerNorm output")
    call("p2t_rmsnorm_fp8", ptr(x), x.stride(0), ptr(w), float(eps), ptr(q), ld_q, ptr(sc), rows, cols, stream())
    return q, sc


def gemm_nt_fp8(a8: torch.Tensor, a_scale: torch.Tensor, w8: torch.Tensor, w_scale: torch.Tensor, bias: torch.Tensor | None = None, *,
                n: int | None = None, k: int | None = None, epilogue: int = _lib.EPI_STORE, out: torch.Tensor | None = None,
                out_dtype: torch.dtype = torch.bfloat16, z: torch.Tensor | None = None, tile: int = 0,
                out_row_scale: torch.Tensor | None = None) -> torch.Tensor:
    """epilogue((a8 * 2^(a_scale-127)) @ (w8 * 2^(w_scale-127)).T) on the fp8 MFMA kernel; a8 [M, lda], w8 [>= n, ldw] uint8."""
    _chk(a8.dtype == torch.uint8 and w8.dtype == torch.uint8 and a8.dim() == 2 and w8.dim() == 2, "gemm_nt_fp8: uint8 (e4m3) operands")
    _chk(a_scale.dtype == torch.uint8 and w_scale.dtype == torch.uint8, "gemm_nt_fp8: uint8 (E8M0) row scales")
    M = a8.shape[0]
    n = w8.shape[0] if n is None else n
    k = min(a8.shape[1], w8.shape[1]) if k is None else k
    _chk(a_scale.numel() >= M and w_scale.numel() >= n, "gemm_nt_fp8: one scale per row")
    n_out = n // 2 if epilogue == _lib.EPI_SWIGLU else n
    if epilogue == _lib.EPI_GELU_FP8:
        _chk(out_row_scale is not None and out_row_scale.dtype == torch.uint8 and out_row_scale.numel() >= M, "EPI_GELU_FP8: one E8M0 byte per output row")
        if out is None:
            out = torch.empty((M, round_up(n_out, 128)), dtype=torch.uint8, device=a8.device)
        call("p2t_gemm_nt_fp8", ptr(a8), a8.stride(0), ptr(a_scale), ptr(w8), w8.stride(0), ptr(w_scale), ptr(bias), ptr(out), out.stride(0),
             None, M, n, k, _lib.BF16, epilogue, 0, int(tile), ptr(out_row_scale), stream())
        return out
    if out is None:
        od = torch.float32 if epilogue in (_lib.EPI_RESID, _lib.EPI_STORE_F32) else out_dtype
        ldc = n_out if epilogue in (_lib.EPI_RESID, _lib.EPI_STORE_F32) else round_up(n_out, 64)
        out = torch.empty((M, ldc), dtype=od, device=a8.device)
    call("p2t_gemm_nt_fp8", ptr(a8), a8.stride(0), ptr(a_scale), ptr(w8), w8.stride(0), ptr(w_scale), ptr(bias), ptr(out), out.stride(0),
         ptr(z), M, n, k, dt_of(out), epilogue, 0, int(tile), None, stream())
    return out


def positions_where(values: torch.Tensor, match: int | None = None):
    """Flat positions (int32, array order) of `values == match` (or `values != 0` when match is None) and their count
    (int32 [1]), both on the device: the enumeration order of torch's boolean-mask indexing."""
    v = values.to(torch.int64).contiguous().view(-1)
    pos = torch.empty((v.numel(),), dtype=torch.int32, device=v.device)
    count = torch.empty((1,), dtype=torch.int32, device=v.device)
    call("p2t_positions_where", ptr(v), v.numel(), 0 if match is not None else 1, int(match or 0), ptr(pos), ptr(count), stream())
    return pos, count


def scatter_rows(dst: torch.Tensor, dst_pos, n_dst, src: torch.Tensor, src_pos, n_src, H: int) -> None:
    """dst[dst_pos[r], :H] = src[src_pos[r], :H] for r < min(n_dst, n_src); dst f32 2-D, src bf16 / f32 2-D."""
    _chk(dst.dim() == 2 and src.dim() == 2 and dst.dtype == torch.float32 and dst.stride(1) == 1 and src.stride(1) == 1,
         "scatter_rows: 2-D row-major operands, f32 destination")
    call("p2t_scatter_rows", ptr(dst), dst.stride(0), ptr(dst_pos), ptr(src), src.stride(0), dt_of(src), ptr(src_pos), ptr(n_dst),
         ptr(n_src), min(dst_pos.numel(), src_pos.numel()), int(H), stream())


def cross_entropy_shifted(logits: torch.Tensor, labels: torch.Tensor, V: int, ignore_index: int = -100):
    """HF causal-LM loss: logits [B, T, ld >= V], labels [B, T] -> (loss f32 [1], n_targets int32 [1])."""
    _chk(logits.dim() == 3 and labels.dim() == 2 and tuple(logits.shape[:2]) == tuple(labels.shape) and logits.stride(2) == 1
         and logits.is_contiguous(), "cross_entropy_shifted: logits [B, T, ld] contiguous, labels [B, T]")
    B, T, ld = logits.shape
    lab = labels.to(torch.int64).contiguous()
    row_loss = torch.empty((B * T,), dtype=torch.float32, device=logits.device)
    row_valid = torch.empty((B * T,), dtype=torch.int32, device=logits.device)
    loss = torch.empty((1,), dtype=torch.float32, device=logits.device)
    count = torch.empty((1,), dtype=torch.int32, device=logits.device)
    call("p2t_cross_entropy_shifted", ptr(logits), ld, dt_of(logits), ptr(lab), B, T, int(V), int(ignore_index), ptr(row_loss),
         ptr(row_valid), ptr(loss), ptr(count), stream())
    return loss, count


def gemm_fix_workspace(device) -> torch.Tensor:
    """Zeroed split-K fix-up workspace for gemm_nt(fix_ws=..., fix_epoch=1, 2, ...)."""
    return torch.zeros((call("p2t_gemm_fix_workspace_bytes"),), dtype=torch.uint8, device=device)


def layernorm(x, w, b, eps, out_dtype=torch.float32, ld_out=None):
    _chk(x.dtype == torch.float32 and x.dim() == 2, "layernorm: x must be f32 [rows, cols]")
    rows, cols = x.shape
    ld = round_up(cols, 64) if ld_out is None else ld_out
    y = torch.empty((rows, ld), dtype=out_dtype, device=x.device)
    call("p2t_layernorm", ptr(x), x.stride(0), ptr(w), ptr(b), float(eps), ptr(y), ld, rows, cols, dt_of(y), stream())
    return y


def rmsnorm(x, w, eps, out_dtype=torch.float32, ld_out=None):
    _chk(x.dtype == torch.float32 and x.dim() == 2, "rmsnorm: x must be f32 [rows, cols]")
    rows, cols = x.shape
    ld = round_up(cols, 64) if ld_out is None else ld_out
    y = torch.empty((rows, ld), dtype=out_dtype, device=x.device)
    call("p2t_rmsnorm", ptr(x), x.stride(0), ptr(w), float(eps), ptr(y), ld, rows, cols, dt_of(y), stream())
    return y


def head_dim_padded(d: int) -> int:
    return 32 if d <= 32 else (64 if d <= 64 else 128)


def mask_prepare(mask: torch.Tensor, ids: torch.Tensor | None = None, mask_id: int = -1, token_dropout: bool = False):
    B, T = mask.shape
    key_mask = torch.empty((B, T), dtype=torch.uint8, device=mask.device)
    kv_info = torch.empty((2 * B,), dtype=torch.int32, device=mask.device)
    emb_scale = torch.empty((2 * B,), dtype=torch.float32, device=mask.device) if ids is not None else None
    call("p2t_mask_prepare", ptr(ids), ptr(mask.contiguous()), B, T, mask_id, int(token_dropout), ptr(key_mask),
         ptr(kv_info), ptr(emb_scale), stream())
    return key_mask, kv_info, emb_scale


def qkv_post(qkv: torch.Tensor, inv_freq: torch.Tensor, B: int, T: int, nh: int, nkv: int, d: int, q_scale: float):
    dp = head_dim_padded(d)
    dev, dty = qkv.device, qkv.dtype
    q = torch.empty((B, nh, T, dp), dtype=dty, device=dev)
    k = torch.empty((B, nkv, T, dp), dtype=dty, device=dev)
    v = torch.empty((B, nkv, T, dp), dtype=dty, device=dev)
    cs = torch.empty((T, d), dtype=torch.float32, device=dev)
    call("p2t_qkv_post", ptr(qkv), qkv.stride(0), ptr(inv_freq), ptr(cs), ptr(q), ptr(k), ptr(v), B, T, nh, nkv, d, dp,
         float(q_scale), dt_of(qkv), stream())
    return q, k, v


def attention(q, k, v, key_mask, kv_info, d: int, scale: float, causal: bool, use_mfma: int = -1, log2_scores: bool = False,
              lse: torch.Tensor | None = None):
    """softmax(scale q k^T + mask) v; log2_scores: q already carries scale * log2(e) (include/p2t_hip.h, p2t_attention).
    lse (optional f32 [B, nh, T]): filled with the rows' log-sum-exps for p2t_attention_backward."""
    B, nh, T, dp = q.shape
    nkv = k.shape[1]
    ld = round_up(nh * d, 64)
    out = torch.empty((B * T, ld), dtype=q.dtype, device=q.device)
    call("p2t_attention", ptr(q), ptr(k), ptr(v), ptr(key_mask), ptr(kv_info), ptr(out), ld, B, T, nh, nkv, d, dp,
         float(scale), int(causal), dt_of(q), use_mfma, int(bool(log2_scores)), ptr(lse), stream())
    return out


def attention_backward(q, k, v, o, d_o, lse, key_mask, kv_info, d: int, scale: float, causal: bool, log2_scores: bool = False, use_mfma: int = -1):
    """(dq, dk, dv) f32 in the layouts of q, k, v (include/p2t_hip.h, p2t_attention_backward); o / d_o: [B*T, ld] as `attention` returns."""
    B, nh, T, dp = q.shape
    nkv = k.shape[1]
    f = lambda h: torch.empty((B, h, T, dp), dtype=torch.float32, device=q.device)
    dq, dk, dv = f(nh), f(nkv), f(nkv)
    D = torch.empty((B, nh, T), dtype=torch.float32, device=q.device)
    call("p2t_attention_backward", ptr(q), ptr(k), ptr(v), ptr(o), o.stride(0), ptr(d_o), d_o.stride(0), ptr(lse), ptr(key_mask), ptr(kv_info),
         ptr(dq), ptr(dk), ptr(dv), ptr(D), B, T, nh, nkv, d, dp, float(scale), int(causal), dt_of(q), int(bool(log2_scores)), int(use_mfma), stream())
    return dq, dk, dv


# ---------------------------------------------------------------------------------------------
def readout(emb: torch.Tensor, mask: torch.Tensor | None, mode: str, D: int | None = None) -> torch.Tensor:
    """readout_embeddings (scripts/train_contrast.py:198-248): emb [B, T, >=D] -> f32 [B, D] or [B, 2D]."""
    _chk(mode in _lib.READOUT, f"readout_fn must be one of {list(_lib.READOUT)}, got {mode!r}")
    _chk(emb.dim() == 3 and emb.stride(2) == 1 and emb.stride(0) == emb.shape[1] * emb.stride(1), "readout: [B, T, ld] layout")
    B, T = emb.shape[:2]
    D = emb.shape[2] if D is None else D
    if mask is not None:
        _chk(tuple(mask.shape) == (B, T), "readout: attention_mask must be [B, T]")
        mask = mask.to(torch.int64).contiguous()
    out = torch.empty((B, 2 * D if mode == "mix" else D), dtype=torch.float32, device=emb.device)
    call("p2t_readout", ptr(emb), dt_of(emb), emb.stride(1), ptr(mask), B, T, D, _lib.READOUT[mode], ptr(out), stream())
    return out


def readout_backward(emb, mask, mode: str, pooled_mix, d_out, D: int | None = None) -> torch.Tensor:
    B, T = emb.shape[:2]
    D = emb.shape[2] if D is None else D
    if mask is not None:
        mask = mask.to(torch.int64).contiguous()
    d_emb = torch.empty((B, T, D), dtype=torch.float32, device=emb.device)
    call("p2t_readout_backward", ptr(emb), dt_of(emb), emb.stride(1), ptr(mask), B, T, D, _lib.READOUT[mode],
         ptr(pooled_mix), ptr(d_out.contiguous()), ptr(d_emb), stream())
    return d_emb


def l2norm_rows(x: torch.Tensor, eps: float = 1e-12):
    _chk(x.dtype == torch.float32 and x.dim() == 2 and x.is_contiguous(), "l2norm_rows: contiguous f32 [rows, cols]")
    y = torch.empty_like(x)
    call("p2t_l2norm_rows", ptr(x), ptr(y), None, x.shape[0], x.shape[1], float(eps), stream())
    return y


def l2norm_rows_backward(x: torch.Tensor, dy: torch.Tensor, eps: float = 1e-12):
    dx = torch.empty_like(x)
    call("p2t_l2norm_rows_backward", ptr(x), ptr(dy.contiguous()), ptr(dx), x.shape[0], x.shape[1], float(eps), stream())
    return dx


def infonce_forward(seg, batch, labels, temperature=0.05, weight=1.0, loss_out=None, accumulate=False):
    """Returns (loss[1] f32, logits [S, N]).  loss (+)= weight * mean_i CE(seg_i batch^T / tau, labels_i)."""
    _chk(seg.dtype == torch.float32 and batch.dtype == torch.float32, "infonce: f32 embeddings")
    _chk(seg.dim() == 2 and batch.dim() == 2 and seg.shape[1] == batch.shape[1], "infonce: [S, D] x [N, D]")
    S, D = seg.shape
    N = batch.shape[0]
    _chk(labels.numel() == S, "infonce: one label per segment row")
    labels = labels.to(torch.int32).contiguous()
    if loss_out is None:
        loss_out = torch.zeros((1,), dtype=torch.float32, device=seg.device)
        accumulate = False
    logits = torch.empty((S, N), dtype=torch.float32, device=seg.device)
    row_loss = torch.empty((S,), dtype=torch.float32, device=seg.device)
    call("p2t_infonce_forward", ptr(seg.contiguous()), ptr(batch.contiguous()), ptr(labels), S, N, D, float(temperature),
         float(weight), int(accumulate), ptr(loss_out), ptr(logits), ptr(row_loss), stream())
    return loss_out, logits


def infonce_backward(batch, labels, logits, temperature=0.05, weight=1.0):
    S, N = logits.shape
    D = batch.shape[1]
    labels = labels.to(torch.int32).contiguous()
    d_seg = torch.empty((S, D), dtype=torch.float32, device=batch.device)
    call("p2t_infonce_backward", ptr(batch.contiguous()), ptr(labels), ptr(logits), S, N, D, float(temperature), float(weight),
         ptr(d_seg), stream())
    return d_seg


def infonce_col_forward(p_all, t_all, temperature=0.05, first=0, count=None, weight=1.0, loss_out=None, accumulate=False, cols=None):
    """Column (text -> protein) term on the global [N, D] embeddings: returns (loss[1], col_lse [N]);
    loss (+)= weight * mean over the columns j in `cols` (device int tensor) or first <= j < first + count of
    (logsumexp_i l_ij - l_jj)."""
    _chk(p_all.dtype == torch.float32 and t_all.dtype == torch.float32 and p_all.dim() == 2 and p_all.shape == t_all.shape,
         "infonce_col: f32 [N, D] protein and text embeddings of the same (global) batch")
    N, D = p_all.shape
    if cols is not None:
        cols = cols.to(torch.int32).contiguous()
        count = cols.numel()
    count = N - first if count is None else count
    if loss_out is None:
        loss_out = torch.zeros((1,), dtype=torch.float32, device=p_all.device)
        accumulate = False
    col_lse = torch.empty((N,), dtype=torch.float32, device=p_all.device)
    scratch = torch.empty((N * N + N,), dtype=torch.float32, device=p_all.device)
    call("p2t_infonce_col_forward", ptr(p_all.contiguous()), ptr(t_all.contiguous()), N, D, float(temperature), ptr(cols), int(first), int(count),
         float(weight), int(accumulate), ptr(loss_out), ptr(col_lse), ptr(scratch), ptr(scratch[N * N:]), stream())
    return loss_out, col_lse


def infonce_col_backward(t_all, labels, logits, col_lse, temperature=0.05, scale=1.0, d_seg=None):
    """d_seg (+)= scale / tau * sum_j (exp(l_ij - col_lse_j) - [j == label_i]) t_j; adds to `d_seg` when given."""
    S, N = logits.shape
    D = t_all.shape[1]
    labels = labels.to(torch.int32).contiguous()
    acc = d_seg is not None
    if d_seg is None:
        d_seg = torch.empty((S, D), dtype=torch.float32, device=t_all.device)
    call("p2t_infonce_col_backward", ptr(t_all.contiguous()), ptr(labels), ptr(logits), ptr(col_lse), S, N, D, float(temperature),
         float(scale), int(acc), ptr(d_seg), stream())
    return d_seg


def clip_adamw_step(params, grads, exp_avg, exp_avg_sq, step: int, *, lr=2e-4, betas=(0.9, 0.999), eps=1e-6,
                    weight_decay=0.01, max_norm=math.inf, shadows=None, scratch=None, grad_norm_out=None):
    """clip_grad_norm_ + AdamW.step on f32 tensors (train_contrast.py:453-465).  shadows[i]: optional 2-D
    tensor (bf16/f32, row stride >= cols) refreshed with the new value of the 2-D params[i]."""
    n = len(params)
    dev = params[0].device
    arr = lambda ts: (C.c_void_p * n)(*[t.data_ptr() if t is not None else None for t in ts])
    numel = (C.c_int64 * n)(*[p.numel() for p in params])
    shadows = shadows or [None] * n
    cols = (C.c_int64 * n)(*[(p.shape[-1] if p.dim() == 2 else p.numel()) for p in params])
    lds = (C.c_int64 * n)(*[(s.stride(0) if s is not None else 1) for s in shadows])
    sdt = next((dt_of(s) for s in shadows if s is not None), F32)
    if scratch is None:
        scratch = torch.empty((256 * n,), dtype=torch.float32, device=dev)
    if grad_norm_out is None:
        grad_norm_out = torch.empty((1,), dtype=torch.float32, device=dev)
    mn = 0.0 if (max_norm is None or math.isinf(max_norm)) else float(max_norm)
    call("p2t_clip_adamw_step", n, arr(params), arr(grads), arr(exp_avg), arr(exp_avg_sq), numel, arr(shadows), cols, lds,
         sdt, int(step), float(lr), float(betas[0]), float(betas[1]), float(eps), float(weight_decay), mn,
         ptr(grad_norm_out), ptr(scratch), stream())
    return grad_norm_out
