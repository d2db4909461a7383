"""GPU parity, kernel level: every HIP kernel through the C ABI against the numpy oracle on the same
seeded inputs.  fp32 kernels: fp32 round-off tolerances.  bf16 MFMA kernels: compared with the oracle
evaluated on the SAME bf16-rounded operands (fp32 accumulate), so the tolerance only covers the
accumulation order and the final bf16 rounding of the output."""
import math

import numpy as np
import pytest
import torch

from oracle import p2t_oracle as O
from gpu_util import bf16r, dev, maxabs, observe, rel, rnd, to_dev, to_np

pytestmark = pytest.mark.gpu

EPI_STORE, EPI_GELU, EPI_RESID, EPI_SWIGLU, EPI_STORE_F32, EPI_GELU_BWD = range(6)


@pytest.fixture(scope="module")
def ops():
    from p2t_hip import ops as _ops
    return _ops


@pytest.fixture
def gemm_policy():
    """p2t_set_gemm_policy for one test (launch-form override of the MFMA GEMM), back to the default afterwards."""
    from p2t_hip import _lib
    yield lambda p: _lib.call("p2t_set_gemm_policy", int(p))
    _lib.call("p2t_set_gemm_policy", 0)


def _no_timeout():
    """The GPU's sticky fault word (include/p2t_hip.h, p2t_fault_status): 0 = no split-K consumer ever gave up waiting."""
    from p2t_hip import _lib
    return _lib.fault_status() == 0


# ---------------------------------------------------------------------------------------------
def test_fill_hash_bit_exact(ops):
    from p2t_hip import synth
    for n, scale, offset in ((1000003, 0.37, 0.0), (4096, 0.1, 1.0), (7, 2.0, -0.5)):
        ref = synth.uniform_f32(11, "t.fill", (n,), scale, offset)
        t = torch.empty((n,), dtype=torch.float32, device=dev())
        ops.fill_hash_(t, 11, "t.fill", scale, offset)
        assert np.array_equal(to_np(t), ref)
        tb = torch.empty((n,), dtype=torch.bfloat16, device=dev())
        ops.fill_hash_(tb, 11, "t.fill", scale, offset)
        assert np.array_equal(to_np(tb), synth.bf16_round(ref))


def test_cast_transpose(ops):
    a = rnd(1, "ct.a", (130, 70), 3.0)
    t = to_dev(a)
    assert np.array_equal(to_np(ops.cast(t, torch.bfloat16)), bf16r(a))
    tt = ops.transpose(t)                       # [70, 192], zero padded
    got = to_np(tt)
    assert np.array_equal(got[:, :130], a.T) and not got[:, 130:].any()
    tb = ops.transpose(to_dev(a, torch.bfloat16))
    assert np.array_equal(to_np(tb)[:, :130], bf16r(a).T)


def _gemm_ref(a, w, bias, epi, resid=None, z=None):
    acc = a.astype(np.float32) @ w.astype(np.float32).T
    if epi == EPI_SWIGLU:
        F = w.shape[0] // 2
        v = acc.reshape(acc.shape[0], F // 32, 2, 32)
        g, u = v[:, :, 0, :].reshape(-1, F), v[:, :, 1, :].reshape(-1, F)
        return (g / (1 + np.exp(-g))) * u
    if bias is not None:
        acc = acc + bias
    if epi == EPI_GELU:
        return O.gelu_erf(acc)
    if epi == EPI_RESID:
        return resid + acc
    if epi == EPI_GELU_BWD:
        return acc * O.gelu_erf_grad(z)
    return acc


@pytest.mark.parametrize("epi", [EPI_STORE, EPI_GELU, EPI_RESID, EPI_SWIGLU, EPI_STORE_F32, EPI_GELU_BWD])
@pytest.mark.parametrize("shape", [(70, 128, 52), (257, 320, 128)])
def test_gemm_fp32(ops, epi, shape):
    M, N, K = shape
    a, w = rnd(2, "g.a", (M, K), 1.0), rnd(2, "g.w", (N, K), 0.5)
    bias = None if epi in (EPI_SWIGLU, EPI_STORE_F32, EPI_GELU_BWD) else rnd(2, "g.b", (N,), 0.3)
    n_out = N // 2 if epi == EPI_SWIGLU else N
    ld = ops.round_up(n_out, 64)
    resid = rnd(2, "g.r", (M, N), 1.0)
    zin = rnd(2, "g.z", (M, ld), 1.5)
    out = z = None
    if epi == EPI_RESID:
        out = to_dev(resid)
    if epi == EPI_GELU_BWD:
        z = to_dev(zin)
    if epi == EPI_GELU:
        z = torch.full((M, ld), 7.0, dtype=torch.float32, device=dev())
    got = ops.gemm_nt(to_dev(a), to_dev(w), to_dev(bias) if bias is not None else None, epilogue=epi, out=out, z=z)
    ref = _gemm_ref(a, w, bias, epi, resid, zin[:, :N])
    g = to_np(got)
    assert rel(g[:, :n_out], ref) < 2e-6
    if g.shape[1] > n_out:
        assert not g[:, n_out:].any()                       # zero K-padding for the next GEMM
    if epi == EPI_GELU:
        assert rel(to_np(z)[:, :N], a @ w.T + bias) < 2e-6


@pytest.mark.parametrize("epi", [EPI_STORE, EPI_GELU, EPI_RESID, EPI_SWIGLU, EPI_STORE_F32, EPI_GELU_BWD])
# the last shapes exercise the persistent-kernel policy (>= 256 tiles, K % 128 == 0): ragged edges in M and N (falls back
# to the per-tile kernel) / whole rounds / 4.25 rounds (blocks 0..63 walk five tiles, the rest four)
@pytest.mark.parametrize("shape", [(300, 320, 320), (1000, 96, 64), (4096, 4096, 128), (2048, 1184, 192), (16500, 2560, 128),
                                   (16500, 2624, 512), (8192, 4096, 384), (16384, 4352, 384)])
def test_gemm_mfma_bf16(ops, epi, shape):
    M, N, K = shape
    if epi == EPI_SWIGLU and N % 64:
        pytest.skip("swiglu needs N % 64 == 0")
    a, w = bf16r(rnd(3, "m.a", (M, K), 1.0)), bf16r(rnd(3, "m.w", (N, K), 0.5))
    bias = None if epi in (EPI_SWIGLU, EPI_STORE_F32, EPI_GELU_BWD) else rnd(3, "m.b", (N,), 0.3)
    n_out = N // 2 if epi == EPI_SWIGLU else N
    ld = ops.round_up(n_out, 64)
    resid = rnd(3, "m.r", (M, N), 1.0)
    zin = bf16r(rnd(3, "m.z", (M, ld), 1.5))
    out = z = None
    if epi == EPI_RESID:
        out = to_dev(resid)
    if epi == EPI_GELU_BWD:
        z = to_dev(zin, torch.bfloat16)
    if epi == EPI_GELU:
        z = torch.full((M, ld), 7.0, dtype=torch.bfloat16, device=dev())
    got = ops.gemm_nt(to_dev(a, torch.bfloat16), to_dev(w, torch.bfloat16), to_dev(bias) if bias is not None else None,
                      epilogue=epi, out=out, z=z, use_mfma=1)
    ref = _gemm_ref(a, w, bias, epi, resid, zin[:, :N])
    g = to_np(got)
    exact_out = epi in (EPI_RESID, EPI_STORE_F32)
    observe(f"gemm_mfma_bf16[epi{epi},{M}x{N}x{K}]", rel(g[:, :n_out], ref), 3e-6 if exact_out else 2.5e-3)   # bf16 output rounding: 2^-9 / sqrt(3)
    assert maxabs(g[:, :n_out], ref) < (1e-4 if exact_out else 0.02 * max(1.0, float(np.abs(ref).max())))
    if g.shape[1] > n_out:
        assert not g[:, n_out:].any()


def test_gemm_mfma_matches_fma_kernel(ops):
    """Same bf16 inputs through both kernels (the fp32-FMA kernel is the exact one)."""
    M, N, K = 1500, 640, 448
    a, w = to_dev(rnd(4, "x.a", (M, K)), torch.bfloat16), to_dev(rnd(4, "x.w", (N, K), 0.2), torch.bfloat16)
    b = to_dev(rnd(4, "x.b", (N,), 0.1))
    r0 = to_np(ops.gemm_nt(a, w, b, epilogue=EPI_STORE, out_dtype=torch.float32, use_mfma=0))
    r1 = to_np(ops.gemm_nt(a, w, b, epilogue=EPI_STORE, out_dtype=torch.float32, use_mfma=1))
    assert rel(r1, r0) < 2e-6


def _pack_d128(w, heads):
    """[heads * 128, K] -> the per-head row order 0..31, 64..95, 32..63, 96..127 of include/p2t_hip.h (p2t_llama_layer)."""
    K = w.shape[1]
    return np.ascontiguousarray(w.reshape(heads, 2, 2, 32, K).transpose(0, 2, 1, 3, 4).reshape(heads * 128, K))


QKV_ROPE_CASES = [  # name, B, T, K, nh, nkv, d, bias, q_scale, rope
    ("esm_d64_bias_qscale", 2, 100, 192, 3, 3, 64, True, 64 ** -0.5, "default"),          # per-tile kernel, edge tiles in M and N
    ("esm_d64_edge_n", 4, 512, 640, 10, 10, 64, True, 64 ** -0.5, "default"),             # 8 x 7.5 tiles: edge tiles in N only
    ("llama_d128_packed_gqa", 3, 70, 256, 4, 2, 128, False, 1.0, "llama3"),
    ("llama_d64_gqa", 2, 130, 128, 8, 2, 64, False, 1.0, "llama3"),
    ("esm3b_qkv_persistent", 16, 1024, 2560, 40, 40, 64, True, 64 ** -0.5, "default"),    # M=16384, N=7680: persistent policy, 7.5 rounds
]


@pytest.mark.parametrize("case", QKV_ROPE_CASES, ids=lambda c: c[0])
@pytest.mark.parametrize("path", ["mfma", "fma"])
def test_qkv_rope_epilogue_vs_reference_arithmetic(ops, case, path):
    """The fused QKV epilogue the benchmark times (EPI_QKV_ROPE) on its own: bias, query scale BEFORE the rotation,
    rotate-half rotary, head split -- against HF's arithmetic restated by the oracle (O.rope_cos_sin / O.rotate_half:
    modeling_esm.py:48-79,345,362-378; modeling_llama.py:130-160,254-259) on the same bf16 operands."""
    name, B, T, K, nh, nkv, d, with_bias, q_scale, rope = case
    if path == "fma" and B * T > 4096:
        pytest.skip("the fp32-FMA kernel at this size adds nothing (same functor, W = 4 lane layout covered by the small cases)")
    M, N = B * T, (nh + 2 * nkv) * d
    a = bf16r(rnd(31, f"qr.a{name}", (M, K), 1.0))
    w = bf16r(rnd(31, f"qr.w{name}", (N, K), 0.06 if K > 1000 else 0.25))
    bias = rnd(31, f"qr.b{name}", (N,), 0.5) if with_bias else None
    inv = O.default_inv_freq(10000.0, d) if rope == "default" else O.llama3_inv_freq(500000.0, d, 8.0, 1.0, 4.0, 64)
    acc = a @ w.T
    if bias is not None:
        acc = acc + bias
    x = acc.reshape(B, T, nh + 2 * nkv, d).transpose(0, 2, 1, 3)
    cos, sin = O.rope_cos_sin(inv, np.arange(T))
    q = x[:, :nh] * np.float32(q_scale)
    k, v = x[:, nh:nh + nkv], x[:, nh + nkv:]
    q = q * cos + O.rotate_half(q) * sin
    k = k * cos + O.rotate_half(k) * sin
    wd = np.concatenate([_pack_d128(w[:nh * d], nh), _pack_d128(w[nh * d:(nh + nkv) * d], nkv), _pack_d128(w[(nh + nkv) * d:], nkv)]) if d == 128 else w
    bd = bias
    if d == 128 and bias is not None:
        bd = np.concatenate([_pack_d128(bias[s][:, None], h)[:, 0] for s, h in ((slice(0, nh * d), nh), (slice(nh * d, (nh + nkv) * d), nkv), (slice((nh + nkv) * d, N), nkv))])
    if path == "mfma":
        got = ops.gemm_qkv_rope(to_dev(a, torch.bfloat16), to_dev(wd, torch.bfloat16), to_dev(bd) if bd is not None else None,
                                to_dev(inv), T, nh, nkv, d, q_scale, use_mfma=1)
        tol = 2.5e-3                       # bf16 output rounding: 2^-9 / sqrt(3) = 1.1e-3 relative L2
    else:
        got = ops.gemm_qkv_rope(to_dev(a), to_dev(wd), to_dev(bd) if bd is not None else None, to_dev(inv), T, nh, nkv, d, q_scale, use_mfma=0)
        tol = 3e-6
    for nm, g, r in zip("qkv", got, (q, k, v)):
        observe(f"qkv_rope[{name},{path}].{nm}", rel(to_np(g), r), tol)
        assert maxabs(to_np(g), r) < (0.03 if path == "mfma" else 1e-4) * max(1.0, float(np.abs(r).max()))


def test_gemm_argument_errors(ops):
    from p2t_hip import _lib
    with pytest.raises(ValueError):
        _lib.call("p2t_set_gemm_policy", 11)
    if not _lib.call("p2t_is_lab_build"):                   # the product library carries the default policy and 9, nothing else
        for forced in (2, 3, 7, 12, 128):
            with pytest.raises(ValueError, match="not in the product library"):
                _lib.call("p2t_set_gemm_policy", forced)
    a, w = torch.zeros((8, 64), device=dev()), torch.zeros((24, 64), device=dev())
    with pytest.raises(ValueError):
        ops.gemm_nt(a, w)                                   # N % 16 != 0
    with pytest.raises(Exception):
        ops.gemm_nt(a, torch.zeros((32, 64), device=dev()), use_mfma=1)   # fp32 cannot take the MFMA kernel


# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("cols", [64, 320, 480, 2560, 4096])
def test_norms(ops, cols):
    rows = 37
    x, w, b = rnd(5, "n.x", (rows, cols), 2.0, 0.3), rnd(5, "n.w", (cols,), 0.1, 1.0), rnd(5, "n.b", (cols,), 0.1)
    for od, tol in ((torch.float32, 2e-6), (torch.bfloat16, 3e-3)):
        y = to_np(ops.layernorm(to_dev(x), to_dev(w), to_dev(b), 1e-5, od))
        assert rel(y[:, :cols], O.layer_norm(x, w, b, 1e-5)) < tol and not y[:, cols:].any()
        y = to_np(ops.rmsnorm(to_dev(x), to_dev(w), 1e-5, od))
        assert rel(y[:, :cols], O.rms_norm(x, w, 1e-5)) < tol and not y[:, cols:].any()


def _attn_ref(q, k, v, mask, scale, causal):
    """q [B,nh,T,d], k/v [B,nkv,T,d] fp32 -> [B,T,nh*d] via the oracle's softmax."""
    B, nh, T, d = q.shape
    rep = nh // k.shape[1]
    k, v = np.repeat(k, rep, axis=1), np.repeat(v, rep, axis=1)
    allowed = (mask[:, None, None, :] != 0)
    if causal:
        allowed = allowed & np.tril(np.ones((T, T), dtype=bool))[None, None]
    s = np.einsum("bhqd,bhkd->bhqk", q, k).astype(np.float32) * np.float32(scale) + np.where(allowed, 0.0, O.NEG).astype(np.float32)
    p = O.softmax_rows(s)
    o = np.einsum("bhqk,bhkd->bhqd", p, v).astype(np.float32)
    return o.transpose(0, 2, 1, 3).reshape(B, T, nh * d)


ATTN_CASES = [  # B, T, nh, nkv, d, causal, lens
    (2, 70, 4, 4, 16, False, [70, 33]),
    (3, 128, 4, 4, 24, False, [128, 77, 5]),
    (2, 200, 3, 3, 64, False, [200, 129]),
    (2, 96, 8, 2, 64, True, [96, 40]),
    (2, 130, 4, 1, 128, True, [130, 64]),
    (1, 64, 2, 2, 32, False, [1]),
]


@pytest.mark.parametrize("case", ATTN_CASES)
@pytest.mark.parametrize("path", ["fp32", "mfma"])
def test_qkv_post_and_attention(ops, case, path):
    B, T, nh, nkv, d, causal, lens = case
    dt = torch.float32 if path == "fp32" else torch.bfloat16
    rq = (lambda x: x) if path == "fp32" else bf16r
    width = (nh + 2 * nkv) * d
    qkv = rq(rnd(6, "a.qkv", (B * T, width), 1.5))
    mask = np.zeros((B, T), dtype=np.int64)
    for b, n in enumerate(lens):
        mask[b, :n] = 1
    inv = O.default_inv_freq(10000.0, d)
    cos, sin = O.rope_cos_sin(inv, np.arange(T))
    x = qkv.reshape(B, T, nh + 2 * nkv, d).transpose(0, 2, 1, 3)
    q_scale = d ** -0.5 if not causal else 1.0
    scale = 1.0 if not causal else d ** -0.5
    q = x[:, :nh] * np.float32(q_scale)
    k, v = x[:, nh:nh + nkv], x[:, nh + nkv:]
    q = rq(q * cos + O.rotate_half(q) * sin)
    k = rq(k * cos + O.rotate_half(k) * sin)
    key_mask, kv_info, _ = ops.mask_prepare(to_dev(mask))
    assert to_np(kv_info)[:B].tolist() == lens and to_np(kv_info)[B:].tolist() == [1] * B
    qd, kd, vd = ops.qkv_post(to_dev(qkv, dt), to_dev(inv), B, T, nh, nkv, d, q_scale)
    tol = 1e-6 if path == "fp32" else 4e-3
    assert rel(to_np(qd)[..., :d], q) < tol and rel(to_np(kd)[..., :d], k) < tol
    assert np.array_equal(to_np(vd)[..., :d], v)
    assert not to_np(qd)[..., d:].any() and not to_np(kd)[..., d:].any() and not to_np(vd)[..., d:].any()
    out = ops.attention(qd, kd, vd, key_mask, kv_info, d, scale, causal, use_mfma=(0 if path == "fp32" else 1))
    got = to_np(out).reshape(B, T, -1)
    ref = _attn_ref(to_np(qd)[..., :d], to_np(kd)[..., :d], v, mask, scale, causal)
    for b, n in enumerate(lens):            # every query row, padded ones included (they see the valid keys)
        assert rel(got[b, :, :nh * d], ref[b]) < (3e-6 if path == "fp32" else 8e-3), (b, n)
    assert not got[..., nh * d:].any()


def test_attention_non_prefix_mask(ops):
    """A mask with a hole (not the right-padded contract) takes the byte-mask branch."""
    B, T, nh, d = 1, 100, 2, 64
    mask = np.ones((B, T), dtype=np.int64)
    mask[0, 10:20] = 0
    mask[0, 90:] = 0
    qkv = bf16r(rnd(7, "h.qkv", (B * T, 3 * nh * d), 1.0))
    inv = O.default_inv_freq(10000.0, d)
    key_mask, kv_info, _ = ops.mask_prepare(to_dev(mask))
    assert to_np(kv_info).tolist() == [90, 0]
    qd, kd, vd = ops.qkv_post(to_dev(qkv, torch.bfloat16), to_dev(inv), B, T, nh, nh, d, 1.0)
    v = qkv.reshape(B, T, 3 * nh, d).transpose(0, 2, 1, 3)[:, 2 * nh:]
    ref = _attn_ref(to_np(qd), to_np(kd), v, mask, 0.125, False)
    for use in (0, 1):
        got = to_np(ops.attention(qd, kd, vd, key_mask, kv_info, d, 0.125, False, use_mfma=use)).reshape(B, T, -1)
        assert rel(got[..., :nh * d], ref) < 8e-3


# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("mode", ["last", "mean", "std", "mix"])
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_readout_forward_backward(ops, mode, dt):
    B, T, D = 5, 37, 72
    emb = rnd(8, "r.e", (B, T, D), 2.0, 0.3)
    if dt == torch.bfloat16:
        emb = bf16r(emb)
    lens = [37, 20, 9, 2, 1]
    mask = np.zeros((B, T), dtype=np.int64)
    for b, n in enumerate(lens):
        mask[b, :n] = 1
    e = to_dev(emb, dt)
    got = to_np(ops.readout(e, to_dev(mask), mode))
    ref = O.readout_embeddings(emb, mask, mode)
    ok = np.isfinite(ref)
    assert rel(got[ok], ref[ok]) < 3e-6
    assert rel(to_np(ops.readout(e, None, mode)), O.readout_embeddings(emb, np.ones_like(mask), mode)) < 3e-6
    d_out = rnd(8, "r.g", ref.shape, 1.0)
    pooled = ops.readout(e, to_dev(mask), "mix") if mode in ("std", "mix") else None
    g = to_np(ops.readout_backward(e, to_dev(mask), mode, pooled, to_dev(d_out)))
    gref = O.readout_backward(emb, mask, mode, d_out)
    rows = [b for b, n in enumerate(lens) if n > 1 or mode in ("last", "mean")]     # std of one token = 0 -> 0/0
    assert rel(g[rows], gref[rows]) < 1e-5


def test_l2norm_and_infonce(ops):
    x = rnd(9, "l.x", (12, 256), 1.0, 0.1)
    y = to_np(ops.l2norm_rows(to_dev(x)))
    assert rel(y, O.l2_normalize(x)) < 1e-6
    dy = rnd(9, "l.g", (12, 256), 1.0)
    assert rel(to_np(ops.l2norm_rows_backward(to_dev(x), to_dev(dy))), O.l2_normalize_backward(x, dy)) < 2e-6
    p, t = O.l2_normalize(rnd(9, "i.p", (6, 128))), O.l2_normalize(rnd(9, "i.t", (10, 128)))
    labels = np.array([3, 9, 0, 4, 4, 7])
    loss, logits = ops.infonce_forward(to_dev(p), to_dev(t), to_dev(labels))
    ref_loss, ref_dp, ref_logits = O.infonce_segmented(p, t, labels, return_grad=True)
    assert abs(float(to_np(loss)[0]) - float(ref_loss)) < 2e-6 * max(1.0, abs(float(ref_loss)))
    assert rel(to_np(logits), ref_logits) < 1e-6
    dp = ops.infonce_backward(to_dev(t), to_dev(labels), logits)
    assert rel(to_np(dp), ref_dp) < 2e-6
    # accumulate + weight (the segment loop of the step)
    acc = torch.zeros((1,), device=dev())
    ops.infonce_forward(to_dev(p[:3]), to_dev(t), to_dev(labels[:3]), weight=0.5, loss_out=acc, accumulate=False)
    ops.infonce_forward(to_dev(p[3:]), to_dev(t), to_dev(labels[3:]), weight=0.5, loss_out=acc, accumulate=True)
    two = 0.5 * (O.infonce_segmented(p[:3], t, labels[:3]) + O.infonce_segmented(p[3:], t, labels[3:]))
    assert abs(float(to_np(acc)[0]) - float(two)) < 3e-6


def test_infonce_column_term_vs_reference_swapped_arguments(ops, golden):
    """Column (text -> protein) term: kernels and the loss modules against the reference's BatchInfoNCELoss applied with
    swapped arguments and torch autograd through it (tests/golden/ops.npz, make_golden.py run_ops)."""
    import p2t_hip as P
    g = golden("ops")
    p, t = g["p"], g["t"]
    n = p.shape[0]
    pd, td = to_dev(p), to_dev(t)
    loss, col_lse = ops.infonce_col_forward(pd, td)
    assert abs(float(to_np(loss)[0]) - float(g["loss_batch_swapped"])) < 2e-6 * max(1.0, abs(float(g["loss_batch_swapped"])))
    lg = (p @ t.T / np.float32(0.05)).astype(np.float64)
    assert rel(to_np(col_lse), np.log(np.exp(lg).sum(0))) < 1e-6
    _, logits = ops.infonce_forward(pd, td, to_dev(np.arange(n)))
    d = ops.infonce_col_backward(td, to_dev(np.arange(n)), logits, col_lse, scale=1.0 / n)
    assert rel(to_np(d), g["grad_swapped_p"]) < 3e-6
    # a sub-range / explicit column list of the mean, accumulated with a weight
    acc = torch.full((1,), 2.0, device=dev())
    ops.infonce_col_forward(pd, td, cols=to_dev(np.array([4, 1])), weight=0.5, loss_out=acc, accumulate=True)
    assert abs(float(to_np(acc)[0]) - (2.0 + 0.5 * float(O.infonce_columns(p, t, [1, 4])))) < 3e-6
    # module surface: symmetric BatchInfoNCELoss and its gradient to the first argument
    pr = to_dev(p).requires_grad_(True)
    sym = P.BatchInfoNCELoss(symmetric=True)(pr, td)
    sym.backward()
    assert abs(float(sym) - float(g["loss_symmetric"])) < 3e-6 * max(1.0, abs(float(g["loss_symmetric"])))
    assert rel(to_np(pr.grad), g["grad_symmetric_p"]) < 3e-6
    # segmented form with the column term: the mean over segments has the whole-batch value and gradient
    pr2 = to_dev(p).requires_grad_(True)
    lf = P.SegmentedBatchInfoNCELoss(column_weight=0.5)
    tot = sum(lf(pr2[a:b], td, torch.arange(a, b, device=dev()), batch_output1=pr2) for a, b in ((0, 2), (2, 4), (4, 6))) / 3
    tot.backward()
    assert abs(float(tot) - float(g["loss_symmetric"])) < 3e-6 * max(1.0, abs(float(g["loss_symmetric"])))
    assert rel(to_np(pr2.grad), g["grad_symmetric_p"]) < 3e-6
    with pytest.raises(ValueError):
        lf(pr2[:2], td, torch.arange(2, device=dev()))                      # column term without the whole first side
    assert float(P.BatchInfoNCELoss()(pr, td)) == pytest.approx(float(g["loss_batch"]), rel=3e-6)     # default: row-only, as upstream


@pytest.mark.parametrize("max_norm", [math.inf, 0.05])
def test_clip_adamw(ops, max_norm):
    names = ["w1", "b1", "w2", "b2"]
    shapes = [(48, 32), (48,), (64, 48), (64,)]
    P = {n: rnd(10, "o.p" + n, s, 0.5) for n, s in zip(names, shapes)}
    G = {n: rnd(10, "o.g" + n, s, 0.02) for n, s in zip(names, shapes)}
    m = {n: np.zeros(s, np.float32) for n, s in zip(names, shapes)}
    v = {n: np.zeros(s, np.float32) for n, s in zip(names, shapes)}
    dp = [to_dev(P[n]) for n in names]
    dm = [torch.zeros_like(t) for t in dp]
    dv = [torch.zeros_like(t) for t in dp]
    sh1 = torch.zeros((48, 64), dtype=torch.bfloat16, device=dev())
    sh2 = torch.zeros((64, 64), dtype=torch.bfloat16, device=dev())
    for step in (1, 2, 3):
        gs = {n: G[n] * np.float32(step) for n in names}
        gn_ref = O.clip_and_adamw(P, gs, m, v, step, max_norm=max_norm)
        gn = ops.clip_adamw_step(dp, [to_dev(gs[n]) for n in names], dm, dv, step, max_norm=max_norm,
                                 shadows=[sh1, None, sh2, None])
        assert abs(float(to_np(gn)[0]) - float(gn_ref)) < 1e-5 * float(gn_ref)
        for t, n in zip(dp, names):
            np.testing.assert_allclose(to_np(t), P[n], rtol=2e-6, atol=1e-7)
    assert np.array_equal(to_np(sh1)[:, :32], bf16r(to_np(dp[0]))) and not to_np(sh1)[:, 32:].any()
    assert np.array_equal(to_np(sh2)[:, :48], bf16r(to_np(dp[2])))


@pytest.mark.parametrize("tile", ["0", "9"])
def test_gemm_persistent_forms_fuzz(ops, tile, gemm_policy):
    """Random whole-tile shapes through the persistent kernels (default policy = four-wave forms; 9 = the eight-wave forms; the forced
    forms -- split-K fix-up always / never, 128-row halves, ... -- run against the lab build: tests/test_gpu_lab_forms.py) against
    the exact fp32-FMA kernel on the same bf16 operands: any stale LDS read or mis-counted wait shows up as a wrong tile."""
    gemm_policy(int(tile))
    rng = np.random.default_rng(int(tile) + 7)
    ws = ops.gemm_fix_workspace(dev())
    epoch = 0
    for _ in range(10):
        tiles = int(rng.integers(256, 700))
        tm = int(rng.choice([d for d in range(4, 65) if tiles // d >= 4]))
        tn = max(4, tiles // tm)
        M, N, K = 256 * tm, 256 * tn, 128 * int(rng.integers(3, 21))
        a = torch.empty((M, K), dtype=torch.bfloat16, device=dev())
        w = torch.empty((N, K), dtype=torch.bfloat16, device=dev())
        ops.fill_hash_(a, 3, f"fz.a{M}x{K}", 1.0)
        ops.fill_hash_(w, 3, f"fz.w{N}x{K}", 0.5)
        b = to_dev(rnd(3, "fz.b", (N,), 0.3))
        for epi in (EPI_STORE, EPI_RESID):
            out0 = torch.ones((M, N), dtype=torch.float32, device=dev()) if epi == EPI_RESID else None
            out1 = torch.ones((M, N), dtype=torch.float32, device=dev()) if epi == EPI_RESID else None
            epoch += 1
            got = ops.gemm_nt(a, w, b, epilogue=epi, out=out1, out_dtype=torch.float32, use_mfma=1, fix_ws=ws, fix_epoch=epoch)
            ref = ops.gemm_nt(a, w, b, epilogue=epi, out=out0, out_dtype=torch.float32, use_mfma=0)
            err = float((got[:, :N] - ref[:, :N]).abs().max() / ref[:, :N].abs().max())
            assert err < 2e-5, (M, N, K, epi, err)
    assert _no_timeout()                     # no split-K consumer timed out


@pytest.mark.parametrize("epi", [EPI_STORE, EPI_RESID])
def test_gemm_four_wave_split_k_pairs_under_the_default_policy(ops, epi, gemm_policy):
    """FFN-down-like shape (K = 6144 >= the default policy's threshold, 1.25 rounds of tiles): the default policy runs the
    partial round as split-K pairs INSIDE the four-wave kernel (producer slab + consumer wait written as one asm statement);
    against the eight-wave kernels (policy 9) on the same operands."""
    M, N, K = 8192, 2560, 6144
    a = torch.empty((M, K), dtype=torch.bfloat16, device=dev())
    w = torch.empty((N, K), dtype=torch.bfloat16, device=dev())
    ops.fill_hash_(a, 5, "w4t.a", 1.0)
    ops.fill_hash_(w, 5, "w4t.w", 0.5)
    b = to_dev(rnd(5, "w4t.b", (N,), 0.3))
    ws = ops.gemm_fix_workspace(dev())
    outs = []
    for rep, pol in enumerate((0, 0, 9)):
        gemm_policy(pol)
        out = torch.ones((M, N), dtype=torch.float32, device=dev()) if epi == EPI_RESID else None
        outs.append(ops.gemm_nt(a, w, b, epilogue=epi, out=out, out_dtype=torch.float32, use_mfma=1, fix_ws=ws, fix_epoch=100 + rep).clone())
    assert torch.equal(outs[0], outs[1])                                   # deterministic
    err = float((outs[0][:, :N] - outs[2][:, :N]).abs().max() / outs[2][:, :N].abs().max())
    assert err < 2e-5, err
    assert _no_timeout()


@pytest.mark.parametrize("epi", [EPI_STORE, EPI_RESID])
@pytest.mark.parametrize("shape", [(2048, 4096, 8192), (2048, 4096, 14336), (1536, 4096, 8192)])
def test_gemm_four_wave_every_tile_as_a_split_k_pair(ops, epi, shape, gemm_policy):
    """Text-tower shapes at 2 048 tokens (128 / 96 tiles = at most half a round): the default policy runs EVERY tile as a split-K
    pair on the four-wave kernel (gemm_w4.hip PAIRS_ONLY); against the eight-wave forms (policy 9) on the same operands."""
    M, N, K = shape
    a = torch.empty((M, K), dtype=torch.bfloat16, device=dev())
    w = torch.empty((N, K), dtype=torch.bfloat16, device=dev())
    ops.fill_hash_(a, 6, f"w4p.a{K}", 1.0)
    ops.fill_hash_(w, 6, f"w4p.w{K}", 0.5)
    b = to_dev(rnd(6, "w4p.b", (N,), 0.3))
    ws = ops.gemm_fix_workspace(dev())
    outs = []
    for rep, pol in enumerate((0, 0, 9)):
        gemm_policy(pol)
        out = torch.ones((M, N), dtype=torch.float32, device=dev()) if epi == EPI_RESID else None
        outs.append(ops.gemm_nt(a, w, b, epilogue=epi, out=out, out_dtype=torch.float32, use_mfma=1, fix_ws=ws, fix_epoch=200 + rep).clone())
    assert torch.equal(outs[0], outs[1])
    err = float((outs[0][:, :N] - outs[2][:, :N]).abs().max() / outs[2][:, :N].abs().max())
    assert err < 2e-5, err
    assert _no_timeout()


def test_gemm_four_wave_long_k_fuzz_vs_eight_wave_forms(ops, gemm_policy):
    """Random shapes with LONG K (up to 16 384: the in-kernel split-K pairs of the default policy, pairs on every tile for grids of at
    most half a round, whole-tile partial rounds) and every four-wave epilogue the towers use: default policy against the eight-wave
    forms (policy 9) on the same operands -- same products, fp32 sums associated differently only where the K split differs."""
    rng = np.random.default_rng(2024)
    ws = ops.gemm_fix_workspace(dev())
    epoch = 1000
    for it in range(12):
        if it % 3 == 2:
            tm, tn = int(rng.integers(4, 9)), int(rng.integers(12, 17))          # 48 .. 128 tiles: at most half a round
        else:
            tiles = int(rng.integers(260, 700))
            tm = int(rng.choice([d for d in range(4, 40) if tiles // d >= 4]))
            tn = max(4, tiles // tm)
        M, N, K = 256 * tm, 256 * tn, 128 * int(rng.integers(40, 129))
        a = torch.empty((M, K), dtype=torch.bfloat16, device=dev())
        w = torch.empty((N, K), dtype=torch.bfloat16, device=dev())
        ops.fill_hash_(a, 4, f"lk.a{it}", 1.0)
        ops.fill_hash_(w, 4, f"lk.w{it}", 0.5)
        b = to_dev(rnd(4, f"lk.b{it}", (N,), 0.3))
        for epi in (EPI_STORE, EPI_RESID, EPI_GELU):
            outs = []
            for pol in (0, 9):
                gemm_policy(pol)
                epoch += 1
                out = torch.ones((M, N), dtype=torch.float32, device=dev()) if epi == EPI_RESID else None
                outs.append(ops.gemm_nt(a, w, b, epilogue=epi, out=out, out_dtype=torch.float32 if epi == EPI_RESID else torch.bfloat16,
                                        use_mfma=1, fix_ws=ws, fix_epoch=epoch).float().clone())
            err = float((outs[0][:, :N] - outs[1][:, :N]).abs().max() / outs[1][:, :N].abs().max())
            assert err < (1e-5 if epi == EPI_RESID else 8e-3), (M, N, K, epi, err)       # bf16 outputs: one rounding step at most
    assert _no_timeout()


def test_attention_hand_placed_kernel_vs_exact_and_general(ops):
    """csrc/attn_fwd64.hip (head_dim padded to 64, log2-scores q; the kernel bench.py's ESM2 towers run) against the exact
    fp32-softmax kernel and the general MFMA kernel on the same bf16 operands: ragged / causal / GQA / non-prefix masks, sequence
    lengths on both sides of the 64-key tile and 256-query block boundaries, one to many blocks per workgroup (the kernel is
    persistent), padded head dims, the pad columns of the last head, and the log-sum-exp output the backward reads."""
    rng = np.random.default_rng(23)
    L2E = 1.4426950408889634
    shapes = [(1, 64, 1, 1, 64), (1, 65, 2, 1, 64), (2, 256, 2, 2, 64), (2, 257, 3, 3, 40), (1, 1, 2, 2, 64), (3, 130, 8, 8, 40), (2, 700, 4, 1, 64),
              (2, 1024, 5, 5, 64), (40, 300, 8, 2, 64), (1, 2048, 2, 2, 64), (2, 511, 3, 1, 48)]
    for it, (B, T, nh, nkv, d) in enumerate(shapes):
        for causal in (False, True):
            lens = [T if b == 0 else int(rng.integers(1, T + 1)) for b in range(B)]
            mask = np.zeros((B, T), dtype=np.int64)
            for b, n in enumerate(lens):
                mask[b, :n] = 1
            non_prefix = it % 4 == 3 and not causal
            if non_prefix:
                mask = (rng.random((B, T)) < 0.6).astype(np.int64)
                mask[:, 0] = 0
                mask[:, T // 2] = 1
            qkv = to_dev(bf16r(rnd(60 + it, "ah.qkv", (B * T, (nh + 2 * nkv) * d), float(rng.choice([0.5, 1.5, 4.0])))), torch.bfloat16)
            inv = to_dev(O.default_inv_freq(10000.0, d))
            key_mask, kv_info, _ = ops.mask_prepare(to_dev(mask))
            q, k, v = ops.qkv_post(qkv, inv, B, T, nh, nkv, d, float(d) ** -0.5 * L2E)
            lse = [torch.zeros((B, nh, T), dtype=torch.float32, device=dev()) for _ in range(2)]
            a = to_np(ops.attention(q, k, v, key_mask, kv_info, d, 1.0, causal, use_mfma=3, log2_scores=True, lse=lse[0])).reshape(B, T, -1)
            r = to_np(ops.attention(q, k, v, key_mask, kv_info, d, 1.0, causal, use_mfma=0, log2_scores=True, lse=lse[1])).reshape(B, T, -1)
            gm = to_np(ops.attention(q, k, v, key_mask, kv_info, d, 1.0, causal, use_mfma=2, log2_scores=True)).reshape(B, T, -1)
            auto = to_np(ops.attention(q, k, v, key_mask, kv_info, d, 1.0, causal, use_mfma=-1, log2_scores=True)).reshape(B, T, -1)
            assert np.array_equal(auto, a)                                 # the default route for this shape IS the hand-placed kernel
            assert np.isfinite(a).all(), (B, T, nh, nkv, d, causal)
            assert not a[:, :, nh * d:].any()                              # pad columns up to the next multiple of 64 are zeroed
            la, lr = to_np(lse[0]), to_np(lse[1])
            for b in range(B):
                rows = np.nonzero(mask[b])[0]
                ra, rr = a[b, rows, :nh * d], r[b, rows, :nh * d]
                assert rel(ra, rr) < 1e-2, (B, T, nh, nkv, d, causal, lens, rel(ra, rr))
                assert rel(gm[b, rows, :nh * d], rr) < 1e-2
                # ln sum exp: the row sums are taken over the bf16-rounded probabilities (the operand of the PV product)
                fin = np.isfinite(lr[b][:, rows])
                assert np.array_equal(fin, np.isfinite(la[b][:, rows]))
                assert np.abs(la[b][:, rows][fin] - lr[b][:, rows][fin]).max() < 1e-2
    with pytest.raises(Exception, match="hand-placed"):
        B, T, nh, d = 1, 32, 2, 32
        qkv = to_dev(bf16r(rnd(1, "ah.r", (B * T, 3 * nh * d), 1.0)), torch.bfloat16)
        key_mask, kv_info, _ = ops.mask_prepare(to_dev(np.ones((B, T), dtype=np.int64)))
        q, k, v = ops.qkv_post(qkv, to_dev(O.default_inv_freq(10000.0, d)), B, T, nh, nh, d, 1.0)
        ops.attention(q, k, v, key_mask, kv_info, d, 1.0, False, use_mfma=3, log2_scores=True)


def test_attention_fuzz_mfma_vs_simple(ops):
    """Random shapes / lengths / causal flags: the MFMA flash kernel (three-buffer prefetch, lazy rescale, XCD block order)
    against the straightforward fp32-softmax kernel on the same bf16 q, k, v."""
    rng = np.random.default_rng(11)
    for it in range(24):
        d = int(rng.choice([16, 24, 32, 64, 64, 128]))
        nkv = int(rng.choice([1, 2, 4]))
        nh = nkv * int(rng.choice([1, 2, 4]))
        B = int(rng.integers(1, 4))
        T = int(rng.integers(1, 330))
        causal = bool(rng.integers(0, 2))
        lens = [int(rng.integers(1, T + 1)) for _ in range(B)]
        if it % 5 == 0:
            lens[0] = T
        mask = np.zeros((B, T), dtype=np.int64)
        for b, n in enumerate(lens):
            mask[b, :n] = 1
        qkv = to_dev(bf16r(rnd(20 + it, "af.qkv", (B * T, (nh + 2 * nkv) * d), float(rng.choice([0.5, 1.5, 4.0])))), torch.bfloat16)
        inv = to_dev(O.default_inv_freq(10000.0, d))
        key_mask, kv_info, _ = ops.mask_prepare(to_dev(mask))
        q, k, v = ops.qkv_post(qkv, inv, B, T, nh, nkv, d, 1.0)
        scale = float(d) ** -0.5
        a = to_np(ops.attention(q, k, v, key_mask, kv_info, d, scale, causal, use_mfma=1)).reshape(B, T, -1)
        r = to_np(ops.attention(q, k, v, key_mask, kv_info, d, scale, causal, use_mfma=0)).reshape(B, T, -1)
        # the towers' form for bf16 models: scale * log2(e) folded into q where it is written, the kernel's exponent is q k^T itself
        q2, _, _ = ops.qkv_post(qkv, inv, B, T, nh, nkv, d, scale * 1.4426950408889634)
        a2 = to_np(ops.attention(q2, k, v, key_mask, kv_info, d, 1.0, causal, use_mfma=1, log2_scores=True)).reshape(B, T, -1)
        r2 = to_np(ops.attention(q2, k, v, key_mask, kv_info, d, 1.0, causal, use_mfma=0, log2_scores=True)).reshape(B, T, -1)
        assert np.isfinite(a).all() and np.isfinite(a2).all(), (it, B, T, nh, nkv, d, causal, lens)
        for b, n in enumerate(lens):
            assert rel(a[b, :n, :nh * d], r[b, :n, :nh * d]) < 1e-2, (it, B, T, nh, nkv, d, causal, lens)
            assert rel(a2[b, :n, :nh * d], r2[b, :n, :nh * d]) < 1e-2, (it, B, T, nh, nkv, d, causal, lens)      # same operands, both kernels
            assert rel(a2[b, :n, :nh * d], r[b, :n, :nh * d]) < 2e-2, (it, B, T, nh, nkv, d, causal, lens)       # q rounded after the fold
