"""Stage-2 training step through the frozen decoder ("LoRA later": the dX chain is what comes first):
`loss = model(**batch).loss; loss.backward()` of the reference (scripts/train_instruct.py:192-213;
models/modeling_esm2llama_instruct.py:195-215) with the towers frozen and the adapter trainable.

* the new kernels one by one against numpy / the oracle (rmsnorm backward, cross-entropy backward, attention log-sum-exp +
  backward with GQA / causal / non-prefix masks / head_dim 16, 64, 128, the gather that undoes the placeholder scatter);
* the whole step against torch autograd through the REFERENCE class (tests/golden/sft_grad_tiny.npz, make_golden.py
  run_sft_backward): loss, gradient at the decoder inputs, the adapter's four gradients -- fp32 at 5e-4 (north_star allows
  1e-3), bf16 with observed tolerances; three decoder shapes (generic head_dim 16, fused QKV + RoPE head_dim 64 with GQA, 128).
"""
import json
import os

import numpy as np
import pytest
import torch

from oracle import p2t_oracle as O
from gpu_util import bf16r, build_model, dev, observe, rel, rnd, to_dev, to_np
from p2t_hip import specs

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ADAPTER = ("fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias")


@pytest.fixture(scope="module")
def g():
    z = np.load(os.path.join(HERE, "golden", "sft_grad_tiny.npz"))
    d = {k: z[k] for k in z.files}
    d["meta"] = json.loads(bytes(d.pop("meta_json")).decode())
    return d


def _model(g, case, dtype):
    m = g["meta"]["cases"][case]
    model = build_model(specs.EsmSpec(**m["esm"]), specs.LlamaSpec(**m["llama"]), specs.AdapterSpec(**m["adapter"]), dtype, 0)
    model.config.placeholder_id = g["meta"]["placeholder_id"]
    model.eval()                                        # no dropout, as the golden
    model.requires_grad_(False)
    model.adapter.requires_grad_(True)                  # stage 2 without LoRA: the adapter is what trains
    return model


def _inputs(g):
    return dict(input_ids=to_dev(g["input_ids"]), attention_mask=to_dev(g["attention_mask"]),
                protein_input_ids=to_dev(g["protein_input_ids"]), protein_attention_mask=to_dev(g["protein_attention_mask"]))


# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dy_dt", [torch.float32, torch.bfloat16])
def test_rmsnorm_backward_vs_oracle(dy_dt):
    from p2t_hip import _lib
    from p2t_hip.ops import ptr, stream
    rows, cols = 37, 320
    x, w = rnd(5, "rb.x", (rows, cols), 2.0, 0.3), rnd(5, "rb.w", (cols,), 0.1, 1.0)
    dy = rnd(5, "rb.dy", (rows, cols), 1.0)
    dyd, xd, wd = to_dev(dy, dy_dt), to_dev(x), to_dev(w)         # (named: a temporary's memory is recycled once ptr() has returned)
    ref = O.rms_norm_backward(x, w, 1e-5, to_np(dyd).astype(np.float32))
    base = rnd(5, "rb.g", (rows, cols), 1.0)
    for acc in (0, 1):
        out = to_dev(base.copy())
        _lib.call("p2t_rmsnorm_backward", ptr(xd), cols, ptr(wd), 1e-5, ptr(dyd), cols, 0 if dy_dt == torch.float32 else 1, ptr(out),
                  cols, rows, cols, acc, stream())
        assert rel(to_np(out), ref + (base if acc else 0)) < 3e-6


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_cross_entropy_backward_vs_numpy(dt):
    from p2t_hip import _lib, ops
    from p2t_hip.ops import ptr, stream
    rng = np.random.default_rng(1)
    B, T, V, ld = 3, 17, 1000, 1024
    logits = (rng.standard_normal((B, T, ld)) * 3).astype(np.float32)
    labels = rng.integers(0, V, size=(B, T)).astype(np.int64)
    labels[0, :5] = -100
    labels[2, -3:] = -100
    lt = to_dev(logits, dt)
    lab = to_dev(labels)
    _, cnt = ops.cross_entropy_shifted(lt, lab, V)
    dl = torch.full_like(lt, 7.0)
    _lib.call("p2t_cross_entropy_shifted_backward", ptr(lt), ld, ops.dt_of(lt), ptr(lab), B, T, V, -100, ptr(cnt), ptr(dl), ld, stream())
    x = to_np(lt).astype(np.float64)[..., :V]
    p = np.exp(x - x.max(-1, keepdims=True))
    p /= p.sum(-1, keepdims=True)
    tgt = np.full((B, T), -100, dtype=np.int64)
    tgt[:, :-1] = labels[:, 1:]
    counted = tgt != -100
    want = p.copy()
    bi, ti = np.nonzero(counted)
    want[bi, ti, tgt[bi, ti]] -= 1.0
    want = want * counted[..., None] / counted.sum()
    got = to_np(dl).astype(np.float64)
    assert not got[..., V:].any()                                   # the K padding of the LM-head dX GEMM
    assert rel(got[..., :V], want) < (2e-6 if dt == torch.float32 else 4e-3)


CASES = [  # name, B, T, nh, nkv, d, causal, lens (None: full), dtype
    ("d16_gqa_causal", 2, 37, 4, 2, 16, True, [37, 20], torch.float32),
    ("d64_causal_leftpad", 2, 70, 4, 2, 64, True, "left", torch.float32),
    ("d128_bidir", 1, 45, 2, 1, 128, False, [33], torch.float32),
    ("d64_bf16", 2, 130, 4, 2, 64, True, [130, 77], torch.bfloat16),
    ("d128_bf16", 2, 96, 4, 1, 128, True, [96, 50], torch.bfloat16),
    ("d128_bf16_long_leftpad", 2, 333, 8, 2, 128, True, "left", torch.bfloat16),      # several key / query tiles, GQA 4:1, non-prefix mask
    ("d64_bf16_bidir", 2, 200, 2, 2, 64, False, [200, 131], torch.bfloat16),
]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_attention_lse_and_backward_vs_numpy(case):
    """p2t_attention(lse=...) + p2t_attention_backward against the textbook formulas in float64 on the same operands; bf16: the
    towers' form (scale * log2 e folded into q, exponent = q k^T)."""
    from p2t_hip import ops
    name, B, T, nh, nkv, d, causal, lens, dt = case
    rng = np.random.default_rng(len(name))
    mask = np.zeros((B, T), dtype=np.int64)
    if lens == "left":                                  # SFT batches: left-padded prompt, right-padded description
        mask[0, 5:T - 10] = 1
        mask[1, 0:T] = 1
    else:
        for b, n in enumerate(lens):
            mask[b, :n] = 1
    scale = d ** -0.5
    l2s = dt == torch.bfloat16
    fold = scale * 1.4426950408889634 if l2s else 1.0
    qkv = to_dev(bf16r(rnd(3, "ab." + name, (B * T, (nh + 2 * nkv) * d), 1.5)), dt)
    inv = to_dev(O.default_inv_freq(10000.0, d))
    key_mask, kv_info, _ = ops.mask_prepare(to_dev(mask))
    q, k, v = ops.qkv_post(qkv, inv, B, T, nh, nkv, d, fold)
    lse = torch.empty((B, nh, T), dtype=torch.float32, device=dev())
    o = ops.attention(q, k, v, key_mask, kv_info, d, scale, causal, use_mfma=(1 if l2s else 0), log2_scores=l2s, lse=lse)
    d_o = to_dev(bf16r(rnd(4, "ab.do." + name, tuple(o.shape), 1.0)), dt)
    d_o[:, nh * d:] = 0
    forms = {"exact": ops.attention_backward(q, k, v, o, d_o, lse, key_mask, kv_info, d, scale, causal, log2_scores=l2s, use_mfma=0)}
    if l2s:                                            # bf16: the MFMA kernels (csrc/attn_bwd_mfma.hip) beside the exact ones
        forms["mfma"] = ops.attention_backward(q, k, v, o, d_o, lse, key_mask, kv_info, d, scale, causal, log2_scores=True, use_mfma=1)
    # float64 reference on the stored operands
    qn, kn, vn = (to_np(t).astype(np.float64)[..., :d] for t in (q, k, v))
    rep = nh // nkv
    kr, vr = np.repeat(kn, rep, 1), np.repeat(vn, rep, 1)
    c_s = np.log(2.0) if l2s else scale
    S = np.einsum("bhid,bhjd->bhij", qn, kr) * c_s
    allowed = (mask[:, None, None, :] != 0) & (np.tril(np.ones((T, T), bool))[None, None] if causal else True)
    S = np.where(allowed, S, -np.inf)
    m = S.max(-1, keepdims=True)
    m = np.where(np.isfinite(m), m, 0.0)
    E = np.exp(S - m)
    l = E.sum(-1, keepdims=True)
    P = np.divide(E, l, out=np.zeros_like(E), where=l > 0)
    want_lse = np.where(l[..., 0] > 0, m[..., 0] + np.log(np.where(l > 0, l, 1.0))[..., 0], np.inf)
    got_lse = to_np(lse).astype(np.float64)
    rows = np.broadcast_to(allowed.any(-1), (B, nh, T))   # query rows with at least one visible key
    assert np.isinf(got_lse[~rows]).all() and (got_lse[~rows] > 0).all()
    assert np.abs(got_lse[rows] - want_lse[rows]).max() < (2e-5 if not l2s else 2e-2)
    On = np.einsum("bhij,bhjd->bhid", P, vr)
    got_o = to_np(o).astype(np.float64).reshape(B, T, -1)[..., :nh * d].reshape(B, T, nh, d).transpose(0, 2, 1, 3)
    assert rel(got_o[rows], On[rows]) < (2e-5 if not l2s else 1e-2)
    dO = to_np(d_o).astype(np.float64).reshape(B, T, -1)[..., :nh * d].reshape(B, T, nh, d).transpose(0, 2, 1, 3)
    Ost = got_o                                          # D uses the STORED output, as the kernel (and torch) do
    D = (dO * Ost).sum(-1, keepdims=True)
    dP = np.einsum("bhid,bhjd->bhij", dO, vr)
    dS = P * (dP - D)
    want_dq = np.einsum("bhij,bhjd->bhid", dS, kr) * c_s
    want_dk = (np.einsum("bhij,bhid->bhjd", dS, qn) * c_s).reshape(B, nkv, rep, T, d).sum(2)
    want_dv = np.einsum("bhij,bhid->bhjd", P, dO).reshape(B, nkv, rep, T, d).sum(2)
    tol = 3e-5 if not l2s else 2e-2                      # bf16: P is rebuilt from an lse of a bf16-probability forward
    for form, (dq, dk, dv) in forms.items():
        for nm, got, want in (("dq", dq, want_dq), ("dk", dk, want_dk), ("dv", dv, want_dv)):
            gn = to_np(got).astype(np.float64)
            assert np.isfinite(gn).all() and not gn[..., d:].any(), (form, nm)
            observe(f"attn_bwd[{name},{form}].{nm}", rel(gn[..., :d], want), tol)


def test_gather_rows_undoes_scatter_rows():
    from p2t_hip import _lib, ops
    from p2t_hip.ops import ptr, stream
    rng = np.random.default_rng(0)
    ids = rng.integers(0, 5, size=(5, 91)).astype(np.int64)
    mask = (rng.random((5, 40)) < 0.5).astype(np.int64)
    dst_pos, n_dst = ops.positions_where(to_dev(ids), 3)
    src_pos, n_src = ops.positions_where(to_dev(mask))
    H = 72
    g = rng.standard_normal((5 * 91, H)).astype(np.float32)
    out, gd = torch.zeros((5 * 40, H), dtype=torch.float32, device=dev()), to_dev(g)
    _lib.call("p2t_gather_rows_f32", ptr(out), H, ptr(src_pos), ptr(gd), H, ptr(dst_pos), ptr(n_src), ptr(n_dst), min(dst_pos.numel(), src_pos.numel()),
              H, stream())
    a, b = np.flatnonzero(ids.reshape(-1) == 3), np.flatnonzero(mask.reshape(-1))
    n = min(len(a), len(b))
    want = np.zeros((5 * 40, H), np.float32)
    want[b[:n]] = g[a[:n]]
    assert np.array_equal(to_np(out), want)


# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("case", ["d16", "d64", "d128"])
def test_fp32_step_matches_reference_autograd(g, case):
    model = _model(g, case, torch.float32)
    labels = to_dev(g["labels"])
    out = model(**_inputs(g), labels=labels)
    assert out.loss.requires_grad
    out.loss.backward()
    assert abs(float(out.loss) - float(g[f"{case}.loss"])) < 2e-4 * float(g[f"{case}.loss"])
    for n in ADAPTER:
        got = dict(model.adapter.named_parameters())[n].grad
        assert rel(to_np(got), g[f"{case}.grad.{n}"]) < 5e-4, n
    assert model.adapter.ln1.weight.grad is None and model.llama_decoder.lm_head.weight.grad is None
    # the gradient at the decoder inputs (after the placeholder scatter), positions under the attention mask
    model.zero_grad(set_to_none=True)
    emb, mask = model(**_inputs(g), return_decoder_inputs=True)
    emb.retain_grad()
    res = model.llama_decoder(inputs_embeds=emb, attention_mask=mask, labels=labels)
    res.loss.backward()
    valid = g["attention_mask"] != 0
    assert rel(to_np(emb.grad)[valid], g[f"{case}.d_inputs_embeds"][valid]) < 3e-4
    # and against the oracle's manual backward on the same weights
    from helpers import model_weights
    m = g["meta"]["cases"][case]
    esm, llama, ad = specs.EsmSpec(**m["esm"]), specs.LlamaSpec(**m["llama"]), specs.AdapterSpec(**m["adapter"])
    ref = O.sft_step(esm, llama, model_weights(esm, llama, ad, 0, lm_head=True), g["protein_input_ids"], g["protein_attention_mask"], g["input_ids"],
                     g["attention_mask"], g["labels"], g["meta"]["placeholder_id"])
    assert rel(to_np(emb.grad)[valid], ref["d_inputs_embeds"][valid]) < 3e-4
    # a scaled loss scales the gradients (the `loss / gradient_accumulation_steps` of train_instruct.py)
    model.zero_grad(set_to_none=True)
    (model(**_inputs(g), labels=labels).loss * 0.25).backward()
    assert rel(to_np(model.adapter.fc2.weight.grad), 0.25 * g[f"{case}.grad.fc2.weight"]) < 5e-4


@pytest.mark.parametrize("case", ["d16", "d64", "d128"])
def test_bf16_step_close_to_reference_autograd(g, case):
    model = _model(g, case, torch.bfloat16)
    out = model(**_inputs(g), labels=to_dev(g["labels"]))
    out.loss.backward()
    observe(f"sft_grad[{case}].bf16.loss", abs(float(out.loss) - float(g[f"{case}.loss"])) / float(g[f"{case}.loss"]), 2e-2)
    for n in ADAPTER:
        got = dict(model.adapter.named_parameters())[n].grad
        observe(f"sft_grad[{case}].bf16.{n}", rel(to_np(got), g[f"{case}.grad.{n}"]), 1e-1)


def test_gradient_checkpointing_flag_warns_once_in_the_stage2_step(g):
    """`gradient_checkpointing_enable()` (reference models/modeling_esm2llama_instruct.py:253-268) is accepted, and the stage-2 step
    -- which keeps its whole activation tape -- says so once instead of silently returning no memory."""
    import warnings
    model = _model(g, "d16", torch.float32)
    model.gradient_checkpointing_enable(gradient_checkpointing_kwargs={"use_reentrant": False})
    with pytest.warns(RuntimeWarning, match="keeps the full activation tape"):
        model(**_inputs(g), labels=to_dev(g["labels"])).loss.backward()
    with warnings.catch_warnings():
        warnings.simplefilter("error", RuntimeWarning)            # once per model, not per step
        model(**_inputs(g), labels=to_dev(g["labels"])).loss.backward()
    assert rel(to_np(model.adapter.fc2.weight.grad), 2.0 * g["d16.grad.fc2.weight"]) < 5e-4      # (two backward passes accumulated)


def test_no_graph_without_trainable_inputs_and_refusals(g):
    model = _model(g, "d16", torch.float32)
    model.adapter.requires_grad_(False)
    out = model(**_inputs(g), labels=to_dev(g["labels"]))       # nothing trainable: the plain forward, no tape
    assert not out.loss.requires_grad
    with torch.no_grad():
        model.adapter.requires_grad_(True)
        assert not model(**_inputs(g), labels=to_dev(g["labels"])).loss.requires_grad
    bf = _model(g, "d64", torch.bfloat16).set_gemm_dtype("fp8")
    with pytest.raises(Exception, match="model dtype|gemm_fp8"):
        bf(**_inputs(g), labels=to_dev(g["labels"])).loss.backward()


def test_fp32_directional_derivative_at_llama8b_layer_shapes():
    """No golden reaches Llama-3.1-8B's layer shapes (hidden 4096, 32 / 8 heads of 128, FFN 14336, llama3 rotary scaling), so the
    chain is checked there by a size-independent property: the directional derivative of the LM loss along a random direction of
    the adapter's fc2.weight, by central differences on the forward, equals <gradient, direction> from the backward (fp32, two
    decoder layers, left-padded prompts)."""
    import p2t_hip as P
    esm = specs.EsmSpec(num_hidden_layers=2, hidden_size=128, intermediate_size=256, num_attention_heads=2)
    llama = specs.LlamaSpec(num_hidden_layers=2, hidden_size=4096, intermediate_size=14336, num_attention_heads=32, num_key_value_heads=8,
                            vocab_size=2048, rope_type="llama3", rope_theta=500000.0, rope_factor=8.0)
    ad = specs.AdapterSpec(128, 192, 4096, 0.0)
    model = build_model(esm, llama, ad, torch.float32, 5).eval()
    model.requires_grad_(False)
    model.adapter.requires_grad_(True)
    ph = 2047
    model.config.placeholder_id = ph
    rng = np.random.default_rng(3)
    from p2t_hip import synth
    lens = [20, 9]
    pid, pmask = synth.protein_batch(9, 2, 22, [n + 2 for n in lens])
    T_prompt, T_desc = 30, 14
    ids = np.full((2, T_prompt + T_desc), 2046, dtype=np.int64)
    mask = np.zeros_like(ids)
    labels = np.full_like(ids, -100)
    for b, n in enumerate(lens):
        prompt = np.concatenate([rng.integers(0, 2000, 3), np.full(n + 2, ph), rng.integers(0, 2000, 2)])
        ids[b, T_prompt - len(prompt):T_prompt] = prompt
        mask[b, T_prompt - len(prompt):T_prompt] = 1
        nd = int(rng.integers(5, T_desc + 1))
        desc = rng.integers(0, 2000, nd)
        ids[b, T_prompt:T_prompt + nd] = desc
        mask[b, T_prompt:T_prompt + nd] = 1
        labels[b, T_prompt:T_prompt + nd] = desc
    kw = dict(input_ids=to_dev(ids), attention_mask=to_dev(mask), labels=to_dev(labels), protein_input_ids=to_dev(pid),
              protein_attention_mask=to_dev(pmask))
    out = model(**kw)
    out.loss.backward()
    w = model.adapter.fc2.weight
    g = w.grad.detach().clone()
    direction = torch.from_numpy(rng.standard_normal(tuple(w.shape)).astype(np.float32)).to(w.device)
    direction *= float(w.detach().norm() / direction.norm())
    want = float((g * direction).sum())
    eps = 2e-3
    with torch.no_grad():
        base = w.detach().clone()
        w.copy_(base + eps * direction)
        lp = float(model(**kw).loss)
        w.copy_(base - eps * direction)
        lm = float(model(**kw).loss)
        w.copy_(base)
    got = (lp - lm) / (2 * eps)
    assert abs(want) > 1e-4 and abs(got - want) < 2e-2 * abs(want), (got, want, lp, lm)
