"""GPU parity of the fp8 path (BASELINE.json configs[4]: fp8-e4m3 weights, CDNA4 fp8 MFMA), through the C ABI:

* p2t_quant_rows_fp8 / p2t_layernorm_fp8 / p2t_rmsnorm_fp8 against the numpy quantiser -- VALUES bit for bit (the scaling is
  a power of two, so the e4m3 rounding is the only rounding and both sides make the same one), E8M0 scale bytes exact;
* p2t_gemm_nt_fp8 (v_mfma_scale_f32_16x16x128_f8f6f4, hardware-applied row scales) against numpy on the SAME quantised
  operands: fp32 accumulation order is the only difference (f32 outputs 1e-6), plus the bf16 rounding of bf16 outputs;
* the towers with fp8 GEMMs against the oracle in Precision("fp8") mode and against the reference goldens (observed
  errors recorded; tolerances in DESIGN.md section 6).
"""
import numpy as np
import pytest
import torch

from oracle import p2t_oracle as O
from gpu_util import bf16r, dev, maxabs, observe, rel, rnd, to_dev, to_np

pytestmark = pytest.mark.gpu
EPI_STORE, EPI_GELU, EPI_RESID, EPI_SWIGLU, EPI_STORE_F32 = range(5)

def _decode(q_bytes):
    """uint8 e4m3fn codes -> f32 values."""
    b = q_bytes.astype(np.int32)
    s, e, m = b >> 7, (b >> 3) & 15, b & 7
    v = np.where(e == 0, m / 8.0 * 2.0 ** -6, (1 + m / 8.0) * np.exp2((e - 7).astype(np.float64)))
    return np.where(s == 1, -v, v).astype(np.float32)


@pytest.fixture(scope="module")
def ops():
    from p2t_hip import ops as _ops
    return _ops


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(37, 200), (300, 2560), (5, 8), (64, 10240)])
def test_quant_rows_bit_exact(ops, dt, shape):
    rows, cols = shape
    x = rnd(40, "q.x", (rows, cols), 1.0) * np.exp(rnd(40, "q.s", (rows, 1), 6.0)).astype(np.float32)
    x[rows // 2] = 0.0
    x[0, 0], x[1, 1] = 448.0, -3.0e-5
    if dt == torch.bfloat16:
        x = bf16r(x)
    q, sc = ops.quant_rows_fp8(to_dev(x, dt))
    deq, E, codes = O.quant_rows_e4m3(x)
    assert np.array_equal(to_np(sc), E)
    got = _decode(to_np(q))
    assert np.array_equal(got[:, :cols], codes) and not to_np(q)[:, cols:].any()
    assert q.shape[1] % 128 == 0


@pytest.mark.parametrize("cols", [64, 320, 480, 2560, 4096])
def test_norms_fp8_output(ops, cols):
    rows = 37
    x, w, b = rnd(5, "n.x", (rows, cols), 2.0, 0.3), rnd(5, "n.w", (cols,), 0.1, 1.0), rnd(5, "n.b", (cols,), 0.1)
    for bias in (b, None):
        q, sc = ops.norm_fp8(to_dev(x), to_dev(w), to_dev(bias) if bias is not None else None, 1e-5)
        y = O.layer_norm(x, w, b, 1e-5) if bias is not None else O.rms_norm(x, w, 1e-5)
        deq, E, _ = O.quant_rows_e4m3(y)
        got = _decode(to_np(q))[:, :cols] * np.exp2(to_np(sc).astype(np.float32) - 127)[:, None]
        # the f32 norm differs from numpy's in the last bit here and there -> an e4m3 code may flip at a rounding boundary
        assert np.abs(to_np(sc).astype(int) - E.astype(int)).max() <= 1
        observe(f"norm_fp8[{cols},{'ln' if bias is not None else 'rms'}]", rel(got, deq), 2e-3)
        assert np.mean(got != deq) < 2e-3
        assert not to_np(q)[:, cols:].any()


def test_layernorm_fp8_bound_scale(ops):
    """The second scale byte the LayerNorm kernel emits: 2^e >= (||y||_2 bound_w + bound_b) / 448 by the same integer rule."""
    rows, cols = 200, 640
    x, w, b = rnd(6, "nb.x", (rows, cols), 2.0, 0.3), rnd(6, "nb.w", (cols,), 0.1, 1.0), rnd(6, "nb.b", (cols,), 0.1)
    x *= np.exp(rnd(6, "nb.s", (rows, 1), 2.0)).astype(np.float32)
    bw, bb = 1.37, 0.21
    q, sc, bs = ops.norm_fp8(to_dev(x), to_dev(w), to_dev(b), 1e-5, bound=(bw, bb))
    q0, sc0 = ops.norm_fp8(to_dev(x), to_dev(w), to_dev(b), 1e-5)
    assert torch.equal(q, q0) and torch.equal(sc, sc0)
    y = O.layer_norm(x, w, b, 1e-5)
    amax = np.sqrt((y.astype(np.float64) ** 2).sum(-1)) * np.float32(bw) + np.float32(bb)
    E = O.e8m0_of_amax(amax.astype(np.float32))
    d = np.abs(to_np(bs).astype(int) - E.astype(int))
    assert d.max() <= 1 and np.mean(d != 0) < 0.02          # a row whose bound sits on a power-of-two boundary may go either way


@pytest.mark.parametrize("tile", [256, 128])
@pytest.mark.parametrize("shape", [(300, 320, 128), (1000, 96, 256), (2048, 1184, 384), (700, 2560, 2560)])
def test_gemm_fp8_gelu_fp8_epilogue(ops, tile, shape):
    """EPI_GELU_FP8: e4m3 codes of gelu(acc + bias) under GIVEN row scales; codes may differ from numpy's only where the f32
    value sits on a rounding boundary (accumulation order), so: dequantised values within half an e4m3 step, few codes differ."""
    M, N, K = shape
    a = rnd(51, "g8.a", (M, K), 1.0) * np.exp(rnd(51, "g8.as", (M, 1), 1.0)).astype(np.float32)
    w = rnd(51, "g8.w", (N, K), 1.0) / np.float32(np.sqrt(K))
    bias = rnd(51, "g8.b", (N,), 0.3)
    a8, sa = ops.quant_rows_fp8(to_dev(a))
    w8, sw = ops.quant_rows_fp8(to_dev(w))
    da, dw = O.quant_rows_e4m3(a)[0], O.quant_rows_e4m3(w)[0]
    E = O.gelu_bound_scale(da, dw / np.float32(O.FP8_OPERAND_GROWTH), bias)       # operands already rounded: no growth factor needed
    ref_f = O.gelu_erf(da @ dw.T + bias)
    assert np.abs(ref_f).max(-1).max() > 0 and (np.abs(ref_f).max(-1) <= 448 * np.exp2(E.astype(np.float32) - 127)).all(), "the bound holds"
    ref = O.quant_rows_e4m3_scaled(ref_f, E)
    out = ops.gemm_nt_fp8(a8, sa, w8, sw, to_dev(bias), n=N, k=a8.shape[1], epilogue=7, tile=tile, out_row_scale=to_dev(E))
    assert out.dtype == torch.uint8 and out.shape == (M, (N + 127) // 128 * 128)
    codes = to_np(out)
    assert not codes[:, N:].any()
    got = _decode(codes[:, :N]) * np.exp2(E.astype(np.float32) - 127)[:, None]
    observe(f"gemm_fp8_gelu_fp8[{M}x{N}x{K},t{tile}]", rel(got, ref), 5e-3)
    assert np.mean(got != ref) < 5e-3
    # how much of e4m3's range the bound scale gives away (Cauchy-Schwarz is loose): recorded, not asserted
    slack = np.log2(448 * np.exp2(E.astype(np.float32) - 127) / np.abs(ref_f).max(-1))
    observe(f"gemm_fp8_gelu_fp8_slack_binades[{M}x{N}x{K},t{tile}]", float(slack.mean()), 12.0, "log2(bound / amax), mean over rows")


def _gemm_ref(a, w, bias, epi, resid=None):
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
    return acc


@pytest.mark.parametrize("epi", [EPI_STORE, EPI_GELU, EPI_RESID, EPI_SWIGLU, EPI_STORE_F32])
@pytest.mark.parametrize("tile", [256, 128])
# one K step / two (the peeled tail) / many; edge tiles in M and N; whole tiles
@pytest.mark.parametrize("shape", [(300, 320, 128), (1000, 96, 256), (256, 256, 1024), (2048, 1184, 384), (700, 2560, 2560)])
def test_gemm_fp8_vs_numpy_on_the_same_quantised_operands(ops, epi, tile, shape):
    M, N, K = shape
    if epi == EPI_SWIGLU and N % 64:
        pytest.skip("swiglu needs N % 64 == 0")
    a = rnd(50, "f8.a", (M, K), 1.0) * np.exp(rnd(50, "f8.as", (M, 1), 2.0)).astype(np.float32)
    w = rnd(50, "f8.w", (N, K), 0.5) * np.exp(rnd(50, "f8.ws", (N, 1), 2.0)).astype(np.float32) / np.float32(np.sqrt(K))
    bias = None if epi in (EPI_SWIGLU, EPI_STORE_F32) else rnd(50, "f8.b", (N,), 0.3)
    resid = rnd(50, "f8.r", (M, N), 1.0)
    a8, sa = ops.quant_rows_fp8(to_dev(a))
    w8, sw = ops.quant_rows_fp8(to_dev(w))
    da, dw = O.quant_rows_e4m3(a)[0], O.quant_rows_e4m3(w)[0]
    n_out = N // 2 if epi == EPI_SWIGLU else N
    out = to_dev(resid) if epi == EPI_RESID else None
    for od in ((torch.float32,) if epi in (EPI_RESID, EPI_STORE_F32) else (torch.float32, torch.bfloat16)):
        got = to_np(ops.gemm_nt_fp8(a8, sa, w8, sw, to_dev(bias) if bias is not None else None, n=N, k=a8.shape[1], epilogue=epi,
                                    out=out, out_dtype=od, tile=tile))
        ref = _gemm_ref(da, dw, bias, epi, resid)
        tol = 3e-5 if od == torch.float32 else 2.5e-3       # f32: the fp8 MFMA sums its 128 products in a narrower adder than f32 FMA chains (observed ~1e-5)
        observe(f"gemm_fp8[epi{epi},{M}x{N}x{K},t{tile},{'f32' if od == torch.float32 else 'bf16'}]", rel(got[:, :n_out], ref), tol)
        if got.shape[1] > n_out:
            assert not got[:, n_out:].any()


def test_gemm_fp8_large_shapes_vs_bf16_kernel(ops):
    """Tower-sized GEMMs (16384 x 7680 x 2560: 1920 tiles; 16384 x 2560 x 10240: long K) -- too big for numpy in a test, so:
    the fp8 kernel on quantised operands against the EXACT fp32-FMA kernel run on the dequantised operands (same values)."""
    for M, N, K in ((16384, 7680, 2560), (8192, 2560, 10240), (4096, 4096, 14336)):
        a = torch.empty((M, K), dtype=torch.float32, device=dev())
        w = torch.empty((N, K), dtype=torch.float32, device=dev())
        ops.fill_hash_(a, 9, f"f8l.a{M}", 1.0)
        ops.fill_hash_(w, 9, f"f8l.w{N}", 0.05)
        a8, sa = ops.quant_rows_fp8(a)
        w8, sw = ops.quant_rows_fp8(w)
        got = ops.gemm_nt_fp8(a8, sa, w8, sw, None, n=N, k=K, epilogue=EPI_STORE, out_dtype=torch.float32)
        # dequantise on the host side of the test (numpy decode of a slab would be slow: use torch's float8 view)
        da = a8[:, :K].view(torch.float8_e4m3fn).float() * torch.exp2(sa.float() - 127)[:, None]
        dw = w8[:, :K].view(torch.float8_e4m3fn).float() * torch.exp2(sw.float() - 127)[:, None]
        ref = ops.gemm_nt(da, dw, None, epilogue=EPI_STORE, out_dtype=torch.float32, use_mfma=0)
        err = float((got[:, :N] - ref[:, :N]).abs().max() / ref[:, :N].abs().max())
        observe(f"gemm_fp8_large[{M}x{N}x{K}]", err, 1e-4, "max/max")


# ---------------------------------------------------------------------------------------------
# towers with fp8 GEMMs
# ---------------------------------------------------------------------------------------------
def _batch(pid, pmask, tid, tmask):
    return dict(protein_input_ids=to_dev(pid), protein_attention_mask=to_dev(pmask),
                description_input_ids=to_dev(tid), description_attention_mask=to_dev(tmask))


@pytest.mark.parametrize("case", ["tiny", "tiny_d24", "tiny_d128", "tiny_d64", "tiny_qwen3"])
def test_fp8_towers_vs_fp8_oracle_and_reference_goldens(golden, case):
    """Pooled embeddings and loss of the fp8-GEMM towers: against the oracle that quantises the same operands the same way
    (what is left: accumulation order, bf16 roundings flipping e4m3 codes at boundaries), and against the reference's fp32
    outputs (the price of fp8 operands; recorded)."""
    import p2t_hip as P
    from helpers import case_setup, model_weights
    from gpu_util import build_model
    g = golden(case)
    meta = g["meta"]
    esm, llama, ad, pid, pmask, tid, tmask = case_setup(meta)
    model = build_model(esm, llama, ad, torch.bfloat16, meta["seed_w"]).eval().set_gemm_dtype("fp8")
    assert model.esm_encoder.gemm_fp8 and model.llama_decoder.model.gemm_fp8
    b = _batch(pid, pmask, tid, tmask)
    W = model_weights(esm, llama, ad, meta["seed_w"])
    k = meta["layers"][-1]
    with torch.no_grad():
        p = P.l2_normalize(P.get_sequence_embeddings(model, b["protein_input_ids"], b["protein_attention_mask"]))
        t = P.l2_normalize(P.get_description_embeddings(model, b["description_input_ids"], b["description_attention_mask"], k))
        loss = float(P.BatchInfoNCELoss()(p, t))
    po = O.protein_embeddings(esm, W, pid, pmask, "mix", prec=O.FP8)
    to_ = O.text_embeddings(llama, W, tid, tmask, k, "mix", prec=O.FP8)
    # an e4m3 code is a 6 % step: one operand element that lands on the other side of a rounding boundary (accumulation order,
    # a bf16 rounding upstream) moves the outputs far more than in bf16, so the same-quantiser comparison is loose too
    observe(f"{case}.fp8_vs_fp8oracle.protein", rel(to_np(p), po), 8e-2)
    observe(f"{case}.fp8_vs_fp8oracle.text", rel(to_np(t), to_), 8e-2)
    observe(f"{case}.fp8_vs_fp8oracle.loss", abs(loss - float(O.infonce_batch(po, to_))), 2e-1, "abs")
    observe(f"{case}.fp8_vs_reference.protein", rel(to_np(p), g["prot_norm_mix"]), 1.5e-1)
    observe(f"{case}.fp8_vs_reference.text", rel(to_np(t), g[f"text_norm_mix_L{k}"]), 1.5e-1)
    observe(f"{case}.fp8_vs_reference.loss", abs(loss - float(g[f"loss_batch_mix_L{k}"])) / max(1.0, float(g[f"loss_batch_mix_L{k}"])), 2e-1,
            "abs/max(1,|ref|)")
    # back to the model dtype: the bf16 engines are rebuilt and give the bf16 numbers again
    model.set_gemm_dtype("model")
    with torch.no_grad():
        p2 = P.l2_normalize(P.get_sequence_embeddings(model, b["protein_input_ids"], b["protein_attention_mask"]))
    assert rel(to_np(p2), O.protein_embeddings(esm, W, pid, pmask, "mix", prec=O.BF16)) < 1e-2


def test_fp8_step_trains_and_is_padding_invariant():
    """The fused step on fp8 towers (adapter in bf16, the only trained block): finite gradients, a falling loss, and the
    size-independent property that padding a batch does not change the valid rows' embeddings."""
    import p2t_hip as P
    from p2t_hip import specs, synth
    from gpu_util import build_model
    esm = specs.EsmSpec(num_hidden_layers=3, hidden_size=512, intermediate_size=1536, num_attention_heads=8)
    llama = specs.LlamaSpec(num_hidden_layers=3, hidden_size=512, intermediate_size=1408, num_attention_heads=8,
                            num_key_value_heads=2, vocab_size=1024)
    ad = specs.AdapterSpec(esm.hidden_size, 256, llama.hidden_size, 0.0)
    B, Tp, Tt = 8, 300, 70
    pid, pmask = synth.protein_batch(5, B, Tp, [300, 299, 180, 64, 65, 33, 24, 17])
    tid, tmask = synth.text_batch(5, B, Tt, 1000, [70, 64, 50, 33, 20, 9, 3, 1], 1023, 1022)
    b = _batch(pid, pmask, tid, tmask)
    model = build_model(esm, llama, ad, torch.bfloat16, 3, ).set_gemm_dtype("fp8")
    with pytest.raises(ValueError):
        build_model(esm, llama, ad, torch.float32, 3).set_gemm_dtype("fp8")          # fp8 GEMMs need a bf16 model
    with torch.no_grad():
        p0 = to_np(P.l2_normalize(P.get_sequence_embeddings(model.eval(), b["protein_input_ids"], b["protein_attention_mask"])))
        # the same rows inside a longer padded batch
        pid2 = np.concatenate([pid, np.ones((B, 84), np.int64)], 1)
        pm2 = np.concatenate([pmask, np.zeros((B, 84), np.int64)], 1)
        p1 = to_np(P.l2_normalize(P.get_sequence_embeddings(model, to_dev(pid2), to_dev(pm2))))
    observe("fp8.padding_invariance.protein", rel(p1, p0), 5e-2)
    bf = build_model(esm, llama, ad, torch.bfloat16, 3).eval()
    with torch.no_grad():
        pb = to_np(P.l2_normalize(P.get_sequence_embeddings(bf, b["protein_input_ids"], b["protein_attention_mask"])))
    observe("fp8_vs_bf16.protein[d64,3 layers]", rel(p0, pb), 1.5e-1)
    # FFN-up output stored under the per-token bound scale (default) against the amax scale of a separate quantise pass:
    # same distance to the bf16 towers (the bound gives binades of e4m3's range away, not mantissa bits)
    enc = model.esm_encoder
    enc.fp8_fused_gelu = False
    enc.invalidate_engine()
    with torch.no_grad():
        pu = to_np(P.l2_normalize(P.get_sequence_embeddings(model, b["protein_input_ids"], b["protein_attention_mask"])))
    enc.fp8_fused_gelu = True
    enc.invalidate_engine()
    observe("fp8_unfused_gelu_vs_bf16.protein[d64,3 layers]", rel(pu, pb), 1.5e-1)
    observe("fp8_fused_vs_unfused_gelu.protein[d64,3 layers]", rel(p0, pu), 1.5e-1)
    tr = P.ContrastiveTrainer(model, num_segments=2, output_llm_layer=3, train_mode=False, lr=2e-4)
    first = float(to_np(tr.step(b))[0])
    assert all(bool(torch.isfinite(g).all()) for g in tr.g)
    for _ in range(5):
        last = float(to_np(tr.step(b))[0])
    assert np.isfinite(last) and last < first


def test_cfg2_fp8_full_models_vs_fp8_oracle():
    """BASELINE.json configs[1] model sizes (esm2_t12_35M + Llama-3.2-1B, 12 + 16 layers) with fp8 GEMMs, 4-pair ragged
    slice, against the fp8 oracle on the GPU model's own weights."""
    import os
    import sys
    import p2t_hip as P
    from p2t_hip import specs, synth
    from gpu_util import build_model
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import GpuWeights
    esm_name, llama_name, _, _, Tp, Tt = specs.CONFIGS["cfg2"]
    esm, llama = specs.esm_spec(esm_name), specs.llama_spec(llama_name)
    ad = specs.adapter_spec(esm, llama)
    model = build_model(esm, llama, ad, torch.bfloat16, 0).eval().set_gemm_dtype("fp8")
    B = 4
    pid, pmask = synth.protein_batch(21, B, Tp, [Tp, 300, 131, 40])
    tid, tmask = synth.text_batch(21, B, Tt, 128000, [Tt, 77, 30, 18], 128002, 128009)
    b = _batch(pid, pmask, tid, tmask)
    layer = min(16, llama.num_hidden_layers)
    with torch.no_grad():
        p = P.l2_normalize(P.get_sequence_embeddings(model, b["protein_input_ids"], b["protein_attention_mask"]))
        t = P.l2_normalize(P.get_description_embeddings(model, b["description_input_ids"], b["description_attention_mask"], layer))
        loss = float(P.BatchInfoNCELoss()(p, t))
    ref = O.contrastive_step(esm, llama, GpuWeights(model), pid, pmask, tid, tmask, layer=layer, num_segments=1, prec=O.FP8)
    observe("cfg2.fp8_vs_fp8oracle.text", rel(to_np(t), ref["text"]), 1e-1)
    observe("cfg2.fp8_vs_fp8oracle.protein", rel(to_np(p), ref["protein"]), 1e-1)
    observe("cfg2.fp8_vs_fp8oracle.loss", abs(loss - float(ref["loss"])) / max(1.0, abs(float(ref["loss"]))), 1e-1, "abs/max(1,|ref|)")
    ref32 = O.contrastive_step(esm, llama, GpuWeights(model), pid, pmask, tid, tmask, layer=layer, num_segments=1)
    observe("cfg2.fp8_vs_fp32oracle.text", rel(to_np(t), ref32["text"]), 2e-1)
    observe("cfg2.fp8_vs_fp32oracle.protein", rel(to_np(p), ref32["protein"]), 2e-1)
    observe("cfg2.fp8_vs_fp32oracle.loss", abs(loss - float(ref32["loss"])) / max(1.0, abs(float(ref32["loss"]))), 2e-1, "abs/max(1,|ref|)")


@pytest.mark.parametrize("epi", [EPI_STORE, EPI_RESID, EPI_SWIGLU, 7])
# one tile per CU and two K steps of pairs (the minimum) / a partial round and ten steps / many rounds, long K
@pytest.mark.parametrize("shape", [(4096, 4096, 512), (5120, 4096, 2560), (16384, 2560, 1024), (4096, 4096, 14336)])
def test_gemm_fp8_four_wave_kernel_vs_per_tile_kernel(ops, epi, shape):
    """gemm_fp8_w4.hip (tile=4: persistent four-wave kernel) against the per-tile eight-wave kernel (tile=256) on the same
    quantised operands: both sum K in the same order per accumulator, so the results are BIT identical; the per-tile
    kernel itself is pinned on numpy and on the exact fp32-FMA kernel above."""
    M, N, K = shape
    a = torch.empty((M, K), dtype=torch.float32, device=dev())
    w = torch.empty((N, K), dtype=torch.float32, device=dev())
    ops.fill_hash_(a, 11, f"f8w4.a{M}x{K}", 1.0)
    ops.fill_hash_(w, 11, f"f8w4.w{N}x{K}", 0.05)
    a *= torch.exp(torch.linspace(-2.0, 2.0, M, device=dev()))[:, None]          # row scales that differ from row to row
    w *= torch.exp(torch.linspace(1.5, -1.5, N, device=dev()))[:, None]
    a8, sa = ops.quant_rows_fp8(a)
    w8, sw = ops.quant_rows_fp8(w)
    bias = None if epi == EPI_SWIGLU else to_dev(rnd(52, "f8w4.b", (N,), 0.3))
    outs = []
    for tile in (4, 256):
        kw = {}
        if epi == EPI_RESID:
            kw["out"] = torch.ones((M, N), dtype=torch.float32, device=dev())
        if epi == 7:
            kw["out_row_scale"] = torch.full((M,), 127 + 2, dtype=torch.uint8, device=dev())
        outs.append(ops.gemm_nt_fp8(a8, sa, w8, sw, bias, n=N, k=K, epilogue=epi, tile=tile, **kw))
    assert outs[0].shape == outs[1].shape and outs[0].dtype == outs[1].dtype
    assert torch.equal(outs[0], outs[1]), float((outs[0].float() - outs[1].float()).abs().max())
    assert bool(outs[0].float().abs().max() > 0)


def test_gemm_fp8_four_wave_kernel_refuses_ineligible_shapes(ops):
    a8, sa = ops.quant_rows_fp8(torch.ones((300, 512), device=dev()))
    w8, sw = ops.quant_rows_fp8(torch.ones((256, 512), device=dev()))
    with pytest.raises(Exception):
        ops.gemm_nt_fp8(a8, sa, w8, sw, None, n=256, k=512, epilogue=EPI_STORE, tile=4)        # edge tile in M
    ops.gemm_nt_fp8(a8, sa, w8, sw, None, n=256, k=512, epilogue=EPI_STORE, tile=0)            # auto: falls back to the per-tile kernel


def test_gemm_fp8_fuzz_vs_exact_kernel(ops):
    """Random whole-tile shapes (many tiles, 2 .. 11 K steps pairs, per-row scales that differ from row to row) through the
    fp8 kernel at both tile heights against the exact fp32-FMA kernel on the dequantised operands: a stale ring slot, a
    mis-counted wait or a wrong scale byte shows up as a wrong tile."""
    rng = np.random.default_rng(5)
    for it in range(8):
        tiles = int(rng.integers(256, 900))
        tm = int(rng.choice([d for d in range(4, 65) if tiles // d >= 4]))
        tn = max(4, tiles // tm)
        M, N, K = 256 * tm, 256 * tn, 128 * int(rng.integers(1, 24))
        a = torch.empty((M, K), dtype=torch.float32, device=dev())
        w = torch.empty((N, K), dtype=torch.float32, device=dev())
        ops.fill_hash_(a, 4, f"pf.a{M}x{K}", 1.0)
        ops.fill_hash_(w, 4, f"pf.w{N}x{K}", 0.5)
        a *= torch.exp2(torch.randint(-6, 7, (M, 1), device=dev()).float())
        w *= torch.exp2(torch.randint(-6, 7, (N, 1), device=dev()).float())
        a8, sa = ops.quant_rows_fp8(a)
        w8, sw = ops.quant_rows_fp8(w)
        da = a8.view(torch.float8_e4m3fn).float() * torch.exp2(sa.float() - 127)[:, None]
        dw = w8.view(torch.float8_e4m3fn).float() * torch.exp2(sw.float() - 127)[:, None]
        b = to_dev(rnd(3, "pf.b", (N,), 0.3))
        for epi in (EPI_STORE, EPI_RESID):
            ref = ops.gemm_nt(da, dw, b, epilogue=epi, out=torch.ones((M, N), dtype=torch.float32, device=dev()) if epi == EPI_RESID else None,
                              out_dtype=torch.float32, use_mfma=0)
            outs = []
            for tile in (0, 128, 256):
                got = ops.gemm_nt_fp8(a8, sa, w8, sw, b, n=N, k=K, epilogue=epi, out_dtype=torch.float32, tile=tile,
                                      out=torch.ones((M, N), dtype=torch.float32, device=dev()) if epi == EPI_RESID else None)
                err = float((got[:, :N] - ref[:, :N]).abs().max() / ref[:, :N].abs().max())
                assert err < 1e-4, (M, N, K, epi, tile, err)
                outs.append(got)
            assert torch.equal(outs[1][:, :N], outs[2][:, :N])          # both tile heights sum in the same order
