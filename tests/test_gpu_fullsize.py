"""Parity at BASELINE.json's own model sizes.

* configs[1] (esm2_t12_35M + Llama-3.2-1B, bf16): the HIP path against the CPU oracle on the FULL models (12 + 16
  layers, head_dim 24 encoder, 128256-token vocabulary), a 4-pair ragged slice of the batch so the oracle finishes in
  seconds; both sides read the same weights (the GPU model's).
* configs[2] (esm2_t36_3B + Llama-3.1-8B, bf16, 1024 residues) and configs[4]'s one-GPU share (the same models, fp8 tower
  GEMMs): two ragged pairs against the CPU oracle at full depth (~25 s of CPU per precision), then -- at batch sizes the oracle
  cannot reach -- size-independent properties of the step: permutation equivariance, padding invariance, segment additivity, agreement of the bf16 MFMA pipeline with
  the exact fp32 pipeline on the same weights, and a falling loss under the fused optimizer step.
"""
import os
import sys

import numpy as np
import pytest
import torch

from gpu_util import build_model, dev, observe, rel, to_dev, to_np
from p2t_hip import specs, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_ORACLE = {}                             # oracle outputs (and the downloaded weights) of the cfg3 pins, per session


def _batch(pid, pmask, tid, tmask):
    return dict(protein_input_ids=to_dev(pid), protein_attention_mask=to_dev(pmask),
                description_input_ids=to_dev(tid), description_attention_mask=to_dev(tmask))


def _embeddings(P, model, b, layer=16):
    with torch.no_grad():
        p = P.l2_normalize(P.get_sequence_embeddings(model, b["protein_input_ids"], b["protein_attention_mask"]))
        t = P.l2_normalize(P.get_description_embeddings(model, b["description_input_ids"], b["description_attention_mask"], layer))
    return to_np(p), to_np(t)


def test_cfg2_full_models_vs_cpu_oracle():
    import p2t_hip as P
    from oracle import p2t_oracle as O
    sys.path.insert(0, ROOT)
    from bench import GpuWeights
    esm_name, llama_name, _, _, Tp, Tt = specs.CONFIGS["cfg2"]
    esm, llama = specs.esm_spec(esm_name), specs.llama_spec(llama_name)
    ad = specs.adapter_spec(esm, llama)
    model = build_model(esm, llama, ad, torch.bfloat16, 0).eval()
    B = 4
    pid, pmask = synth.protein_batch(21, B, Tp, [Tp, 300, 131, 40])
    tid, tmask = synth.text_batch(21, B, Tt, 128000, [Tt, 77, 30, 18], 128002, 128009)
    b = _batch(pid, pmask, tid, tmask)
    layer = min(16, llama.num_hidden_layers)
    with torch.no_grad():
        p = P.l2_normalize(P.get_sequence_embeddings(model, b["protein_input_ids"], b["protein_attention_mask"]))
        t = P.l2_normalize(P.get_description_embeddings(model, b["description_input_ids"], b["description_attention_mask"], layer))
        loss = float(P.BatchInfoNCELoss()(p, t))
    ref = O.contrastive_step(esm, llama, GpuWeights(model), pid, pmask, tid, tmask, layer=layer, num_segments=1, prec=O.BF16)
    observe("cfg2.bf16_vs_bf16oracle.text", rel(to_np(t), ref["text"]), 2e-2)
    observe("cfg2.bf16_vs_bf16oracle.protein", rel(to_np(p), ref["protein"]), 2e-2)
    observe("cfg2.bf16_vs_bf16oracle.loss", abs(loss - float(ref["loss"])) / max(1.0, abs(float(ref["loss"]))), 2e-2, "abs/max(1,|ref|)")
    # the same slice against the fp32 oracle (= the reference's CPU arithmetic): the bf16-vs-reference figure BASELINE.md quotes
    ref32 = O.contrastive_step(esm, llama, GpuWeights(model), pid, pmask, tid, tmask, layer=layer, num_segments=1)
    observe("cfg2.bf16_vs_fp32oracle.text", rel(to_np(t), ref32["text"]), 3e-2)
    observe("cfg2.bf16_vs_fp32oracle.protein", rel(to_np(p), ref32["protein"]), 3e-2)
    observe("cfg2.bf16_vs_fp32oracle.loss", abs(loss - float(ref32["loss"])) / max(1.0, abs(float(ref32["loss"]))), 3e-2, "abs/max(1,|ref|)")


@pytest.fixture(scope="module")
def cfg3():
    import p2t_hip as P
    esm_name, llama_name, _, _, Tp, Tt = specs.CONFIGS["cfg3"]
    esm, llama = specs.esm_spec(esm_name), specs.llama_spec(llama_name)
    ad = specs.adapter_spec(esm, llama)
    model = build_model(esm, llama, ad, torch.bfloat16, 0).eval()
    B = 4
    lens_p, lens_t = [600, 411, 250, 37], [100, 64, 33, 9]
    pid, pmask = synth.protein_batch(31, B, Tp, lens_p)
    tid, tmask = synth.text_batch(31, B, Tt, 128000, lens_t, 128002, 128009)
    yield dict(P=P, model=model, esm=esm, llama=llama, ad=ad, Tp=Tp, Tt=Tt, pid=pid, pmask=pmask, tid=tid, tmask=tmask)
    _ORACLE.clear()                     # (the cached weight view holds the parameters)
    del model
    torch.cuda.empty_cache()


def _cfg3_oracle_pairs(cfg3):
    """Two ragged pairs at the benchmarked sizes: a full-length protein (1024 residues) beside a 411-residue one, a
    full-length description (128 tokens) beside a 33-token one."""
    Tp, Tt = cfg3["Tp"], cfg3["Tt"]
    pid, pmask = synth.protein_batch(51, 2, Tp, [Tp, 411])
    tid, tmask = synth.text_batch(51, 2, Tt, 128000, [Tt, 33], 128002, 128009)
    return pid, pmask, tid, tmask


def _cfg3_oracle(cfg3, name, prec):
    """The oracle's contrastive step on the two pairs with the GPU model's weights (each precision computed once per session).
    Every pair is encoded at its OWN length -- rows never interact and padding changes nothing (the padding-invariance test below
    checks that on the GPU side) -- so the 411-residue pair costs a sixth of the 1024-residue one instead of the same."""
    if name not in _ORACLE:
        from oracle import p2t_oracle as O
        sys.path.insert(0, ROOT)
        from bench import GpuWeights
        pid, pmask, tid, tmask = _cfg3_oracle_pairs(cfg3)
        if "W" not in _ORACLE:
            _ORACLE["W"] = GpuWeights(cfg3["model"], cache=True)        # downloaded once for all precisions and both pairs
        W = _ORACLE["W"]
        if prec.kind == "fp8":
            # one padded call: the fp8 oracle quantises every weight matrix per call (the dominant cost), so the pairs share it
            _ORACLE[name] = O.contrastive_step(cfg3["esm"], cfg3["llama"], W, pid, pmask, tid, tmask, layer=16, num_segments=1, prec=prec)
            return _ORACLE[name]
        ps, ts = [], []
        for i in range(pid.shape[0]):
            n_p, n_t = int(pmask[i].sum()), int(tmask[i].sum())
            ps.append(O.protein_embeddings(cfg3["esm"], W, pid[i:i + 1, :n_p], pmask[i:i + 1, :n_p], "mix", False, prec))
            ts.append(O.text_embeddings(cfg3["llama"], W, tid[i:i + 1, :n_t], tmask[i:i + 1, :n_t], 16, "mix", prec))
        p, t = np.concatenate(ps, 0), np.concatenate(ts, 0)
        _ORACLE[name] = {"protein": p, "text": t, "loss": O.contrastive_loss(p, t, 1)}
    return _ORACLE[name]


def _step_outputs(P, model, b):
    with torch.no_grad():
        p = P.l2_normalize(P.get_sequence_embeddings(model, b["protein_input_ids"], b["protein_attention_mask"]))
        t = P.l2_normalize(P.get_description_embeddings(model, b["description_input_ids"], b["description_attention_mask"], 16))
        loss = float(P.BatchInfoNCELoss()(p, t))
    return to_np(p), to_np(t), loss


def test_cfg3_full_models_vs_cpu_oracle(cfg3):
    """BASELINE.json configs[2] -- the configuration bench.py times -- against the CPU oracle (the restatement of
    scripts/train_contrast.py:284-310,345-379 that tests/test_oracle_golden.py pins on the reference's own outputs): the full
    36-layer esm2_t36_3B + 16 layers of Llama-3.1-8B, same weights on both sides (the GPU model's), pooled embeddings and loss.
    Once against the oracle that rounds where the HIP path stores bf16, once against the fp32 oracle (= the reference's CPU
    arithmetic)."""
    from oracle import p2t_oracle as O
    P, model = cfg3["P"], cfg3["model"]
    pid, pmask, tid, tmask = _cfg3_oracle_pairs(cfg3)
    p, t, loss = _step_outputs(P, model, _batch(pid, pmask, tid, tmask))
    for name, prec, caps in (("bf16oracle", O.BF16, (1e-2, 1e-2)), ("fp32oracle", O.FP32, (1e-2, 3e-2))):
        ref = _cfg3_oracle(cfg3, name, prec)
        observe(f"cfg3.bf16_vs_{name}.text", rel(t, ref["text"]), caps[0])
        observe(f"cfg3.bf16_vs_{name}.protein", rel(p, ref["protein"]), caps[0])
        observe(f"cfg3.bf16_vs_{name}.loss", abs(loss - float(ref["loss"])) / max(1.0, abs(float(ref["loss"]))), caps[1], "abs/max(1,|ref|)")


def test_cfg5_fp8_full_models_vs_cpu_oracle(cfg3):
    """BASELINE.json configs[4] (one GPU's share): the cfg3 models with fp8 tower GEMMs against the oracle that quantises the
    same operands the same way (O.FP8) and against the fp32 oracle."""
    from oracle import p2t_oracle as O
    P, model = cfg3["P"], cfg3["model"]
    pid, pmask, tid, tmask = _cfg3_oracle_pairs(cfg3)
    try:
        model.set_gemm_dtype("fp8")
        p, t, loss = _step_outputs(P, model, _batch(pid, pmask, tid, tmask))
    finally:
        model.set_gemm_dtype("model")
    for name, prec, caps in (("fp8oracle", O.FP8, (1e-1, 1e-1)), ("fp32oracle", O.FP32, (2e-1, 2e-1))):
        ref = _cfg3_oracle(cfg3, name, prec)
        observe(f"cfg5.fp8_vs_{name}.text", rel(t, ref["text"]), caps[0])
        observe(f"cfg5.fp8_vs_{name}.protein", rel(p, ref["protein"]), caps[0])
        observe(f"cfg5.fp8_vs_{name}.loss", abs(loss - float(ref["loss"])) / max(1.0, abs(float(ref["loss"]))), caps[1], "abs/max(1,|ref|)")


def test_cfg3_permutation_equivariance_and_padding_invariance(cfg3):
    P, model = cfg3["P"], cfg3["model"]
    pid, pmask, tid, tmask = cfg3["pid"], cfg3["pmask"], cfg3["tid"], cfg3["tmask"]
    mix = lambda b: (to_np(P.l2_normalize(P.readout_embeddings(
        model(protein_input_ids=b["protein_input_ids"], protein_attention_mask=b["protein_attention_mask"], return_adapter_outputs=True)[0],
        b["protein_attention_mask"], "mix"))),
        to_np(P.l2_normalize(P.get_description_embeddings(model, b["description_input_ids"], b["description_attention_mask"], 16))))
    with torch.no_grad():
        p0, t0 = mix(_batch(pid, pmask, tid, tmask))
        perm = [2, 0, 3, 1]
        p1, t1 = mix(_batch(pid[perm], pmask[perm], tid[perm], tmask[perm]))
        assert rel(p1, p0[perm]) < 1e-3 and rel(t1, t0[perm]) < 1e-3           # rows do not see each other
        # every sequence fits in 640 residues / 104 tokens: dropping the all-padding tail changes nothing
        p2, t2 = mix(_batch(pid[:, :640], pmask[:, :640], tid[:, :104], tmask[:, :104]))
        assert rel(p2, p0) < 1e-3 and rel(t2, t0) < 1e-3
    assert np.isfinite(p0).all() and np.isfinite(t0).all()
    assert np.allclose(np.linalg.norm(p0, axis=1), 1.0, atol=1e-3)


def test_cfg3_segments_and_fused_trainer(cfg3):
    P, model = cfg3["P"], cfg3["model"]
    b = _batch(cfg3["pid"], cfg3["pmask"], cfg3["tid"], cfg3["tmask"])
    model.train()
    model.adapter.dropout.p = 0.0
    losses = {}
    grads = {}
    for nseg in (1, 2, 4):
        tr = P.ContrastiveTrainer(model, num_segments=nseg, train_mode=False, lr=1e-3)
        losses[nseg] = float(to_np(tr.forward_backward(b))[0])
        grads[nseg] = to_np(tr.flat_g).copy()
    # the segmented loss is the mean of per-segment means of equally sized segments = the batch mean; gradients add up
    assert abs(losses[2] - losses[1]) < 1e-4 * max(1.0, abs(losses[1])) and abs(losses[4] - losses[1]) < 1e-4 * max(1.0, abs(losses[1]))
    assert rel(grads[2], grads[1]) < 2e-2 and rel(grads[4], grads[1]) < 2e-2
    assert 0.0 < losses[1] < 2.0 * np.log(4) + 1.0
    # reference hyper-parameters (lr 2e-4): ten times that collapses a feature of a sequence to a constant after one step
    # and the eps-free std readout then back-propagates 0/0, exactly as the reference's would
    tr = P.ContrastiveTrainer(model, num_segments=1, train_mode=False, lr=2e-4)
    first = float(to_np(tr.step(b))[0])
    for _ in range(5):
        last = float(to_np(tr.step(b))[0])
    assert np.isfinite(last) and last < 0.6 * first                             # the fused step optimises the loss it reports
    model.eval()


def test_cfg3_bf16_pipeline_vs_exact_fp32_pipeline(cfg3):
    """Same synthetic weights (fp32 masters -> the bf16 model holds their roundings), two pairs, full depth."""
    P = cfg3["P"]
    m32 = build_model(cfg3["esm"], cfg3["llama"], cfg3["ad"], torch.float32, 0).eval()
    b = _batch(cfg3["pid"][:2], cfg3["pmask"][:2], cfg3["tid"][:2], cfg3["tmask"][:2])
    p16, t16 = _embeddings(P, cfg3["model"], b)
    p32, t32 = _embeddings(P, m32, b)
    del m32
    torch.cuda.empty_cache()
    observe("cfg3.bf16_vs_fp32pipeline.text", rel(t16, t32), 3e-2)
    observe("cfg3.bf16_vs_fp32pipeline.protein", rel(p16, p32), 5e-2)             # 36 layers of bf16 storage


def test_cfg3_fp32_pipeline_vs_fp32_oracle_at_north_star_tolerance(cfg3):
    """BASELINE.json north_star: "pooled embeddings, loss match the reference CPU path within 1e-3 relative fp32 tolerance" --
    asserted HERE at configs[2]'s own size with an INLINE bound (no tolerance table): the fp32 HIP pipeline (full esm2_t36_3B +
    16 layers of Llama-3.1-8B on the fp32-FMA kernels) against the fp32 oracle on the same fp32 weights, two ragged pairs
    (1024 + 411 residues / 128 + 33 tokens).  The bf16 pipeline cannot meet 1e-3 (bf16 storage: DESIGN.md section 6 holds its
    observed error and where it comes from); the fp32 path is what carries the claim at this size.
    Reference arithmetic: scripts/train_contrast.py:284-310,345-379."""
    from oracle import p2t_oracle as O
    sys.path.insert(0, ROOT)
    from bench import GpuWeights
    P = cfg3["P"]
    m32 = build_model(cfg3["esm"], cfg3["llama"], cfg3["ad"], torch.float32, 0).eval()
    pid, pmask, tid, tmask = _cfg3_oracle_pairs(cfg3)
    p, t, loss = _step_outputs(P, m32, _batch(pid, pmask, tid, tmask))
    W = GpuWeights(m32, cache=True)
    ps, ts = [], []
    for i in range(pid.shape[0]):                       # every pair at its own length (padding changes nothing: tested above)
        n_p, n_t = int(pmask[i].sum()), int(tmask[i].sum())
        ps.append(O.protein_embeddings(cfg3["esm"], W, pid[i:i + 1, :n_p], pmask[i:i + 1, :n_p], "mix", False, O.FP32))
        ts.append(O.text_embeddings(cfg3["llama"], W, tid[i:i + 1, :n_t], tmask[i:i + 1, :n_t], 16, "mix", O.FP32))
    rp, rt = np.concatenate(ps, 0), np.concatenate(ts, 0)
    rl = float(O.contrastive_loss(rp, rt, 1))
    del W, m32
    torch.cuda.empty_cache()
    ep, et, el = rel(p, rp), rel(t, rt), abs(loss - rl) / max(1.0, abs(rl))
    print(f"cfg3 fp32 HIP vs fp32 oracle: protein {ep:.2e} text {et:.2e} loss {el:.2e}")
    observe("cfg3.fp32_vs_fp32oracle.protein", ep, 1e-3)
    observe("cfg3.fp32_vs_fp32oracle.text", et, 1e-3)
    observe("cfg3.fp32_vs_fp32oracle.loss", el, 1e-3, "abs/max(1,|ref|)")
    assert ep < 1e-3 and et < 1e-3 and el < 1e-3, (ep, et, el)          # north_star's bound, whatever the table says


def test_cfg3_batch_size_invariance_across_kernel_forms(cfg3):
    """The same sequences inside a large batch (persistent GEMM kernels: 16 x 1024 residues, 64 x 128 tokens) and inside a
    small one (per-tile kernels) give the same embeddings: rows never interact, whichever launch form computes them."""
    P, model = cfg3["P"], cfg3["model"]
    Tp, Tt = cfg3["Tp"], cfg3["Tt"]
    pid, pmask = synth.protein_batch(41, 16, Tp, [Tp - 7 * i for i in range(16)])
    tid, tmask = synth.text_batch(41, 64, Tt, 128000, [Tt - (i % 50) for i in range(64)], 128002, 128009)
    with torch.no_grad():
        pe = lambda i, m: to_np(P.l2_normalize(P.get_sequence_embeddings(model, to_dev(i), to_dev(m))))
        te = lambda i, m: to_np(P.l2_normalize(P.get_description_embeddings(model, to_dev(i), to_dev(m), 16)))
        p_big, t_big = pe(pid, pmask), te(tid, tmask)
        p_small, t_small = pe(pid[5:7], pmask[5:7]), te(tid[20:24], tmask[20:24])
    assert rel(p_big[5:7], p_small) < 1e-3
    assert rel(t_big[20:24], t_small) < 1e-3


def test_cfg3_four_wave_kernels_are_bit_identical_to_the_eight_wave_forms(cfg3):
    """The whole cfg3 forward (16 x 1024 residues: every encoder GEMM on the persistent four-wave kernel, FFN-down with in-kernel
    split-K pairs; 64 x 128 tokens: text-tower QKV one tile per block, FFN-down as pairs on every tile) under the default policy
    and with the four-wave forms switched off (policy 9): every accumulator sums its K chunks in the same order in both, so the
    pooled embeddings are BIT identical -- any stale LDS read, mis-counted wait or wrong epilogue in the new kernels shows here,
    in the shapes and epilogues the benchmark runs.  Then the same with fp8 tower GEMMs (gemm_fp8_w4.hip vs gemm_fp8.hip)."""
    from p2t_hip import _lib
    P, model = cfg3["P"], cfg3["model"]
    Tp, Tt = cfg3["Tp"], cfg3["Tt"]
    pid, pmask = synth.protein_batch(43, 16, Tp, [Tp - 11 * i for i in range(16)])
    tid, tmask = synth.text_batch(43, 64, Tt, 128000, [Tt - (i % 40) for i in range(64)], 128002, 128009)

    def both():
        with torch.no_grad():
            return (P.l2_normalize(P.get_sequence_embeddings(model, to_dev(pid), to_dev(pmask))).clone(),
                    P.l2_normalize(P.get_description_embeddings(model, to_dev(tid), to_dev(tmask), 16)).clone())
    try:
        for gemm_dtype in ("model", "fp8"):
            model.set_gemm_dtype(gemm_dtype)
            _lib.call("p2t_set_gemm_policy", 0)
            p0, t0 = both()
            _lib.call("p2t_set_gemm_policy", 9)
            p9, t9 = both()
            assert bool(torch.isfinite(p0).all()) and bool(torch.isfinite(t0).all())
            assert torch.equal(p0, p9), (gemm_dtype, float((p0 - p9).abs().max()))
            assert torch.equal(t0, t9), (gemm_dtype, float((t0 - t9).abs().max()))
    finally:
        _lib.call("p2t_set_gemm_policy", 0)
        model.set_gemm_dtype("model")


def test_cfg3_ragged_batch_trimmed_segments_equal_padded_step(cfg3):
    """16 pairs with log-normal lengths at full model size: the length-sorted step with per-segment padded lengths
    (persistent GEMMs with half-tile / split-K tails on odd row counts, text tower in its own length order) reproduces the
    padded step's loss and adapter gradients up to bf16 accumulation-order noise."""
    from p2t_hip.data import sort_batch_by_length
    P, model, Tp, Tt = cfg3["P"], cfg3["model"], cfg3["Tp"], cfg3["Tt"]
    rs = np.random.RandomState(3)
    B = 16
    lens = np.clip(np.round(rs.lognormal(5.75, 0.6, B)), 16, Tp).astype(int).tolist()
    tl = np.clip(np.round(rs.lognormal(4.0, 0.5, B)), 4, Tt).astype(int).tolist()
    pid, pmask = synth.protein_batch(77, B, Tp, lens)
    tid, tmask = synth.text_batch(77, B, Tt, 128000, tl, 128002, 128009)
    Tmax, Ttmax = max(lens), max(tl)
    host = dict(protein_input_ids=torch.from_numpy(pid[:, :Tmax].copy()), protein_attention_mask=torch.from_numpy(pmask[:, :Tmax].copy()),
                description_input_ids=torch.from_numpy(tid[:, :Ttmax].copy()), description_attention_mask=torch.from_numpy(tmask[:, :Ttmax].copy()))
    to_cuda = lambda b: {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in b.items()}
    plain = P.ContrastiveTrainer(model, train_mode=False)
    l0 = float(to_np(plain.forward_backward(to_cuda(host)))[0])
    g0 = to_np(plain.flat_g).copy()
    del plain
    trim = P.ContrastiveTrainer(model, train_mode=False, trim_padding=True, trim_floor_tokens=2048)
    srt = to_cuda(sort_batch_by_length(host))
    segs = trim._segments(srt, B, Tmax)
    assert len(segs) >= 2 and sum((b - a) * t for a, b, t, _ in segs) < 0.8 * B * Tmax
    l1 = float(to_np(trim.forward_backward(srt))[0])
    g1 = to_np(trim.flat_g).copy()
    assert np.isfinite(l0)
    observe("cfg3.ragged_trim_vs_padded.loss", abs(l1 - l0) / max(1.0, abs(l0)), 2e-2, "abs/max(1,|ref|)")
    observe("cfg3.ragged_trim_vs_padded.grads", rel(g1, g0), 6e-2)
    torch.cuda.empty_cache()


def test_cfg3_generate_cached_steps_equal_the_cacheless_forward(cfg3):
    """`generate` at full model size (ESM2-3B -> adapter -> placeholder scatter -> all 32 layers of Llama-3.1-8B): the ids come back
    as the reference returns them, and the logits of every cached decode step equal the model's own cache-LESS forward over the
    compacted prompt + the generated ids -- the size-independent property that pins the KV-cache path (prefill cache write, stream
    copies of the weights, fused rotation + append, single-query attention, graph replay) where the CPU oracle cannot go; the forward
    it is compared with is pinned on the oracle by the tests above and tests/test_gpu_sft_forward.py."""
    model, Tp = cfg3["model"], cfg3["Tp"]
    model.set_gemm_dtype("model")
    B, n_prompt, n_new = 3, 24, 6
    lens = [Tp, 411, 37]
    pid, pmask = synth.protein_batch(91, B, Tp, lens)
    ph = model.config.placeholder_id
    rs = np.random.RandomState(5)
    T = Tp + n_prompt
    ids = np.full((B, T), 128002, dtype=np.int64)
    mask = np.zeros((B, T), dtype=np.int64)
    for b, n in enumerate(lens):                       # left padded, as the reference's collater builds its prompts
        row = np.concatenate([rs.randint(0, 128000, 8), np.full(n, ph), rs.randint(0, 128000, n_prompt - 8)])
        ids[b, T - len(row):] = row
        mask[b, T - len(row):] = 1
    kw = dict(inputs=to_dev(ids), attention_mask=to_dev(mask), protein_input_ids=to_dev(pid), protein_attention_mask=to_dev(pmask))
    out = model.generate(**kw, max_new_tokens=n_new, eos_token_id=None, pad_token_id=128002, do_sample=False, return_dict_in_generate=True,
                         output_logits=True)
    toks, lg = to_np(out.sequences), to_np(torch.stack(out.logits, 0))
    assert toks.shape == (B, n_new) and np.isfinite(lg).all()
    eager = model.generate(**kw, max_new_tokens=n_new, eos_token_id=None, pad_token_id=128002, do_sample=False, use_graph=False, stream_copy=False)
    assert np.array_equal(to_np(eager), toks)           # graph replay + stream copies == eager launches on the row-major weights
    emb, m2 = model(input_ids=kw["inputs"], attention_mask=kw["attention_mask"], protein_input_ids=kw["protein_input_ids"],
                    protein_attention_mask=kw["protein_attention_mask"], return_decoder_inputs=True)
    dec = model.llama_decoder
    for b in range(B):
        valid = torch.from_numpy(mask[b] != 0).to(emb.device)
        row = torch.cat([emb[b][valid], dec.model.embed(to_dev(toks[b:b + 1, :-1]))[0]], 0)[None]
        full = to_np(dec(inputs_embeds=row).logits.float())[0]
        n0 = int(mask[b].sum())
        observe(f"cfg3.generate_cache_vs_full_forward.row{b}", rel(lg[:, b], full[n0 - 1:n0 - 1 + n_new]), 3e-2)
    torch.cuda.empty_cache()
