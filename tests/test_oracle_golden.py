"""CPU: the numpy oracle (oracle/p2t_oracle.py) against golden vectors produced by the reference
itself (tests/golden/make_golden.py).  Tolerances are fp32 round-off: both sides are fp32 CPU."""
import numpy as np
import pytest

from oracle import p2t_oracle as O
from helpers import case_setup, model_weights, rel_err

TOL = 2e-5     # relative L2, fp32 vs fp32 with different summation orders


def test_ops_readout_and_losses(golden):
    g = golden("ops")
    for ro in ("last", "mean", "std", "mix"):
        np.testing.assert_allclose(O.readout_embeddings(g["emb"], g["mask"], ro), g[f"readout_{ro}"],
                                   rtol=2e-5, atol=2e-6)
    assert abs(O.infonce_batch(g["p"], g["t"]) - g["loss_batch"]) < 1e-5
    assert abs(O.infonce_batch(g["p"], g["t"], 0.1) - g["loss_batch_t01"]) < 1e-5
    assert abs(O.infonce_segmented(g["p"][3:6], g["t"], np.array([3, 4, 5])) - g["loss_seg"]) < 1e-5
    # column (text -> protein) term == the reference's BatchInfoNCELoss with swapped arguments, values and autograd gradients
    loss_c, gcol = O.infonce_columns(g["p"], g["t"], None, return_grad=True)
    n = g["p"].shape[0]
    assert abs(loss_c - g["loss_batch_swapped"]) < 1e-5
    assert rel_err(gcol / n, g["grad_swapped_p"]) < 1e-5
    _, grow, _ = O.infonce_segmented(g["p"], g["t"], np.arange(n), return_grad=True)
    assert abs(0.5 * (O.infonce_batch(g["p"], g["t"]) + loss_c) - g["loss_symmetric"]) < 1e-5
    assert rel_err(0.5 * grow + 0.5 * gcol / n, g["grad_symmetric_p"]) < 1e-5
    assert abs(O.infonce_columns(g["p"], g["t"], [1, 4]) - np.mean([O.infonce_columns(g["p"], g["t"], [j]) for j in (1, 4)])) < 1e-6


@pytest.mark.parametrize("case", ["tiny", "tiny_d24", "tiny_d128", "tiny_d64", "tiny_qwen3"])
def test_towers_and_step(golden, case):
    g = golden(case)
    meta = g["meta"]
    esm, llama, ad, pid, pmask, tid, tmask = case_setup(meta)
    W = model_weights(esm, llama, ad, meta["seed_w"])
    enc = O.esm2_forward(esm, W, pid, pmask, prefix="esm_encoder.")
    assert rel_err(enc, g["esm_last_hidden"]) < TOL
    keep = {}
    ad_out = O.adapter_forward(W, enc, prefix="adapter.", keep=keep)
    assert rel_err(ad_out, g["adapter_out"]) < TOL
    for ro in ("last", "mean", "std", "mix"):
        assert rel_err(O.readout_embeddings(ad_out, pmask, ro), g[f"prot_pooled_{ro}"]) < TOL
    assert rel_err(O.readout_embeddings(ad_out, np.ones_like(pmask), "mix"), g["prot_pooled_mix_onesmask"]) < TOL
    for k in meta["layers"]:
        hs = O.llama_hidden_state(llama, W, tid, tmask, k, prefix="llama_decoder.")
        assert rel_err(hs, g[f"text_hidden_L{k}"]) < TOL
        t = O.text_embeddings(llama, W, tid, tmask, k, "mix")
        assert rel_err(t, g[f"text_norm_mix_L{k}"]) < TOL
        p = O.protein_embeddings(esm, W, pid, pmask, "mix")
        assert rel_err(p, g["prot_norm_mix"]) < TOL
        assert rel_err(p @ t.T / 0.05, g[f"logits_mix_L{k}"]) < TOL
        for nseg in (1, 2):
            assert abs(O.contrastive_loss(p, t, nseg) - g[f"loss_seg{nseg}_mix_L{k}"]) < 2e-5
        assert abs(O.infonce_batch(p, t) - g[f"loss_batch_mix_L{k}"]) < 2e-5
        p1 = O.protein_embeddings(esm, W, pid, pmask, "mix", ones_mask=True)
        assert abs(O.infonce_batch(p1, t) - g[f"loss_batch_mix_onesmask_L{k}"]) < 2e-5
        pm = O.protein_embeddings(esm, W, pid, pmask, "mean")
        tm = O.text_embeddings(llama, W, tid, tmask, k, "mean")
        assert abs(O.infonce_batch(pm, tm) - g[f"loss_batch_mean_L{k}"]) < 2e-5
    # adapter gradients through readout / normalise / loss (reference autograd)
    k = meta["layers"][-1]
    for nseg in (1, 2):
        out = O.contrastive_step(esm, llama, W, pid, pmask, tid, tmask, layer=k, num_segments=nseg, with_grads=True)
        for n in ("fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias"):
            assert rel_err(out["grads"]["adapter." + n], g[f"grad_seg{nseg}_{n}"]) < 1e-4, n


@pytest.mark.parametrize("case", ["tiny", "tiny_d24", "tiny_d128", "tiny_d64", "tiny_qwen3"])
def test_clip_adamw_step(golden, case):
    g = golden(case)
    meta = g["meta"]
    esm, llama, ad, pid, pmask, tid, tmask = case_setup(meta)
    W = model_weights(esm, llama, ad, meta["seed_w"])
    out = O.contrastive_step(esm, llama, W, pid, pmask, tid, tmask, layer=meta["layers"][-1], with_grads=True)
    names = ["adapter." + n for n in ("fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias")]
    params = {n: W[n].copy() for n in names}
    m = {n: np.zeros_like(params[n]) for n in names}
    v = {n: np.zeros_like(params[n]) for n in names}
    gn = O.clip_and_adamw(params, out["grads"], m, v, step=1, max_norm=0.05)
    assert abs(gn - g["opt_gradnorm"]) < 1e-4 * max(1.0, float(g["opt_gradnorm"]))
    for n in names:
        np.testing.assert_allclose(params[n], g["opt_after_" + n[len("adapter."):]], rtol=1e-5, atol=2e-7)


def test_cfg1_shapes_against_reference(golden):
    """BASELINE.json configs[0]: esm2_t6_8M + Llama-3.2-1B shapes, B=4, T=128/64 (the reference's own
    CPU-runnable case).  Weights (1.2 B parameters) are regenerated layer by layer from the hash generator."""
    g = golden("cfg1")
    meta = g["meta"]
    esm, llama, ad, pid, pmask, tid, tmask = case_setup(meta)
    W = model_weights(esm, llama, ad, meta["seed_w"], cache=False)
    out = O.contrastive_step(esm, llama, W, pid, pmask, tid, tmask, layer=16, num_segments=2, with_grads=True)
    assert rel_err(out["protein"], g["prot_norm_mix"]) < 5e-5
    assert rel_err(out["text"], g["text_norm_mix_L16"]) < 5e-5
    assert abs(out["loss"] - g["loss_seg2_mix_L16"]) < 5e-5
    for n in ("fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias"):
        got = out["grads"]["adapter." + n]
        if got.ndim == 2:
            assert abs(np.linalg.norm(got.astype(np.float64)) - g[f"gradnorm_seg2_{n}"]) < 1e-3 * g[f"gradnorm_seg2_{n}"]
            got = got[::7, ::5]
        assert rel_err(got, g[f"grad_seg2_{n}"]) < 1e-3, n


@pytest.mark.parametrize("case", ["d16", "d64", "d128"])
def test_sft_backward_oracle_vs_reference_autograd(case):
    """Stage-2 step (oracle.sft_step: LM loss through the frozen decoder, gradient back to the adapter) against torch autograd
    through the reference class (tests/golden/sft_grad_tiny.npz, make_golden.py run_sft_backward)."""
    from oracle import p2t_oracle as O
    from helpers import model_weights, rel_err
    from p2t_hip import specs
    from conftest import load_golden
    g = load_golden("sft_grad_tiny")
    m = g["meta"]["cases"][case]
    esm, llama, ad = specs.EsmSpec(**m["esm"]), specs.LlamaSpec(**m["llama"]), specs.AdapterSpec(**m["adapter"])
    W = model_weights(esm, llama, ad, 0, lm_head=True)
    out = O.sft_step(esm, llama, W, g["protein_input_ids"], g["protein_attention_mask"], g["input_ids"], g["attention_mask"], g["labels"],
                     g["meta"]["placeholder_id"])
    assert abs(float(out["loss"]) - float(g[f"{case}.loss"])) < 2e-5 * max(1.0, abs(float(g[f"{case}.loss"])))
    valid = g["attention_mask"] != 0          # padded positions: the reference back-propagates through a uniform softmax row there
    assert rel_err(out["d_inputs_embeds"][valid], g[f"{case}.d_inputs_embeds"][valid]) < 2e-4
    for n in ("fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias"):
        assert rel_err(out["grads"]["adapter." + n], g[f"{case}.grad.{n}"]) < 5e-4, n


@pytest.mark.parametrize("case", ["d16", "d64", "d128", "qwen3"])
def test_generate_oracle_vs_reference_generate(case):
    """oracle.generate_greedy / generate_beam (cache-less restatement of HF `_sample` / `_beam_search` under the reference's
    `generate`, models/modeling_esm2llama_instruct.py:217-251) against `model.generate` of the reference class itself
    (tests/golden/generate_tiny.npz, make_golden.py run_generate): left-padded prompts with protein placeholders; greedy ids exact
    (the fixture's smallest top-2 logit gap is >= 5e-3) with per-step logits to 2e-4, the finished-row padding rule under an eos id,
    and beam search (3 beams, two length penalties): ids exact, scores to 1e-4."""
    from oracle import p2t_oracle as O
    from helpers import model_weights
    from p2t_hip import specs
    from conftest import load_golden
    g = load_golden("generate_tiny")
    meta = g["meta"]
    m = meta["cases"][case]
    esm, llama, ad = specs.EsmSpec(**m["esm"]), specs.LlamaSpec(**m["llama"]), specs.AdapterSpec(**m["adapter"])
    W = model_weights(esm, llama, ad, m["weight_seed"], lm_head=True)
    enc = O.esm2_forward(esm, W, g["protein_input_ids"], g["protein_attention_mask"], O.FP32, prefix="esm_encoder.")
    emb = O.sft_decoder_inputs(llama, W, g["input_ids"], O.adapter_forward(W, enc, O.FP32, prefix="adapter."), g["protein_attention_mask"],
                               meta["placeholder_id"])
    n, pad, eos = meta["max_new_tokens"], meta["pad_id"], m["eos"]
    toks, logits = O.generate_greedy(llama, W, emb, g["attention_mask"], n, (), pad)
    assert np.array_equal(toks, g[f"{case}.greedy"])
    assert np.abs(logits - g[f"{case}.greedy_logits"]).max() < 2e-4
    toks_e, _ = O.generate_greedy(llama, W, emb, g["attention_mask"], n, (eos,), pad)
    assert np.array_equal(toks_e, g[f"{case}.greedy_eos"])
    for lp in ("1.0", "0.5"):
        seq, sc = O.generate_beam(llama, W, emb, g["attention_mask"], n, 3, (eos,), pad, float(lp))
        assert np.array_equal(seq, g[f"{case}.beam3_lp{lp}"]), lp
        assert np.abs(sc - g[f"{case}.beam3_lp{lp}_scores"]).max() < 1e-4, lp
    # two eos ids and a `max_length` budget (HF counts the padded prompt in it), then early_stopping=True / "never" with two
    # returned hypotheses per prompt
    toks2, _ = O.generate_greedy(llama, W, emb, g["attention_mask"], n - 3, tuple(m["eos2"]), pad)
    assert np.array_equal(toks2, g[f"{case}.greedy_eos2_maxlen"])
    for es in (True, "never"):
        seq, sc = O.generate_beam(llama, W, emb, g["attention_mask"], n, 3, tuple(m["eos2"]), pad, 0.8, early_stopping=es, num_return_sequences=2)
        assert np.array_equal(seq, g[f"{case}.beam3_es{es}"]), es
        assert np.abs(sc - g[f"{case}.beam3_es{es}_scores"]).max() < 1e-4, es
