"""Edge shapes of the contrastive step against the oracle (fp32 towers, eval mode): the smallest legal inputs (<cls><eos>
proteins, one-token descriptions, a batch of one pair), lengths that are not multiples of any tile (65, 130, 7, 1), rows
of very different lengths in one batch, and every readout mode.  Loss and adapter gradients through ContrastiveTrainer
(the C ABI: p2t_esm2_forward, p2t_llama_hidden_forward, p2t_adapter_*, p2t_readout*, p2t_infonce_*)."""
import numpy as np
import pytest
import torch

from oracle import p2t_oracle as O
from helpers import case_setup, model_weights
from gpu_util import build_model, observe, rel, to_dev, to_np
from p2t_hip import synth

pytestmark = pytest.mark.gpu
NAMES = ("adapter.fc1.weight", "adapter.fc1.bias", "adapter.fc2.weight", "adapter.fc2.bias")

SHAPES = [
    # B, T_p, protein lengths, T_t, text lengths
    (1, 2, [2], 1, [1]),                           # one pair, <cls><eos>, one text token: loss exactly 0, zero gradients
    (3, 65, [65, 2, 33], 7, [7, 1, 3]),
    (2, 130, [130, 129], 2, [2, 1]),
    (5, 31, [31, 30, 2, 17, 9], 19, [19, 2, 1, 18, 10]),
]


# "tiny": encoder head_dim 16 (unfused rotary pass); "tiny_d64": encoder head_dim 64 = the fused QKV epilogue of the timed path
@pytest.mark.parametrize("case", ["tiny", "tiny_d64"])
@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: f"B{s[0]}_Tp{s[1]}_Tt{s[3]}")
@pytest.mark.parametrize("readout", ["mix", "mean", "last"])
def test_edge_shapes_match_oracle(golden, shape, readout, case):
    import p2t_hip as P
    B, Tp, plens, Tt, tlens = shape
    if case != "tiny" and readout != "mix":
        pytest.skip("readout modes are covered on the first case")
    meta = golden(case)["meta"]
    esm, llama, ad, *_ = case_setup(meta)
    k = meta["layers"][-1]
    pid, pmask = synth.protein_batch(41, B, Tp, plens)
    tid, tmask = synth.text_batch(41, B, Tt, meta["id_high"], tlens, meta["pad_id"], meta["eos_id"])
    W = model_weights(esm, llama, ad, meta["seed_w"])
    ref = O.contrastive_step(esm, llama, W, pid, pmask, tid, tmask, layer=k, readout=readout, with_grads=True)
    model = build_model(esm, llama, ad, torch.float32, meta["seed_w"]).eval()
    tr = P.ContrastiveTrainer(model, output_llm_layer=k, readout_fn=readout, train_mode=False)
    batch = dict(protein_input_ids=to_dev(pid), protein_attention_mask=to_dev(pmask), description_input_ids=to_dev(tid),
                 description_attention_mask=to_dev(tmask))
    loss = float(to_np(tr.forward_backward(batch))[0])
    assert np.isfinite(loss) and abs(loss - float(ref["loss"])) < 1e-4 * max(1.0, abs(float(ref["loss"])))
    if B == 1:
        assert abs(loss) < 1e-6 and all(float(g.abs().max()) < 1e-6 for g in tr.g)      # a single candidate: nothing to learn
        return
    for g, name in zip(tr.g, NAMES):
        assert rel(to_np(g), ref["grads"][name]) < 5e-4, name
    # pooled embeddings through the module surface too
    p = P.l2_normalize(P.get_sequence_embeddings(model, batch["protein_input_ids"], batch["protein_attention_mask"], readout))
    t = P.l2_normalize(P.get_description_embeddings(model, batch["description_input_ids"], batch["description_attention_mask"], k, readout))
    assert rel(to_np(p), ref["protein"]) < 2e-4 and rel(to_np(t), ref["text"]) < 2e-4


@pytest.mark.parametrize("case", ["tiny", "tiny_d64"])
@pytest.mark.parametrize("shape", SHAPES[1:], ids=lambda s: f"B{s[0]}_Tp{s[1]}_Tt{s[3]}")
def test_edge_shapes_bf16_mfma_path(golden, shape, case):
    """Same shapes through the bf16 kernels (MFMA GEMM edge tiles, flash attention with ragged tails, LDS-DMA staging of
    rows near the end of the buffers) against the oracle with bf16 rounding at the same points."""
    import p2t_hip as P
    B, Tp, plens, Tt, tlens = shape
    meta = golden(case)["meta"]
    esm, llama, ad, *_ = case_setup(meta)
    k = meta["layers"][-1]
    pid, pmask = synth.protein_batch(41, B, Tp, plens)
    tid, tmask = synth.text_batch(41, B, Tt, meta["id_high"], tlens, meta["pad_id"], meta["eos_id"])
    W = model_weights(esm, llama, ad, meta["seed_w"])
    po = O.protein_embeddings(esm, W, pid, pmask, "mix", prec=O.BF16)
    to_ = O.text_embeddings(llama, W, tid, tmask, k, "mix", prec=O.BF16)
    model = build_model(esm, llama, ad, torch.bfloat16, meta["seed_w"]).eval()
    tr = P.ContrastiveTrainer(model, output_llm_layer=k, train_mode=False)
    batch = dict(protein_input_ids=to_dev(pid), protein_attention_mask=to_dev(pmask), description_input_ids=to_dev(tid),
                 description_attention_mask=to_dev(tmask))
    loss = float(to_np(tr.forward_backward(batch))[0])
    with torch.no_grad():
        p = P.l2_normalize(P.get_sequence_embeddings(model, batch["protein_input_ids"], batch["protein_attention_mask"]))
        t = P.l2_normalize(P.get_description_embeddings(model, batch["description_input_ids"], batch["description_attention_mask"], k))
    observe(f"edge[{case},B{B}_Tp{Tp}].bf16_vs_bf16oracle.protein", rel(to_np(p), po), 1e-2)
    observe(f"edge[{case},B{B}_Tp{Tp}].bf16_vs_bf16oracle.text", rel(to_np(t), to_), 1e-2)
    observe(f"edge[{case},B{B}_Tp{Tp}].bf16_vs_bf16oracle.loss", abs(loss - float(O.infonce_batch(po, to_))), 5e-2, "abs")     # logits = 20 x cosine: 6e-3 on the embeddings is 2-3e-2 on this loss
    # gradients: finite, unless the reference's own eps-free std readout hits 0 / 0 -- a two-token protein whose two
    # bf16 adapter rows agree exactly in some column has variance 0 there, and d sqrt(0) is NaN upstream as well
    # (scripts/train_contrast.py:223-235)
    with torch.no_grad():
        y, _ = model(protein_input_ids=batch["protein_input_ids"], protein_attention_mask=batch["protein_attention_mask"],
                     return_adapter_outputs=True)
    y, m = to_np(y).astype(np.float64), pmask.astype(np.float64)[:, :, None]
    mean = (y * m).sum(1, keepdims=True) / m.sum(1, keepdims=True)
    var = (((y - mean) ** 2) * m).sum(1) / m.sum(1)
    finite = all(bool(torch.isfinite(g).all()) for g in tr.g)
    assert finite or bool((var == 0).any()), "non-finite gradients without a zero-variance column"
    if not (var == 0).any():
        assert finite


@pytest.mark.parametrize("nseg", [1, 2])
def test_column_term_trainer_vs_oracle(golden, nseg):
    """ContrastiveTrainer(column_weight=0.5) and teacher_forcing_forward_pass(column_weight=0.5): loss and adapter
    gradients against the oracle's symmetric step (the reference's loss module with swapped arguments for the column half)."""
    import p2t_hip as P
    meta = golden("tiny_d64")["meta"]
    esm, llama, ad, pid, pmask, tid, tmask = case_setup(meta)
    pid, pmask, tid, tmask = pid[:4], pmask[:4], tid[:4], tmask[:4]
    k = meta["layers"][-1]
    W = model_weights(esm, llama, ad, meta["seed_w"])
    ref = O.contrastive_step(esm, llama, W, pid, pmask, tid, tmask, layer=k, with_grads=True, column_weight=0.5, num_segments=nseg)
    row_only = O.contrastive_step(esm, llama, W, pid, pmask, tid, tmask, layer=k, num_segments=nseg)
    assert abs(float(ref["loss"]) - float(row_only["loss"])) > 1e-3          # the column half changes the value
    model = build_model(esm, llama, ad, torch.float32, meta["seed_w"])
    batch = dict(protein_input_ids=to_dev(pid), protein_attention_mask=to_dev(pmask), description_input_ids=to_dev(tid),
                 description_attention_mask=to_dev(tmask))
    tr = P.ContrastiveTrainer(model, output_llm_layer=k, train_mode=False, num_segments=nseg, column_weight=0.5)
    loss = float(to_np(tr.forward_backward(batch))[0])
    assert abs(loss - float(ref["loss"])) < 1e-4 * max(1.0, abs(float(ref["loss"])))
    for g, name in zip(tr.g, NAMES):
        assert rel(to_np(g), ref["grads"][name]) < 5e-4, name
    assert float(to_np(tr.evaluate(batch))[0]) == pytest.approx(loss, rel=1e-6)
    # autograd surface
    model.esm_encoder.requires_grad_(False); model.llama_decoder.requires_grad_(False)
    model.adapter.requires_grad_(True)
    model.train(); model.adapter.dropout.p = 0.0
    l2 = P.teacher_forcing_forward_pass(0, model, batch, nseg, output_llm_layer=k, column_weight=0.5)
    l2.backward()
    assert abs(float(l2) - float(ref["loss"])) < 1e-4 * max(1.0, abs(float(ref["loss"])))
    prm = dict(model.adapter.named_parameters())
    for name in NAMES:
        assert rel(to_np(prm[name[len("adapter."):]].grad), ref["grads"][name]) < 5e-4, name
    with pytest.raises(ValueError):
        P.ContrastiveTrainer(model, column_weight=1.5)
    with pytest.raises(ValueError):
        P.ContrastiveTrainer(model, num_segments=3, column_weight=0.5).forward_backward(batch)      # 4 pairs, 3 segments
