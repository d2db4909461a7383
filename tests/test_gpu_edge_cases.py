"""Edge shapes of the contrastive step against the oracle (fp32 towers, eval mode): the smallest legal inputs (<cls><eos>
proteins, one-token descriptions, a batch of one pair), lengths that are not multiples of any tile (65, 130, 7, 1), rows
of very different lengths in one batch, and every readout mode.  Loss and adapter gradients through ContrastiveTrainer
(the C ABI: p2t_esm2_forward, p2t_llama_hidden_forward, p2t_adapter_*, p2t_readout*, p2t_infonce_*)."""
import numpy as np
import pytest
import torch

from oracle import p2t_oracle as O
from helpers import case_setup, model_weights
from gpu_util import build_model, rel, to_dev, to_np
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


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: f"B{s[0]}_Tp{s[1]}_Tt{s[3]}")
@pytest.mark.parametrize("readout", ["mix", "mean", "last"])
def test_edge_shapes_match_oracle(golden, shape, readout):
    import p2t_hip as P
    B, Tp, plens, Tt, tlens = shape
    meta = golden("tiny")["meta"]
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
