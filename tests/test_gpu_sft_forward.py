"""Decoder half of Esm2LlamaInstructForCausalLM.forward against the reference class run in
this container (tests/golden/sft_tiny.npz, make_golden.py run_sft): placeholder scatter, logits, shifted LM loss; the
small kernels behind it against numpy."""
import json
import os

import numpy as np
import pytest
import torch

from gpu_util import build_model, dev, rel, to_dev, to_np, observe
from p2t_hip import specs

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def g():
    z = np.load(os.path.join(HERE, "golden", "sft_tiny.npz"))
    d = {k: z[k] for k in z.files}
    d["meta"] = json.loads(bytes(d.pop("meta_json")).decode())
    return d


def _model(g, dtype):
    m = g["meta"]
    model = build_model(specs.EsmSpec(**m["esm"]), specs.LlamaSpec(**m["llama"]), specs.AdapterSpec(**m["adapter"]), dtype, 0)
    model.config.placeholder_id = m["placeholder_id"]
    return model.eval()


def _inputs(g):
    return dict(input_ids=to_dev(g["input_ids"]), attention_mask=to_dev(g["attention_mask"]),
                protein_input_ids=to_dev(g["protein_input_ids"]), protein_attention_mask=to_dev(g["protein_attention_mask"]))


def test_positions_where_and_scatter_rows():
    from p2t_hip import ops
    rng = np.random.default_rng(0)
    ids = rng.integers(0, 5, size=(7, 333)).astype(np.int64)
    pos, cnt = ops.positions_where(to_dev(ids), 3)
    want = np.flatnonzero(ids.reshape(-1) == 3)
    assert int(cnt.item()) == len(want) and np.array_equal(to_np(pos)[:len(want)], want)
    mask = (rng.random((7, 333)) < 0.4).astype(np.int64)
    pos2, cnt2 = ops.positions_where(to_dev(mask))
    want2 = np.flatnonzero(mask.reshape(-1))
    assert int(cnt2.item()) == len(want2) and np.array_equal(to_np(pos2)[:len(want2)], want2)
    H = 70
    dst = rng.standard_normal((7 * 333, H)).astype(np.float32)
    src = rng.standard_normal((7 * 333, 128)).astype(np.float32)
    for sdt in (torch.float32, torch.bfloat16):
        d = to_dev(dst.copy())
        s = to_dev(src, sdt)
        ops.scatter_rows(d, pos, cnt, s, pos2, cnt2, H)
        n = min(len(want), len(want2))
        ref = dst.copy()
        ref[want[:n]] = to_np(s)[want2[:n], :H]
        assert np.array_equal(to_np(d), ref)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_cross_entropy_shifted_vs_numpy(dt):
    from p2t_hip import ops
    rng = np.random.default_rng(1)
    B, T, V, ld = 3, 17, 1000, 1024
    logits = (rng.standard_normal((B, T, ld)) * 3).astype(np.float32)
    labels = rng.integers(0, V, size=(B, T)).astype(np.int64)
    labels[0, :5] = -100
    labels[2, -3:] = -100
    lt = to_dev(logits, dt)
    loss, cnt = ops.cross_entropy_shifted(lt, to_dev(labels), V)
    x = to_np(lt).astype(np.float64)[:, :-1, :V]
    y = labels[:, 1:]
    lse = np.log(np.exp(x - x.max(-1, keepdims=True)).sum(-1)) + x.max(-1)
    valid = y != -100
    picked = np.take_along_axis(x, np.where(valid, y, 0)[..., None], -1)[..., 0]
    want = ((lse - picked) * valid).sum() / valid.sum()
    assert int(cnt.item()) == int(valid.sum())
    assert abs(float(loss.item()) - want) < 2e-5 * abs(want)
    none, cnt0 = ops.cross_entropy_shifted(lt, to_dev(np.full((B, T), -100, dtype=np.int64)), V)
    assert np.isnan(float(none.item())) and int(cnt0.item()) == 0


def test_fp32_decoder_inputs_logits_and_loss_match_reference(g):
    model = _model(g, torch.float32)
    emb, mask = model(**_inputs(g), return_decoder_inputs=True)
    assert torch.equal(mask, to_dev(g["attention_mask"]))
    assert rel(to_np(emb), g["inputs_embeds"]) < 2e-4
    ph = g["input_ids"] == g["meta"]["placeholder_id"]
    np.testing.assert_allclose(to_np(emb)[~ph], g["inputs_embeds"][~ph], rtol=0, atol=0)     # untouched rows: exact table rows
    out = model(**_inputs(g), labels=to_dev(g["labels"]))
    # positions under attention_mask == 0 (left padding) are unspecified: a padded query sees no key at all, where HF's
    # eager path softmaxes a row of equal finite minima and this kernel returns zeros; neither feeds the loss
    valid = g["attention_mask"].astype(bool)
    assert rel(to_np(out.logits)[valid], g["logits"][valid]) < 2e-4
    assert abs(float(out.loss) - float(g["loss"])) < 2e-4 * float(g["loss"])
    assert out[0] is out.loss and out["logits"] is out.logits
    nolabel = model(**_inputs(g))
    assert nolabel.loss is None and rel(to_np(nolabel.logits)[valid], g["logits"][valid]) < 2e-4
    # the decoder alone, from token ids (no placeholders replaced), is a plain LlamaForCausalLM forward
    plain = model.llama_decoder(input_ids=to_dev(g["input_ids"]), attention_mask=to_dev(g["attention_mask"]))
    assert plain.logits.shape == out.logits.shape and rel(to_np(plain.logits)[valid], g["logits"][valid]) > 1e-3


def test_bf16_forward_close_to_reference(g):
    model = _model(g, torch.bfloat16)
    out = model(**_inputs(g), labels=to_dev(g["labels"]))
    assert out.logits.dtype == torch.bfloat16
    valid = g["attention_mask"].astype(bool)
    observe("sft_tiny.bf16_vs_reference.logits", rel(to_np(out.logits)[valid], g["logits"][valid]), 3e-2)
    observe("sft_tiny.bf16_vs_reference.loss", abs(float(out.loss) - float(g["loss"])) / float(g["loss"]), 2e-2)


def test_placeholder_count_mismatch_raises(g):
    model = _model(g, torch.float32)
    inp = _inputs(g)
    inp["protein_attention_mask"] = inp["protein_attention_mask"].clone()
    inp["protein_attention_mask"][0, -1] = 0                      # one encoder token fewer than placeholders
    with pytest.raises(RuntimeError, match="shape mismatch"):
        model(**inp, return_decoder_inputs=True)
    with pytest.raises(ValueError):
        model.llama_decoder(input_ids=inp["input_ids"], inputs_embeds=torch.zeros(3, 27, 64, device=dev()))
