"""Ragged batches: running each segment of a length-sorted batch at its own padded length
(`ContrastiveTrainer(num_segments=k, trim_padding=True)` on `data.sort_batch_by_length` output) must give the loss and
the adapter gradients of the plain step on the original batch: the in-batch InfoNCE loss is invariant to the order of
the pairs, equal segments average to the unsegmented loss (scripts/train_contrast.py:94-114,367-379), and with the
mask-aware readout nothing reads a padded position."""
import json
import os

import numpy as np
import pytest
import torch

from gpu_util import build_model, rel, to_np, observe
from p2t_hip import specs, synth
from p2t_hip.data import sort_batch_by_length

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
LENGTHS = [64, 256, 17, 60, 64, 5, 33, 50]


def _model(dtype):
    with open(os.path.join(HERE, "golden", "train_state.json")) as f:
        meta = json.load(f)
    esm, llama, ad = specs.EsmSpec(**meta["esm"]), specs.LlamaSpec(**meta["llama"]), specs.AdapterSpec(**meta["adapter"])
    return build_model(esm, llama, ad, dtype, 0), meta


def _host_batch():
    pid, pmask = synth.protein_batch(7, 8, 256, LENGTHS)
    tid, tmask = synth.text_batch(7, 8, 160, 500, [160, 9, 5, 70, 12, 130, 3, 11], 510, 509)
    return {"name": [f"P{i}" for i in range(8)], "protein_input_ids": torch.from_numpy(pid),
            "protein_attention_mask": torch.from_numpy(pmask), "description_input_ids": torch.from_numpy(tid),
            "description_attention_mask": torch.from_numpy(tmask)}


def _to_dev(batch):
    return {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in batch.items()}


@pytest.mark.parametrize("dtype,tol_loss,tol_grad", [(torch.float32, 2e-5, 2e-4), (torch.bfloat16, 5e-3, 5e-2)])
def test_trimmed_segments_equal_the_padded_step(dtype, tol_loss, tol_grad):
    import p2t_hip as P
    model, meta = _model(dtype)
    host = _host_batch()
    plain = P.ContrastiveTrainer(model, output_llm_layer=meta["layer"], train_mode=False)
    loss0 = float(to_np(plain.forward_backward(_to_dev(host)))[0])
    g0 = to_np(plain.flat_g).copy()

    srt = sort_batch_by_length(host)
    assert srt["protein_lengths"] == sorted(LENGTHS, reverse=True) and srt["name"][0] == "P1"
    trimmed = P.ContrastiveTrainer(model, output_llm_layer=meta["layer"], train_mode=False, num_segments=2, trim_padding=True,
                                   trim_multiple=64, trim_floor_tokens=0, overlap_streams=False)
    segs = trimmed._segments(srt, 8, 256)
    assert [(a, b, t) for a, b, t, _ in segs] == [(0, 1, 256), (1, 8, 64)]       # unequal ranges, own lengths
    assert sum(w for *_, w in segs) == pytest.approx(1.0) and all((b - a) * t <= 4 * 256 for a, b, t, _ in segs)
    # the text side runs in its own length order (description_order), cut the same way, and is scattered back
    assert srt["description_order"].tolist()[:3] == [srt["description_lengths"].index(v) for v in (160, 130, 70)]
    t_plain = to_np(plain.text_embeddings(*[_to_dev(srt)[k] for k in ("description_input_ids", "description_attention_mask")]))
    t_trim = to_np(trimmed.text_embeddings(*[_to_dev(srt)[k] for k in ("description_input_ids", "description_attention_mask")], _to_dev(srt)))
    tag = "fp32" if dtype == torch.float32 else "bf16"
    observe(f"ragged[{tag}].text_trim_vs_padded", rel(t_trim, t_plain), 1e-5 if dtype == torch.float32 else 2e-2)
    loss1 = float(to_np(trimmed.forward_backward(_to_dev(srt)))[0])
    g1 = to_np(trimmed.flat_g).copy()
    observe(f"ragged[{tag}].loss_trim_vs_padded", abs(loss1 - loss0) / max(1.0, abs(loss0)), tol_loss * 1.0000001, "abs/max(1,|ref|)")
    observe(f"ragged[{tag}].grads_trim_vs_padded", rel(g1, g0), tol_grad)

    # the same trainer on two streams (segments alternate between encode streams): the default of trim_padding=True
    assert P.ContrastiveTrainer(model, trim_padding=True).overlap_streams and not P.ContrastiveTrainer(model).overlap_streams
    trimmed.overlap_streams = True
    loss2 = float(to_np(trimmed.forward_backward(_to_dev(srt)))[0])
    assert abs(loss2 - loss1) <= tol_loss * max(1.0, abs(loss1)) and rel(to_np(trimmed.flat_g), g1) < tol_grad


def test_autograd_path_with_trimmed_segments():
    """teacher_forcing_forward_pass(trim_padding=True): same loss and adapter gradients as the padded call."""
    import p2t_hip as P
    model, meta = _model(torch.float32)
    model.eval()                                            # dropout masks depend on the row layout
    srt = _to_dev(sort_batch_by_length(_host_batch()))
    grads = []
    for trim in (False, True):
        model.adapter.zero_grad(set_to_none=True)
        loss = P.teacher_forcing_forward_pass(0, model, srt, 4, output_llm_layer=meta["layer"], trim_padding=trim, trim_multiple=64)
        loss.backward()
        grads.append((float(loss.detach()), [to_np(p.grad) for p in (model.adapter.fc1.weight, model.adapter.fc1.bias,
                                                              model.adapter.fc2.weight, model.adapter.fc2.bias)]))
    assert abs(grads[0][0] - grads[1][0]) < 2e-5 * max(1.0, abs(grads[0][0]))
    for a, b in zip(grads[0][1], grads[1][1]):
        assert rel(b, a) < 2e-4
    with pytest.raises(ValueError, match="protein_lengths"):
        P.teacher_forcing_forward_pass(0, model, {k: v for k, v in srt.items() if k != "protein_lengths"}, 4,
                                       output_llm_layer=meta["layer"], trim_padding=True)
    with pytest.raises(ValueError, match="mask-aware"):
        P.teacher_forcing_forward_pass(0, model, srt, 4, trim_padding=True, ones_mask=True)


def test_trim_padding_argument_errors():
    import p2t_hip as P
    model, meta = _model(torch.float32)
    tr = P.ContrastiveTrainer(model, output_llm_layer=meta["layer"], num_segments=2, trim_padding=True)
    dev_batch = _to_dev(_host_batch())
    with pytest.raises(ValueError, match="protein_lengths"):
        tr.forward_backward(dev_batch)
    with pytest.raises(ValueError, match="host"):
        tr.forward_backward(dict(dev_batch, protein_lengths=torch.tensor(LENGTHS).cuda()))
    with pytest.raises(ValueError, match="entries"):
        tr.forward_backward(dict(dev_batch, protein_lengths=LENGTHS[:4]))
    with pytest.raises(ValueError, match="only 256 wide"):
        tr.forward_backward(dict(dev_batch, protein_lengths=[300] * 8))


def test_prefetcher_sort_transform_feeds_the_trimming_trainer():
    """Host batches -> DevicePrefetcher(transform=sort_batch_by_length) -> ContrastiveTrainer(trim_padding=True): the order
    tensor arrives on the device with the batch, the length lists stay on the host, and each step's loss equals the plain
    padded step on the same (unsorted) host batch."""
    import p2t_hip as P
    model, meta = _model(torch.float32)
    hosts = [_host_batch(), {k: (v.flip(0) if torch.is_tensor(v) else v[::-1]) for k, v in _host_batch().items()}]
    plain = P.ContrastiveTrainer(model, output_llm_layer=meta["layer"], train_mode=False)
    want = [float(to_np(plain.forward_backward(_to_dev(h)))[0]) for h in hosts]
    trim = P.ContrastiveTrainer(model, output_llm_layer=meta["layer"], train_mode=False, trim_padding=True, trim_multiple=64,
                                trim_floor_tokens=0)
    got = []
    for batch in P.DevicePrefetcher(hosts, "cuda:0", transform=sort_batch_by_length):
        assert batch["description_order"].is_cuda and isinstance(batch["protein_lengths"], list)
        got.append(float(to_np(trim.forward_backward(batch))[0]))
    assert got == pytest.approx(want, rel=2e-5) and want[0] == pytest.approx(want[1], rel=1e-5)     # order of pairs is irrelevant
