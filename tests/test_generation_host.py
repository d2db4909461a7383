"""Host-side pieces of `generate` (p2t_hip/generation.py) that need no GPU: the sampling filters against the installed HF logits
warpers (transformers/generation/logits_process.py: temperature -> top-k -> top-p, the order GenerationMixin applies them in), the
trimming rule of the returned ids, and the eos-id normalisation."""
import numpy as np
import pytest
import torch


def test_filter_logits_equals_the_hf_warpers():
    from transformers.generation.logits_process import TemperatureLogitsWarper, TopKLogitsWarper, TopPLogitsWarper
    from p2t_hip.generation import filter_logits
    g = torch.Generator().manual_seed(0)
    scores = torch.randn((6, 500), generator=g) * 3.0
    scores[2, 10:20] = scores[2, 10]                    # ties inside the kept set
    ids = torch.zeros((6, 1), dtype=torch.long)
    for temperature, top_k, top_p in [(1.0, None, None), (0.7, 50, 1.0), (1.3, 5, 0.9), (1.0, 0, 0.5), (0.5, 1, 1.0), (2.0, 1000, 0.99), (1.0, None, 0.1)]:
        want = scores.clone()
        if temperature != 1.0:
            want = TemperatureLogitsWarper(temperature)(ids, want)
        if top_k:
            want = TopKLogitsWarper(top_k=top_k)(ids, want)
        if top_p is not None and top_p < 1.0:
            want = TopPLogitsWarper(top_p=top_p)(ids, want)
        got = filter_logits(scores.clone(), temperature, top_k, top_p)
        assert torch.equal(torch.isinf(got), torch.isinf(want)), (temperature, top_k, top_p)
        keep = ~torch.isinf(want)
        assert torch.allclose(got[keep], want[keep], rtol=0, atol=0), (temperature, top_k, top_p)
        assert (keep.sum(1) >= 1).all()
    with pytest.raises(ValueError):
        filter_logits(scores, temperature=0.0)
    with pytest.raises(ValueError):
        filter_logits(scores, top_p=-0.1)


def test_trim_and_eos_list():
    from p2t_hip.generation import _as_id_list, _trim
    toks = torch.tensor([[5, 9, 7, 0, 0, 0], [5, 5, 5, 9, 0, 0], [1, 2, 3, 4, 5, 6]])
    assert _trim(toks, [9], 6).shape[1] == 6                                   # a row without eos keeps the full width
    assert _trim(toks[:2], [9], 6).tolist() == [[5, 9, 7, 0], [5, 5, 5, 9]]     # HF stops right after the last row's eos
    assert _trim(toks[:2], [9, 7], 5).shape[1] == 4 and _trim(toks, [], 4).shape[1] == 4
    assert _as_id_list(None) == [] and _as_id_list(7) == [7] and _as_id_list([1, 2]) == [1, 2] and _as_id_list(torch.tensor([3, 4])) == [3, 4]
    assert np.array_equal(_trim(toks, [42], 3).numpy(), toks[:, :3].numpy())
