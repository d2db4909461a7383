"""Host-side batch pipeline: tokeniser vs the installed HF EsmTokenizer, collater vs the
reference's own Prot2TextLightCollater (tests/golden/collate.json, written by make_golden.py), prefetcher ordering."""
import json
import os
import random

import pytest
import torch

from p2t_hip import data

HERE = os.path.dirname(os.path.abspath(__file__))


def tiny_text_tokenizer(words):
    """Same local word-level tokenizer the golden script gave the reference collater."""
    from tokenizers import Tokenizer, models, pre_tokenizers
    from transformers import PreTrainedTokenizerFast
    vocab = {w: i for i, w in enumerate(["<pad>", "<unk>", "<eos>"] + list(words))}
    tk = Tokenizer(models.WordLevel(vocab=vocab, unk_token="<unk>"))
    tk.pre_tokenizer = pre_tokenizers.WhitespaceSplit()
    return PreTrainedTokenizerFast(tokenizer_object=tk, pad_token="<pad>", unk_token="<unk>", eos_token=" <eos>")


@pytest.fixture(scope="module")
def hf_esm_tokenizer(tmp_path_factory):
    from transformers import EsmTokenizer
    path = tmp_path_factory.mktemp("esm") / "vocab.txt"
    path.write_text("\n".join(data.ESM_VOCAB) + "\n")
    return EsmTokenizer(str(path))


def test_esm_vocab_ids():
    tok = data.EsmSequenceTokenizer()
    assert len(data.ESM_VOCAB) == 33
    assert (tok.cls_token_id, tok.pad_token_id, tok.eos_token_id, tok.unk_token_id, tok.mask_token_id) == (0, 1, 2, 3, 32)


def test_esm_tokenizer_matches_hf(hf_esm_tokenizer):
    tok = data.EsmSequenceTokenizer()
    fixed = ["MKTAYIAKQR", "AJJB", "AjB", "A B", "AC-.XZ", "", "M<mask>K", "JJJ", "AJ J", "<cls>A<eos>", "<A", "A<", "<null_1>",
             "  M  K ", "UZOBX"]
    rng = random.Random(5)
    alphabet = "LAGVSERTIDPKQNFYMHWCXBUZO.-" + "Jj1 <>"
    rand = ["".join(rng.choice(alphabet) for _ in range(rng.randint(0, 60))) for _ in range(300)]
    for s in fixed + rand:
        want = hf_esm_tokenizer([s], add_special_tokens=True, return_attention_mask=False)["input_ids"][0]
        assert tok.encode(s) == want, repr(s)
        want_ns = hf_esm_tokenizer([s], add_special_tokens=False, return_attention_mask=False)["input_ids"][0]
        assert tok.encode(s, add_special_tokens=False) == want_ns, repr(s)


def test_esm_batch_padding_matches_hf(hf_esm_tokenizer):
    seqs = ["MKTAYIAKQRQISFVKSHFSRQ", "MK", "", "ACDEFGHIKLMNPQRSTVWY"]
    got = data.EsmSequenceTokenizer()(seqs)
    want = hf_esm_tokenizer(seqs, padding=True, return_tensors="pt")
    assert torch.equal(got["input_ids"], want["input_ids"]) and torch.equal(got["attention_mask"], want["attention_mask"])
    assert got["input_ids"].dtype == torch.int64 and got["input_ids"][1, 4:].eq(1).all()


def test_pad_sequences():
    a, b = torch.tensor([1, 2, 3]), torch.tensor([4])
    assert data.pad_sequences([a, b], 9, "right").tolist() == [[1, 2, 3], [4, 9, 9]]
    assert data.pad_sequences([a, b], -100, "left").tolist() == [[1, 2, 3], [-100, -100, 4]]
    with pytest.raises(ValueError):
        data.pad_sequences([a, b], 0, "middle")


def test_collater_matches_reference_golden():
    with open(os.path.join(HERE, "golden", "collate.json")) as f:
        g = json.load(f)
    rows = [{k: (float("nan") if v is None else v) for k, v in r.items()} for r in g["rows"]]
    col = data.ContrastiveCollater(tiny_text_tokenizer(g["words"]), **g["params"])
    seq_tok = data.EsmSequenceTokenizer()
    for exp in g["expected"]:
        random.seed(exp["seed"])
        b = col(rows)
        assert b["protein_sequences"] == exp["protein_sequences"]                        # same random crop windows
        assert b["description_input_ids"].tolist() == exp["description_input_ids"]
        assert b["description_attention_mask"].tolist() == exp["description_attention_mask"]
        assert b["name"] == [r["AlphaFoldDB"] for r in rows]
        # protein half of the contract: <cls> seq <eos>, right padded with id 1 (dataset/dataloader.py:113-123)
        want = seq_tok(exp["protein_sequences"])
        assert torch.equal(b["protein_input_ids"], want["input_ids"])
        assert torch.equal(b["protein_attention_mask"], want["attention_mask"])
        assert b["protein_input_ids"].shape[1] == g["params"]["max_sequence_length"] + 2
        assert b["description_input_ids"].shape[1] <= g["params"]["max_description_length"]
    crops = {tuple(e["protein_sequences"]) for e in g["expected"]}
    assert len(crops) > 1                                                               # the seeds really differ


def test_prefetcher_keeps_order_and_content():
    batches = [{"protein_input_ids": torch.full((2, 3), i), "name": [f"n{i}"]} for i in range(5)]
    seen = list(data.DevicePrefetcher(batches, "cpu", transform=lambda b: {**b, "extra": b["protein_input_ids"] + 1}))
    assert [b["name"][0] for b in seen] == [f"n{i}" for i in range(5)]
    assert all(int(b["extra"][0, 0]) == i + 1 for i, b in enumerate(seen))
    assert list(data.DevicePrefetcher([], "cpu")) == []


def test_sort_batch_by_length_permutes_pairs_together():
    from p2t_hip.data import sort_batch_by_length
    mask = torch.tensor([[1, 1, 0, 0], [1, 1, 1, 1], [1, 0, 0, 0], [1, 1, 1, 0]])
    batch = {"name": ["a", "b", "c", "d"], "protein_input_ids": torch.arange(16).view(4, 4), "protein_attention_mask": mask,
             "description_input_ids": torch.arange(8).view(4, 2) * 10, "description_attention_mask": torch.ones(4, 2, dtype=torch.int64),
             "other": 7}
    out = sort_batch_by_length(batch)
    assert out["name"] == ["b", "d", "a", "c"] and out["protein_lengths"] == [4, 3, 2, 1] and out["other"] == 7
    assert out["protein_input_ids"][:, 0].tolist() == [4, 12, 0, 8]
    assert out["description_input_ids"][:, 0].tolist() == [20, 60, 0, 40]          # text rows follow their proteins
    assert out["description_lengths"] == [2, 2, 2, 2] and out["description_order"].tolist() == [0, 1, 2, 3]
    assert sort_batch_by_length(batch, descending=False)["name"] == ["c", "a", "d", "b"]
    assert batch["name"] == ["a", "b", "c", "d"]                                   # input untouched


def test_trim_padding_rejects_the_all_ones_readout():
    from p2t_hip.contrastive import ContrastiveTrainer
    with pytest.raises(ValueError, match="mask-aware"):
        ContrastiveTrainer(None, ones_mask=True, trim_padding=True)
    with pytest.raises(ValueError, match="multiple of 64"):
        ContrastiveTrainer(None, trim_padding=True, trim_multiple=96)


def test_plan_length_segments():
    from p2t_hip.contrastive import plan_length_segments as plan
    assert plan([], 128) == [] and plan([5], 1024) == [(0, 1, 128)]
    # equal lengths: one segment unless the token cap splits it
    assert plan([1024] * 16, 1024) == [(0, 16, 1024)]
    assert plan([1024] * 16, 1024, max_tokens=4 * 1024) == [(0, 4, 1024), (4, 8, 1024), (8, 12, 1024), (12, 16, 1024)]
    # a row longer than the cap still gets its own segment
    assert plan([1000, 10], 1024, max_tokens=512, floor_tokens=0) == [(0, 1, 1024), (1, 2, 128)]
    lens = sorted([1024, 900, 700, 500, 400, 390, 380, 300, 250, 240, 200, 130, 120, 100, 60, 30] * 4, reverse=True)
    segs = plan(lens, 1024, multiple=128, floor_tokens=4096)
    assert segs[0][0] == 0 and segs[-1][1] == len(lens) and all(a[1] == b[0] for a, b in zip(segs, segs[1:]))
    assert all(t % 128 == 0 and t >= max(lens[a:b]) for a, b, t in segs)
    padded = len(lens) * 1024
    assert sum((b - a) * t for a, b, t in segs) < 0.6 * padded              # the point of it
    # a huge floor means splitting never pays
    assert plan(lens, 1024, floor_tokens=10 ** 9) == [(0, len(lens), 1024)]


def test_prefetcher_with_length_sort_transform_keeps_host_lengths():
    """DevicePrefetcher(transform=sort_batch_by_length): tensors move, `protein_lengths` / `description_lengths` stay
    host lists, `description_order` travels as a tensor; every batch of the loader comes out once, in order."""
    from p2t_hip.data import DevicePrefetcher, sort_batch_by_length

    def make(i):
        mask = torch.zeros(3, 6, dtype=torch.int64)
        for r, n in enumerate([2 + i, 6, 3]):
            mask[r, :n] = 1
        return {"name": [f"b{i}r{r}" for r in range(3)], "protein_input_ids": torch.arange(18).view(3, 6) + 100 * i,
                "protein_attention_mask": mask, "description_input_ids": torch.arange(12).view(3, 4),
                "description_attention_mask": torch.tensor([[1, 1, 0, 0], [1, 1, 1, 1], [1, 0, 0, 0]])}

    out = list(DevicePrefetcher([make(0), make(1), make(2)], "cpu", transform=sort_batch_by_length))
    assert [b["name"][0] for b in out] == ["b0r1", "b1r1", "b2r1"] and len(out) == 3
    for i, b in enumerate(out):
        assert b["protein_lengths"] == sorted([2 + i, 6, 3], reverse=True) and isinstance(b["protein_lengths"], list)
        assert b["protein_attention_mask"].sum(1).tolist() == b["protein_lengths"]
        assert isinstance(b["description_lengths"], list) and torch.is_tensor(b["description_order"])
        lens = b["description_lengths"]
        assert [lens[j] for j in b["description_order"].tolist()] == sorted(lens, reverse=True)
    assert list(DevicePrefetcher([], "cpu")) == []


def test_sort_batch_by_length_refuses_left_padding():
    """trim_padding slices ids[:, :T_s]: only valid for right-padded rows, verified on the host (dataloader_derived.py:142-147
    can produce left-padded descriptions)."""
    import torch
    from p2t_hip.data import sort_batch_by_length
    ok = {"protein_input_ids": torch.zeros((2, 4), dtype=torch.int64), "protein_attention_mask": torch.tensor([[1, 1, 1, 0], [1, 0, 0, 0]]),
          "description_input_ids": torch.zeros((2, 3), dtype=torch.int64), "description_attention_mask": torch.tensor([[1, 1, 0], [1, 1, 1]])}
    assert sort_batch_by_length(ok)["protein_lengths"] == [3, 1]
    for key, bad in (("protein_attention_mask", torch.tensor([[0, 1, 1, 1], [1, 0, 0, 0]])),
                     ("description_attention_mask", torch.tensor([[1, 0, 1], [1, 1, 1]]))):
        with pytest.raises(ValueError, match="right-padded"):
            sort_batch_by_length(dict(ok, **{key: bad}))
