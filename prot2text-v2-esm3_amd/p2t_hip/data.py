"""Host-side batch pipeline of the contrastive step.

What the reference does before the hot path sees a batch, restated without its dataset classes:

* protein side -- `dataset/dataset.py:392-397`: HF `EsmTokenizer` (33-token vocabulary, `<cls> seq <eos>`), then
  `dataset/dataloader.py:113-123`: right padding with the tokenizer's pad id and a 0/1 attention mask;
* description side -- `dataset/dataloader_light.py:222-239`: `description + eos`, no BOS, truncation to
  `max_description_length`, right padding to the longest, attention mask;
* sequence cropping -- `dataset/dataloader_light.py:172-179`: a random window of `max_sequence_length` residues.

`EsmSequenceTokenizer` is pinned against the installed `transformers.EsmTokenizer` (tests/test_data_pipeline.py);
`ContrastiveCollater` against the reference's own `Prot2TextLightCollater` run on the same rows with the same tiny
tokenizer (tests/golden/collate.json, written by tests/golden/make_golden.py).  `DevicePrefetcher` stages batches
through pinned memory and a copy stream so the H2D transfer of batch i+1 overlaps the step of batch i.

Ragged lengths: real proteins are much shorter than the longest of a batch.  `sort_batch_by_length` (host side, before
the H2D copy) orders the pairs of a batch by protein length and records the lengths, so that
`ContrastiveTrainer(num_segments=k, trim_padding=True)` can run each segment at ITS longest length instead of the
batch's.  The in-batch InfoNCE loss does not depend on the order of the pairs, and the segmented loss of equal segments
equals the unsegmented one (`scripts/train_contrast.py:94-114,367-379`), so the step's result is unchanged.
"""
from __future__ import annotations

import random
from typing import Any, Callable, Dict, Iterable, Iterator, List, Optional, Sequence

import torch

# facebook/esm2_* vocabulary (public, 33 entries; ids = positions).  SURVEY.md section 8: cls 0, pad 1, eos 2, mask 32.
ESM_VOCAB = ["<cls>", "<pad>", "<eos>", "<unk>", "L", "A", "G", "V", "S", "E", "R", "T", "I", "D", "P", "K", "Q", "N", "F",
             "Y", "M", "H", "W", "C", "X", "B", "U", "Z", "O", ".", "-", "<null_1>", "<mask>"]


def pad_sequences(sequences: Sequence[torch.Tensor], padding_value: int, padding_side: str = "right") -> torch.Tensor:
    """Stack 1-D tensors of different lengths (dataset/dataloader.py:199-228)."""
    if padding_side not in ("left", "right"):
        raise ValueError(f"Invalid padding side: {padding_side}")
    n = max(int(s.shape[-1]) for s in sequences)
    out = torch.full((len(sequences), n), padding_value, dtype=sequences[0].dtype)
    for i, s in enumerate(sequences):
        k = int(s.shape[-1])
        if padding_side == "right":
            out[i, :k] = s
        else:
            out[i, n - k:] = s
    return out


class EsmSequenceTokenizer:
    """Amino-acid tokenizer with the behaviour of HF `EsmTokenizer` on sequence strings: every vocabulary entry is
    matched wherever it occurs (special-token strings included), white space separates, and a maximal run of other
    characters becomes ONE `<unk>`."""

    def __init__(self, vocab: Sequence[str] = ESM_VOCAB):
        self.vocab = list(vocab)
        self.token_to_id = {t: i for i, t in enumerate(self.vocab)}
        self.cls_token_id, self.pad_token_id = self.token_to_id["<cls>"], self.token_to_id["<pad>"]
        self.eos_token_id, self.unk_token_id = self.token_to_id["<eos>"], self.token_to_id["<unk>"]
        self.mask_token_id = self.token_to_id["<mask>"]
        self._multi = sorted((t for t in self.vocab if len(t) > 1), key=len, reverse=True)

    def encode(self, sequence: str, add_special_tokens: bool = True) -> List[int]:
        ids: List[int] = [self.cls_token_id] if add_special_tokens else []
        i, n, in_unk = 0, len(sequence), False
        while i < n:
            ch = sequence[i]
            if ch.isspace():
                in_unk = False
                i += 1
                continue
            hit = None
            if ch == "<":
                hit = next((t for t in self._multi if sequence.startswith(t, i)), None)
            if hit is None and ch in self.token_to_id:
                hit = ch
            if hit is not None:
                ids.append(self.token_to_id[hit])
                in_unk = False
                i += len(hit)
            else:
                if not in_unk:
                    ids.append(self.unk_token_id)
                in_unk = True
                i += 1
        if add_special_tokens:
            ids.append(self.eos_token_id)
        return ids

    def __call__(self, sequences: Sequence[str], add_special_tokens: bool = True) -> Dict[str, torch.Tensor]:
        """Right-padded `input_ids` / `attention_mask` (int64), the protein half of the batch contract."""
        toks = [torch.tensor(self.encode(s, add_special_tokens), dtype=torch.int64) for s in sequences]
        return {"input_ids": pad_sequences(toks, self.pad_token_id, "right"),
                "attention_mask": pad_sequences([torch.ones_like(t) for t in toks], 0, "right")}


class ContrastiveCollater:
    """Rows {"AlphaFoldDB", "sequence", "function", ...} -> the batch dict `teacher_forcing_forward_pass` consumes.

    `description_tokenizer` is any HF tokenizer (the reference passes the LLM's); it is called exactly as upstream does.
    The random crop draws from Python's `random` in the upstream order (two draws per row for the name / taxon
    dropout, which this path does not otherwise use, then one per over-long sequence), so a seeded run sees the same
    windows as the reference collater."""

    def __init__(self, description_tokenizer, sequence_tokenizer: Optional[EsmSequenceTokenizer] = None,
                 max_sequence_length: int = 1021, max_description_length: int = 512, name_dropout: float = 0.8,
                 taxonomy_dropout: float = 0.8):
        self.description_tokenizer = description_tokenizer
        self.sequence_tokenizer = sequence_tokenizer or EsmSequenceTokenizer()
        self.max_sequence_length, self.max_description_length = max_sequence_length, max_description_length
        self.name_dropout, self.taxonomy_dropout = name_dropout, taxonomy_dropout

    def __call__(self, batch: List[Dict[str, Any]]) -> Dict[str, Any]:
        for key in ("Full Name", "taxon"):                  # RNG parity with the upstream collater (:159-170)
            for item in batch:
                if isinstance(item.get(key), str):
                    random.random()
        sequences = []
        for item in batch:
            seq = item["sequence"]
            if len(seq) > self.max_sequence_length:
                start = random.randint(0, len(seq) - self.max_sequence_length)
                seq = seq[start:start + self.max_sequence_length]
            sequences.append(seq)
        prot = self.sequence_tokenizer(sequences)
        tok = self.description_tokenizer
        tok.padding_side = "right"
        desc = tok([item["function"] + tok.eos_token for item in batch], add_special_tokens=False, truncation=True,
                   padding="longest", max_length=self.max_description_length, return_tensors="pt")
        d_ids, d_mask = desc["input_ids"], desc["attention_mask"]
        if d_ids.size(1) > self.max_description_length:
            d_ids, d_mask = d_ids[:, :self.max_description_length], d_mask[:, :self.max_description_length]
        return {"name": [item["AlphaFoldDB"] for item in batch], "protein_sequences": sequences,
                "protein_input_ids": prot["input_ids"], "protein_attention_mask": prot["attention_mask"],
                "description_input_ids": d_ids, "description_attention_mask": d_mask}


def sort_batch_by_length(batch: Dict[str, Any], descending: bool = True) -> Dict[str, Any]:
    """Reorder the pairs of a collated (host) batch by protein length and add `protein_lengths` (list of ints, host),
    `description_lengths` (host ints) and `description_order` (int64 tensor: rows by falling description length).
    Every per-pair entry (tensors with leading dimension B, lists of length B) is permuted the same way.
    The lengths feed ContrastiveTrainer(trim_padding=True), which cuts `ids[:, :T_s]`: that is only valid for RIGHT-padded
    rows (the contract of the contrastive batch, dataset/dataloader.py:113-123,139-149), so both masks are checked to be
    prefixes here, on the host; a left-padded batch (dataset/dataloader_derived.py:142-147 can build one) is refused."""
    mask = batch["protein_attention_mask"]
    if mask.is_cuda:
        raise ValueError("sort_batch_by_length works on the host batch (before the H2D copy)")
    for key in ("protein_attention_mask", "description_attention_mask"):
        m = batch.get(key)
        if m is not None and m.dim() == 2 and m.shape[1] > 1 and not bool((m[:, :-1] >= m[:, 1:]).all()):
            raise ValueError(f"{key} is not right-padded (a row has a 1 after a 0): trimming by length would cut valid tokens")
    lengths = mask.sum(dim=1)
    order = torch.argsort(lengths, descending=descending, stable=True)
    B = int(mask.shape[0])
    idx = order.tolist()
    out: Dict[str, Any] = {}
    for k, v in batch.items():
        if torch.is_tensor(v) and v.dim() >= 1 and v.shape[0] == B:
            out[k] = v[order]
        elif isinstance(v, (list, tuple)) and len(v) == B:
            out[k] = [v[i] for i in idx]
        else:
            out[k] = v
    out["protein_lengths"] = [int(lengths[i]) for i in idx]
    if "description_attention_mask" in out:
        # the text tower may run in its own length order (rows are independent there): lengths in the new batch order
        # (host) and the rows by falling length (a tensor, so it travels to the device with the batch)
        dlen = out["description_attention_mask"].sum(dim=1)
        out["description_lengths"] = [int(v) for v in dlen]
        out["description_order"] = torch.argsort(dlen, descending=True, stable=True)
    return out


class DevicePrefetcher:
    """Iterate device-resident batches one step ahead of the consumer: tensors go through pinned host memory and are
    copied on a dedicated stream; the consumer's stream only waits on the copy event of the batch it is handed.
    On a CPU-only host it degrades to a plain iterator (tests)."""

    def __init__(self, batches: Iterable[Dict[str, Any]], device: torch.device | str,
                 transform: Optional[Callable[[Dict[str, Any]], Dict[str, Any]]] = None):
        self.batches, self.device, self.transform = batches, torch.device(device), transform
        self.cuda = self.device.type == "cuda" and torch.cuda.is_available()
        self.copy_stream = torch.cuda.Stream(device=self.device) if self.cuda else None

    def _stage(self, batch: Dict[str, Any]):
        if self.transform is not None:
            batch = self.transform(batch)
        if not self.cuda:
            return {k: (v.to(self.device) if torch.is_tensor(v) else v) for k, v in batch.items()}, None
        out = {}
        with torch.cuda.stream(self.copy_stream):
            for k, v in batch.items():
                out[k] = v.pin_memory().to(self.device, non_blocking=True) if torch.is_tensor(v) else v
            ready = self.copy_stream.record_event()
        return out, ready

    def __iter__(self) -> Iterator[Dict[str, Any]]:
        it = iter(self.batches)
        try:
            nxt = self._stage(next(it))
        except StopIteration:
            return
        while nxt is not None:
            cur, ready = nxt
            try:
                nxt = self._stage(next(it))                 # enqueue the next copy before handing out this batch
            except StopIteration:
                nxt = None
            if ready is not None:
                torch.cuda.current_stream(self.device).wait_event(ready)
                for v in cur.values():
                    if torch.is_tensor(v):
                        v.record_stream(torch.cuda.current_stream(self.device))
            yield cur
