"""The contrastive-alignment step: host-side mirror of the reference's scripts/train_contrast.py.

Same public names and argument meaning as the reference functions they replace:

    BatchInfoNCELoss / SegmentedBatchInfoNCELoss     scripts/train_contrast.py:72-114
    readout_embeddings                               :198-248
    get_sequence_embeddings                          :251-281  (ESM2 variant: ids + mask)
    get_description_embeddings                       :284-310
    teacher_forcing_forward_pass                     :313-379

plus `ContrastiveTrainer`, the fused step the benchmark times: text tower (16 layers) -> ESM2 ->
adapter -> readout -> normalise -> InfoNCE -> adapter backward -> clip + AdamW, enqueued without a
host sync, with the batch sharded over ranks (each rank encodes its slice, text embeddings are
all-gathered over RCCL so every rank scores its rows against the GLOBAL batch).

All arithmetic runs in libp2t_hip.so; autograd.Functions below only wire the hand-written
backward kernels into torch's graph so `loss.backward()` works in an existing training loop.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Any, Dict, Literal, Optional

import torch
import torch.distributed as dist
from torch import nn

from . import _lib, ops, sharding
from ._lib import call
from .ops import ptr, round_up, stream


# ---------------------------------------------------------------------------------------------
# autograd wiring
# ---------------------------------------------------------------------------------------------
class _ReadoutFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, emb, mask, mode):
        out = ops.readout(emb, mask, mode)
        if mode in ("std", "mix"):
            pooled = out if mode == "mix" else ops.readout(emb, mask, "mix")
        else:
            pooled = None
        ctx.mode, ctx.mask = mode, mask
        ctx.save_for_backward(emb, pooled) if pooled is not None else ctx.save_for_backward(emb)
        return out

    @staticmethod
    def backward(ctx, d_out):
        saved = ctx.saved_tensors
        emb, pooled = saved[0], (saved[1] if len(saved) > 1 else None)
        d_emb = ops.readout_backward(emb, ctx.mask, ctx.mode, pooled, d_out.float())
        return (d_emb if emb.dtype == torch.float32 else ops.cast(d_emb, emb.dtype)), None, None


class _NormalizeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return ops.l2norm_rows(x)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return ops.l2norm_rows_backward(x, dy)


class _InfoNCEFn(torch.autograd.Function):
    """(1 - cw) * row term + cw * column term.  Row term: the reference's SegmentedBatchInfoNCELoss.  Column term (cw > 0):
    mean over the segment's OWN columns of logsumexp_i(l_ij) - l_jj with i over `all1` (every first-side embedding of the
    batch, detached); the backward gives each segment row its share of the column term of EVERY column, so that the
    mean over segments, as teacher_forcing_forward_pass forms it, has the exact gradient of the whole-batch loss."""

    @staticmethod
    def forward(ctx, seg, batch, labels, temperature, all1, cw):
        loss, logits = ops.infonce_forward(seg, batch, labels, temperature, 1.0 - cw)
        col_lse = None
        if cw > 0.0:
            _, col_lse = ops.infonce_col_forward(all1, batch, temperature, cols=labels, weight=cw, loss_out=loss, accumulate=True)
        ctx.save_for_backward(batch, labels, logits, col_lse) if col_lse is not None else ctx.save_for_backward(batch, labels, logits)
        ctx.temperature, ctx.cw = temperature, cw
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        batch, labels, logits = ctx.saved_tensors[:3]
        d = ops.infonce_backward(batch, labels, logits, ctx.temperature, 1.0 - ctx.cw)
        if ctx.cw > 0.0:
            ops.infonce_col_backward(batch, labels, logits, ctx.saved_tensors[3], ctx.temperature, ctx.cw / logits.shape[0], d_seg=d)
        call("p2t_scale_by_device_scalar", ptr(d), d.numel(), ptr(g.float().reshape(1).contiguous()), stream())
        return d, None, None, None, None, None


def l2_normalize(x: torch.Tensor) -> torch.Tensor:
    """torch.nn.functional.normalize(x, p=2, dim=-1) for [rows, cols] embeddings (train_contrast.py:354,365)."""
    if x.dim() != 2:
        raise ValueError("l2_normalize expects pooled [batch, dim] embeddings")
    x = x.float().contiguous()
    return _NormalizeFn.apply(x) if x.requires_grad and torch.is_grad_enabled() else ops.l2norm_rows(x)


def readout_embeddings(embeddings: torch.Tensor, attention_mask: Optional[torch.Tensor],
                       readout_fn: Literal["last", "mean", "std", "mix"]) -> torch.Tensor:
    """Readout over the sequence axis given the attention mask -> f32 (bsz, hidden) or (bsz, 2*hidden).
    "last" assumes right padding, "std" is the population std with no eps (reference :207-248)."""
    if readout_fn not in ("last", "mean", "std", "mix"):
        raise ValueError(f"readout_fn must be one of 'last', 'mean', 'std', 'mix', got {readout_fn!r}")
    if embeddings.dim() != 3:
        raise ValueError("embeddings must be (bsz, seq_len, hidden_dim)")
    emb = embeddings
    if emb.stride(2) != 1 or emb.stride(0) != emb.shape[1] * emb.stride(1):
        emb = emb.contiguous()
    if emb.requires_grad and torch.is_grad_enabled():
        return _ReadoutFn.apply(emb, attention_mask, readout_fn)
    return ops.readout(emb, attention_mask, readout_fn)


class SegmentedBatchInfoNCELoss(nn.Module):
    """Row-wise InfoNCE of a segment against the whole batch (reference :94-114):
    logits = seg @ batch.T / temperature; loss = -mean_i log softmax(logits_i)[labels_i].

    `column_weight` (default 0 = the reference's row-only loss) adds the column (text -> protein) term named by
    BASELINE.json north_star: loss = (1 - cw) * row + cw * column, column = F.cross_entropy(logits^T, arange) -- the
    reference's own BatchInfoNCELoss with swapped arguments (:72-91) -- restricted to the segment's own columns; it needs
    `batch_output1`, the first-side embeddings of the WHOLE batch (values only).  Gradients reach `segment_output1` only:
    the text side is frozen on this path (:186-187, :348-354)."""

    def __init__(self, temperature: float = 0.05, column_weight: float = 0.0):
        super().__init__()
        if not 0.0 <= column_weight <= 1.0:
            raise ValueError("column_weight must lie in [0, 1]")
        self.temperature, self.column_weight = temperature, float(column_weight)

    def forward(self, segment_output1: torch.Tensor, batch_output2: torch.Tensor, labels: torch.Tensor,
                batch_output1: Optional[torch.Tensor] = None):
        if segment_output1.dim() != 2 or batch_output2.dim() != 2 or segment_output1.shape[1] != batch_output2.shape[1]:
            raise ValueError("expected segment_output1 (segment_size, dim) and batch_output2 (bsz, dim)")
        if labels.numel() != segment_output1.shape[0]:
            raise ValueError("labels must hold one target index per segment row")
        if batch_output2.requires_grad and torch.is_grad_enabled():
            raise NotImplementedError("gradients flow to the first argument only (the text side is frozen on this path)")
        seg = segment_output1.float().contiguous()
        bat = batch_output2.detach().float().contiguous()
        all1 = None
        if self.column_weight > 0.0:
            if batch_output1 is None or tuple(batch_output1.shape) != tuple(bat.shape):
                raise ValueError("the column term needs batch_output1: the (bsz, dim) first-side embeddings of the whole batch")
            all1 = batch_output1.detach().float().contiguous()
        return _InfoNCEFn.apply(seg, bat, labels, float(self.temperature), all1, self.column_weight)


class BatchInfoNCELoss(nn.Module):
    """In-batch InfoNCE, positives on the diagonal (reference :72-91).  `symmetric=True` (off by default, not in the
    reference loop): the mean of the row and the column cross-entropy, 0.5 * (L(out1, out2) + L(out2, out1))."""

    def __init__(self, temperature: float = 0.05, symmetric: bool = False):
        super().__init__()
        self.temperature, self.symmetric = temperature, bool(symmetric)

    def forward(self, batch_output1: torch.Tensor, batch_output2: torch.Tensor):
        if batch_output1.shape != batch_output2.shape:
            raise ValueError("BatchInfoNCELoss expects two (bsz, dim) tensors of the same shape")
        labels = torch.arange(batch_output1.shape[0], device=batch_output1.device)
        cw = 0.5 if self.symmetric else 0.0
        return SegmentedBatchInfoNCELoss(self.temperature, cw)(batch_output1, batch_output2, labels,
                                                               batch_output1 if self.symmetric else None)


# ---------------------------------------------------------------------------------------------
# embedding getters + the step, as the reference spells them
# ---------------------------------------------------------------------------------------------
def get_sequence_embeddings(model, protein_input_ids: torch.Tensor, protein_attention_mask: torch.Tensor,
                            readout_fn: str = "mix", ones_mask: bool = False) -> torch.Tensor:
    """Pooled adapter outputs for contrastive learning (reference :251-281).  `ones_mask=True` reproduces
    the fork's all-ones readout mask (:269-275); the default pools valid tokens only."""
    adapter_output, _ = model(protein_input_ids=protein_input_ids, protein_attention_mask=protein_attention_mask,
                              return_adapter_outputs=True)
    mask = None if ones_mask else protein_attention_mask
    return readout_embeddings(adapter_output, mask, readout_fn)


def get_description_embeddings(model, description_input_ids: torch.Tensor, description_attention_mask: torch.Tensor,
                               output_llm_layer: int = 16, readout_fn: str = "mix") -> torch.Tensor:
    """hidden_states[output_llm_layer] of the frozen text tower, pooled (reference :284-310)."""
    decoder = getattr(model, "llama_decoder", None) or model.llm_decoder
    with torch.no_grad():
        outputs = decoder.model(input_ids=description_input_ids, attention_mask=description_attention_mask,
                                use_cache=False, output_attentions=False, output_hidden_states=True, return_dict=True)
        hidden_states = outputs.hidden_states[output_llm_layer]
    return readout_embeddings(hidden_states, description_attention_mask, readout_fn)


_gather_text = sharding.gather_rows       # all-gather of the (no-grad) normalised text embeddings, rank order


def teacher_forcing_forward_pass(rank, model, data_batch: Dict[str, Any], contrastive_num_segments: int, *,
                                 output_llm_layer: int = 16, readout_fn: str = "mix", ones_mask: bool = False,
                                 global_negatives: bool = False, temperature: float = 0.05,
                                 trim_padding: bool = False, trim_multiple: int = 128, column_weight: float = 0.0) -> torch.Tensor:
    """One forward of the contrastive step (reference :313-379); the returned loss carries the autograd
    graph through the adapter.  `global_negatives=True` scores against the all-gathered global batch
    (the reference uses per-rank negatives; with world size 1 the two coincide).
    `trim_padding=True`: each (equal) segment runs at the padded length of ITS longest protein, read from the host-side
    `data_batch["protein_lengths"]` (data.sort_batch_by_length) -- same loss with the mask-aware readout.
    `column_weight` > 0 adds the column (text -> protein) term (SegmentedBatchInfoNCELoss): every segment is encoded first,
    the protein embeddings are all-gathered like the text ones, then the segments are scored."""
    if trim_padding and ones_mask:
        raise ValueError("trim_padding needs the mask-aware readout (ones_mask=False)")
    base = model.module if hasattr(model, "module") else model
    dev = next(base.adapter.parameters()).device
    pid = data_batch["protein_input_ids"].to(dev)
    pmask = data_batch["protein_attention_mask"].to(dev)
    tid = data_batch["description_input_ids"].to(dev)
    tmask = data_batch["description_attention_mask"].to(dev)
    batch_size = pid.shape[0]
    segment_size = batch_size // contrastive_num_segments
    if segment_size * contrastive_num_segments != batch_size:
        print("WARNING: Given batch size is not divisible by the number of segments for contrastive learning.")
    with torch.no_grad():
        description_output = l2_normalize(get_description_embeddings(base, tid, tmask, output_llm_layer, readout_fn))
    offset = 0
    if global_negatives:
        description_output, offset = _gather_text(description_output)
    loss_fn = SegmentedBatchInfoNCELoss(temperature, column_weight)
    acc_loss = torch.zeros([], device=dev)
    lengths = data_batch.get("protein_lengths") if trim_padding else None
    if trim_padding and (lengths is None or len(lengths) != batch_size or (torch.is_tensor(lengths) and lengths.is_cuda)):
        raise ValueError("trim_padding=True needs data_batch['protein_lengths']: one host int per pair")
    if column_weight > 0.0 and segment_size * contrastive_num_segments != batch_size:
        raise ValueError("the column term needs every pair on both sides: batch size must be divisible by the segments")

    def encode_segment(s):
        sl = slice(s * segment_size, (s + 1) * segment_size)
        Ts = pid.shape[1]
        if lengths is not None:
            Ts = min(Ts, round_up(max(1, max(int(v) for v in lengths[sl])), trim_multiple))
        return l2_normalize(get_sequence_embeddings(base, pid[sl, :Ts], pmask[sl, :Ts], readout_fn, ones_mask))

    segs, protein_all = None, None
    if column_weight > 0.0:                  # the column log-sum-exps run over EVERY protein: encode all segments first
        segs = [encode_segment(s) for s in range(contrastive_num_segments)]
        protein_all = torch.cat([x.detach() for x in segs], 0)
        if global_negatives:
            protein_all, _ = sharding.gather_rows(protein_all)
    for s in range(contrastive_num_segments):
        seg = segs[s] if segs is not None else encode_segment(s)
        labels = torch.arange(s * segment_size, (s + 1) * segment_size, device=dev) + offset
        acc_loss = acc_loss + loss_fn(segment_output1=seg, batch_output2=description_output, labels=labels,
                                      batch_output1=protein_all)
    return acc_loss / contrastive_num_segments


# ---------------------------------------------------------------------------------------------
# fused training step
# ---------------------------------------------------------------------------------------------
def plan_length_segments(lengths, T: int, *, multiple: int = 128, max_tokens: Optional[int] = None,
                         floor_tokens: int = 4096, overhead_tokens: int = 512):
    """Split rows 0..B-1 (in the given order; sort by length first for the best result) into contiguous segments, each
    run at its own padded length T_s = min(T, round_up(longest row of the segment, multiple)).

    Dynamic programme over the cut points, minimising  sum_s [ max(n_s * T_s, floor_tokens) + overhead_tokens ]:
    n_s * T_s is the segment's padded token count (what the encoder's GEMMs process), `floor_tokens` the size below
    which a launch sequence no longer gets faster on 256 CUs (fewer tiles than CUs; 4096 measured best on the cfg3 step,
    tools/ragged_sweep.py -> profiles/r01_ragged_sweep.log), and
    `overhead_tokens` the per-segment fixed cost (weights re-read from HBM, ~200 launches) in token equivalents.
    `max_tokens` caps a segment (activation memory); a single row is always allowed.  -> [(start, stop, T_s)]."""
    n = len(lengths)
    if n == 0:
        return []
    pad = [min(T, round_up(max(1, int(v)), multiple)) for v in lengths]
    best = [0.0] + [math.inf] * n
    cut = [0] * (n + 1)
    for j in range(1, n + 1):
        longest = 0
        for i in range(j - 1, -1, -1):
            longest = max(longest, pad[i])
            tokens = round_up((j - i) * longest, 256)          # GEMM row tiles
            if max_tokens is not None and tokens > max_tokens and j - i > 1:
                break
            cost = best[i] + max(tokens, floor_tokens) + overhead_tokens
            if cost < best[j]:
                best[j], cut[j] = cost, i
    out, j = [], n
    while j > 0:
        i = cut[j]
        out.append((i, j, max(pad[i:j])))
        j = i
    return out[::-1]


class ContrastiveTrainer:
    """The timed step: forward + adapter backward + clip + AdamW in one enqueue sequence.

    Adapter parameters are kept as fp32 masters (with fp32 Adam moments) in ONE flat buffer -- so
    the multi-GPU gradient exchange is a single RCCL all-reduce per optimizer step -- plus GEMM-layout
    copies in the tower dtype that the AdamW kernel refreshes.  `sync_to_module()` writes the masters
    back into `model.adapter` (checkpoint keys fc1/fc2.weight/bias as upstream).
    Hyper-parameters default to the reference's: AdamW(lr=2e-4, eps=1e-6, betas=(0.9, 0.999)), weight decay
    0.01 (torch default), no clipping, dropout from the adapter config (train mode).
    """

    def __init__(self, model, *, lr=2e-4, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.01, max_norm=None,
                 num_segments: int = 1, output_llm_layer: int = 16, readout_fn: str = "mix", ones_mask: bool = False,
                 temperature: float = 0.05, train_mode: bool = True, global_negatives: bool = True, process_group=None,
                 overlap_streams: Optional[bool] = None, schedule=None, trim_padding: bool = False, trim_multiple: int = 128,
                 trim_floor_tokens: int = 4096, gradient_accumulation_steps: int = 1, column_weight: float = 0.0,
                 schedule_step: str = "epoch"):
        if trim_padding and ones_mask:
            raise ValueError("trim_padding needs the mask-aware readout: with ones_mask=True (the fork's quirk, "
                             "train_contrast.py:269-275) the padded positions are part of the result")
        if trim_multiple <= 0 or trim_multiple % 64:
            raise ValueError("trim_multiple must be a positive multiple of 64")
        if not 0.0 <= column_weight <= 1.0:
            raise ValueError("column_weight must lie in [0, 1]")
        if schedule_step not in ("epoch", "step"):
            raise ValueError("schedule_step must be 'epoch' (the reference: scheduler.step() once per epoch) or 'step'")
        self.column_weight, self.schedule_step = float(column_weight), schedule_step
        self.trim_padding, self.trim_multiple, self.trim_floor_tokens = trim_padding, trim_multiple, trim_floor_tokens
        self.model = model
        self.schedule = schedule                   # training_state.CosineWarmupSchedule or None (constant lr)
        self.hp = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay,
                       max_norm=math.inf if max_norm is None else max_norm)
        self.num_segments, self.layer, self.readout_fn = num_segments, output_llm_layer, readout_fn
        self.ones_mask, self.temperature, self.train_mode = ones_mask, temperature, train_mode
        self.global_negatives, self.group = global_negatives, process_group
        # towers / segments on separate HIP streams: default on for ragged segments (short segments leave CUs idle: +8 %
        # measured), off otherwise (+2.4 %, but per-kernel timings then include the co-runner -- bench.py measures without)
        self.overlap_streams, self._streams = (trim_padding if overlap_streams is None else bool(overlap_streams)), {}
        self._pf, self._next = None, None            # towers of the next batch already in flight (prefetch_towers) / the batch step() announced
        if gradient_accumulation_steps < 1:
            raise ValueError("gradient_accumulation_steps must be >= 1")
        self.gradient_accumulation_steps, self._micro = int(gradient_accumulation_steps), 0
        self.step_count = 0
        ad = model.adapter
        c = ad.config
        self.c = c
        dev = ad.fc1.weight.device
        self.dev = dev
        self.tdt = model.esm_encoder.dtype
        shapes = [(c.intermediate_dim, c.input_dim), (c.intermediate_dim,), (c.output_dim, c.intermediate_dim), (c.output_dim,)]
        sizes = [math.prod(s) for s in shapes]
        tot = sum(sizes)
        self.flat_p = torch.empty((tot,), dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros((tot,), dtype=torch.float32, device=dev)
        self.flat_m = torch.zeros((tot,), dtype=torch.float32, device=dev)
        self.flat_v = torch.zeros((tot,), dtype=torch.float32, device=dev)
        offs = [0]
        for n in sizes:
            offs.append(offs[-1] + n)
        view = lambda flat: [flat[offs[i]:offs[i + 1]].view(shapes[i]) for i in range(4)]
        self.p, self.g, self.m, self.v = view(self.flat_p), view(self.flat_g), view(self.flat_m), view(self.flat_v)
        with torch.no_grad():
            for dst, src in zip(self.p, (ad.fc1.weight, ad.fc1.bias, ad.fc2.weight, ad.fc2.bias)):
                dst.copy_(src.detach().float())
        K1, ld1 = round_up(c.input_dim, 64), round_up(c.intermediate_dim, 64)
        self.w1 = torch.zeros((c.intermediate_dim, K1), dtype=self.tdt, device=dev)
        self.w2 = torch.zeros((c.output_dim, ld1), dtype=self.tdt, device=dev)
        self.w1[:, : c.input_dim].copy_(self.p[0])
        self.w2[:, : c.intermediate_dim].copy_(self.p[2])
        self.scratch = torch.empty((256 * 4,), dtype=torch.float32, device=dev)
        self.grad_norm = torch.zeros((1,), dtype=torch.float32, device=dev)
        self.loss = torch.zeros((1,), dtype=torch.float32, device=dev)
        self._bufs = {}
        self._col_lse = None

    # ------------------------------------------------------------------------------------------
    def _buffers(self, Bs: int, T: int, slot: int = 0):
        """Adapter activations / gradients for M = Bs * T rows: ONE grow-only set (per slot) sized for the largest M seen,
        handed out as leading-row views, so ragged segment shapes (trim_padding) do not multiply the footprint.  Slots > 0
        exist only with the column term, where every segment's forward state lives until its backward."""
        M = Bs * T
        bufs = self._bufs.setdefault(slot, {})
        cap = bufs.get("cap", 0)
        if M > cap:
            c, dev, dt = self.c, self.dev, self.tdt
            ld1, ld2 = round_up(c.intermediate_dim, 64), round_up(c.output_dim, 64)
            e = lambda *s, d=dt: torch.empty(s, dtype=d, device=dev)
            full = dict(z1=e(M, ld1), h1=e(M, ld1), z2=e(M, ld2), g2=e(M, ld2), y=e(M, ld2),
                        inv=e(M, d=torch.float32), dY=e(M, c.output_dim, d=torch.float32))
            cfg = _lib.AdapterConfigC(input_dim=c.input_dim, intermediate_dim=c.intermediate_dim, output_dim=c.output_dim,
                                      dropout_p=0.0, dropout_seed=0, dtype=ops.dt_of(dt))
            nbytes = call("p2t_adapter_backward_workspace_bytes", C.byref(cfg), M)
            full["ws"] = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
            full["saved"] = _lib.AdapterSavedC(z1=full["z1"].data_ptr(), h1=full["h1"].data_ptr(), z2=full["z2"].data_ptr(),
                                               g2=full["g2"].data_ptr(), inv_norm=full["inv"].data_ptr())
            bufs.clear()
            bufs.update({"cap": M, "full": full})
        full = bufs["full"]
        b = {k: (v[:M] if k in ("z1", "h1", "z2", "g2", "y", "inv", "dY") else v) for k, v in full.items()}
        return b

    def _stream(self, i: int) -> "torch.cuda.Stream":
        if i not in self._streams:
            self._streams[i] = torch.cuda.Stream(device=self.dev)
        return self._streams[i]

    def text_embeddings(self, tid, tmask, batch: Optional[Dict[str, Any]] = None) -> torch.Tensor:
        """L2-normalised pooled hidden_states[layer] of the descriptions, rows in batch order.
        trim_padding with `description_order` (device int64 [B]: rows by falling description length) and
        `description_lengths` (host ints, batch order) in the batch -- data.sort_batch_by_length adds both, consistently:
        the order must be the argsort of the lengths, which only a host copy lets this method verify -- runs the
        text tower in that order, cut into segments at their own padded length like the protein side, and scatters
        the pooled rows back; rows are independent in the text tower (causal attention within a row, padding masked)."""
        order = batch.get("description_order") if (batch is not None and self.trim_padding) else None
        lengths = batch.get("description_lengths") if order is not None else None
        if order is None or lengths is None:
            hs = self.model.llama_decoder.model.hidden_state(tid, tmask, self.layer)
            return ops.l2norm_rows(ops.readout(hs, tmask, self.readout_fn))
        B, T = tid.shape
        if torch.is_tensor(lengths):
            lengths = lengths.tolist()
        if len(lengths) != B or order.numel() != B or max(int(v) for v in lengths) > T:
            raise ValueError("description_lengths / description_order do not match the description batch")
        by_len = sorted((int(v) for v in lengths), reverse=True)        # the lengths in `order`'s row order
        if not order.is_cuda and [int(lengths[i]) for i in order.tolist()] != by_len:   # host copy: cheap to verify
            raise ValueError("description_order must list the rows by falling description_lengths")
        order = order.to(device=tid.device, dtype=torch.int64)
        ids, mask = tid.index_select(0, order), tmask.index_select(0, order)
        D = self.model.llama_decoder.model.spec.hidden_size * (2 if self.readout_fn == "mix" else 1)
        t_sorted = torch.empty((B, D), dtype=torch.float32, device=tid.device)
        for a, b, Ts in plan_length_segments(by_len, T, multiple=min(self.trim_multiple, 64), floor_tokens=self.trim_floor_tokens // 2):
            hs = self.model.llama_decoder.model.hidden_state(ids[a:b, :Ts].contiguous(), mask[a:b, :Ts].contiguous(), self.layer)
            t_sorted[a:b] = ops.l2norm_rows(ops.readout(hs, mask[a:b, :Ts], self.readout_fn))
        return torch.empty_like(t_sorted).index_copy_(0, order, t_sorted)

    def forward_backward(self, batch: Dict[str, torch.Tensor], accumulate: bool = False, backward: bool = True,
                         grad_scale: float = 1.0, reduce: bool = True) -> torch.Tensor:
        """Loss (device scalar, f32 [1]) and adapter gradients into self.g (summed over segments,
        all-reduce-averaged over ranks).  No host synchronisation.  backward=False stops after the loss.
        accumulate: add the gradients to self.g instead of overwriting (the loss is always this batch's own);
        grad_scale: factor on the gradients only (1 / gradient_accumulation_steps: `loss / GA` before `backward()`,
        train_contrast.py:428-448); reduce=False leaves the cross-rank average to a later call (sums commute with it).
        The rank logic (gather, label offsets, gradient average) is sharding.sharded_forward_backward; the closures below
        are the kernels between its collectives."""
        m, c = self.model, self.c
        pid, pmask = batch["protein_input_ids"], batch["protein_attention_mask"]
        tid, tmask = batch["description_input_ids"], batch["description_attention_mask"]
        B, T = pid.shape
        segs = self._segments(batch, B, T)
        cw = self.column_weight
        if cw > 0.0 and (segs[0][0] != 0 or segs[-1][1] != B or tid.shape[0] != B):
            raise ValueError("the column term needs every pair on both sides: batch size must be divisible by num_segments")
        # packed weight copies are built on the CALLER's stream, before any side stream reads them
        m.esm_encoder.ensure_engine()
        m.llama_decoder.model.ensure_engine(self.layer)
        # The text tower and the ESM2 encodes of the segments are independent until the loss: they are enqueued on
        # separate HIP streams so the tail of one kernel's grid (M = 2048 text GEMMs, 2.5-"round" encoder GEMMs)
        # is filled by another stream's blocks instead of idling CUs.  Everything joins on the caller's stream.
        main = torch.cuda.current_stream()
        encs, pre = {}, {}
        pf, self._pf = self._pf, None
        if pf is not None and pf["batch"] is batch and pf["segs"] == segs:
            encs, pre = pf["encs"], pf["pre"]          # the frozen towers of THIS batch were enqueued during the previous step's tail
        elif self.overlap_streams:
            encs, pre = self._launch_towers(batch, segs, main.record_event())
        if self._next is not None and backward:
            # (the side streams run it behind this batch's towers: it meets the backward / optimizer tail on the caller's stream)
            self.prefetch_towers(self._next, getattr(self, "_next_ready", None))

        def text_fn():
            if pre:
                main.wait_event(pre["ev"])
                return pre["text"]
            return self.text_embeddings(tid, tmask, batch)

        p_drop = float(m.adapter.dropout.p) if self.train_mode else 0.0
        wts = _lib.AdapterWeightsC(fc1_w=self.w1.data_ptr(), fc1_b=self.p[1].data_ptr(), fc2_w=self.w2.data_ptr(),
                                   fc2_b=self.p[3].data_ptr())
        mode = self.readout_fn
        states = {}

        def seg_forward(s, r0, r1, Ts, slot):
            """encoder -> adapter forward -> readout -> normalise for rows r0..r1 at padded length Ts; activations in
            buffer set `slot` (one shared set when segments are finished one after the other)."""
            sl, Bs = slice(r0, r1), r1 - r0
            ids_s, mask_s = pid[sl, :Ts], pmask[sl, :Ts]
            if s in encs:
                enc, ev = encs[s]
                main.wait_event(ev)
            else:
                enc = m.esm_encoder.encode(ids_s, mask_s)                              # [Bs, Ts, Hp]
            Hp, M = enc.shape[2], Bs * Ts
            b = self._buffers(Bs, Ts, slot)
            seed = m.adapter._next_seed() if p_drop > 0 else 0
            cfg = _lib.AdapterConfigC(input_dim=c.input_dim, intermediate_dim=c.intermediate_dim, output_dim=c.output_dim,
                                      dropout_p=p_drop, dropout_seed=seed, dtype=ops.dt_of(self.tdt))
            call("p2t_adapter_forward", C.byref(cfg), C.byref(wts), ptr(enc), Hp, M, ptr(b["y"]), C.byref(b["saved"]), stream())
            y3 = b["y"].view(Bs, Ts, -1)
            rmask = None if self.ones_mask else mask_s
            pooled_mix = ops.readout(y3, rmask, "mix", D=c.output_dim) if mode in ("std", "mix") else None
            pooled = pooled_mix if mode == "mix" else ops.readout(y3, rmask, mode, D=c.output_dim)
            return dict(enc=enc, Hp=Hp, M=M, b=b, cfg=cfg, y3=y3, rmask=rmask, pooled_mix=pooled_mix, pooled=pooled,
                        p=ops.l2norm_rows(pooled), Bs=Bs, Ts=Ts)

        def protein_fn():
            """Column term only: the forward of EVERY segment first (activations kept per segment for the backward)."""
            for s, (r0, r1, Ts, _) in enumerate(segs):
                states[s] = seg_forward(s, r0, r1, Ts, slot=s)
            return torch.cat([states[s]["p"] for s in range(len(segs))], 0)

        def column_fn(p_all, t_all, offset):
            _, self._col_lse = ops.infonce_col_forward(p_all, t_all, self.temperature, first=offset, count=B, weight=cw,
                                                       loss_out=self.loss, accumulate=False)

        def segment_fn(s, r0, r1, Ts, weight, t_all, labels, offset):
            st = states.pop(s) if s in states else seg_forward(s, r0, r1, Ts, slot=0)
            _, logits = ops.infonce_forward(st["p"], t_all, labels, self.temperature, weight * (1.0 - cw), self.loss,
                                            accumulate=s > 0 or cw > 0.0)
            if not backward:
                return
            dp = ops.infonce_backward(t_all, labels, logits, self.temperature, weight * (1.0 - cw) * grad_scale)
            if cw > 0.0:      # every column's log-sum-exp depends on this row: scale cw * weight / n_s = cw / B_loc per row
                ops.infonce_col_backward(t_all, labels, logits, self._col_lse, self.temperature,
                                         cw * weight * grad_scale / st["Bs"], d_seg=dp)
            dpooled = ops.l2norm_rows_backward(st["pooled"], dp)
            b, y3, rmask = st["b"], st["y3"], st["rmask"]
            call("p2t_readout_backward", ptr(y3), ops.dt_of(y3), y3.stride(1), ptr(rmask.to(torch.int64).contiguous()) if rmask is not None else None,
                 st["Bs"], st["Ts"], c.output_dim, _lib.READOUT[mode], ptr(st["pooled_mix"]), ptr(dpooled), ptr(b["dY"]), stream())
            call("p2t_adapter_backward", C.byref(st["cfg"]), C.byref(wts), ptr(st["enc"]), st["Hp"], st["M"], C.byref(b["saved"]), ptr(b["dY"]),
                 ptr(self.g[0]), ptr(self.g[1]), ptr(self.g[2]), ptr(self.g[3]), int(s > 0 or accumulate), ptr(b["ws"]),
                 b["ws"].numel(), stream())

        def prefetch_fn(s, r0, r1, Ts):
            """The first segment's encoder -> adapter -> readout, enqueued before the wait for the gathered text embeddings."""
            states[s] = seg_forward(s, r0, r1, Ts, slot=0)

        sharding.sharded_forward_backward(text_fn=text_fn, segment_fn=segment_fn, segments=segs,
                                          global_negatives=self.global_negatives, flat_g=self.flat_g, backward=backward,
                                          reduce=reduce, group=self.group, protein_fn=protein_fn if cw > 0.0 else None,
                                          column_fn=column_fn if cw > 0.0 else None,
                                          prefetch_fn=prefetch_fn if (cw == 0.0 and self.global_negatives and sharding.world_info(self.group)[1] > 1) else None)
        return self.loss

    def _launch_towers(self, batch, segs, start):
        """Text tower on side stream 0, the encoder of segment s on side stream 1 + s % 2; `start`: an event of the caller's stream
        the side streams wait for (None: nothing of the caller's stream feeds them -- `prefetch_towers`)."""
        m, main = self.model, torch.cuda.current_stream()
        pid, pmask = batch["protein_input_ids"], batch["protein_attention_mask"]
        tid, tmask = batch["description_input_ids"], batch["description_attention_mask"]
        encs, pre = {}, {}
        with torch.cuda.stream(self._stream(0)):
            if start is not None:
                torch.cuda.current_stream().wait_event(start)
            pre["text"] = self.text_embeddings(tid, tmask, batch)
            pre["text"].record_stream(main)
            pre["ev"] = torch.cuda.current_stream().record_event()
        for s, (r0, r1, Ts, _) in enumerate(segs):
            with torch.cuda.stream(self._stream(1 + s % 2)):
                if start is not None:
                    torch.cuda.current_stream().wait_event(start)
                enc = m.esm_encoder.encode(pid[r0:r1, :Ts], pmask[r0:r1, :Ts])
                enc.record_stream(main)
                encs[s] = (enc, torch.cuda.current_stream().record_event())
        return encs, pre

    def prefetch_towers(self, next_batch: Dict[str, torch.Tensor], ready: Optional["torch.cuda.Event"] = None):
        """Enqueue the FROZEN towers of the next batch now, on the side streams: neither the ESM2 encoder nor the text tower depends
        on the optimizer step that is still running on the caller's stream, so the tail of this step (adapter backward, readout,
        loss, clip + AdamW: small grids and HBM-bound passes that leave most CUs idle -- and, across ranks, the gradient
        all-reduce) overlaps with the next step's first GEMMs.  The next `step(next_batch)` (the same dict object) picks the
        results up; any other batch discards them.  `ready`: an event after which next_batch's tensors are valid (e.g. the copy
        stream's event of a DevicePrefetcher); None = they already are.  A no-op without overlap_streams."""
        if not self.overlap_streams or next_batch is None:
            return
        B, T = next_batch["protein_input_ids"].shape
        segs = self._segments(next_batch, B, T)
        self.model.esm_encoder.ensure_engine()
        self.model.llama_decoder.model.ensure_engine(self.layer)
        if ready is not None:
            for i in range(3):
                self._stream(i).wait_event(ready)
        encs, pre = self._launch_towers(next_batch, segs, None)
        self._pf = dict(batch=next_batch, segs=segs, encs=encs, pre=pre)

    def global_loss(self, loss: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Mean over ranks of `loss` (default: the last step's) = the loss of the global batch; one tiny all-reduce, no host
        sync.  `step()` itself returns the rank-local loss, as the reference's teacher_forcing_forward_pass does -- the
        reference only all-reduces its epoch sums (train_contrast.py:468), which is what loop.train_epoch does too."""
        return sharding.average_loss(self.loss if loss is None else loss, self.group)

    def _segments(self, batch, B: int, T: int):
        """[(start, stop, T_s, weight)]: the row ranges of the protein side, the padded length each runs at and its
        share of the loss.  Default: `num_segments` equal ranges at the batch's T, weight 1 / num_segments (tail rows
        beyond num_segments * (B // num_segments) are dropped from the protein side, as upstream :339-343).
        trim_padding: ranges from plan_length_segments over the HOST-side `protein_lengths` of the batch
        (data.sort_batch_by_length) -- never from a device read, which would serialise host and GPU every step --
        capped at the token count of one default segment; weight n_s / B, so the total is the mean over all rows,
        which is what equal segments average to."""
        nseg = self.num_segments
        Bs = B // nseg
        if not self.trim_padding:
            return [(s * Bs, (s + 1) * Bs, T, 1.0 / nseg) for s in range(nseg)]
        lengths = batch.get("protein_lengths")
        if lengths is None:
            raise ValueError("trim_padding=True needs batch['protein_lengths'] (host ints; data.sort_batch_by_length adds it)")
        if torch.is_tensor(lengths):
            if lengths.is_cuda:
                raise ValueError("batch['protein_lengths'] must live on the host")
            lengths = lengths.tolist()
        if len(lengths) != B:
            raise ValueError(f"batch['protein_lengths'] has {len(lengths)} entries for a batch of {B}")
        if max(int(v) for v in lengths) > T:
            raise ValueError(f"protein_lengths says {max(int(v) for v in lengths)} but the batch is only {T} wide")
        plan = plan_length_segments(lengths, T, multiple=self.trim_multiple, max_tokens=-(-B // nseg) * T,
                                    floor_tokens=self.trim_floor_tokens)
        return [(a, b, Ts, (b - a) / B) for a, b, Ts in plan]

    def optimizer_step(self):
        """clip_grad_norm_ -> AdamW.step -> zero_grad (train_contrast.py:453-465) at the schedule's current learning rate.
        The reference advances its LambdaLR ONCE PER EPOCH (`scheduler.step()` after `train_epoch`, :654-662), although
        the schedule is sized in optimizer steps (:626-637): call `end_epoch()` where the reference does.
        `schedule_step="step"` advances it here instead, after every optimizer step (the conventional reading)."""
        self.step_count += 1
        hp = dict(self.hp)
        if self.schedule is not None:
            hp["lr"] = self.schedule.lr()
        ops.clip_adamw_step(self.p, self.g, self.m, self.v, self.step_count, shadows=[self.w1, None, self.w2, None],
                            scratch=self.scratch, grad_norm_out=self.grad_norm, **hp)
        if self.schedule is not None and self.schedule_step == "step":
            self.schedule.step()
        return self.grad_norm

    def end_epoch(self):
        """`scheduler.step()` of the reference's epoch loop (train_contrast.py:662); a no-op with schedule_step="step"."""
        if self.schedule is not None and self.schedule_step == "epoch":
            self.schedule.step()

    def step(self, batch: Dict[str, torch.Tensor], next_batch: Optional[Dict[str, torch.Tensor]] = None, next_ready=None) -> torch.Tensor:
        """One micro-batch of `train_epoch` (train_contrast.py:417-465): gradients of loss / GA are accumulated, and every
        `gradient_accumulation_steps` calls clip + AdamW run.  Returns this batch's (unscaled) loss.
        next_batch (optional, already on the device): its frozen towers are enqueued on the side streams as soon as this step's
        forward has consumed the current ones, i.e. they run beside this step's backward / optimizer tail (`prefetch_towers`);
        next_ready: an event after which its tensors are valid (None: they already are)."""
        accumulate, grad_scale, reduce, do_step = sharding.micro_step_plan(self._micro, self.gradient_accumulation_steps)
        self._next, self._next_ready = next_batch, next_ready
        loss = self.forward_backward(batch, accumulate=accumulate, grad_scale=grad_scale, reduce=reduce)
        self._next = None
        self._micro += 1
        if do_step:
            self.optimizer_step()
            self._micro = 0
        return loss

    @torch.no_grad()
    def evaluate(self, batch: Dict[str, torch.Tensor]) -> torch.Tensor:
        """Forward-only loss (eval mode, no dropout) on this rank's rows."""
        saved = self.train_mode
        self.train_mode = False
        try:
            loss = self.forward_backward(batch, backward=False).clone()
        finally:
            self.train_mode = saved
        return loss

    @torch.no_grad()
    def sync_from_module(self):
        """Module parameters (e.g. a checkpoint just loaded) -> fp32 masters and GEMM-layout copies."""
        ad, c = self.model.adapter, self.c
        for dst, src in zip(self.p, (ad.fc1.weight, ad.fc1.bias, ad.fc2.weight, ad.fc2.bias)):
            dst.copy_(src.detach().float())
        self.w1[:, : c.input_dim].copy_(self.p[0])
        self.w2[:, : c.intermediate_dim].copy_(self.p[2])

    @torch.no_grad()
    def sync_to_module(self):
        ad = self.model.adapter
        for src, dst in zip(self.p, (ad.fc1.weight, ad.fc1.bias, ad.fc2.weight, ad.fc2.bias)):
            dst.copy_(src.to(dst.dtype))
