"""`generate` for the decoder half of Esm2LlamaInstructForCausalLM (reference models/modeling_esm2llama_instruct.py:217-251:
prompt embeddings from `forward(return_decoder_inputs=True)`, then `llama_decoder.generate(inputs_embeds=, attention_mask=,
**kwargs)`; call sites scripts/generate_instruct.py:72-87: max_new_tokens, eos_token_id, pad_token_id, num_beams, length_penalty,
temperature, do_sample, top_p, top_k, return_dict_in_generate=False).

What runs where: the prompt compaction, the prefill (all decoder layers + KV-cache write), every decode step (RMSNorm, the
projections, rotation + cache append, single-query attention over the cache, LM head) and the greedy choice are kernels of
libp2t_hip (csrc/llama_decode.hip).  After the first step the greedy loop replays ONE captured HIP graph per token (every length
lives in device memory), and the host looks at the finished flags only every `sync_every` tokens.  Sampling (temperature / top-k /
top-p filters + multinomial) and the beam bookkeeping (log-softmax, top-k over beams x vocabulary, gathers of the token tables) are
torch device ops on the [rows, vocab] logits -- selection bookkeeping, restated from transformers/generation/utils.py (`_sample`,
`_beam_search` and its helpers) and transformers/generation/logits_process.py; the decoder arithmetic never goes through torch.

Like HF with `inputs_embeds` only, the returned ids hold the NEW tokens only ([batch, <= max_new_tokens])."""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import torch

from . import _lib, ops
from ._lib import call
from .ops import ptr, round_up, stream


class GenerateOutput:
    """`return_dict_in_generate=True`: the fields of HF's GenerateDecoderOnlyOutput / GenerateBeamDecoderOnlyOutput this path fills."""

    def __init__(self, sequences, scores=None, logits=None, sequences_scores=None, beam_indices=None):
        self.sequences, self.scores, self.logits = sequences, scores, logits
        self.sequences_scores, self.beam_indices = sequences_scores, beam_indices
        self.past_key_values = None

    def __getitem__(self, k):
        return getattr(self, k)


def _as_id_list(v) -> list:
    if v is None:
        return []
    if isinstance(v, torch.Tensor):
        return [int(x) for x in v.reshape(-1).tolist()]
    if isinstance(v, (list, tuple)):
        return [int(x) for x in v]
    return [int(v)]


# ---------------------------------------------------------------------------------------------------------------------------
# logits filters of the sampling mode (transformers/generation/logits_process.py: TemperatureLogitsWarper, TopKLogitsWarper,
# TopPLogitsWarper; applied in that order by GenerationMixin._get_logits_processor)
# ---------------------------------------------------------------------------------------------------------------------------
def filter_logits(scores: torch.Tensor, temperature: float = 1.0, top_k: Optional[int] = None, top_p: Optional[float] = None,
                  min_tokens_to_keep: int = 1) -> torch.Tensor:
    """f32 [rows, vocab] -> filtered scores (removed entries = -inf)."""
    if temperature is not None and temperature != 1.0:
        if not temperature > 0:
            raise ValueError(f"`temperature` (={temperature}) has to be a strictly positive float")
        scores = scores / temperature
    if top_k is not None and top_k > 0:
        k = min(max(int(top_k), min_tokens_to_keep), scores.shape[-1])
        kth = torch.topk(scores, k)[0][..., -1, None]
        scores = scores.masked_fill(scores < kth, -float("inf"))
    if top_p is not None and top_p < 1.0:
        if top_p < 0:
            raise ValueError(f"`top_p` has to be a float > 0 and < 1, but is {top_p}")
        sorted_logits, sorted_indices = torch.sort(scores, descending=False)
        cumulative = sorted_logits.softmax(dim=-1).cumsum(dim=-1)
        remove = cumulative <= (1 - top_p)
        remove[..., -min_tokens_to_keep:] = False
        scores = scores.masked_fill(remove.scatter(1, sorted_indices, remove), -float("inf"))
    return scores


# ---------------------------------------------------------------------------------------------------------------------------
def preshuffle(w: torch.Tensor, n: int) -> torch.Tensor:
    """bf16 [>= n, K] (or e4m3 bytes as uint8, K % 128 == 0) -> the stream order of the decode GEMM (p2t_preshuffle_w[_fp8]): the
    1 KB one MFMA consumes is contiguous, whole-line loads."""
    K = w.shape[1]
    out = torch.empty((round_up(n, 16) * K,), dtype=w.dtype, device=w.device)
    call("p2t_preshuffle_w_fp8" if w.dtype == torch.uint8 else "p2t_preshuffle_w", ptr(w), w.stride(0), n, K, ptr(out), stream())
    return out


def stream_weights(decoder):
    """Second copies of every decoder projection + the LM head in the stream order, built once per model (dropped with the engine
    when weights are loaded): Llama-3.1-8B: 15 GB next to the 15 GB the prefill reads -- HBM is sized for it."""
    m, s = decoder.model, decoder.spec
    e = m.ensure_engine(s.num_hidden_layers)
    lm = decoder._lm_head_padded()
    # (the engine object is rebuilt when weights are loaded; _version catches an in-place update of the LM head that kept the engine)
    key = (id(e), lm.data_ptr(), lm._version, decoder.model.embed_tokens.weight._version, bool(m.gemm_fp8))
    st = getattr(m, "_stream_engine", None)
    if st is not None and st["key"] == key:
        return st
    d, nh, nkv = s.head_dim, s.num_attention_heads, s.num_key_value_heads
    rows = dict(qkv_w=(nh + 2 * nkv) * d, o_w=s.hidden_size, gu_w=2 * s.intermediate_size, down_w=s.hidden_size)
    layers, keep = (_lib.LlamaLayerStreamC * s.num_hidden_layers)(), []
    for i in range(s.num_hidden_layers):
        t = e["keep"][i]
        for name, n in rows.items():
            w = preshuffle(t[name], n)
            keep.append(w)
            setattr(layers[i], name, w.data_ptr())
    st = dict(key=key, layers=layers, keep=keep, lm_head=preshuffle(lm, s.vocab_size))
    m._stream_engine = st
    return st


class DecodeEngine:
    """Cache + buffers of ONE generate() call: B0 prompts, `group` rows per prompt (beams), capacities Tp / G (multiples of 64)."""

    def __init__(self, decoder, B0: int, group: int, T: int, max_new_tokens: int, stream_copy: bool = True, fuse_rope: bool = True):
        m, s = decoder.model, decoder.spec
        self.flags = 0 if fuse_rope else 1          # P2T_DECODE_NO_ROPE_FUSION
        if s.hidden_size % 64:
            raise ValueError("the LM head path needs hidden_size % 64 == 0")
        self.decoder, self.spec, self.dtype = decoder, s, m.dtype
        self.dev = m.embed_tokens.weight.device
        self.e = m.ensure_engine(s.num_hidden_layers)
        self.B0, self.group, self.BB = B0, group, B0 * group
        self.Tp, self.G = round_up(max(T, 1), 64), round_up(max(max_new_tokens, 1), 64)
        L, nkv, dp = s.num_hidden_layers, s.num_key_value_heads, ops.head_dim_padded(s.head_dim)
        z = lambda *shape: torch.zeros(shape, dtype=self.dtype, device=self.dev)
        self.k_prompt, self.vt_prompt = z(L, B0, nkv, self.Tp, dp), z(L, B0, nkv, dp, self.Tp)
        self.k_gen, self.vt_gen = z(L, self.BB, nkv, self.G, dp), z(L, self.BB, nkv, dp, self.G)
        self.k_alt = self.vt_alt = None                                  # second generated segment, beam search only
        self.lens = torch.zeros((B0,), dtype=torch.int32, device=self.dev)
        self.step = torch.zeros((1,), dtype=torch.int32, device=self.dev)
        self.ld_logits = round_up(s.vocab_size, 64)
        self.logits = torch.zeros((self.BB, self.ld_logits), dtype=self.dtype, device=self.dev)
        self.x = torch.zeros((self.BB, s.hidden_size), dtype=torch.float32, device=self.dev)
        self.next_tokens = torch.zeros((self.BB,), dtype=torch.int64, device=self.dev)
        self.finished = torch.zeros((self.BB,), dtype=torch.int32, device=self.dev)
        self.out_tokens = torch.zeros((self.BB, self.G), dtype=torch.int64, device=self.dev)
        self.ws = torch.empty((call("p2t_llama_decode_workspace_bytes", C.byref(self.e["cfg"]), self.BB, self.Tp, self.G),), dtype=torch.uint8,
                              device=self.dev)
        self.lm_head = decoder._lm_head_padded()
        # the stream-order copies are read by the skinny GEMMs only (<= 64 rows): beyond that every projection and the LM head
        # run on the row-major weights (the LM head had no row-major fallback and returned wrong logits)
        self.stream = stream_weights(decoder) if (stream_copy and self.dtype == torch.bfloat16 and B0 * group <= 64) else None
        self._cache_struct()

    def _cache_struct(self):
        self.cache = _lib.KvCacheC(k_prompt=self.k_prompt.data_ptr(), vt_prompt=self.vt_prompt.data_ptr(), k_gen=self.k_gen.data_ptr(),
                                   vt_gen=self.vt_gen.data_ptr(), prompt_len=self.lens.data_ptr(), step=self.step.data_ptr(), B0=self.B0,
                                   group=self.group, Tp=self.Tp, G=self.G)

    # -- prompt -------------------------------------------------------------------------------------------------------------
    def compact(self, inputs_embeds: torch.Tensor, attention_mask: torch.Tensor):
        """Valid tokens first (positions = HF's cumsum(mask) - 1 on them); -> (embeds, prefix mask) trimmed to the longest row."""
        B, T, H = inputs_embeds.shape
        x = inputs_embeds.to(device=self.dev, dtype=torch.float32).contiguous()
        mask = attention_mask.to(device=self.dev, dtype=torch.int64).contiguous()
        out, out_mask = torch.empty_like(x), torch.empty_like(mask)
        scratch = torch.empty((B * T,), dtype=torch.int32, device=self.dev)
        call("p2t_compact_rows", ptr(x), ptr(mask), B, T, H, ptr(out), ptr(out_mask), ptr(self.lens), ptr(scratch), stream())
        longest = int(self.lens.max().item())                            # the one host read of the prompt phase
        if longest < 1:
            raise ValueError("every prompt row needs at least one token under its attention mask")
        if longest < T:
            out, out_mask = out[:, :longest].contiguous(), out_mask[:, :longest].contiguous()
        return out, out_mask

    def prefill(self, embeds: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
        """-> logits of the first new token, model dtype [B0, ld] (row b from prompt b's last valid token)."""
        B, T, H = embeds.shape
        cfg = self.e["cfg"]
        ws = torch.empty((call("p2t_llama_prefill_workspace_bytes", C.byref(cfg), B, T),), dtype=torch.uint8, device=self.dev)
        last = torch.empty((B, H), dtype=torch.float32, device=self.dev)
        call("p2t_llama_prefill", C.byref(cfg), C.byref(self.e["w"]), ptr(embeds), ptr(mask), B, T, C.byref(self.cache), ptr(last), ptr(ws),
             ws.numel(), stream())
        a = last if self.dtype == torch.float32 else ops.cast(last, self.dtype)
        return ops.gemm_nt(a, self.lm_head, None, n=self.spec.vocab_size, k=H, out_dtype=self.dtype)

    # -- steps --------------------------------------------------------------------------------------------------------------
    def feed(self, tokens: torch.Tensor):
        """x <- embedding rows of the chosen tokens (i64 [BB], device)."""
        call("p2t_llama_embed_tokens", C.byref(self.e["cfg"]), C.byref(self.e["w"]), ptr(tokens), self.BB, ptr(self.x), stream())

    def decode_step(self):
        """One token per row from self.x -> self.logits; the device step counter advances."""
        st = self.stream
        layers = C.cast(st["layers"], C.POINTER(_lib.LlamaLayerStreamC)) if st else None
        head = st["lm_head"] if st else self.lm_head
        call("p2t_llama_decode_step", C.byref(self.e["cfg"]), C.byref(self.e["w"]), layers, ptr(head), self.lm_head.stride(0), int(st is not None),
             C.byref(self.cache), ptr(self.x), ptr(self.logits), self.ld_logits, self.flags, ptr(self.ws), self.ws.numel(), stream())

    def greedy_select(self, logits: torch.Tensor, eos: torch.Tensor, pad_id: int):
        call("p2t_greedy_select", ptr(logits), ops.dt_of(logits), logits.stride(0), self.spec.vocab_size, self.BB, ptr(eos) if eos.numel() else None,
             eos.numel(), int(pad_id), ptr(self.finished), ptr(self.next_tokens), ptr(self.out_tokens), self.G, ptr(self.step), self.G, stream())

    def reorder(self, src_rows: torch.Tensor):
        """Generated segment row r <- row src_rows[r] (beam re-ordering; the prompt segment is shared by a prompt's beams)."""
        if self.k_alt is None:
            self.k_alt, self.vt_alt = torch.zeros_like(self.k_gen), torch.zeros_like(self.vt_gen)
        call("p2t_kv_reorder", C.byref(self.e["cfg"]), C.byref(self.cache), ptr(src_rows.to(torch.int64).contiguous()), ptr(self.k_alt),
             ptr(self.vt_alt), stream())
        self.k_gen, self.k_alt, self.vt_gen, self.vt_alt = self.k_alt, self.k_gen, self.vt_alt, self.vt_gen
        self._cache_struct()


def _trim(tokens: torch.Tensor, eos_ids: Sequence[int], n: int) -> torch.Tensor:
    """HF stops right after the step in which the last row finishes: cut the columns produced after that."""
    tokens = tokens[:, :n]
    if not eos_ids or n == 0:
        return tokens
    hit = torch.zeros_like(tokens, dtype=torch.bool)
    for e in eos_ids:
        hit |= tokens == e
    first = torch.where(hit.any(1), hit.int().argmax(1) + 1, torch.full((tokens.shape[0],), n, device=tokens.device))
    return tokens[:, : int(first.max().item())]


@torch.no_grad()
def generate(decoder, inputs_embeds: Optional[torch.Tensor] = None, attention_mask: Optional[torch.Tensor] = None, *,
             input_ids: Optional[torch.Tensor] = None, max_new_tokens: Optional[int] = None, max_length: Optional[int] = None,
             eos_token_id=None, pad_token_id: Optional[int] = None, do_sample: bool = False, temperature: Optional[float] = 1.0,
             top_k: Optional[int] = 50, top_p: Optional[float] = 1.0, num_beams: int = 1, length_penalty: float = 1.0,
             early_stopping=False, num_return_sequences: int = 1, return_dict_in_generate: bool = False, output_scores: bool = False,
             output_logits: bool = False, use_graph: bool = True, sync_every: int = 16, generator: Optional[torch.Generator] = None,
             stream_copy: bool = True, fuse_rope: bool = True, **unused):
    """`LlamaForCausalLM.generate` for prompts given as embeddings (or ids): greedy, sampling (temperature / top-k / top-p) and beam
    search with length penalty.  Returns the new token ids i64 [batch * num_return_sequences, n] (or a GenerateOutput)."""
    if (inputs_embeds is None) == (input_ids is None):
        raise ValueError("pass exactly one of inputs_embeds / input_ids")
    if unused:
        bad = [k for k in unused if unused[k] is not None and k not in ("use_cache", "output_attentions", "output_hidden_states", "synced_gpus")]
        if bad:
            raise NotImplementedError(f"generate(): unsupported arguments {bad}")
    if input_ids is not None:
        inputs_embeds = decoder.model.embed(input_ids)
    if inputs_embeds.dim() != 3 or inputs_embeds.shape[2] != decoder.spec.hidden_size:
        raise ValueError(f"inputs_embeds must be [batch, seq_len, {decoder.spec.hidden_size}]")
    B, T, _ = inputs_embeds.shape
    if attention_mask is None:
        attention_mask = torch.ones((B, T), dtype=torch.int64, device=inputs_embeds.device)
    if tuple(attention_mask.shape) != (B, T):
        raise ValueError(f"attention_mask shape {tuple(attention_mask.shape)} != inputs {(B, T)}")
    if max_new_tokens is None:
        if max_length is None:
            raise ValueError("generate() needs max_new_tokens (or max_length)")
        max_new_tokens = max_length - T      # HF: `max_length` counts the (padded) prompt, for embeddings too (generate(): max_length -= inputs_embeds.shape[1])
    if max_new_tokens < 1:
        raise ValueError("max_new_tokens must be >= 1")
    eos_ids = _as_id_list(eos_token_id)
    if pad_token_id is None:
        if not eos_ids:
            pad_token_id = 0
        else:
            pad_token_id = eos_ids[0]                                        # HF's fallback (with a warning)
    if num_beams < 1 or num_return_sequences < 1 or num_return_sequences > num_beams and num_beams > 1:
        raise ValueError("num_beams >= 1 and 1 <= num_return_sequences <= num_beams")
    if num_beams > 1:
        if do_sample:
            raise NotImplementedError("beam sampling (num_beams > 1 with do_sample=True) is not built")
        return _beam_search(decoder, inputs_embeds, attention_mask, max_new_tokens, eos_ids, int(pad_token_id), num_beams, float(length_penalty),
                            early_stopping, num_return_sequences, return_dict_in_generate, output_scores, output_logits, stream_copy)
    R = int(num_return_sequences)
    if R != 1 and not do_sample:
        raise ValueError("greedy decoding returns one sequence per prompt: num_return_sequences > 1 needs do_sample=True or num_beams > 1")
    # R samples per prompt (HF repeats the prompt R times): R rows that SHARE the prompt segment of the cache, like the beams of a prompt
    eng = DecodeEngine(decoder, B, R, T, max_new_tokens, stream_copy, fuse_rope)
    embeds, mask = eng.compact(inputs_embeds, attention_mask)
    logits = eng.prefill(embeds, mask)
    if R > 1:
        logits = logits.repeat_interleave(R, dim=0)
    V = decoder.spec.vocab_size
    eos = torch.tensor(eos_ids, dtype=torch.int64, device=eng.dev)
    if do_sample and return_dict_in_generate and output_scores:
        # HF returns the PROCESSED scores (after temperature / top-k / top-p) under sampling; nothing here is pinned on that
        # (the goldens cover greedy and beam search), so the combination is refused rather than answered with the raw logits
        raise NotImplementedError("generate: output_scores with do_sample is not supported (ask for output_logits: the raw LM-head rows)")
    keep_logits = [] if (return_dict_in_generate and (output_logits or output_scores)) else None
    pad = int(pad_token_id)

    def choose(lg: torch.Tensor, col: int):
        if do_sample:
            scores = filter_logits(lg[:, :V].float(), temperature, top_k, top_p)
            nxt = torch.multinomial(scores.softmax(dim=-1), num_samples=1, generator=generator).squeeze(1)
            nxt = torch.where(eng.finished.bool(), torch.full_like(nxt, pad), nxt)
            eng.next_tokens.copy_(nxt)
            eng.out_tokens[:, col] = nxt
            if eos_ids:
                eng.finished |= torch.isin(nxt, eos).int()
        else:
            eng.greedy_select(lg, eos, pad)              # writes column step[0] == col

    def advance():                                        # chosen tokens -> embeddings -> all layers over the cache -> logits -> choice
        eng.feed(eng.next_tokens)
        eng.decode_step()
        if not do_sample:
            eng.greedy_select(eng.logits, eos, pad)

    choose(logits, 0)
    if keep_logits is not None:
        keep_logits.append(logits[:, :V].float().clone())
    graph, n = None, 1
    while n < max_new_tokens:
        if eos_ids and (n == 1 or n % sync_every == 0) and bool(eng.finished.all().item()):
            break
        if use_graph and not do_sample and n >= 2:        # the first step ran eagerly (lazy one-time initialisation inside the library)
            if graph is None:
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                    advance()
            graph.replay()
        else:
            advance()
        if do_sample:
            choose(eng.logits, n)
        if keep_logits is not None:
            keep_logits.append(eng.logits[:, :V].float().clone())
        n += 1
    seqs = _trim(eng.out_tokens, eos_ids, n)
    if not return_dict_in_generate:
        return seqs
    keep = tuple(keep_logits[: seqs.shape[1]]) if keep_logits is not None else None
    return GenerateOutput(seqs, scores=keep if output_scores else None, logits=keep if output_logits else None)


# ---------------------------------------------------------------------------------------------------------------------------
# beam search: transformers/generation/utils.py `_beam_search` (5.x, vectorised form) with an empty id prompt (decoder_prompt_len 0)
# ---------------------------------------------------------------------------------------------------------------------------
def _gather_beams(t: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    while idx.dim() < t.dim():
        idx = idx.unsqueeze(-1)
    return torch.take_along_dim(t, idx, dim=1)


def _beam_search(decoder, inputs_embeds, attention_mask, max_length: int, eos_ids, pad_id: int, nb: int, length_penalty: float, early_stopping,
                 num_return_sequences: int, return_dict: bool, output_scores: bool, output_logits: bool, stream_copy: bool = True):
    B = inputs_embeds.shape[0]
    eng = DecodeEngine(decoder, B, nb, inputs_embeds.shape[1], max_length, stream_copy)
    dev, V = eng.dev, decoder.spec.vocab_size
    embeds, mask = eng.compact(inputs_embeds, attention_mask)
    first_logits = eng.prefill(embeds, mask)                                 # [B, ld]: the beams of a prompt start from the same state
    eos = torch.tensor(eos_ids, dtype=torch.int64, device=dev)
    keep = max(2, 1 + len(eos_ids)) * nb
    top_mask = torch.cat((torch.ones(nb, dtype=torch.bool), torch.zeros(keep - nb, dtype=torch.bool))).to(dev)
    fill = pad_id
    running = torch.full((B, nb, max_length), fill, dtype=torch.int64, device=dev)
    sequences = running.clone()
    running_scores = torch.zeros((B, nb), dtype=torch.float32, device=dev)
    running_scores[:, 1:] = -1e9
    beam_scores = torch.full((B, nb), -1e9, dtype=torch.float32, device=dev)
    is_finished = torch.zeros((B, nb), dtype=torch.bool, device=dev)
    heuristic_open = torch.ones((B, 1), dtype=torch.bool, device=dev)
    running_idx = torch.full((B, nb, max_length), -1, dtype=torch.int32, device=dev)
    beam_idx_tab = running_idx.clone()
    all_scores, all_logits = [], []
    cur = 0
    while True:
        if cur == 0:
            logits = first_logits[:, :V].float().repeat_interleave(nb, dim=0)     # HF expands the prompt nb times before its prefill
        else:
            logits = eng.logits[:, :V].float()
        log_probs = torch.log_softmax(logits, dim=-1)
        if return_dict and output_logits:
            all_logits.append(logits.clone())
        if return_dict and output_scores:
            all_scores.append(log_probs.clone())
        acc = (log_probs.view(B, nb, V) + running_scores[:, :, None]).reshape(B, nb * V)
        topk_lp, topk_i = torch.topk(acc, k=keep)
        src_beam = topk_i // V
        topk_idx = _gather_beams(running_idx, src_beam)
        topk_seq = _gather_beams(running, src_beam)
        topk_seq[:, :, cur] = topk_i % V
        topk_idx[:, :, cur] = (src_beam + torch.arange(B, device=dev).view(-1, 1) * nb).to(torch.int32)
        # stopping criteria on the candidates: max length, eos
        hits = torch.full((B, keep), cur + 1 >= max_length, dtype=torch.bool, device=dev)
        if eos_ids:
            hits |= torch.isin(topk_seq[:, :, cur], eos)
        # beams that go on
        live_lp = topk_lp + hits.float() * -1.0e9
        nxt = torch.topk(live_lp, k=nb)[1]
        running = _gather_beams(topk_seq, nxt)
        running_scores = _gather_beams(live_lp, nxt)
        running_idx = _gather_beams(topk_idx, nxt)
        # finished hypotheses
        just_done = hits & top_mask[None, :]
        fin_lp = topk_lp / ((cur + 1) ** length_penalty)
        full = torch.all(is_finished, dim=-1, keepdim=True) & (early_stopping is True)
        fin_lp = fin_lp + full.float() * -1.0e9 + (~heuristic_open).float() * -1.0e9 + (~just_done).float() * -1.0e9
        m_seq, m_sc = torch.cat((sequences, topk_seq), 1), torch.cat((beam_scores, fin_lp), 1)
        m_idx, m_fin = torch.cat((beam_idx_tab, topk_idx), 1), torch.cat((is_finished, just_done), 1)
        sel = torch.topk(m_sc, k=nb)[1]
        sequences, beam_scores = _gather_beams(m_seq, sel), _gather_beams(m_sc, sel)
        beam_idx_tab, is_finished = _gather_beams(m_idx, sel), _gather_beams(m_fin, sel)
        # the rows the next step continues from
        src_rows = running_idx[:, :, cur].reshape(-1).to(torch.int64)
        cur += 1
        best_len = max_length if (early_stopping == "never" and length_penalty > 0.0) else cur
        best_running = running_scores[:, :1] / (best_len ** length_penalty)
        worst_done = torch.where(is_finished, torch.min(beam_scores, dim=1, keepdim=True)[0], torch.full_like(beam_scores, -1.0e9))
        heuristic_open = heuristic_open & torch.any(best_running > worst_done, dim=-1, keepdim=True)
        go_on = torch.any(heuristic_open) & ~(torch.all(is_finished) & (early_stopping is True)) & ~torch.all(hits)
        if not bool(go_on.item()) or cur >= max_length:
            break
        if cur > 1:
            eng.reorder(src_rows)                       # generated keys / values follow their beams (nothing is cached before step 1)
        eng.next_tokens.copy_(running[:, :, cur - 1].reshape(-1))
        eng.feed(eng.next_tokens)
        eng.decode_step()
    R = num_return_sequences
    sequences, beam_scores, beam_idx_tab = sequences[:, :R].reshape(B * R, -1), beam_scores[:, :R].reshape(-1), beam_idx_tab[:, :R].reshape(B * R, -1)
    n = int(((beam_idx_tab + 1).bool()).sum(dim=1).max().item())
    sequences = sequences[:, :n]
    if not return_dict:
        return sequences
    return GenerateOutput(sequences, scores=tuple(all_scores) if output_scores else None, logits=tuple(all_logits) if output_logits else None,
                          sequences_scores=beam_scores if output_scores else None, beam_indices=beam_idx_tab[:, :n])
