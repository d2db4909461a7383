"""`generate` of Esm2LlamaInstructForCausalLM (reference models/modeling_esm2llama_instruct.py:217-251; call site
scripts/generate_instruct.py:72-87) on the KV-cache decode path (csrc/llama_decode.hip, p2t_hip/generation.py).

* the kernels one by one against numpy: prompt compaction, single-query attention over the two cache segments (both dtypes, head_dim
  16 / 48 / 64 / 128, GQA groups 1 / 2 / 4 / 5, split-KV on and off, beams sharing a prompt segment), beam re-ordering of the
  generated segment, the greedy choice with ties / finished rows;
* the whole call against `model.generate` of the REFERENCE class (tests/golden/generate_tiny.npz, make_golden.py run_generate):
  left-padded prompts holding protein placeholders; fp32: greedy ids exact and per-step logits to 2e-4 (eager steps and the replayed
  HIP graph), the finished-row rule under an eos id, beam search (3 beams, two length penalties) ids exact and scores to 1e-4; four
  decoder shapes (generic head_dim 16, 64 with GQA, 128 with the packed rows, Qwen3 with q/k norms and head_dim 48);
* bf16: per-step logits against the bf16-rounding oracle fed the SAME tokens (teacher forcing), and against the model's own
  cache-less forward over prompt + generated tokens."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import p2t_oracle as O
from gpu_util import build_model, dev, observe, rel, rnd, to_dev, to_np
from helpers import model_weights
from p2t_hip import specs

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
CASES = ["d16", "d64", "d128", "qwen3"]


@pytest.fixture(scope="module")
def g():
    z = np.load(os.path.join(HERE, "golden", "generate_tiny.npz"))
    d = {k: z[k] for k in z.files}
    d["meta"] = json.loads(bytes(d.pop("meta_json")).decode())
    return d


def _model(g, case, dtype):
    m = g["meta"]["cases"][case]
    model = build_model(specs.EsmSpec(**m["esm"]), specs.LlamaSpec(**m["llama"]), specs.AdapterSpec(**m["adapter"]), dtype, m["weight_seed"])
    model.config.placeholder_id = g["meta"]["placeholder_id"]
    model.eval()
    return model


def _inputs(g):
    return dict(inputs=to_dev(g["input_ids"]), attention_mask=to_dev(g["attention_mask"]), protein_input_ids=to_dev(g["protein_input_ids"]),
                protein_attention_mask=to_dev(g["protein_attention_mask"]))


# ---------------------------------------------------------------------------------------------
def test_compact_rows_vs_numpy():
    from p2t_hip import _lib
    from p2t_hip.ops import ptr, stream
    B, T, H = 5, 300, 24
    x = rnd(3, "cr.x", (B, T, H), 1.0)
    rs = np.random.RandomState(0)
    mask = (rs.rand(B, T) < 0.6).astype(np.int64)
    mask[1] = 0; mask[1, 200:] = 1            # left padding
    mask[2] = 1                               # nothing to move
    mask[3] = 0; mask[3, 7] = 1               # a single token
    mask[4] = 0                               # an empty row (the caller refuses it; the kernel must not misbehave)
    xd, md = to_dev(x), to_dev(mask)
    out, om = torch.full((B, T, H), 7.0, device=dev()), torch.empty_like(md)
    lens, scr = torch.empty((B,), dtype=torch.int32, device=dev()), torch.empty((B * T,), dtype=torch.int32, device=dev())
    _lib.call("p2t_compact_rows", ptr(xd), ptr(md), B, T, H, ptr(out), ptr(om), ptr(lens), ptr(scr), stream())
    o, l, m2 = to_np(out), to_np(lens), to_np(om)
    for b in range(B):
        n = int(mask[b].sum())
        assert l[b] == n
        assert np.array_equal(o[b, :n], x[b][mask[b] != 0]) and not o[b, n:].any()
        assert np.array_equal(m2[b], (np.arange(T) < n).astype(np.int64))


def _attn_ref(q, kp, vp, kg, vg, lens, step, group, scale):
    """q [BB, nh, d]; prompt keys/values [B0, nkv, Tp, d], generated [BB, nkv, G, d] -> [BB, nh, d]"""
    BB, nh, d = q.shape
    nkv = kp.shape[1]
    G = nh // nkv
    out = np.zeros((BB, nh, d), dtype=np.float32)
    for bb in range(BB):
        b0 = bb // group
        for h in range(nh):
            kv = h // G
            k = np.concatenate([kp[b0, kv, :lens[b0]], kg[bb, kv, :step + 1]], 0)
            v = np.concatenate([vp[b0, kv, :lens[b0]], vg[bb, kv, :step + 1]], 0)
            s = (k @ q[bb, h]) * scale
            p = np.exp(s - s.max())
            out[bb, h] = (p / p.sum()) @ v
    return out


@pytest.mark.parametrize("dt,use_mfma", [(torch.float32, 0), (torch.bfloat16, 0), (torch.bfloat16, 1)])
@pytest.mark.parametrize("shape", [(16, 4, 2, 1, 100, 3), (64, 4, 2, 2, 700, 70), (128, 8, 2, 1, 1500, 130), (48, 5, 1, 3, 64, 0),
                                   (128, 2, 2, 1, 5, 63), (64, 16, 1, 1, 200, 9)])
def test_decode_attention_vs_numpy(dt, use_mfma, shape):
    """p2t_attention_decode (the decode step's attention on its own) against softmax(q k^T) v over the concatenated valid keys; NaN
    behind every valid prefix proves nothing outside it is read into a result."""
    from p2t_hip import _lib, ops
    from p2t_hip.ops import ptr, stream
    d, nh, nkv, group, n0max, step = shape
    dp = ops.head_dim_padded(d)
    B0 = 3
    BB = B0 * group
    Tp, G = ops.round_up(n0max, 64), ops.round_up(step + 1, 64)
    lens = np.array([n0max, max(1, n0max // 3), 1], dtype=np.int32)
    rs = np.random.RandomState(d + nh)
    f = lambda *s: (rs.randn(*s) * 0.7).astype(np.float32)
    q, kp, vp, kg, vg = f(BB, nh, d), f(B0, nkv, Tp, d), f(B0, nkv, Tp, d), f(BB, nkv, G, d), f(BB, nkv, G, d)
    # garbage (NaN) behind the valid prefixes: the kernel must neither read it into a result nor multiply it by zero
    for b in range(B0):
        kp[b, :, lens[b]:] = np.nan; vp[b, :, lens[b]:] = np.nan
    kg[:, :, step + 1:] = np.nan; vg[:, :, step + 1:] = np.nan
    pad = lambda a: np.concatenate([a, np.zeros(a.shape[:-1] + (dp - d,), np.float32)], -1)
    rd = lambda a: to_np(to_dev(a, dt)).astype(np.float32)
    qd = to_dev(pad(q), dt)
    kpd, kgd = to_dev(pad(kp), dt), to_dev(pad(kg), dt)
    vtp, vtg = to_dev(np.ascontiguousarray(pad(vp).transpose(0, 1, 3, 2)), dt), to_dev(np.ascontiguousarray(pad(vg).transpose(0, 1, 3, 2)), dt)
    lens_d, step_d = to_dev(lens), torch.tensor([step], dtype=torch.int32, device=dev())
    scale = d ** -0.5
    l2s = dt == torch.bfloat16
    QO = ops.round_up(nh * d, 64)
    out = torch.zeros((BB, QO), dtype=dt, device=dev())
    _lib.call("p2t_attention_decode", ptr(qd), ptr(kpd), ptr(vtp), ptr(kgd), ptr(vtg), ptr(lens_d), ptr(step_d), B0, group, nh, nkv, d, Tp, G,
              float(scale), int(l2s), ops.dt_of(dt), use_mfma, ptr(out), QO, stream())
    qq = rd(pad(q))[..., :d]
    ref = _attn_ref(qq, np.nan_to_num(rd(pad(kp)))[..., :d], np.nan_to_num(rd(pad(vp)))[..., :d], np.nan_to_num(rd(pad(kg)))[..., :d],
                    np.nan_to_num(rd(pad(vg)))[..., :d], lens, step, group, (np.log(2.0) if l2s else scale))
    got = to_np(out).astype(np.float32)[:, : nh * d].reshape(BB, nh, d)
    assert np.isfinite(got).all()
    if dt == torch.float32:
        assert rel(got, ref) < 3e-6
    else:
        observe(f"decode_attn[d{d},nh{nh},nkv{nkv},g{group},n{n0max},s{step}].bf16{'_mfma' if use_mfma else ''}", rel(got, ref), 1.2e-2)


@pytest.mark.parametrize("shape", [(1, 512, 64), (3, 48, 96), (8, 4096, 4096), (16, 6144, 4096), (17, 1000, 512), (32, 28672, 1024), (40, 4096, 14336),
                                   (64, 33000, 256)])
def test_skinny_gemm_vs_numpy(shape):
    """p2t_gemm_nt_skinny (the decode step's weight-streaming GEMM) against numpy on the same bf16 operands: f32 store, bf16 store,
    residual accumulate, SwiGLU over the gate/up interleave; row counts 1..64, ragged N, K down to two MFMA steps."""
    from p2t_hip import _lib
    from p2t_hip.ops import ptr, stream
    M, N, K = shape
    bf = lambda a: to_np(to_dev(a, torch.bfloat16)).astype(np.float32)
    x, w = bf(rnd(7, "sk.x", (M, K), 2.0)), bf(rnd(7, "sk.w", (N, K), 2.0 / np.sqrt(K)))
    xd, wd = to_dev(x, torch.bfloat16), to_dev(w, torch.bfloat16)
    ref = x @ w.T
    ldc = (N + 3) // 4 * 4
    out = torch.full((M, ldc), 7.0, dtype=torch.float32, device=dev())
    _lib.call("p2t_gemm_nt_skinny", ptr(xd), K, ptr(wd), K, 0, ptr(out), ldc, M, N, K, _lib.F32, _lib.EPI_STORE, stream())
    assert rel(to_np(out)[:, :N], ref) < 3e-6 and (to_np(out)[:, N:] == 7.0).all()
    # the same product from the pre-shuffled stream copy of W: bit-identical (same fragments, same order)
    from p2t_hip.generation import preshuffle
    ws = preshuffle(wd, N)
    out2 = torch.full((M, ldc), 7.0, dtype=torch.float32, device=dev())
    _lib.call("p2t_gemm_nt_skinny", ptr(xd), K, ptr(ws), 0, 1, ptr(out2), ldc, M, N, K, _lib.F32, _lib.EPI_STORE, stream())
    assert torch.equal(out, out2)
    base = rnd(7, "sk.r", (M, ldc), 1.0)
    acc = to_dev(base.copy())
    _lib.call("p2t_gemm_nt_skinny", ptr(xd), K, ptr(ws), 0, 1, ptr(acc), ldc, M, N, K, _lib.F32, _lib.EPI_RESID, stream())
    assert rel(to_np(acc)[:, :N], base[:, :N] + ref) < 3e-6
    ob = torch.zeros((M, ldc), dtype=torch.bfloat16, device=dev())
    _lib.call("p2t_gemm_nt_skinny", ptr(xd), K, ptr(wd), K, 0, ptr(ob), ldc, M, N, K, _lib.BF16, _lib.EPI_STORE, stream())
    assert np.array_equal(to_np(ob).astype(np.float32)[:, :N], bf(to_np(out)[:, :N]))       # the same accumulators, rounded once
    if N % 64 == 0:
        F = N // 2
        blocks = ref.reshape(M, N // 64, 2, 32)                       # 32-row gate / up blocks of p2t_llama_layer.gu_w
        gate, up = blocks[:, :, 0].reshape(M, F), blocks[:, :, 1].reshape(M, F)
        act = torch.zeros((M, F), dtype=torch.bfloat16, device=dev())
        _lib.call("p2t_gemm_nt_skinny", ptr(xd), K, ptr(wd), K, 0, ptr(act), F, M, N, K, _lib.BF16, _lib.EPI_SWIGLU, stream())
        act2 = torch.zeros((M, F), dtype=torch.bfloat16, device=dev())
        _lib.call("p2t_gemm_nt_skinny", ptr(xd), K, ptr(ws), 0, 1, ptr(act2), F, M, N, K, _lib.BF16, _lib.EPI_SWIGLU, stream())
        assert torch.equal(act, act2)
        want = gate / (1.0 + np.exp(-gate)) * up
        assert rel(to_np(act).astype(np.float32), want) < 4e-3
    with pytest.raises(_lib.P2TError):
        _lib.call("p2t_gemm_nt_skinny", ptr(xd), K, ptr(wd), K, 0, ptr(out), ldc, 65, N, K, _lib.F32, _lib.EPI_STORE, stream())


@pytest.mark.parametrize("shape", [(1, 256, 128), (5, 96, 256), (8, 4096, 4096), (16, 6144, 4096), (19, 1000, 512), (32, 2048, 1792), (64, 640, 384)])
def test_skinny_gemm_fp8_vs_numpy_on_the_same_quantised_operands(shape):
    """p2t_gemm_nt_skinny_fp8 (the decode step's GEMM of gemm_fp8 models: e4m3 operands, one E8M0 scale per row, block-scaled MFMA)
    against numpy on the de-quantised operands; row-major weights and the pre-shuffled stream copy give the same bits."""
    from p2t_hip import _lib, ops
    from p2t_hip.generation import preshuffle
    from p2t_hip.ops import ptr, stream
    M, N, K = shape
    a = rnd(52, "s8.a", (M, K), 1.0) * np.exp(rnd(52, "s8.as", (M, 1), 2.0)).astype(np.float32)
    w = rnd(52, "s8.w", (N, K), 0.5) * np.exp(rnd(52, "s8.ws", (N, 1), 2.0)).astype(np.float32) / np.float32(np.sqrt(K))
    a8, sa = ops.quant_rows_fp8(to_dev(a))
    w8, sw = ops.quant_rows_fp8(to_dev(w))
    Kq = a8.shape[1]
    ref = O.quant_rows_e4m3(a)[0] @ O.quant_rows_e4m3(w)[0].T
    ldc = (N + 3) // 4 * 4
    outs = []
    for pre, wt in ((0, w8), (1, preshuffle(w8, N))):
        out = torch.full((M, ldc), 7.0, dtype=torch.float32, device=dev())
        _lib.call("p2t_gemm_nt_skinny_fp8", ptr(a8), Kq, ptr(sa), ptr(wt), Kq, ptr(sw), pre, ptr(out), ldc, M, N, Kq, _lib.F32, _lib.EPI_STORE, stream())
        outs.append(out)
    assert torch.equal(outs[0], outs[1]) and (to_np(outs[0])[:, N:] == 7.0).all()
    observe(f"skinny_fp8[{M}x{N}x{K}].f32", rel(to_np(outs[0])[:, :N], ref), 3e-5)
    base = rnd(52, "s8.r", (M, ldc), 1.0)
    acc = to_dev(base.copy())
    _lib.call("p2t_gemm_nt_skinny_fp8", ptr(a8), Kq, ptr(sa), ptr(w8), Kq, ptr(sw), 0, ptr(acc), ldc, M, N, Kq, _lib.F32, _lib.EPI_RESID, stream())
    assert np.allclose(to_np(acc)[:, :N], base[:, :N] + to_np(outs[0])[:, :N], rtol=0, atol=1e-5 * np.abs(ref).max())
    if N % 64 == 0:
        F = N // 2
        blocks = to_np(outs[0])[:, :N].reshape(M, N // 64, 2, 32)
        gate, up = blocks[:, :, 0].reshape(M, F), blocks[:, :, 1].reshape(M, F)
        act = torch.zeros((M, F), dtype=torch.bfloat16, device=dev())
        _lib.call("p2t_gemm_nt_skinny_fp8", ptr(a8), Kq, ptr(sa), ptr(w8), Kq, ptr(sw), 0, ptr(act), F, M, N, Kq, _lib.BF16, _lib.EPI_SWIGLU, stream())
        assert rel(to_np(act).astype(np.float32), gate / (1.0 + np.exp(-gate)) * up) < 4e-3


def test_greedy_select_ties_and_finished_rows():
    from p2t_hip import _lib, ops
    from p2t_hip.ops import ptr, stream
    V, ld, BB, G = 1000, 1024, 4, 64
    for dt in (torch.float32, torch.bfloat16):
        lg = np.full((BB, ld), -5.0, dtype=np.float32)
        lg[0, [900, 17, 400]] = 3.0                   # a three-way tie: torch.argmax returns the lowest index
        lg[1, 999] = 1.0; lg[1, 1001] = 9.0           # columns >= V are padding
        lg[2, 5] = 2.0                                # an eos id
        lg[3, 8] = 4.0                                # finished before: emits pad
        lgd = to_dev(lg, dt)
        eos = torch.tensor([5, 77], dtype=torch.int64, device=dev())
        fin = torch.tensor([0, 0, 0, 1], dtype=torch.int32, device=dev())
        nxt, out = torch.zeros((BB,), dtype=torch.int64, device=dev()), torch.full((BB, G), -1, dtype=torch.int64, device=dev())
        step = torch.tensor([3], dtype=torch.int32, device=dev())
        _lib.call("p2t_greedy_select", ptr(lgd), ops.dt_of(dt), ld, V, BB, ptr(eos), 2, 555, ptr(fin), ptr(nxt), ptr(out), G, ptr(step), G, stream())
        assert to_np(nxt).tolist() == [17, 999, 5, 555]
        assert to_np(out)[:, 3].tolist() == [17, 999, 5, 555] and (to_np(out)[:, :3] == -1).all()
        assert to_np(fin).tolist() == [0, 0, 1, 1]
        assert to_np(nxt).tolist() == torch.where(fin.cpu().bool() & torch.tensor([False, False, False, True]), torch.tensor(555),
                                                   torch.from_numpy(lg[:, :V]).argmax(1)).tolist()


# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("use_graph", [False, True])
def test_fp32_greedy_vs_reference_generate(g, case, use_graph):
    model = _model(g, case, torch.float32)
    meta = g["meta"]
    n, pad = meta["max_new_tokens"], meta["pad_id"]
    out = model.generate(**_inputs(g), max_new_tokens=n, eos_token_id=None, pad_token_id=pad, do_sample=False, num_beams=1,
                         return_dict_in_generate=True, output_logits=True, use_graph=use_graph)
    assert np.array_equal(to_np(out.sequences), g[f"{case}.greedy"])
    lg = torch.stack(out.logits, 0)
    assert np.abs(to_np(lg) - g[f"{case}.greedy_logits"]).max() < 2e-4
    # the argument set of scripts/generate_instruct.py:72-87 (sampling switches passed but off), with rows that stop early
    eos = meta["cases"][case]["eos"]
    seq = model.generate(**_inputs(g), max_new_tokens=n, eos_token_id=eos, pad_token_id=pad, return_dict_in_generate=False, num_beams=1,
                         length_penalty=1.0, temperature=1.0, do_sample=False, top_p=1.0, top_k=50, use_graph=use_graph, sync_every=4)
    assert np.array_equal(to_np(seq), g[f"{case}.greedy_eos"])


@pytest.mark.parametrize("case", CASES)
def test_fp32_beam_search_vs_reference_generate(g, case):
    model = _model(g, case, torch.float32)
    meta = g["meta"]
    for lp in ("1.0", "0.5"):
        out = model.generate(**_inputs(g), max_new_tokens=meta["max_new_tokens"], eos_token_id=meta["cases"][case]["eos"], pad_token_id=meta["pad_id"],
                             do_sample=False, num_beams=3, length_penalty=float(lp), return_dict_in_generate=True, output_scores=True)
        assert np.array_equal(to_np(out.sequences), g[f"{case}.beam3_lp{lp}"]), lp
        assert np.abs(to_np(out.sequences_scores) - g[f"{case}.beam3_lp{lp}_scores"]).max() < 1e-4, lp
    # two eos ids, early_stopping=True / "never", two hypotheses per prompt; and greedy under a `max_length` budget
    eos2, T = meta["cases"][case]["eos2"], g["input_ids"].shape[1]
    for es in (True, "never"):
        out = model.generate(**_inputs(g), max_new_tokens=meta["max_new_tokens"], eos_token_id=eos2, pad_token_id=meta["pad_id"], do_sample=False,
                             num_beams=3, length_penalty=0.8, early_stopping=es, num_return_sequences=2, return_dict_in_generate=True, output_scores=True)
        assert np.array_equal(to_np(out.sequences), g[f"{case}.beam3_es{es}"]), es
        assert np.abs(to_np(out.sequences_scores) - g[f"{case}.beam3_es{es}_scores"]).max() < 1e-4, es
    seq = model.generate(**_inputs(g), max_length=T + meta["max_new_tokens"] - 3, eos_token_id=eos2, pad_token_id=meta["pad_id"], do_sample=False)
    assert np.array_equal(to_np(seq), g[f"{case}.greedy_eos2_maxlen"])


@pytest.mark.parametrize("case", CASES)
def test_bf16_steps_vs_bf16_oracle_and_own_forward(g, case):
    """bf16: the greedy ids may leave the fp32 path's at a near-tie, so the comparison is per step on the SAME tokens: the oracle
    with bf16 rounding is teacher-forced with the ids the GPU chose; and the model's own cache-less forward over prompt + those ids
    must give the same last-position logits (KV-cache path == full forward)."""
    model = _model(g, case, torch.bfloat16)
    meta = g["meta"]
    m = meta["cases"][case]
    n, pad = 8, meta["pad_id"]
    out = model.generate(**_inputs(g), max_new_tokens=n, eos_token_id=None, pad_token_id=pad, do_sample=False, return_dict_in_generate=True,
                         output_logits=True)
    toks, lg = to_np(out.sequences), to_np(torch.stack(out.logits, 0))
    # the decoder streamed from its own matrices instead of the pre-shuffled copies: the same fragments in the same order
    out_n = model.generate(**_inputs(g), max_new_tokens=n, eos_token_id=None, pad_token_id=pad, do_sample=False, return_dict_in_generate=True,
                           output_logits=True, stream_copy=False)
    assert torch.equal(out.sequences, out_n.sequences) and torch.equal(torch.stack(out.logits, 0), torch.stack(out_n.logits, 0))
    # ... and with the rotation + cache append as its own launch instead of the QKV GEMM's epilogue (head_dim 64 / 128): bit for bit
    out_u = model.generate(**_inputs(g), max_new_tokens=n, eos_token_id=None, pad_token_id=pad, do_sample=False, return_dict_in_generate=True,
                           output_logits=True, fuse_rope=False)
    assert torch.equal(out.sequences, out_u.sequences) and torch.equal(torch.stack(out.logits, 0), torch.stack(out_u.logits, 0))
    esm, llama, ad = specs.EsmSpec(**m["esm"]), specs.LlamaSpec(**m["llama"]), specs.AdapterSpec(**m["adapter"])
    W = model_weights(esm, llama, ad, m["weight_seed"], lm_head=True)
    emb, mask = model(input_ids=to_dev(g["input_ids"]), attention_mask=to_dev(g["attention_mask"]), protein_input_ids=to_dev(g["protein_input_ids"]),
                      protein_attention_mask=to_dev(g["protein_attention_mask"]), return_decoder_inputs=True)
    _, ref = O.generate_greedy(llama, W, to_np(emb), to_np(mask), n, (), pad, prec=O.BF16, forced=toks)
    observe(f"generate[{case}].bf16_vs_bf16oracle.logits", rel(lg, ref), 4e-2)
    # own cache-less forward: prompt (still left padded: positions differ from the compacted ones only on rows WITHOUT padding ...
    # so compare on the compacted prompt) + the generated ids
    dec = model.llama_decoder
    for b in range(toks.shape[0]):
        valid = to_np(mask)[b] != 0
        row = torch.cat([emb[b][torch.from_numpy(valid).to(emb.device)], dec.model.embed(to_dev(toks[b:b + 1, :-1]))[0]], 0)[None]
        full = to_np(dec(inputs_embeds=row).logits.float())[0]
        n0 = int(valid.sum())
        observe(f"generate[{case}].bf16_cache_vs_full_forward.row{b}", rel(lg[:, b], full[n0 - 1:n0 - 1 + n]), 3e-2)


def test_sampling_is_seeded_and_respects_the_filters(g):
    model = _model(g, "d64", torch.float32)
    meta = g["meta"]
    kw = dict(max_new_tokens=6, eos_token_id=None, pad_token_id=meta["pad_id"], do_sample=True, temperature=0.7, top_k=5, top_p=0.9)
    gen = torch.Generator(device=dev())
    a = model.generate(**_inputs(g), **kw, generator=gen.manual_seed(3))
    b = model.generate(**_inputs(g), **kw, generator=gen.manual_seed(3))
    c = model.generate(**_inputs(g), **kw, generator=gen.manual_seed(4))
    assert torch.equal(a, b) and a.shape == (3, 6) and not torch.equal(a, c)
    # top_k = 1 is greedy whatever the seed
    d = model.generate(**_inputs(g), max_new_tokens=6, eos_token_id=None, pad_token_id=meta["pad_id"], do_sample=True, top_k=1, generator=gen.manual_seed(9))
    assert np.array_equal(to_np(d), g["d64.greedy"][:, :6])
    # several samples per prompt: rows b * R .. b * R + R - 1 share prompt b's cache segment (HF order: prompt-major)
    r3 = model.generate(**_inputs(g), max_new_tokens=6, eos_token_id=None, pad_token_id=meta["pad_id"], do_sample=True, top_k=1, num_return_sequences=3,
                        generator=gen.manual_seed(1))
    assert np.array_equal(to_np(r3), np.repeat(g["d64.greedy"][:, :6], 3, axis=0))
    s3 = model.generate(**_inputs(g), **kw, num_return_sequences=3, generator=gen.manual_seed(5))
    assert s3.shape == (9, 6) and not torch.equal(s3[0], s3[1])
    with pytest.raises(ValueError):
        model.generate(**_inputs(g), max_new_tokens=2, num_return_sequences=2)


def test_generate_refusals(g):
    model = _model(g, "d16", torch.float32)
    with pytest.raises(ValueError):
        model.llama_decoder.generate(inputs_embeds=torch.zeros((1, 4, 64), device=dev()), attention_mask=torch.zeros((1, 4), dtype=torch.int64, device=dev()),
                                     max_new_tokens=2)                       # a row without a single token under the mask
    with pytest.raises(ValueError):
        model.llama_decoder.generate(inputs_embeds=torch.zeros((1, 4, 64), device=dev()))       # no length budget
    with pytest.raises(NotImplementedError):
        model.generate(**_inputs(g), max_new_tokens=2, num_beams=2, do_sample=True)
    # ids instead of embeddings: the decoder alone (HF would return prompt + new tokens; here the new tokens, documented)
    ids = to_dev(np.array([[3, 4, 5, 6]], dtype=np.int64))
    out = model.llama_decoder.generate(ids, max_new_tokens=3, pad_token_id=0)
    assert out.shape == (1, 3)


def test_inference_epoch_writes_the_reference_json(g, tmp_path):
    """scripts/generate_instruct.py:90-147 on the fixture's batch: `iterative_generation_loop` returns the golden greedy ids, and
    `inference_epoch` writes generation_{postfix}_rank{r}.json = {name: {"true", "pred"}} through the tokenizer's batch_decode."""
    import p2t_hip as P
    model = _model(g, "d64", torch.float32)
    meta = g["meta"]
    n, pad, eos = meta["max_new_tokens"], meta["pad_id"], meta["cases"]["d64"]["eos"]

    class Tok:                                      # batch_decode(ids, skip_special_tokens=True): ids as words, specials dropped
        def batch_decode(self, ids, skip_special_tokens=False):
            return [" ".join(str(int(t)) for t in row if not (skip_special_tokens and int(t) in (pad, eos))) for row in ids]

    batch = dict(name=["P1", "P2", "P3"], input_ids=torch.from_numpy(g["input_ids"]), attention_mask=torch.from_numpy(g["attention_mask"]),
                 protein_input_ids=torch.from_numpy(g["protein_input_ids"]), protein_attention_mask=torch.from_numpy(g["protein_attention_mask"]),
                 description_input_ids=torch.tensor([[7, 8, pad], [9, eos, pad], [1, 2, 3]]))
    out = P.iterative_generation_loop(0, model, batch, n, eos_token_id=eos, pad_token_id=pad)
    assert np.array_equal(to_np(out), g["d64.greedy_eos"])
    args = dict(max_generation_length=n, num_beams=1, length_penalty=1.0, temperature=1.0, do_sample=False, top_p=1.0, top_k=50,
                save_generation_dir=str(tmp_path), save_generation_postfix_identifier="test")
    # (upstream hard-codes the Llama-3 ids 128009 / 128002, which this 512-token fixture does not have: no row stops early here)
    path = P.inference_epoch(0, model, [batch], Tok(), args)
    rec = json.load(open(path))
    assert os.path.basename(path) == "generation_test_rank0.json" and list(rec) == ["P1", "P2", "P3"]
    assert rec["P1"]["true"] == "7 8" and rec["P2"]["true"] == "9" and rec["P3"]["true"] == "1 2 3"
    assert rec["P2"]["pred"].split() == [str(t) for t in g["d64.greedy"][1] if t not in (pad, eos)]


@pytest.mark.parametrize("case", ["d64", "d128", "d16"])
def test_fp8_gemm_model_generates_like_its_own_cacheless_forward(g, case):
    """`set_gemm_dtype("fp8")` models: the decode step streams e4m3 weights (p2t_gemm_nt_skinny_fp8, rows quantised
    by the RMSNorm / a quantise pass as in the prefill).  Per-step logits against the model's own cache-less fp8 forward over prompt
    + generated ids, and against the fp8 oracle fed the same tokens; stream copies == row-major weights bit for bit."""
    model = _model(g, case, torch.bfloat16)
    model.set_gemm_dtype("fp8")
    meta = g["meta"]
    m = meta["cases"][case]
    n, pad = 6, meta["pad_id"]
    kw = dict(max_new_tokens=n, eos_token_id=None, pad_token_id=pad, do_sample=False, return_dict_in_generate=True, output_logits=True)
    out = model.generate(**_inputs(g), **kw)
    out_n = model.generate(**_inputs(g), **kw, stream_copy=False, use_graph=False)
    toks, lg = to_np(out.sequences), to_np(torch.stack(out.logits, 0))
    assert torch.equal(out.sequences, out_n.sequences) and torch.equal(torch.stack(out.logits, 0), torch.stack(out_n.logits, 0))
    emb, mask = model(input_ids=to_dev(g["input_ids"]), attention_mask=to_dev(g["attention_mask"]), protein_input_ids=to_dev(g["protein_input_ids"]),
                      protein_attention_mask=to_dev(g["protein_attention_mask"]), return_decoder_inputs=True)
    dec = model.llama_decoder
    for b in range(toks.shape[0]):
        valid = to_np(mask)[b] != 0
        row = torch.cat([emb[b][torch.from_numpy(valid).to(emb.device)], dec.model.embed(to_dev(toks[b:b + 1, :-1]))[0]], 0)[None]
        full = to_np(dec(inputs_embeds=row).logits.float())[0]
        n0 = int(valid.sum())
        observe(f"generate[{case}].fp8_cache_vs_full_forward.row{b}", rel(lg[:, b], full[n0 - 1:n0 - 1 + n]), 8e-2)
    model.set_gemm_dtype("model")
    esm, llama, ad = specs.EsmSpec(**m["esm"]), specs.LlamaSpec(**m["llama"]), specs.AdapterSpec(**m["adapter"])
    W = model_weights(esm, llama, ad, m["weight_seed"], lm_head=True)
    _, ref = O.generate_greedy(llama, W, to_np(emb), to_np(mask), n, (), pad, prec=O.FP8, forced=toks)
    observe(f"generate[{case}].fp8_vs_fp8oracle.logits", rel(lg, ref), 1e-1)


@pytest.mark.parametrize("case", ["d64", "qwen3"])
def test_long_generation_crosses_the_tile_boundaries_of_the_generated_segment(g, case):
    """70 new tokens (the generated segment grows past its first 64-key tile, capacity 128) and a single new token: fp32 logits of every
    step against the cache-less oracle fed the same tokens (2e-4), ids equal to the oracle's own greedy choice wherever its top-2 gap is
    not a near-tie."""
    model = _model(g, case, torch.float32)
    meta = g["meta"]
    m = meta["cases"][case]
    n, pad = 70, meta["pad_id"]
    out = model.generate(**_inputs(g), max_new_tokens=n, eos_token_id=None, pad_token_id=pad, do_sample=False, return_dict_in_generate=True,
                         output_logits=True, sync_every=32)
    toks, lg = to_np(out.sequences), to_np(torch.stack(out.logits, 0))
    assert toks.shape == (3, n)
    esm, llama, ad = specs.EsmSpec(**m["esm"]), specs.LlamaSpec(**m["llama"]), specs.AdapterSpec(**m["adapter"])
    W = model_weights(esm, llama, ad, m["weight_seed"], lm_head=True)
    emb, mask = model(input_ids=to_dev(g["input_ids"]), attention_mask=to_dev(g["attention_mask"]), protein_input_ids=to_dev(g["protein_input_ids"]),
                      protein_attention_mask=to_dev(g["protein_attention_mask"]), return_decoder_inputs=True)
    _, ref = O.generate_greedy(llama, W, to_np(emb), to_np(mask), n, (), pad, forced=toks)
    assert np.abs(lg - ref).max() < 2e-4
    top2 = np.sort(ref, axis=-1)[..., -2:]
    clear = (top2[..., 1] - top2[..., 0]) > 1e-3                       # [n, B]
    assert np.array_equal(toks.T[clear], ref.argmax(-1)[clear]) and clear.mean() > 0.95
    one = model.generate(**_inputs(g), max_new_tokens=1, eos_token_id=None, pad_token_id=pad, do_sample=False)
    assert np.array_equal(to_np(one), toks[:, :1])


def test_long_beam_search_reorders_more_than_one_tile_of_generated_keys(g):
    """3 beams x 70 new tokens: the generated cache segment that follows the beams (p2t_kv_reorder) grows past 64 keys.  fp32 ids and
    scores against the cache-less oracle's beam search (kernels are deterministic: the comparison is stable once it holds)."""
    case = "d64"
    model = _model(g, case, torch.float32)
    meta = g["meta"]
    m = meta["cases"][case]
    n, pad = 70, meta["pad_id"]
    out = model.generate(**_inputs(g), max_new_tokens=n, eos_token_id=m["eos"], pad_token_id=pad, do_sample=False, num_beams=3, length_penalty=1.0,
                         return_dict_in_generate=True, output_scores=True)
    esm, llama, ad = specs.EsmSpec(**m["esm"]), specs.LlamaSpec(**m["llama"]), specs.AdapterSpec(**m["adapter"])
    W = model_weights(esm, llama, ad, m["weight_seed"], lm_head=True)
    emb, mask = model(input_ids=to_dev(g["input_ids"]), attention_mask=to_dev(g["attention_mask"]), protein_input_ids=to_dev(g["protein_input_ids"]),
                      protein_attention_mask=to_dev(g["protein_attention_mask"]), return_decoder_inputs=True)
    seq, sc = O.generate_beam(llama, W, to_np(emb), to_np(mask), n, 3, (m["eos"],), pad, 1.0)
    assert np.array_equal(to_np(out.sequences), seq) and seq.shape[1] > 64
    assert np.abs(to_np(out.sequences_scores) - sc).max() < 1e-4


def test_more_than_64_rows_decode_on_the_row_major_weights(g):
    """with the default stream copies a bf16 `generate` over more than 64 rows (prompts x beams) handed the
    PRE-SHUFFLED LM head to the general GEMM, which read it as a row-major matrix: wrong tokens, no error.  72 greedy rows (the three
    golden prompts tiled 24 times) must give, row for row, the tokens and logits of the same prompts in a 3-row call (rows never
    interact); 16 prompts x 5 beams = 80 rows must score like the same prompts run four at a time (20 rows); and the C entry point refuses a
    pre-shuffled LM head beyond 64 rows instead of mis-reading it."""
    import ctypes as C
    from p2t_hip import _lib, generation
    case = "d64"
    model = _model(g, case, torch.bfloat16)
    meta = g["meta"]
    pad, n = meta["pad_id"], 6
    kw = _inputs(g)
    small = model.generate(**kw, max_new_tokens=n, eos_token_id=None, pad_token_id=pad, do_sample=False, return_dict_in_generate=True, output_logits=True)
    rep = 24
    big_kw = {k: v.repeat(rep, *([1] * (v.dim() - 1))) for k, v in kw.items()}
    big = model.generate(**big_kw, max_new_tokens=n, eos_token_id=None, pad_token_id=pad, do_sample=False, return_dict_in_generate=True, output_logits=True)
    ts, tb = to_np(small.sequences), to_np(big.sequences)
    assert tb.shape == (3 * rep, n)
    ls, lb = to_np(torch.stack(small.logits, 0)), to_np(torch.stack(big.logits, 0))
    for r in range(rep):
        # the skinny kernels (3 rows) and the general GEMM (72 rows) sum in different orders: compare logits, and tokens where the top-2 gap is clear
        assert rel(lb[:, 3 * r:3 * r + 3], ls) < 2e-2, r
        top2 = np.sort(ls.astype(np.float64), axis=-1)[..., -2:]
        clear = (top2[..., 1] - top2[..., 0]) > 0.05 * np.abs(top2[..., 1]).clip(1e-3)
        assert np.array_equal(tb[3 * r:3 * r + 3].T[clear], ts.T[clear]), r
    # beams: 16 prompts x 5 beams = 80 rows against the same prompts run 4 at a time (20 rows: the skinny kernels)
    rep = 6
    b_kw = {k: v.repeat(rep, *([1] * (v.dim() - 1)))[:16] for k, v in kw.items()}
    m = meta["cases"][case]
    wide = model.generate(**b_kw, max_new_tokens=n, eos_token_id=m["eos"], pad_token_id=pad, do_sample=False, num_beams=5, return_dict_in_generate=True,
                          output_scores=True)
    for i in range(0, 16, 4):
        part = {k: v[i:i + 4] for k, v in b_kw.items()}
        narrow = model.generate(**part, max_new_tokens=n, eos_token_id=m["eos"], pad_token_id=pad, do_sample=False, num_beams=5,
                                return_dict_in_generate=True, output_scores=True)
        diff = np.abs(to_np(wide.sequences_scores)[i:i + 4] - to_np(narrow.sequences_scores))
        assert np.isfinite(diff).all() and np.median(diff) < 5e-2 and diff.max() < 0.3, (i, diff)     # bf16 logits, two GEMM kernels: near-tied beams may swap
    # the C boundary itself: a pre-shuffled LM head with 65 rows is an argument error
    eng = generation.DecodeEngine(model.llama_decoder, 65, 1, 8, 8, stream_copy=True)
    assert eng.stream is None                                              # the engine did not even build the copies
    st = generation.stream_weights(model.llama_decoder)
    layers = C.cast(st["layers"], C.POINTER(_lib.LlamaLayerStreamC))
    from p2t_hip.ops import ptr, stream
    with pytest.raises(Exception, match="at most 64 rows"):
        _lib.call("p2t_llama_decode_step", C.byref(eng.e["cfg"]), C.byref(eng.e["w"]), layers, ptr(st["lm_head"]), eng.lm_head.stride(0), 1,
                  C.byref(eng.cache), ptr(eng.x), ptr(eng.logits), eng.ld_logits, eng.flags, ptr(eng.ws), eng.ws.numel(), stream())
