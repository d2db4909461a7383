"""Stage-2 step with LoRA adapters on the decoder ("LoRA later") and with Qwen3's per-head q / k
RMSNorm (row 4, the text half): p2t_hip/decoder_train.py against torch autograd through the REFERENCE class with every target
projection wrapped by a hand-written LoRA linear (tests/golden/sft_lora_tiny.npz, make_golden.py run_sft_lora: r = 4, alpha = 8,
dropout 0) -- loss, gradient at the decoder inputs (through the adapter's four gradients), dA / dB of every (layer, target); fp32 at
5e-4 (north_star allows 1e-3), bf16 with observed tolerances.  `peft` is not importable here: parity against peft's own code is
unpinned, the arithmetic is LoRA's published one (scripts/train_instruct.py:146-183 is where the reference configures it).
The branch's input dropout (lora_dropout = 0.1 upstream) cannot be pinned on a torch mask: it is checked by what any mask must
satisfy -- same mask forward and backward (directional derivative == <gradient, direction>), expectation preserved."""
import json
import os

import numpy as np
import pytest
import torch

from gpu_util import build_model, dev, observe, rel, to_dev, to_np
from p2t_hip import specs, synth

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ADAPTER = ("fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias")


@pytest.fixture(scope="module")
def g():
    z = np.load(os.path.join(HERE, "golden", "sft_lora_tiny.npz"))
    d = {k: z[k] for k in z.files}
    d["meta"] = json.loads(bytes(d.pop("meta_json")).decode())
    return d


def _model(g, case, dtype, dropout=0.0):
    meta = g["meta"]
    m = meta["cases"][case]
    model = build_model(specs.EsmSpec(**m["esm"]), specs.LlamaSpec(**m["llama"]), specs.AdapterSpec(**m["adapter"]), dtype, 0)
    model.config.placeholder_id = meta["placeholder_id"]
    model.eval()
    model.requires_grad_(False)
    lora = None
    if m["lora"]:
        lora = model.add_lora(meta["r"], meta["alpha"], dropout, meta["targets"])
        P = dict(model.llama_decoder.model.named_parameters())
        with torch.no_grad():
            for i in range(model.llama_decoder.spec.num_hidden_layers):
                for t in meta["targets"]:
                    a, b = lora.get(i, t)
                    w = P[f"layers.{i}.{t}.weight"]
                    a.copy_(to_dev(synth.uniform_f32(meta["lora_seed"], f"lora.{i}.{t}.A", (meta["r"], w.shape[1]), 0.25)))
                    b.copy_(to_dev(synth.uniform_f32(meta["lora_seed"], f"lora.{i}.{t}.B", (w.shape[0], meta["r"]), 0.25)))
    model.adapter.requires_grad_(True)                  # modules_to_save = adapter.fc1 / fc2
    return model, lora


def _inputs(g):
    return dict(input_ids=to_dev(g["input_ids"]), attention_mask=to_dev(g["attention_mask"]),
                protein_input_ids=to_dev(g["protein_input_ids"]), protein_attention_mask=to_dev(g["protein_attention_mask"]))


@pytest.mark.parametrize("case", ["d16", "d64", "d128", "qwen3", "qwen3_lora"])
def test_fp32_lora_step_matches_reference_autograd(g, case):
    model, lora = _model(g, case, torch.float32)
    out = model(**_inputs(g), labels=to_dev(g["labels"]))
    assert abs(float(out.loss) - float(g[f"{case}.loss"])) < 2e-5 * max(1.0, float(g[f"{case}.loss"]))
    out.loss.backward()
    for n in ADAPTER:
        got = dict(model.adapter.named_parameters())[n].grad
        assert rel(to_np(got), g[f"{case}.grad.{n}"]) < 5e-4, (case, n, rel(to_np(got), g[f"{case}.grad.{n}"]))
    if lora is not None:
        worst = 0.0
        for i in range(model.llama_decoder.spec.num_hidden_layers):
            for t in g["meta"]["targets"]:
                a, b = lora.get(i, t)
                ea, eb = rel(to_np(a.grad), g[f"{case}.lora.{i}.{t}.dA"]), rel(to_np(b.grad), g[f"{case}.lora.{i}.{t}.dB"])
                worst = max(worst, ea, eb)
                assert ea < 5e-4 and eb < 5e-4, (case, i, t, ea, eb)
        print(f"{case}: worst LoRA gradient error {worst:.2e}")
    else:
        assert model.llama_decoder.spec.qk_norm                   # the Qwen3 case without LoRA pins the q / k-norm backward on its own


@pytest.mark.parametrize("case", ["d64", "d128", "qwen3_lora"])
def test_bf16_lora_step_close_to_reference_autograd(g, case):
    model, lora = _model(g, case, torch.bfloat16)
    out = model(**_inputs(g), labels=to_dev(g["labels"]))
    out.loss.backward()
    observe(f"sft_lora[{case}].bf16.loss", abs(float(out.loss) - float(g[f"{case}.loss"])) / float(g[f"{case}.loss"]), 3e-2)
    for n in ADAPTER:
        observe(f"sft_lora[{case}].bf16.{n}", rel(to_np(dict(model.adapter.named_parameters())[n].grad), g[f"{case}.grad.{n}"]), 1.5e-1)
    ga, gb, ra, rb = [], [], [], []
    for i in range(model.llama_decoder.spec.num_hidden_layers):
        for t in g["meta"]["targets"]:
            a, b = lora.get(i, t)
            ga.append(to_np(a.grad).ravel()); ra.append(g[f"{case}.lora.{i}.{t}.dA"].ravel())
            gb.append(to_np(b.grad).ravel()); rb.append(g[f"{case}.lora.{i}.{t}.dB"].ravel())
    observe(f"sft_lora[{case}].bf16.dA", rel(np.concatenate(ga), np.concatenate(ra)), 1.5e-1)
    observe(f"sft_lora[{case}].bf16.dB", rel(np.concatenate(gb), np.concatenate(rb)), 1.5e-1)


def test_lora_dropout_mask_is_shared_by_forward_and_backward(g):
    """lora_dropout = 0.3 on the d16 case, fp32: along a random direction of one A and one B matrix the central difference of the loss
    (same step counter, hence the same regenerated masks) equals <gradient, direction>; with B = 0 (peft's start) the loss does not
    depend on A at all and the step equals the LoRA-free one."""
    case = "d16"
    model, lora = _model(g, case, torch.float32, dropout=0.3)
    lora.eval()                                         # keeps the step counter (the mask seed) fixed between evaluations
    batch, labels = _inputs(g), to_dev(g["labels"])
    loss = model(**batch, labels=labels).loss
    loss.backward()
    assert abs(float(loss) - float(g[f"{case}.loss"])) > 1e-4          # the mask changed the branch
    rs = np.random.RandomState(0)
    for i, t, which in ((1, "self_attn.v_proj", 0), (2, "mlp.down_proj", 1), (0, "mlp.gate_proj", 0), (1, "self_attn.o_proj", 1)):
        prm = lora.get(i, t)[which]
        v = to_dev(rs.standard_normal(tuple(prm.shape)).astype(np.float32))
        want = float((prm.grad * v).sum())
        eps = 2e-3
        with torch.no_grad():
            prm.add_(eps * v)
            lp = float(model(**batch, labels=labels).loss)
            prm.add_(-2 * eps * v)
            lm = float(model(**batch, labels=labels).loss)
            prm.add_(eps * v)
        fd = (lp - lm) / (2 * eps)
        assert abs(fd - want) < 2e-2 * max(abs(want), 1e-2), (i, t, which, fd, want)
    # B = 0: the branch is silent
    with torch.no_grad():
        for i in range(model.llama_decoder.spec.num_hidden_layers):
            for t in g["meta"]["targets"]:
                lora.get(i, t)[1].zero_()
    base, _ = _model(g, case, torch.float32)
    base.llama_decoder.lora = None
    l0 = float(base(**batch, labels=labels).loss)
    assert abs(float(model(**batch, labels=labels).loss) - l0) < 1e-5


def test_peft_style_state_dict_round_trips_through_the_merge_loader(g, tmp_path):
    """DecoderLora.peft_state_dict() -> p2t_hip.lora.load_and_merge_adapter (the inference-time merge): generation-free check that the
    merged decoder's loss equals the LoRA decoder's (fp32, dropout off)."""
    case = "d16"
    model, lora = _model(g, case, torch.float32)
    batch, labels = _inputs(g), to_dev(g["labels"])
    with torch.no_grad():
        pass
    l_lora = float(model(**batch, labels=labels).loss)
    sd = lora.peft_state_dict()
    merged, _ = _model(g, case, torch.float32)
    merged.llama_decoder.lora = None
    P = dict(merged.llama_decoder.model.named_parameters())
    with torch.no_grad():                               # W + (alpha / r) B A, what a merge does
        for i in range(merged.llama_decoder.spec.num_hidden_layers):
            for t in g["meta"]["targets"]:
                a = sd[f"base_model.model.llama_decoder.model.layers.{i}.{t}.lora_A.weight"]
                b = sd[f"base_model.model.llama_decoder.model.layers.{i}.{t}.lora_B.weight"]
                P[f"layers.{i}.{t}.weight"].add_((lora.scale * (b @ a)).to(P[f"layers.{i}.{t}.weight"].dtype))
    merged.llama_decoder.model.invalidate_engine()
    with torch.no_grad():
        l_merged = float(merged(**batch, labels=labels).loss)
    assert abs(l_lora - l_merged) < 2e-5 * max(1.0, abs(l_merged))


def test_lora_at_llama8b_layer_shapes_agrees_with_the_fused_frozen_chain_and_its_own_finite_differences():
    """Llama-3.1-8B's layer shapes (hidden 4096, 32 / 8 heads of 128, FFN 14336, llama3 rotary scaling; two layers, fp32), where no
    golden reaches: (1) with B = 0 (peft's start) the per-layer LoRA path gives the loss and the adapter gradients of the fused
    frozen-decoder chain (p2t_llama_train_forward / _backward) -- two independent drivers of the same kernels; (2) with B != 0, rank 16,
    the directional derivative of the loss along a random direction of one A and one B equals <gradient, direction>."""
    esm = specs.EsmSpec(num_hidden_layers=2, hidden_size=128, intermediate_size=256, num_attention_heads=2)
    llama = specs.LlamaSpec(num_hidden_layers=2, hidden_size=4096, intermediate_size=14336, num_attention_heads=32, num_key_value_heads=8,
                            vocab_size=2048, rope_type="llama3", rope_theta=500000.0, rope_factor=8.0)
    ad = specs.AdapterSpec(128, 192, 4096, 0.0)
    model = build_model(esm, llama, ad, torch.float32, 5).eval()
    model.requires_grad_(False)
    model.adapter.requires_grad_(True)
    ph = 2047
    model.config.placeholder_id = ph
    rng = np.random.default_rng(3)
    lens = [20, 9]
    pid, pmask = synth.protein_batch(9, 2, 22, [n + 2 for n in lens])
    T_prompt, T_desc = 30, 14
    ids = np.full((2, T_prompt + T_desc), 2046, dtype=np.int64)
    mask = np.zeros_like(ids)
    labels = np.full_like(ids, -100)
    for b, n in enumerate(lens):
        prompt = np.concatenate([rng.integers(0, 2000, 3), np.full(n + 2, ph), rng.integers(0, 2000, 2)])
        ids[b, T_prompt - len(prompt):T_prompt] = prompt
        mask[b, T_prompt - len(prompt):T_prompt] = 1
        nd = int(rng.integers(5, T_desc + 1))
        desc = rng.integers(0, 2000, nd)
        ids[b, T_prompt:T_prompt + nd] = desc
        mask[b, T_prompt:T_prompt + nd] = 1
        labels[b, T_prompt:T_prompt + nd] = desc
    kw = dict(input_ids=to_dev(ids), attention_mask=to_dev(mask), labels=to_dev(labels), protein_input_ids=to_dev(pid),
              protein_attention_mask=to_dev(pmask))
    fused = model(**kw)
    fused.loss.backward()
    g_fused = {n: p.grad.detach().clone() for n, p in model.adapter.named_parameters() if p.grad is not None}
    model.zero_grad(set_to_none=True)
    lora = model.add_lora(16, lora_dropout=0.0)
    model.adapter.requires_grad_(True)
    out = model(**kw)
    out.loss.backward()
    assert abs(float(out.loss) - float(fused.loss)) < 2e-5 * max(1.0, abs(float(fused.loss)))
    for n, p in model.adapter.named_parameters():
        if n in g_fused:
            assert rel(to_np(p.grad), to_np(g_fused[n])) < 2e-4, n
    a0, b0 = lora.get(0, "mlp.down_proj")
    assert float(b0.grad.abs().max()) > 0 and float(a0.grad.abs().max()) == 0.0           # B = 0: dA vanishes, dB does not
    # (2) non-zero B, directional derivatives
    with torch.no_grad():
        for q in lora.parameters():
            if q.shape[1] == 16:                        # the B matrices
                q.copy_(torch.from_numpy(rng.standard_normal(tuple(q.shape)).astype(np.float32) * 0.02).to(q.device))
    model.zero_grad(set_to_none=True)
    model(**kw).loss.backward()
    for i, t, which in ((1, "self_attn.q_proj", 0), (0, "mlp.up_proj", 1)):
        prm = lora.get(i, t)[which]
        v = torch.from_numpy(rng.standard_normal(tuple(prm.shape)).astype(np.float32)).to(prm.device)
        v *= float(max(prm.detach().norm(), 1e-2) / v.norm())
        want = float((prm.grad * v).sum())
        eps = 2e-2
        with torch.no_grad():
            prm.add_(eps * v)
            lp = float(model(**kw).loss)
            prm.add_(-2 * eps * v)
            lm = float(model(**kw).loss)
            prm.add_(eps * v)
        got = (lp - lm) / (2 * eps)
        assert abs(want) > 1e-5 and abs(got - want) < 3e-2 * abs(want), (i, t, which, got, want)
