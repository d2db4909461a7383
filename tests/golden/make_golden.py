#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REFERENCE ITSELF (run in the build container only).

The reference ships no tests or golden vectors, so parity is pinned on
outputs of the reference's own Python for this path, imported from /root/reference exactly as
SURVEY.md Appendix A describes: synthetic `models` / `dataset` / `scripts` packages (so the
package __init__ files that need torch_geometric/graphein are not executed) plus name-only stubs
for the absent third-party `esm` package that scripts/train_contrast.py imports but this path
never calls.  What runs is the reference's code, verbatim:

    models.modeling_esm2llama_instruct.Esm2LlamaInstructForCausalLM / ModalityAdapter
    models.configuration_esm2llama_instruct.Esm2LlamaInstructConfig, models.modality_config
    scripts.train_contrast.{readout_embeddings, get_description_embeddings, BatchInfoNCELoss,
                            SegmentedBatchInfoNCELoss}

on random-init towers built from local config objects (no from_pretrained, no network) whose
weights come from the repo's counter-hash generator (p2t_hip.synth / p2t_hip.specs), so the
fixtures hold only inputs' seeds and expected outputs -- never weights, never reference source.

Usage (from /root/repo):  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py [--only tiny,cfg1]
This script and /root/reference never travel to the GPU box; only the .npz files do.
"""
from __future__ import annotations

import argparse
import importlib
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "prot2text-v2-esm3_amd"))
from p2t_hip import specs, synth  # noqa: E402

REF = "/root/reference"


def load_reference():
    def _stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    _stub("esm").__path__ = []
    _stub("esm.models").__path__ = []
    _stub("esm.models.esmc", ESMC=type("ESMC", (), {}))
    _stub("esm.utils", encoding=types.SimpleNamespace(tokenize_sequence=None)).__path__ = []
    _stub("esm.utils.misc", stack_variable_length_tensors=None)
    for p in ("models", "dataset", "scripts"):
        _stub(p).__path__ = [f"{REF}/{p}"]
    mm = importlib.import_module("models.modeling_esm2llama_instruct")
    mc = importlib.import_module("models.modality_config")
    cc = importlib.import_module("models.configuration_esm2llama_instruct")
    ec = importlib.import_module("models.esmc_config")
    eq = importlib.import_module("models.esmc_qwen_arc")
    sys.modules["models"].__dict__.update(ModalityAdapter=mm.ModalityAdapter,
                                          ModalityAdapterConfig=mc.ModalityAdapterConfig,
                                          ESMCConfig=ec.ESMCConfig, ESMCQwen=eq.ESMCQwen)
    dl = importlib.import_module("dataset.dataloader_light")
    sys.modules["dataset"].__dict__.update(Prot2TextLightDataset=dl.Prot2TextLightDataset,
                                           Prot2TextLightCollater=dl.Prot2TextLightCollater)
    tc = importlib.import_module("scripts.train_contrast")
    return types.SimpleNamespace(mm=mm, mc=mc, cc=cc, tc=tc)


def build_reference_model(ref, esm: specs.EsmSpec, llama: specs.LlamaSpec, ad: specs.AdapterSpec, seed=0):
    from transformers import EsmConfig, LlamaConfig
    from transformers.models.esm.modeling_esm import EsmModel
    from transformers.models.llama import LlamaForCausalLM

    ecfg = EsmConfig(vocab_size=esm.vocab_size, hidden_size=esm.hidden_size,
                     num_hidden_layers=esm.num_hidden_layers, num_attention_heads=esm.num_attention_heads,
                     intermediate_size=esm.intermediate_size, hidden_dropout_prob=0.0,
                     attention_probs_dropout_prob=0.0, max_position_embeddings=esm.max_position_embeddings,
                     layer_norm_eps=esm.layer_norm_eps, position_embedding_type="rotary",
                     token_dropout=esm.token_dropout, emb_layer_norm_before=esm.emb_layer_norm_before,
                     pad_token_id=esm.pad_token_id, mask_token_id=esm.mask_token_id)
    ecfg._attn_implementation = "eager"
    rope = {"rope_type": llama.rope_type, "rope_theta": llama.rope_theta}
    if llama.rope_type == "llama3":
        rope.update(factor=llama.rope_factor, low_freq_factor=llama.rope_low_freq_factor,
                    high_freq_factor=llama.rope_high_freq_factor,
                    original_max_position_embeddings=llama.rope_original_max_position_embeddings)
    lcfg = LlamaConfig(vocab_size=llama.vocab_size, hidden_size=llama.hidden_size,
                       intermediate_size=llama.intermediate_size, num_hidden_layers=llama.num_hidden_layers,
                       num_attention_heads=llama.num_attention_heads,
                       num_key_value_heads=llama.num_key_value_heads, rms_norm_eps=llama.rms_norm_eps,
                       max_position_embeddings=llama.max_position_embeddings, rope_parameters=rope,
                       tie_word_embeddings=llama.tie_word_embeddings, attention_bias=False, mlp_bias=False,
                       attention_dropout=0.0, pad_token_id=None, bos_token_id=None, eos_token_id=None)
    lcfg._attn_implementation = "eager"
    if llama.qk_norm:
        # the text tower the fork's scripts instantiate is a Qwen3 (models/esmc_config.py:9): random-init HF Qwen3ForCausalLM
        # from a local config -- same decoder interface (`.model(..., output_hidden_states=True)`) as the Llama one
        from transformers import Qwen3Config, Qwen3ForCausalLM
        lcfg = Qwen3Config(vocab_size=llama.vocab_size, hidden_size=llama.hidden_size, intermediate_size=llama.intermediate_size,
                           num_hidden_layers=llama.num_hidden_layers, num_attention_heads=llama.num_attention_heads,
                           num_key_value_heads=llama.num_key_value_heads, head_dim=llama.head_dim, rms_norm_eps=llama.rms_norm_eps,
                           max_position_embeddings=llama.max_position_embeddings, rope_parameters=rope,
                           tie_word_embeddings=llama.tie_word_embeddings, attention_bias=False, attention_dropout=0.0,
                           use_sliding_window=False, pad_token_id=None, bos_token_id=None, eos_token_id=None)
        lcfg._attn_implementation = "eager"
    with torch.device("cpu"):
        esm_m = EsmModel(ecfg, add_pooling_layer=False)
        llama_m = Qwen3ForCausalLM(lcfg) if llama.qk_norm else LlamaForCausalLM(lcfg)
        acfg = ref.mc.ModalityAdapterConfig(input_dim=ad.input_dim, intermediate_dim=ad.intermediate_dim,
                                            output_dim=ad.output_dim, dropout_rate=ad.dropout_rate)
        adapter = ref.mm.ModalityAdapter(acfg)
    model = ref.mm.Esm2LlamaInstructForCausalLM(esm_encoder=esm_m, adapter=adapter, llama_decoder=llama_m)

    def load(module, tensors, allowed_missing):
        sd = {k: torch.from_numpy(v) for k, v in specs.materialize(tensors, seed).items()}
        res = module.load_state_dict(sd, strict=False)
        bad = [k for k in res.missing_keys if not any(a in k for a in allowed_missing)]
        assert not bad and not res.unexpected_keys, (bad, res.unexpected_keys)

    load(model, specs.esm_tensors(esm, "esm_encoder."), ["adapter.", "llama_decoder.", "inv_freq", "contact_head"])
    load(model, specs.adapter_tensors(ad, "adapter."), ["esm_encoder.", "llama_decoder."])
    load(model, specs.llama_tensors(llama, "llama_decoder."), ["esm_encoder.", "adapter.", "lm_head", "inv_freq"])
    model.eval()
    model.requires_grad_(False)
    return model


def run_case(ref, name, esm, llama, ad, B, T_p, T_t, p_lens, t_lens, layers, id_high, pad_id, eos_id,
             store_hidden=True, grads=True, seed_w=0, seed_in=1234):
    tc = ref.tc
    model = build_reference_model(ref, esm, llama, ad, seed_w)
    pid, pmask = synth.protein_batch(seed_in, B, T_p, p_lens)
    tid, tmask = synth.text_batch(seed_in, B, T_t, id_high, t_lens, pad_id, eos_id)
    pid_t, pmask_t = torch.from_numpy(pid), torch.from_numpy(pmask)
    tid_t, tmask_t = torch.from_numpy(tid), torch.from_numpy(tmask)
    out = {}
    F = torch.nn.functional
    with torch.no_grad():
        enc = model(protein_input_ids=pid_t, protein_attention_mask=pmask_t, return_encoder_outputs=True)[0]
        ad_out, ad_mask = model(protein_input_ids=pid_t, protein_attention_mask=pmask_t,
                                return_adapter_outputs=True)
        assert torch.equal(ad_mask, pmask_t)
        shim = types.SimpleNamespace(llm_decoder=model.llama_decoder)
        ones = torch.ones_like(pmask_t)
        for ro in ("last", "mean", "std", "mix"):
            out[f"prot_pooled_{ro}"] = tc.readout_embeddings(ad_out, pmask_t, ro).numpy()
        out["prot_pooled_mix_onesmask"] = tc.readout_embeddings(ad_out, ones, "mix").numpy()
        p_mix = F.normalize(tc.readout_embeddings(ad_out, pmask_t, "mix"), p=2, dim=-1)
        p_mix_ones = F.normalize(tc.readout_embeddings(ad_out, ones, "mix"), p=2, dim=-1)
        p_mean = F.normalize(tc.readout_embeddings(ad_out, pmask_t, "mean"), p=2, dim=-1)
        out["prot_norm_mix"], out["prot_norm_mix_onesmask"] = p_mix.numpy(), p_mix_ones.numpy()
        out["prot_norm_mean"] = p_mean.numpy()
        hs_all = model.llama_decoder.model(input_ids=tid_t, attention_mask=tmask_t, use_cache=False,
                                           output_attentions=False, output_hidden_states=True,
                                           return_dict=True).hidden_states
        assert len(hs_all) == llama.num_hidden_layers + 1
        for k in layers:
            # the reference's own text-side function, verbatim (readout "mix")
            d = tc.get_description_embeddings(shim, tid_t, tmask_t, output_llm_layer=k)
            out[f"text_pooled_mix_L{k}"] = d.numpy()
            t_mix = F.normalize(d, p=2, dim=-1)
            out[f"text_norm_mix_L{k}"] = t_mix.numpy()
            t_mean = F.normalize(tc.readout_embeddings(hs_all[k], tmask_t, "mean"), p=2, dim=-1)
            out[f"text_norm_mean_L{k}"] = t_mean.numpy()
            if store_hidden:
                out[f"text_hidden_L{k}"] = hs_all[k].numpy()
            labels_all = torch.arange(B)
            out[f"loss_batch_mix_L{k}"] = tc.BatchInfoNCELoss()(p_mix, t_mix).numpy()
            out[f"loss_batch_mix_onesmask_L{k}"] = tc.BatchInfoNCELoss()(p_mix_ones, t_mix).numpy()
            out[f"loss_batch_mean_L{k}"] = tc.BatchInfoNCELoss()(p_mean, t_mean).numpy()
            for nseg in (1, 2):
                seg = B // nseg
                acc = torch.zeros([])
                for s in range(nseg):
                    acc += tc.SegmentedBatchInfoNCELoss()(p_mix[s * seg:(s + 1) * seg], t_mix,
                                                          labels_all[s * seg:(s + 1) * seg])
                out[f"loss_seg{nseg}_mix_L{k}"] = (acc / nseg).numpy()
            out[f"logits_mix_L{k}"] = (p_mix @ t_mix.t() / 0.05).numpy()
        if store_hidden:
            out["esm_last_hidden"] = enc.numpy()
            out["adapter_out"] = ad_out.numpy()
        else:   # strided sample keeps the fixture small
            out["esm_last_hidden_s"] = enc[:, ::7, ::5].contiguous().numpy()
            out["adapter_out_s"] = ad_out[:, ::7, ::13].contiguous().numpy()
    if grads:
        # adapter gradients through the reference loss: train-mode forward with dropout_rate = 0
        k = layers[-1]
        model.adapter.requires_grad_(True)
        model.adapter.dropout.p = 0.0
        model.adapter.train()
        ad_out, _ = model(protein_input_ids=pid_t, protein_attention_mask=pmask_t, return_adapter_outputs=True)
        t_mix = torch.from_numpy(out[f"text_norm_mix_L{k}"])
        for nseg in (1, 2):
            model.adapter.zero_grad()
            p_mix = F.normalize(tc.readout_embeddings(ad_out, pmask_t, "mix"), p=2, dim=-1)
            seg = B // nseg
            acc = torch.zeros([])
            for s in range(nseg):
                acc = acc + tc.SegmentedBatchInfoNCELoss()(p_mix[s * seg:(s + 1) * seg], t_mix,
                                                           torch.arange(s * seg, (s + 1) * seg))
            (acc / nseg).backward(retain_graph=True)
            for n, prm in model.adapter.named_parameters():
                if prm.grad is not None:
                    gnp = prm.grad.numpy()
                    if not store_hidden and gnp.ndim == 2:     # big model: strided sample + exact norm
                        out[f"gradnorm_seg{nseg}_{n}"] = np.float32(np.linalg.norm(gnp.astype(np.float64)))
                        gnp = gnp[::7, ::5]
                    out[f"grad_seg{nseg}_{n}"] = gnp.copy()
                else:
                    assert n.startswith("ln"), n      # ln1/ln2 are unused
    if grads and store_hidden:
        # one clip + AdamW step with the reference's optimizer settings (train_contrast.py:621-626)
        model.adapter.zero_grad()
        p_mix = F.normalize(tc.readout_embeddings(ad_out, pmask_t, "mix"), p=2, dim=-1)
        loss = tc.SegmentedBatchInfoNCELoss()(p_mix, t_mix, torch.arange(B))
        loss.backward()
        opt = torch.optim.AdamW(model.adapter.parameters(), lr=2e-4, eps=1e-6, betas=(0.9, 0.999))
        gn = torch.nn.utils.clip_grad_norm_(model.adapter.parameters(), max_norm=0.05)
        opt.step()
        out["opt_gradnorm"] = gn.numpy()
        for n in ("fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias"):
            out[f"opt_after_{n}"] = dict(model.adapter.named_parameters())[n].detach().numpy().copy()
    meta = dict(name=name, B=B, T_p=T_p, T_t=T_t, p_lens=list(p_lens), t_lens=list(t_lens), layers=list(layers),
                id_high=id_high, pad_id=pad_id, eos_id=eos_id, seed_w=seed_w, seed_in=seed_in,
                esm=specs.spec_dict(esm), llama=specs.spec_dict(llama), adapter=specs.spec_dict(ad))
    import json
    out["meta_json"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    path = os.path.join(HERE, f"{name}.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {os.path.getsize(path) / 1e6:.2f} MB, {len(out)} arrays")


def run_ops(ref):
    """Function-level known answers from the reference's own readout / loss code on random input."""
    tc = ref.tc
    out = {}
    B, T, D = 5, 11, 24
    emb = synth.uniform_f32(7, "ops.emb", (B, T, D), 2.0, 0.3)
    lens = [11, 8, 5, 2, 1]
    mask = np.zeros((B, T), dtype=np.int64)
    for b, n in enumerate(lens):
        mask[b, :n] = 1
    out["emb"], out["mask"] = emb, mask
    for ro in ("last", "mean", "std", "mix"):
        out[f"readout_{ro}"] = tc.readout_embeddings(torch.from_numpy(emb), torch.from_numpy(mask), ro).numpy()
    p = torch.nn.functional.normalize(torch.from_numpy(synth.uniform_f32(7, "ops.p", (6, 32), 1.0)), dim=-1)
    t = torch.nn.functional.normalize(torch.from_numpy(synth.uniform_f32(7, "ops.t", (6, 32), 1.0)), dim=-1)
    out["p"], out["t"] = p.numpy(), t.numpy()
    out["loss_batch"] = tc.BatchInfoNCELoss()(p, t).numpy()
    out["loss_batch_t01"] = tc.BatchInfoNCELoss(temperature=0.1)(p, t).numpy()
    labels = torch.tensor([3, 4, 5])
    out["loss_seg"] = tc.SegmentedBatchInfoNCELoss()(p[3:6], t, labels).numpy()
    # column (text -> protein) term: the reference's own loss module with its arguments swapped; gradients to the protein
    # side by torch autograd through the reference's forward (the text side is frozen on this path)
    out["loss_batch_swapped"] = tc.BatchInfoNCELoss()(t, p).numpy()
    p_req = p.clone().requires_grad_(True)
    sym = 0.5 * (tc.BatchInfoNCELoss()(p_req, t) + tc.BatchInfoNCELoss()(t, p_req))
    sym.backward()
    out["loss_symmetric"], out["grad_symmetric_p"] = sym.detach().numpy(), p_req.grad.numpy().copy()
    p_req.grad = None
    tc.BatchInfoNCELoss()(t, p_req).backward()
    out["grad_swapped_p"] = p_req.grad.numpy().copy()
    path = os.path.join(HERE, "ops.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}")


def tiny_text_tokenizer(words):
    """A local word-level HF tokenizer (no files, no network) so that the reference collater can run here."""
    from tokenizers import Tokenizer, models, pre_tokenizers
    from transformers import PreTrainedTokenizerFast
    vocab = {w: i for i, w in enumerate(["<pad>", "<unk>", "<eos>"] + list(words))}
    tk = Tokenizer(models.WordLevel(vocab=vocab, unk_token="<unk>"))
    tk.pre_tokenizer = pre_tokenizers.WhitespaceSplit()
    tok = PreTrainedTokenizerFast(tokenizer_object=tk, pad_token="<pad>", unk_token="<unk>", eos_token=" <eos>")
    tok.chat_template = "{% for m in messages %}{{ m['content'] }} {% endfor %}"
    return tok


def run_collate(ref):
    """The reference's own Prot2TextLightCollater (dataset/dataloader_light.py:95-290) on a few rows: the cropped
    sequences and the description ids / masks are what p2t_hip.data.ContrastiveCollater must reproduce.  The collater's
    `_calculate_sequence_length` needs the absent `esm` package; it only sizes the chat prompt (not compared here), so
    that one method is replaced on the instance by len + 2."""
    import json
    import random
    from dataset import Prot2TextLightCollater
    words = "binds atp and catalyzes the transfer of phosphate to serine residues in membrane proteins kinase".split()
    nan = float("nan")
    rows = [
        {"AlphaFoldDB": "P1", "Full Name": "Kinase A", "taxon": "Homo sapiens", "sequence": "MKTAYIAKQRQISFVKSHFSRQLEERLGLIEV",
         "function": "binds atp and catalyzes the transfer of phosphate"},
        {"AlphaFoldDB": "P2", "Full Name": nan, "taxon": "Mus musculus", "sequence": "MSTNPKPQRKTK", "function": "kinase"},
        {"AlphaFoldDB": "P3", "Full Name": "Membrane protein", "taxon": nan,
         "sequence": "MALWMRLLPLLALLALWGPDPAAAFVNQHLCGSHLVEALYLVCGERGFFYTPKT",
         "function": "membrane proteins binds atp in serine residues and the kinase catalyzes transfer of phosphate to proteins"},
        {"AlphaFoldDB": "P4", "Full Name": "X", "taxon": "E coli", "sequence": "MKVLAAG", "function": "unknownword binds atp"},
        {"AlphaFoldDB": "P5", "Full Name": nan, "taxon": nan, "sequence": "MGSSHHHHHHSSGLVPRGSHMASMTGG", "function": "transfer of phosphate to serine"},
    ]
    params = dict(max_sequence_length=20, max_description_length=8, name_dropout=0.8, taxonomy_dropout=0.8)
    tok = tiny_text_tokenizer(words)
    col = Prot2TextLightCollater(description_tokenizer=tok, esm_tokenizer=None, mode="train", **params)
    col._calculate_sequence_length = lambda seq: len(seq) + 2
    expected = []
    for seed in (0, 1, 2):
        random.seed(seed)
        b = col(rows)
        expected.append({"seed": seed, "protein_sequences": b["protein_sequences"],
                         "description_input_ids": b["description_input_ids"].tolist(),
                         "description_attention_mask": b["description_attention_mask"].tolist()})
    path = os.path.join(HERE, "collate.json")
    rows_json = [{k: (None if isinstance(v, float) else v) for k, v in r.items()} for r in rows]
    with open(path, "w") as f:
        json.dump({"words": words, "rows": rows_json, "params": params, "expected": expected}, f, indent=1)
    print(f"wrote {path}")


def run_train_state(ref):
    """Three epochs of two optimizer steps of the reference's training recipe on the tiny model, IN THE REFERENCE'S CALL
    ORDER (train_contrast.py:417-465 inside an epoch; `scheduler.step()` once per epoch after it, :654-662):
    AdamW over model.parameters() + HF cosine schedule with warm-up, loss through the reference's readout / InfoNCE,
    dropout 0 for determinism.  With a warm-up of 2 the learning rate is 0 throughout epoch 1 (LambdaLR factor(0)), half
    in epoch 2, full in epoch 3 -- the reference's behaviour, reproduced as is.  Stores the per-step loss and lr, the
    final adapter tensors, the parameter order and the optimizer / scheduler state dicts in the format
    train_contrast.py:692-698 saves (`last_epoch` = epochs completed)."""
    import json
    from transformers import get_cosine_schedule_with_warmup
    tc = ref.tc
    F = torch.nn.functional
    esm = specs.EsmSpec(num_hidden_layers=2, hidden_size=64, intermediate_size=128, num_attention_heads=4)
    llama = specs.LlamaSpec(num_hidden_layers=3, hidden_size=64, intermediate_size=160, num_attention_heads=4,
                            num_key_value_heads=2, vocab_size=512)
    ad = specs.AdapterSpec(64, 96, 64, 0.0)
    B, T_p, T_t, layer = 4, 24, 12, 2
    model = build_reference_model(ref, esm, llama, ad, 0)
    model.adapter.requires_grad_(True)
    model.adapter.train()
    names = [n for n, _ in model.named_parameters()]
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, eps=1e-6, betas=(0.9, 0.999))
    total_steps, warmup = 10, 2
    sched = get_cosine_schedule_with_warmup(opt, num_warmup_steps=warmup, num_training_steps=total_steps)
    shim = types.SimpleNamespace(llm_decoder=model.llama_decoder)
    out = {"losses": [], "lrs": []}
    steps_per_epoch = 2
    for step in range(6):
        pid, pmask = synth.protein_batch(100 + step, B, T_p, [24, 15, 7, 3])
        tid, tmask = synth.text_batch(100 + step, B, T_t, 500, [12, 9, 5, 2], 510, 509)
        pid_t, pmask_t, tid_t, tmask_t = (torch.from_numpy(a) for a in (pid, pmask, tid, tmask))
        with torch.no_grad():
            t = F.normalize(tc.get_description_embeddings(shim, tid_t, tmask_t, output_llm_layer=layer), p=2, dim=-1)
        ad_out, _ = model(protein_input_ids=pid_t, protein_attention_mask=pmask_t, return_adapter_outputs=True)
        p = F.normalize(tc.readout_embeddings(ad_out, pmask_t, "mix"), p=2, dim=-1)
        loss = tc.SegmentedBatchInfoNCELoss()(p, t, torch.arange(B))
        out["lrs"].append(opt.param_groups[0]["lr"])
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=float("inf"))
        opt.step()
        opt.zero_grad(set_to_none=True)
        out["losses"].append(float(loss))
        if (step + 1) % steps_per_epoch == 0:
            sched.step()                       # once per epoch, after train_epoch (train_contrast.py:662)
    final = {n: prm.detach().clone() for n, prm in model.adapter.named_parameters()}
    torch.save({"optimizer_state_dict": opt.state_dict(), "scheduler_state_dict": sched.state_dict()},
               os.path.join(HERE, "train_state_optimizer_scheduler.pt"))
    torch.save(final, os.path.join(HERE, "train_state_model.pt"))
    meta = dict(param_names=names, losses=out["losses"], lrs=out["lrs"], lr=1e-3, total_steps=total_steps, warmup=warmup,
                steps_per_epoch=steps_per_epoch,
                B=B, T_p=T_p, T_t=T_t, layer=layer, esm=specs.spec_dict(esm), llama=specs.spec_dict(llama),
                adapter=specs.spec_dict(ad))
    with open(os.path.join(HERE, "train_state.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print("wrote train_state.json / train_state_model.pt / train_state_optimizer_scheduler.pt", out)



def sft_batch(B, lens, T_prompt, T_desc, placeholder_id, pad_id, id_high, seed):
    """Deterministic SFT-style batch: left-padded prompts holding len+2 placeholders, right-padded descriptions."""
    rng = np.random.default_rng(seed)
    T_p = max(lens) + 2
    pid, pmask = synth.protein_batch(seed, B, T_p, [n + 2 for n in lens])
    ids = np.full((B, T_prompt + T_desc), pad_id, dtype=np.int64)
    mask = np.zeros((B, T_prompt + T_desc), dtype=np.int64)
    labels = np.full((B, T_prompt + T_desc), -100, dtype=np.int64)
    for b, n in enumerate(lens):
        head = rng.integers(0, id_high, size=3)
        prompt = np.concatenate([head[:2], np.full(n + 2, placeholder_id), head[2:]])
        ids[b, T_prompt - len(prompt):T_prompt] = prompt
        mask[b, T_prompt - len(prompt):T_prompt] = 1
        nd = int(rng.integers(2, T_desc + 1))
        desc = rng.integers(0, id_high, size=nd)
        ids[b, T_prompt:T_prompt + nd] = desc
        mask[b, T_prompt:T_prompt + nd] = 1
        labels[b, T_prompt:T_prompt + nd] = desc
    return pid, pmask, ids, mask, labels


def run_sft(ref):
    """Full forward of the reference class with labels (models/modeling_esm2llama_instruct.py:141-215): decoder inputs
    after the placeholder scatter, logits and the LM loss of HF LlamaForCausalLM, tiny towers."""
    esm = specs.EsmSpec(num_hidden_layers=2, hidden_size=64, intermediate_size=128, num_attention_heads=4)
    llama = specs.LlamaSpec(num_hidden_layers=3, hidden_size=64, intermediate_size=160, num_attention_heads=4,
                            num_key_value_heads=2, vocab_size=512)
    ad = specs.AdapterSpec(64, 96, 64, 0.3)
    model = build_reference_model(ref, esm, llama, ad, 0)
    placeholder_id = 511
    model.config.placeholder_id = placeholder_id
    lens = [10, 6, 3]
    pid, pmask, ids, mask, labels = sft_batch(3, lens, 18, 9, placeholder_id, 510, 500, 7)
    t = lambda a: torch.from_numpy(a)
    with torch.no_grad():
        emb, m2 = model(input_ids=t(ids), attention_mask=t(mask), protein_input_ids=t(pid), protein_attention_mask=t(pmask),
                        return_decoder_inputs=True)
        out = model(input_ids=t(ids), attention_mask=t(mask), labels=t(labels), protein_input_ids=t(pid),
                    protein_attention_mask=t(pmask))
        out_nolabel = model(input_ids=t(ids), attention_mask=t(mask), protein_input_ids=t(pid), protein_attention_mask=t(pmask))
    assert torch.equal(m2, t(mask)) and out_nolabel.loss is None
    meta = dict(esm=specs.spec_dict(esm), llama=specs.spec_dict(llama), adapter=specs.spec_dict(ad), placeholder_id=placeholder_id,
                lens=lens)
    import json
    path = os.path.join(HERE, "sft_tiny.npz")
    np.savez_compressed(path, protein_input_ids=pid, protein_attention_mask=pmask, input_ids=ids, attention_mask=mask, labels=labels,
                        inputs_embeds=emb.numpy(), logits=out.logits.numpy(), loss=np.float32(out.loss.item()),
                        meta_json=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8))
    print(f"wrote {path}: loss {float(out.loss):.6f}, logits {tuple(out.logits.shape)}")


def run_sft_backward(ref):
    """Stage-2 step of the reference (scripts/train_instruct.py:192-213: `loss = model(**batch).loss; loss.backward()`) with the
    decoder frozen and the adapter trainable, through torch autograd on the reference class (eval mode: no dropout): the LM loss,
    its gradient with respect to the decoder inputs (after the placeholder scatter) and the adapter's four gradients.  Three
    decoder shapes: head_dim 16 (the generic path), 64 with GQA and 128 (the fused QKV + rotary epilogues)."""
    import json
    cases = {
        "d16": (specs.LlamaSpec(num_hidden_layers=3, hidden_size=64, intermediate_size=160, num_attention_heads=4, num_key_value_heads=2, vocab_size=512), 64),
        "d64": (specs.LlamaSpec(num_hidden_layers=2, hidden_size=256, intermediate_size=320, num_attention_heads=4, num_key_value_heads=2, vocab_size=512), 256),
        "d128": (specs.LlamaSpec(num_hidden_layers=2, hidden_size=256, intermediate_size=288, num_attention_heads=2, num_key_value_heads=1, vocab_size=512,
                                 rope_type="default", rope_theta=10000.0), 256),
    }
    out, metas = {}, {}
    placeholder_id = 511
    lens = [10, 6, 3]
    pid, pmask, ids, mask, labels = sft_batch(3, lens, 18, 9, placeholder_id, 510, 500, 7)
    t = lambda a: torch.from_numpy(a)
    for name, (llama, H) in cases.items():
        esm = specs.EsmSpec(num_hidden_layers=2, hidden_size=64, intermediate_size=128, num_attention_heads=4)
        ad = specs.AdapterSpec(64, 96, H, 0.3)
        model = build_reference_model(ref, esm, llama, ad, 0)
        model.config.placeholder_id = placeholder_id
        model.adapter.requires_grad_(True)
        emb, _ = model(input_ids=t(ids), attention_mask=t(mask), protein_input_ids=t(pid), protein_attention_mask=t(pmask),
                       return_decoder_inputs=True)
        emb.retain_grad()
        res = model.llama_decoder(inputs_embeds=emb, attention_mask=t(mask), labels=t(labels))
        res.loss.backward()
        with torch.no_grad():
            full = model(input_ids=t(ids), attention_mask=t(mask), labels=t(labels), protein_input_ids=t(pid), protein_attention_mask=t(pmask))
        assert abs(float(full.loss) - float(res.loss)) < 1e-6
        out[f"{name}.loss"] = np.float32(res.loss.item())
        out[f"{name}.d_inputs_embeds"] = emb.grad.numpy().copy()
        for n in ("fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias"):
            g = dict(model.adapter.named_parameters())[n].grad
            out[f"{name}.grad.{n}"] = g.numpy().copy()
        assert model.adapter.ln1.weight.grad is None                       # the unused LayerNorms receive nothing, as in stage 1
        metas[name] = dict(esm=specs.spec_dict(esm), llama=specs.spec_dict(llama), adapter=specs.spec_dict(ad))
        print(f"sft_grad {name}: loss {float(res.loss):.6f} |d emb| {float(emb.grad.norm()):.4e} |g fc2.w| {float(model.adapter.fc2.weight.grad.norm()):.4e}")
    meta = dict(cases=metas, placeholder_id=placeholder_id, lens=lens)
    path = os.path.join(HERE, "sft_grad_tiny.npz")
    np.savez_compressed(path, protein_input_ids=pid, protein_attention_mask=pmask, input_ids=ids, attention_mask=mask, labels=labels,
                        meta_json=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8), **out)
    print(f"wrote {path}")

LORA_TARGETS = ("self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj", "self_attn.o_proj", "mlp.gate_proj", "mlp.up_proj", "mlp.down_proj")


class _LoraLinear(torch.nn.Module):
    """What `peft.LoraConfig(r, lora_alpha=2r, lora_dropout=p, bias="none")` (scripts/train_instruct.py:155-176) makes of a target
    nn.Linear: y = W x + (alpha / r) B (A drop(x)), W frozen, A [r, in] and B [out, r] trained.  `peft` is not importable here: this
    is the published LoRA arithmetic on the reference class's own decoder, not peft's code -- parity against peft itself is unpinned."""

    def __init__(self, base: torch.nn.Linear, r: int, alpha: float, a: np.ndarray, b: np.ndarray):
        super().__init__()
        self.base, self.scale = base, alpha / r
        self.lora_A = torch.nn.Parameter(torch.from_numpy(a.copy()))
        self.lora_B = torch.nn.Parameter(torch.from_numpy(b.copy()))

    def forward(self, x):
        return self.base(x) + self.scale * ((x @ self.lora_A.t()) @ self.lora_B.t())


def lora_init(seed: int, layer: int, target: str, r: int, n_out: int, n_in: int):
    """A and B of one target from the counter-hash generator (both non-zero: peft's B = 0 start would make every dA vanish)."""
    from p2t_hip import synth
    a = synth.uniform_f32(seed, f"lora.{layer}.{target}.A", (r, n_in), 0.25)
    b = synth.uniform_f32(seed, f"lora.{layer}.{target}.B", (n_out, r), 0.25)
    return a, b


def run_sft_lora(ref):
    """Stage-2 step WITH LoRA adapters on the seven decoder projections (scripts/train_instruct.py:146-183; r = 4, alpha = 2r,
    dropout 0 for a deterministic target) and the modality adapter trainable (`modules_to_save`): torch autograd through the
    reference class with every target wrapped by _LoraLinear.  Loss, gradient at the decoder inputs, the adapter's four gradients
    and dA / dB of every (layer, target).  Also the Qwen3 decoder (per-head q / k RMSNorm) WITHOUT LoRA: its adapter gradients pin
    the q / k-norm backward.  Decoder shapes as sft_grad_tiny."""
    import json
    r = 4
    cases = {
        "d16": (specs.LlamaSpec(num_hidden_layers=3, hidden_size=64, intermediate_size=160, num_attention_heads=4, num_key_value_heads=2, vocab_size=512), 64, True),
        "d64": (specs.LlamaSpec(num_hidden_layers=2, hidden_size=256, intermediate_size=320, num_attention_heads=4, num_key_value_heads=2, vocab_size=512), 256, True),
        "d128": (specs.LlamaSpec(num_hidden_layers=2, hidden_size=256, intermediate_size=288, num_attention_heads=2, num_key_value_heads=1, vocab_size=512,
                                 rope_type="default", rope_theta=10000.0), 256, True),
        "qwen3": (specs.LlamaSpec(num_hidden_layers=2, hidden_size=128, intermediate_size=192, num_attention_heads=4, num_key_value_heads=2, vocab_size=512,
                                  head_dim=48, qk_norm=True, rope_type="default", rope_theta=10000.0, tie_word_embeddings=False), 128, False),
        "qwen3_lora": (specs.LlamaSpec(num_hidden_layers=2, hidden_size=128, intermediate_size=192, num_attention_heads=4, num_key_value_heads=2, vocab_size=512,
                                       head_dim=48, qk_norm=True, rope_type="default", rope_theta=10000.0, tie_word_embeddings=False), 128, True),
    }
    out, metas = {}, {}
    placeholder_id = 511
    lens = [10, 6, 3]
    pid, pmask, ids, mask, labels = sft_batch(3, lens, 18, 9, placeholder_id, 510, 500, 7)
    t = lambda a: torch.from_numpy(a)
    for name, (llama, H, lora) in cases.items():
        esm = specs.EsmSpec(num_hidden_layers=2, hidden_size=64, intermediate_size=128, num_attention_heads=4)
        ad = specs.AdapterSpec(64, 96, H, 0.3)
        model = build_reference_model(ref, esm, llama, ad, 0)
        model.config.placeholder_id = placeholder_id
        model.adapter.requires_grad_(True)
        wrapped = {}
        if lora:
            for i, layer in enumerate(model.llama_decoder.model.layers):
                for tgt in LORA_TARGETS:
                    parent, leaf = tgt.split(".")
                    base = getattr(getattr(layer, parent), leaf)
                    a, b = lora_init(5, i, tgt, r, base.out_features, base.in_features)
                    w = _LoraLinear(base, r, 2.0 * r, a, b)
                    setattr(getattr(layer, parent), leaf, w)
                    wrapped[(i, tgt)] = w
        emb, _ = model(input_ids=t(ids), attention_mask=t(mask), protein_input_ids=t(pid), protein_attention_mask=t(pmask),
                       return_decoder_inputs=True)
        emb.retain_grad()
        res = model.llama_decoder(inputs_embeds=emb, attention_mask=t(mask), labels=t(labels))
        res.loss.backward()
        out[f"{name}.loss"] = np.float32(res.loss.item())
        out[f"{name}.d_inputs_embeds"] = emb.grad.numpy().copy()
        for n in ("fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias"):
            out[f"{name}.grad.{n}"] = dict(model.adapter.named_parameters())[n].grad.numpy().copy()
        for (i, tgt), w in wrapped.items():
            out[f"{name}.lora.{i}.{tgt}.dA"] = w.lora_A.grad.numpy().copy()
            out[f"{name}.lora.{i}.{tgt}.dB"] = w.lora_B.grad.numpy().copy()
        metas[name] = dict(esm=specs.spec_dict(esm), llama=specs.spec_dict(llama), adapter=specs.spec_dict(ad), lora=bool(lora))
        gl = max((float(w.lora_A.grad.norm()) for w in wrapped.values()), default=0.0)
        print(f"sft_lora {name}: loss {float(res.loss):.6f} |d emb| {float(emb.grad.norm()):.4e} |g fc2.w| {float(model.adapter.fc2.weight.grad.norm()):.4e} max |dA| {gl:.3e}")
    meta = dict(cases=metas, placeholder_id=placeholder_id, lens=lens, r=r, alpha=2.0 * r, lora_seed=5, targets=list(LORA_TARGETS))
    path = os.path.join(HERE, "sft_lora_tiny.npz")
    np.savez_compressed(path, protein_input_ids=pid, protein_attention_mask=pmask, input_ids=ids, attention_mask=mask, labels=labels,
                        meta_json=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8), **out)
    print(f"wrote {path}")


def run_generate(ref):
    """`Esm2LlamaInstructForCausalLM.generate` of the reference (models/modeling_esm2llama_instruct.py:217-251) on tiny random-init
    towers, driven as scripts/generate_instruct.py:72-87 drives it: left-padded prompts holding protein placeholders, greedy decoding
    (with and without an eos id that some rows hit early), and beam search with two length penalties.  Per-step logits of the greedy
    run are kept so that a port can be checked step by step; the smallest top-1 / top-2 logit gap of every greedy run is asserted to
    be far above fp32 noise, so the token ids are a stable target."""
    import json
    cases = {
        "d16": (specs.LlamaSpec(num_hidden_layers=3, hidden_size=64, intermediate_size=160, num_attention_heads=4, num_key_value_heads=2, vocab_size=512), 64),
        "d64": (specs.LlamaSpec(num_hidden_layers=2, hidden_size=256, intermediate_size=320, num_attention_heads=4, num_key_value_heads=2, vocab_size=512), 256),
        "d128": (specs.LlamaSpec(num_hidden_layers=2, hidden_size=256, intermediate_size=288, num_attention_heads=2, num_key_value_heads=1, vocab_size=512,
                                 rope_type="default", rope_theta=10000.0), 256),
        "qwen3": (specs.LlamaSpec(num_hidden_layers=2, hidden_size=128, intermediate_size=192, num_attention_heads=4, num_key_value_heads=2, vocab_size=512,
                                  head_dim=48, qk_norm=True, rope_type="default", rope_theta=10000.0, tie_word_embeddings=False), 128),
    }
    out, metas = {}, {}
    placeholder_id, pad_id, n_new = 511, 510, 14
    lens = [10, 6, 3]
    T_prompt = 18
    pid, pmask, ids, mask, _ = sft_batch(3, lens, T_prompt, 9, placeholder_id, pad_id, 500, 11)
    ids, mask = ids[:, :T_prompt].copy(), mask[:, :T_prompt].copy()          # the [prompt] only, left padded
    t = lambda a: torch.from_numpy(a)
    for name, (llama, H) in cases.items():
        esm = specs.EsmSpec(num_hidden_layers=2, hidden_size=64, intermediate_size=128, num_attention_heads=4)
        ad = specs.AdapterSpec(64, 96, H, 0.3)
        kw = dict(inputs=t(ids), attention_mask=t(mask), protein_input_ids=t(pid), protein_attention_mask=t(pmask), pad_token_id=pad_id,
                  return_dict_in_generate=True)
        for wseed in range(8):                                               # first weight seed whose greedy choices are all clear-cut
            model = build_reference_model(ref, esm, llama, ad, wseed)
            model.config.placeholder_id = placeholder_id
            with torch.no_grad():
                g0 = model.generate(**kw, max_new_tokens=n_new, eos_token_id=None, do_sample=False, num_beams=1, output_logits=True)
            seq0 = g0.sequences.numpy()
            lg = torch.stack(g0.logits, 0).numpy()                           # [n_new, B, V]
            top2 = np.sort(lg, axis=-1)[..., -2:]
            gap = float((top2[..., 1] - top2[..., 0]).min())
            if gap > 5e-3:
                break
        assert seq0.shape == (3, n_new) and gap > 5e-3, (seq0.shape, gap)
        eos = int(seq0[1, 4])                                                # row 1 stops after five tokens (others whenever they emit it)
        with torch.no_grad():
            g1 = model.generate(**kw, max_new_tokens=n_new, eos_token_id=eos, do_sample=False, num_beams=1)
            # the argument set of scripts/generate_instruct.py:72-87 (sampling switches passed but off)
            g1b = model.generate(**kw, max_new_tokens=n_new, eos_token_id=eos, num_beams=1, length_penalty=1.0, temperature=1.0, do_sample=False,
                                 top_p=1.0, top_k=50)
            assert torch.equal(g1.sequences, g1b.sequences)
            beams = {}
            for lp in (1.0, 0.5):
                gb = model.generate(**kw, max_new_tokens=n_new, eos_token_id=eos, do_sample=False, num_beams=3, length_penalty=lp, output_scores=True)
                beams[lp] = (gb.sequences.numpy(), gb.sequences_scores.numpy())
        seq1 = g1.sequences.numpy()
        assert (seq1[1, :5] == seq0[1, :5]).all() and (seq1[1, 5:] == pad_id).all()
        # rarer switches of the same call: two eos ids + `max_length` instead of `max_new_tokens` (greedy); beam search with
        # early_stopping=True / "never" and two returned hypotheses per prompt
        eos2 = [eos, int(seq0[2, 7])]
        with torch.no_grad():
            g2 = model.generate(**kw, max_length=T_prompt + n_new - 3, eos_token_id=eos2, do_sample=False, num_beams=1)   # counts the (padded) prompt
            gx = {}
            for es in (True, "never"):
                gb = model.generate(**kw, max_new_tokens=n_new, eos_token_id=eos2, do_sample=False, num_beams=3, length_penalty=0.8, early_stopping=es,
                                    num_return_sequences=2, output_scores=True)
                gx[es] = (gb.sequences.numpy(), gb.sequences_scores.numpy())
        out[f"{name}.greedy_eos2_maxlen"] = g2.sequences.numpy()
        for es, (sq, sc) in gx.items():
            out[f"{name}.beam3_es{es}"] = sq
            out[f"{name}.beam3_es{es}_scores"] = sc.astype(np.float32)
        out[f"{name}.greedy"] = seq0
        out[f"{name}.greedy_logits"] = lg.astype(np.float32)
        out[f"{name}.greedy_eos"] = seq1
        for lp, (sq, sc) in beams.items():
            out[f"{name}.beam3_lp{lp}"] = sq
            out[f"{name}.beam3_lp{lp}_scores"] = sc.astype(np.float32)
        metas[name] = dict(esm=specs.spec_dict(esm), llama=specs.spec_dict(llama), adapter=specs.spec_dict(ad), eos=eos, eos2=eos2, min_top2_gap=gap, weight_seed=wseed)
        print(f"generate {name}: greedy {seq0[0, :6]}..., min top-2 gap {gap:.3e}, eos {eos} -> widths {seq1.shape[1]}, beams "
              f"{[(k, v[0].shape, np.round(v[1], 4).tolist()) for k, v in beams.items()]}")
    meta = dict(cases=metas, placeholder_id=placeholder_id, pad_id=pad_id, lens=lens, max_new_tokens=n_new)
    path = os.path.join(HERE, "generate_tiny.npz")
    np.savez_compressed(path, protein_input_ids=pid, protein_attention_mask=pmask, input_ids=ids, attention_mask=mask,
                        meta_json=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8), **out)
    print(f"wrote {path}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="ops,tiny,tiny_d24,tiny_d128,tiny_d64,tiny_qwen3,cfg1,collate,train_state,sft,sft_grad,sft_lora,generate")
    args = ap.parse_args()
    only = set(args.only.split(","))
    torch.manual_seed(0)
    torch.set_num_threads(8)
    ref = load_reference()
    if "ops" in only:
        run_ops(ref)
    if "collate" in only:
        run_collate(ref)
    if "train_state" in only:
        run_train_state(ref)
    if "sft" in only:
        run_sft(ref)
    if "sft_grad" in only:
        run_sft_backward(ref)
    if "sft_lora" in only:
        run_sft_lora(ref)
    if "generate" in only:
        run_generate(ref)
    if "tiny" in only:
        esm = specs.EsmSpec(num_hidden_layers=2, hidden_size=64, intermediate_size=128, num_attention_heads=4)
        llama = specs.LlamaSpec(num_hidden_layers=3, hidden_size=64, intermediate_size=160, num_attention_heads=4,
                                num_key_value_heads=2, vocab_size=512)
        ad = specs.AdapterSpec(64, 96, 64, 0.3)
        run_case(ref, "tiny", esm, llama, ad, B=4, T_p=24, T_t=12, p_lens=[24, 15, 7, 3], t_lens=[12, 9, 5, 2],
                 layers=[2, 3], id_high=500, pad_id=510, eos_id=509)
    if "tiny_d24" in only:
        esm = specs.EsmSpec(num_hidden_layers=3, hidden_size=96, intermediate_size=192, num_attention_heads=4)
        llama = specs.LlamaSpec(num_hidden_layers=2, hidden_size=128, intermediate_size=256, num_attention_heads=4,
                                num_key_value_heads=1, vocab_size=300, rope_type="default", rope_theta=10000.0)
        ad = specs.AdapterSpec(96, 80, 128, 0.3)
        run_case(ref, "tiny_d24", esm, llama, ad, B=6, T_p=40, T_t=20, p_lens=[40, 33, 21, 10, 4, 2],
                 t_lens=[20, 20, 13, 7, 3, 1], layers=[1, 2], id_high=290, pad_id=299, eos_id=298)
    if "tiny_d128" in only:
        # text tower with head_dim 128 (Llama-3.1-8B's): fused QKV + rotary epilogue with the packed row order, GQA 2:1
        esm = specs.EsmSpec(num_hidden_layers=2, hidden_size=64, intermediate_size=128, num_attention_heads=4)
        llama = specs.LlamaSpec(num_hidden_layers=2, hidden_size=256, intermediate_size=192, num_attention_heads=2,
                                num_key_value_heads=1, vocab_size=300)
        ad = specs.AdapterSpec(64, 80, 256, 0.3)
        run_case(ref, "tiny_d128", esm, llama, ad, B=4, T_p=24, T_t=40, p_lens=[24, 15, 7, 3], t_lens=[40, 33, 9, 2],
                 layers=[1, 2], id_high=290, pad_id=299, eos_id=298)
    if "tiny_d64" in only:
        # ENCODER with head_dim 64 (esm2_t33_650M / esm2_t36_3B's): the towers take the fused QKV epilogue with bias +
        # q * d^-1/2 BEFORE the rotation + rotate-half + head split (HF modeling_esm.py:345,362-378) -- the path bench.py
        # times.  Text tower head_dim 64 with GQA 2:1 (Llama-3.2-1B's), llama3 rope.
        esm = specs.EsmSpec(num_hidden_layers=3, hidden_size=128, intermediate_size=320, num_attention_heads=2)
        llama = specs.LlamaSpec(num_hidden_layers=2, hidden_size=128, intermediate_size=192, num_attention_heads=2,
                                num_key_value_heads=1, vocab_size=300)
        ad = specs.AdapterSpec(128, 96, 128, 0.3)
        run_case(ref, "tiny_d64", esm, llama, ad, B=5, T_p=70, T_t=24, p_lens=[70, 64, 33, 9, 3], t_lens=[24, 17, 8, 2, 1],
                 layers=[1, 2], id_high=290, pad_id=299, eos_id=298)
    if "tiny_qwen3" in only:
        # SURVEY.md section 8f row 4, the half that can be pinned offline: a Qwen3 text tower (per-head q / k RMSNorm, head_dim 128
        # that is NOT hidden / heads, GQA 2:1, default rope with theta 1e6) driven through the reference's own
        # get_description_embeddings; the protein side stays the ESM2 encoder (ESM-C needs the absent `esm` package)
        esm = specs.EsmSpec(num_hidden_layers=2, hidden_size=64, intermediate_size=128, num_attention_heads=4)
        llama = specs.LlamaSpec(num_hidden_layers=3, hidden_size=192, intermediate_size=320, num_attention_heads=4,
                                num_key_value_heads=2, vocab_size=300, head_dim=128, rms_norm_eps=1e-6, rope_theta=1e6,
                                rope_type="default", rope_factor=1.0, qk_norm=True)
        ad = specs.AdapterSpec(64, 80, 192, 0.3)
        run_case(ref, "tiny_qwen3", esm, llama, ad, B=4, T_p=24, T_t=40, p_lens=[24, 15, 7, 3], t_lens=[40, 33, 9, 2],
                 layers=[2, 3], id_high=290, pad_id=299, eos_id=298)
    if "cfg1" in only:
        name_e, name_l, _, B, T_p, T_t = specs.CONFIGS["cfg1"]
        esm, llama = specs.esm_spec(name_e), specs.llama_spec(name_l)
        ad = specs.adapter_spec(esm, llama)
        run_case(ref, "cfg1", esm, llama, ad, B=B, T_p=T_p, T_t=T_t, p_lens=[128, 77, 32, 5], t_lens=[64, 40, 17, 3],
                 layers=[16], id_high=128000, pad_id=128002, eos_id=128009, store_hidden=False, grads=True)


if __name__ == "__main__":
    main()
