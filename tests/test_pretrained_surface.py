"""CPU: the model class is a transformers.PreTrainedModel like the reference's (models/modeling_esm2llama_instruct.py:71-106,
253-268): isinstance, config_class, save_pretrained / from_pretrained with the keywords HF callers use (state_dict=,
safe_serialization=, torch_dtype=), gradient_checkpointing_enable(gradient_checkpointing_kwargs=...), and
transformers.Trainer.save_model -> from_pretrained round trip.  Parameters only: no kernel is launched (no GPU here)."""
import os

import pytest
import torch


def _tiny():
    import p2t_hip as P
    from p2t_hip import specs
    esm = specs.EsmSpec(num_hidden_layers=2, hidden_size=64, intermediate_size=128, num_attention_heads=4)
    llama = specs.LlamaSpec(num_hidden_layers=2, hidden_size=64, intermediate_size=128, num_attention_heads=4, num_key_value_heads=2, vocab_size=256)
    m = P.Esm2LlamaInstructForCausalLM.from_specs(esm, llama, specs.AdapterSpec(64, 32, 64, 0.1), dtype=torch.float32, device="cpu", seed=None)
    g = torch.Generator().manual_seed(0)
    with torch.no_grad():
        for p in m.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * 0.02)
    return P, m


def test_is_a_pretrained_model_with_the_reference_surface():
    import transformers
    P, m = _tiny()
    assert isinstance(m, transformers.PreTrainedModel) and isinstance(m, torch.nn.Module)
    assert type(m).config_class is P.Esm2LlamaInstructConfig and isinstance(m.config, transformers.PretrainedConfig)
    assert m.config.model_type == "esm2llama_instruct" and m.config.placeholder_id == 128003
    assert [n for n, _ in m.named_children()] == ["esm_encoder", "adapter", "llama_decoder"]
    assert not m.is_gradient_checkpointing
    m.gradient_checkpointing_enable(gradient_checkpointing_kwargs={"use_reentrant": False})      # as transformers.Trainer calls it
    assert m.is_gradient_checkpointing
    m.gradient_checkpointing_disable()
    assert not m.is_gradient_checkpointing
    m.gradient_checkpointing_enable()                                                             # as the reference's scripts call it
    import inspect
    sig = inspect.signature(m.generate)                                                           # reference :217-225
    assert list(sig.parameters)[:5] == ["inputs", "attention_mask", "protein_input_ids", "protein_attention_mask", "protein_inputs_embeds"]
    assert callable(m.llama_decoder.generate)


def test_save_and_load_with_hf_keywords(tmp_path):
    P, m = _tiny()
    sd = m.state_dict()
    m.save_pretrained(str(tmp_path / "a"), state_dict=sd, safe_serialization=True, is_main_process=True)
    assert sorted(os.listdir(tmp_path / "a")) == ["config.json", "model.safetensors"]
    m.save_pretrained(str(tmp_path / "skip"), is_main_process=False)
    assert not os.path.exists(tmp_path / "skip")
    same = P.Esm2LlamaInstructForCausalLM.from_pretrained(str(tmp_path / "a"), device="cpu")
    assert not same.training and all(torch.equal(a, b) for a, b in zip(sd.values(), same.state_dict().values()))
    half = P.Esm2LlamaInstructForCausalLM.from_pretrained(str(tmp_path / "a"), torch_dtype=torch.bfloat16, device="cpu",
                                                          low_cpu_mem_usage=True, attn_implementation="eager")
    assert half.esm_encoder.dtype == torch.bfloat16 and half.adapter.fc1.weight.dtype == torch.bfloat16
    assert torch.equal(half.adapter.fc1.weight, sd["adapter.fc1.weight"].to(torch.bfloat16))
    m.save_pretrained(str(tmp_path / "b"), safe_serialization=False)
    assert os.path.exists(tmp_path / "b" / "pytorch_model.bin")
    again = P.Esm2LlamaInstructForCausalLM.from_pretrained(str(tmp_path / "b"), dtype="float32", device="cpu")
    assert all(torch.equal(a, b) for a, b in zip(sd.values(), again.state_dict().values()))
    with pytest.raises(ValueError):
        P.Esm2LlamaInstructForCausalLM.from_pretrained("meta-llama/not-a-directory")
    with pytest.raises(ValueError):
        m.save_pretrained(str(tmp_path / "c"), push_to_hub=True)


def test_transformers_trainer_can_hold_and_save_the_model(tmp_path):
    """`north_star`: "drops into the existing HuggingFace Trainer loop" -- the Trainer accepts the model (a PreTrainedModel on its
    device), and its save path (Trainer.save_model -> model.save_pretrained(output_dir, state_dict=..., safe_serialization=...))
    writes a directory from_pretrained reads back bit for bit."""
    from transformers import Trainer, TrainingArguments
    P, m = _tiny()
    args = TrainingArguments(output_dir=str(tmp_path / "out"), use_cpu=True, report_to=[], save_strategy="no", gradient_checkpointing=True)
    tr = Trainer(model=m, args=args)
    assert tr.model is m
    tr.save_model(str(tmp_path / "saved"))
    back = P.Esm2LlamaInstructForCausalLM.from_pretrained(str(tmp_path / "saved"), device="cpu")
    assert all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), back.state_dict().values()))
    assert back.config.to_dict()["adapter_config"]["intermediate_dim"] == 32
