"""CPU: host-side logic that needs no GPU -- synthetic generator, tower specs / FLOP model,
config classes of the drop-in surface, batch contract helpers."""
import numpy as np
import pytest

from p2t_hip import specs, synth


def test_hash_generator_is_deterministic_and_random_access():
    a = synth.uniform_f32(3, "x.weight", (7, 11), 0.5, 1.0)
    b = synth.uniform_f32(3, "x.weight", (7, 11), 0.5, 1.0)
    assert np.array_equal(a, b) and a.dtype == np.float32
    assert not np.array_equal(a, synth.uniform_f32(4, "x.weight", (7, 11), 0.5, 1.0))
    assert not np.array_equal(a, synth.uniform_f32(3, "y.weight", (7, 11), 0.5, 1.0))
    assert (a >= 0.5).all() and (a < 1.5).all()
    rows = synth.uniform_rows_f32(3, "x.weight", [5, 0, 5], 11, 0.5, 1.0)
    assert np.array_equal(rows, a[[5, 0, 5]])
    big = synth.uniform_f32(0, "stats", (200000,), 1.0)
    assert abs(big.mean()) < 0.01 and abs(big.std() - 1 / np.sqrt(3)) < 0.01


def test_bf16_rounding_matches_torch():
    torch = pytest.importorskip("torch")
    x = synth.uniform_f32(1, "bf", (4096,), 3.0)
    x[:4] = [0.0, -0.0, 1.0000001, 65504.0]
    ref = torch.from_numpy(x).to(torch.bfloat16).float().numpy()
    assert np.array_equal(synth.bf16_round(x), ref)
    assert np.array_equal(synth.bf16_bits(x), torch.from_numpy(x).to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16))


def test_batch_contract():
    ids, mask = synth.protein_batch(1234, 3, 16, [16, 9, 2])
    assert ids.dtype == np.int64 and mask.dtype == np.int64
    assert (ids[:, 0] == synth.ESM_CLS).all()
    for b, n in enumerate([16, 9, 2]):
        assert ids[b, n - 1] == synth.ESM_EOS and (ids[b, n:] == synth.ESM_PAD).all() and mask[b].sum() == n
        assert ((ids[b, 1:n - 1] >= 4) & (ids[b, 1:n - 1] < 24)).all()
    tid, tmask = synth.text_batch(1234, 2, 8, 500, [8, 3], 510, 509)
    assert tid[1, 2] == 509 and (tid[1, 3:] == 510).all() and tmask[1].sum() == 3 and tid[:, :2].max() < 500


def test_flop_model_matches_survey():
    """cfg 3 = 7.153 TF / sample (ESM 6.185, Llama x16 0.896, adapter 0.073)."""
    esm, llama = specs.esm_spec("esm2_t36_3B"), specs.llama_spec("Llama-3.1-8B-Instruct")
    f = specs.flops_per_sample(esm, llama, specs.adapter_spec(esm, llama), 1024, 128)
    assert abs(f["esm"] / 1e12 - 6.185) < 0.005 and abs(f["llama"] / 1e12 - 0.896) < 0.005
    assert abs(f["adapter"] / 1e12 - 0.073) < 0.002 and abs(f["total"] / 1e12 - 7.153) < 0.01
    e1, l1 = specs.esm_spec("esm2_t6_8M"), specs.llama_spec("Llama-3.2-1B")
    assert abs(specs.flops_per_sample(e1, l1, specs.adapter_spec(e1, l1), 128, 64)["total"] / 1e12 - 0.130) < 0.005


def test_spec_tables_cover_hf_state_dict_names():
    esm = specs.EsmSpec(num_hidden_layers=2, hidden_size=64, intermediate_size=128, num_attention_heads=4)
    names = [n for n, *_ in specs.esm_tensors(esm, "esm_encoder.")]
    assert "esm_encoder.encoder.layer.1.attention.self.query.weight" in names
    assert "esm_encoder.encoder.emb_layer_norm_after.bias" in names and len(names) == 1 + 2 * 16 + 2
    llama = specs.LlamaSpec(num_hidden_layers=2, hidden_size=64, intermediate_size=96, num_attention_heads=4,
                            num_key_value_heads=2, vocab_size=100)
    shapes = {n: s for n, s, *_ in specs.llama_tensors(llama, "llama_decoder.")}
    assert shapes["llama_decoder.model.layers.0.self_attn.k_proj.weight"] == (32, 64)
    assert shapes["llama_decoder.lm_head.weight"] == (100, 64)
    ad = [n for n, *_ in specs.adapter_tensors(specs.AdapterSpec(64, 96, 64), "adapter.")]
    assert ad == [f"adapter.{m}.{p}" for m in ("fc1", "fc2", "ln1", "ln2") for p in ("weight", "bias")]


def test_config_surface_matches_reference_fields():
    from p2t_hip.configuration import (Esm2LlamaInstructConfig, ModalityAdapterConfig, esm_config_from_spec,
                                       esm_spec_from_config, llama_config_from_spec, llama_spec_from_config)
    a = ModalityAdapterConfig(input_dim=320, intermediate_dim=2048, output_dim=2048)
    assert a.model_type == "modality_adapter" and a.dropout_rate == 0.3
    esm, llama = specs.esm_spec("esm2_t6_8M"), specs.llama_spec("Llama-3.2-1B")
    c = Esm2LlamaInstructConfig(esm_config_from_spec(esm), a, llama_config_from_spec(llama))
    assert c.model_type == "esm2llama_instruct" and c.placeholder_id == 128003
    assert esm_spec_from_config(c.esm_config) == esm
    assert llama_spec_from_config(c.llama_config) == llama
    bad = esm_config_from_spec(esm)
    bad.position_embedding_type = "absolute"
    with pytest.raises(ValueError, match="rotary"):
        esm_spec_from_config(bad)


def test_assembled_config_save_pretrained_round_trip(tmp_path):
    """HF save_pretrained / from_pretrained of Esm2LlamaInstructConfig.  PretrainedConfig writes the
    sub-configs as nested dicts; they come back as config objects with the same tower shapes."""
    from p2t_hip.configuration import (Esm2LlamaInstructConfig, ModalityAdapterConfig, esm_config_from_spec, esm_spec_from_config,
                                       llama_config_from_spec, llama_spec_from_config)
    from transformers import EsmConfig, LlamaConfig
    esm = esm_config_from_spec(specs.esm_spec("esm2_t6_8M"))
    llama = llama_config_from_spec(specs.llama_spec("Llama-3.2-1B"))
    cfg = Esm2LlamaInstructConfig(esm, ModalityAdapterConfig(esm.hidden_size, 2048, llama.hidden_size, dropout_rate=0.25), llama,
                                  placeholder_id=128003)
    cfg.save_pretrained(tmp_path)
    back = Esm2LlamaInstructConfig.from_pretrained(tmp_path)
    assert isinstance(back.esm_config, EsmConfig) and isinstance(back.llama_config, LlamaConfig)
    assert isinstance(back.adapter_config, ModalityAdapterConfig) and back.placeholder_id == 128003
    assert esm_spec_from_config(back.esm_config) == esm_spec_from_config(esm)
    assert llama_spec_from_config(back.llama_config) == llama_spec_from_config(llama)
    assert back.adapter_config.to_spec() == cfg.adapter_config.to_spec()
    assert back.model_type == "esm2llama_instruct" and back.adapter_config.model_type == "modality_adapter"
