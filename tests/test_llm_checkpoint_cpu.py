"""Checkpoint key mapping of the language-model side (msr3d_amd/llm/checkpoint.py): a transformers LlamaForCausalLM
state dict loads into LoRALlamaStack by its own keys, peft's wrapped / adapter-file spellings are accepted, and the
exports round-trip.  Host logic only -- no kernel runs."""
import pytest
import torch


def _tiny():
    from msr3d_amd.llm import LoRALlamaStack
    return LoRALlamaStack(2, hidden_size=128, num_heads=2, intermediate_size=192, vocab_size=256)


def test_transformers_state_dict_loads_by_its_own_keys():
    from transformers.models.llama.modeling_llama import LlamaConfig, LlamaForCausalLM
    from msr3d_amd.llm import hf_state_dict, load_hf_state_dict
    cfg = LlamaConfig(hidden_size=128, num_attention_heads=2, num_key_value_heads=2, intermediate_size=192,
                      num_hidden_layers=2, vocab_size=256, attention_bias=False, mlp_bias=False, tie_word_embeddings=False)
    torch.manual_seed(0)
    hf = LlamaForCausalLM(cfg)
    with torch.no_grad():
        hf.model.layers[1].input_layernorm.weight.normal_()
        hf.model.norm.weight.normal_()
    sd = hf.state_dict()
    net = _tiny()
    embed = torch.empty(256, 128, dtype=torch.bfloat16)
    unused = load_hf_state_dict(net, sd, embed_out=embed)
    assert unused == [], unused
    bf = lambda t: t.to(torch.bfloat16)          # noqa: E731
    assert torch.equal(net.layers[1].mlp["down_proj"].weight, bf(sd["model.layers.1.mlp.down_proj.weight"]))
    assert torch.equal(net.layers[1].mlp["down_proj"].weight_t, bf(sd["model.layers.1.mlp.down_proj.weight"]).t())
    assert torch.equal(net.layers[1].input_layernorm_weight, bf(sd["model.layers.1.input_layernorm.weight"]))
    assert torch.equal(net.norm_weight, bf(sd["model.norm.weight"])) and torch.equal(embed, bf(sd["model.embed_tokens.weight"]))
    assert torch.equal(net.lm_head.weight, bf(sd["lm_head.weight"]))
    # export under the same keys: identical key set (the rotary buffers are not persistent in current transformers)
    out = hf_state_dict(net, embed)
    assert set(out) == set(sd)
    assert all(torch.equal(out[k], bf(sd[k])) for k in sd)
    # a checkpoint that lacks a frozen weight is refused
    sd2 = {k: v for k, v in sd.items() if "layers.0.self_attn.k_proj" not in k}
    with pytest.raises(KeyError):
        load_hf_state_dict(_tiny(), sd2)


def test_peft_spellings_round_trip():
    from msr3d_amd.llm import load_hf_state_dict, peft_adapter_state_dict
    torch.manual_seed(1)
    a, b = _tiny(), _tiny()
    with torch.no_grad():
        for p in a.lora_parameters():
            p.normal_()
    ad = peft_adapter_state_dict(a)
    assert len(ad) == 2 * 7 * 2
    assert "base_model.model.model.layers.1.mlp.up_proj.lora_B.weight" in ad
    assert load_hf_state_dict(b, ad) == []                       # adapter file: only LoRA tensors, no frozen weights needed
    for pa, pb in zip(a.lora_parameters(), b.lora_parameters()):
        assert torch.equal(pa, pb)
    # the in-memory spelling of a wrapped model: `.lora_A.default.weight`, `.base_layer.weight`
    wrapped = {k.replace(".lora_A.weight", ".lora_A.default.weight").replace(".lora_B.weight", ".lora_B.default.weight"): v
               for k, v in ad.items()}
    wrapped["base_model.model.model.layers.0.self_attn.q_proj.base_layer.weight"] = torch.randn(128, 128)
    c = _tiny()
    load_hf_state_dict(c, wrapped, strict=False)
    assert torch.equal(c.layers[0].self_attn["q_proj"].weight,
                       wrapped["base_model.model.model.layers.0.self_attn.q_proj.base_layer.weight"].to(torch.bfloat16))
    assert torch.equal(c.layers[1].mlp["gate_proj"].lora_A.weight, a.layers[1].mlp["gate_proj"].lora_A.weight)


def test_reference_trainer_checkpoint_spelling_round_trips_and_nothing_loads_silently():
    """leo_trainer.py:445-454 saves the trainable tensors of the whole MSR3D module: `llm_model.` + peft's in-memory
    spelling.  Such a file loads; a state dict of which NO key names a tensor of the stack is refused (it would leave the
    LoRA matrices at their initial values without a word)."""
    from msr3d_amd.llm import load_hf_state_dict, reference_trainer_state_dict
    torch.manual_seed(2)
    a, b = _tiny(), _tiny()
    with torch.no_grad():
        for p in a.lora_parameters():
            p.normal_()
    sd = reference_trainer_state_dict(a)
    assert "llm_model.base_model.model.model.layers.0.self_attn.q_proj.lora_A.default.weight" in sd and len(sd) == 2 * 7 * 2
    assert load_hf_state_dict(b, sd) == []
    for pa, pb in zip(a.lora_parameters(), b.lora_parameters()):
        assert torch.equal(pa, pb)
    with pytest.raises(KeyError, match="none of the"):
        load_hf_state_dict(_tiny(), {"visual_prompter.anchor_feat": torch.zeros(3), "some.other.key": torch.zeros(1)})
