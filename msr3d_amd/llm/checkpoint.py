"""Key mapping between `LoRALlamaStack` and the checkpoints the reference works with
(/root/reference/model/msr3d/msr3d.py:71 `LlamaForCausalLM.from_pretrained`, :103-112 `get_peft_model`;
trainer/leo_trainer.py:445-462 saves / loads the trainable tensors by name).

`LoRALlamaStack`'s own state-dict keys are flat (`layers.0.self_attn.q_proj.weight`,
`layers.0.self_attn.q_proj.lora_A.weight`, `layers.0.input_layernorm_weight`, `norm_weight`, `lm_head.weight`):
the norm weights are buffers of the layer, not sub-modules.  These helpers translate, in both directions:

  Hugging Face LlamaForCausalLM      model.layers.{i}.self_attn.q_proj.weight, model.layers.{i}.input_layernorm.weight,
                                     model.norm.weight, lm_head.weight, model.embed_tokens.weight
  peft (LoraConfig on q/k/v/o/gate/  base_model.model.model.layers.{i}.self_attn.q_proj.base_layer.weight (wrapped model),
  up/down)                           ...q_proj.lora_A.default.weight (in memory) / ...q_proj.lora_A.weight (adapter file)

Only names are mapped: dtypes follow the destination (frozen weights bf16, LoRA matrices fp32)."""
import re

import torch

PROJ = {"q_proj": "self_attn", "k_proj": "self_attn", "v_proj": "self_attn", "o_proj": "self_attn",
        "gate_proj": "mlp", "up_proj": "mlp", "down_proj": "mlp"}
_PEFT_PREFIX = "base_model.model."


def _strip(key):
    """Any of the accepted spellings -> ('layer', i, rest) | ('norm',) | ('head',) | ('embed',) | None."""
    k = key
    if k.startswith("llm_model."):            # the reference trainer's pytorch_model.bin (leo_trainer.py:445-454)
        k = k[len("llm_model."):]
    if k.startswith(_PEFT_PREFIX):
        k = k[len(_PEFT_PREFIX):]
    if k.startswith("model."):
        k = k[len("model."):]
    if k == "norm.weight":
        return ("norm",)
    if k == "lm_head.weight":
        return ("head",)
    if k == "embed_tokens.weight":
        return ("embed",)
    m = re.match(r"layers\.(\d+)\.(.+)$", k)
    return ("layer", int(m.group(1)), m.group(2)) if m else None


def load_hf_state_dict(stack, state_dict, embed_out=None, strict=True):
    """Copy a Hugging Face `LlamaForCausalLM` / peft-wrapped state dict (or a peft adapter file) into `stack`
    (LoRALlamaStack).  embed_out: optional (V, H) tensor that receives `model.embed_tokens.weight`
    (MSR3DFullStep.embed_tokens).  Returns the list of keys that were not consumed; with strict, raises if a frozen
    weight of the stack was not supplied by a full (non-adapter) checkpoint."""
    unused, seen = [], set()
    with torch.no_grad():
        for key, v in state_dict.items():
            t = _strip(key)
            if t is None:
                unused.append(key)
                continue
            if t[0] == "norm":
                stack.norm_weight.copy_(v)
                seen.add("norm")
            elif t[0] == "head":
                stack.lm_head.load_weight(v.to(stack.lm_head.weight.device))
                seen.add("head")
            elif t[0] == "embed":
                if embed_out is not None:
                    embed_out.copy_(v)
                else:
                    unused.append(key)
            else:
                _, i, rest = t
                if i >= len(stack.layers):
                    unused.append(key)
                    continue
                layer = stack.layers[i]
                if rest == "input_layernorm.weight":
                    layer.input_layernorm_weight.copy_(v)
                elif rest == "post_attention_layernorm.weight":
                    layer.post_attention_layernorm_weight.copy_(v)
                else:
                    m = re.match(r"(self_attn|mlp)\.(\w+)\.(base_layer\.weight|weight|lora_[AB](?:\.default)?\.weight)$", rest)
                    if not m or PROJ.get(m.group(2)) != m.group(1):
                        unused.append(key)            # rotary_emb.inv_freq and the like
                        continue
                    mod = getattr(layer, m.group(1))[m.group(2)]
                    what = m.group(3)
                    if what in ("weight", "base_layer.weight"):
                        mod.load_base_weight(v.to(mod.weight.device))
                        seen.add((i, m.group(2)))
                    elif what.startswith("lora_A"):
                        mod.lora_A.weight.copy_(v)
                    else:
                        mod.lora_B.weight.copy_(v)
                    mod.invalidate_shadows()
    if state_dict and len(unused) == len(state_dict):
        # nothing matched: a silent no-op load would leave the LoRA matrices at their initial values
        raise KeyError(f"none of the {len(state_dict)} keys names a tensor of the stack (first key: {next(iter(state_dict))!r})")
    full = any(isinstance(s, tuple) for s in seen)
    if strict and full:
        missing = [(i, n) for i in range(len(stack.layers)) for n in PROJ if (i, n) not in seen]
        if missing or "norm" not in seen or "head" not in seen:
            raise KeyError(f"checkpoint lacks frozen weights: {missing[:4]}{' ...' if len(missing) > 4 else ''}"
                           f"{'' if 'norm' in seen else ' model.norm.weight'}{'' if 'head' in seen else ' lm_head.weight'}")
    return unused


def hf_state_dict(stack, embed=None):
    """`stack` under Hugging Face LlamaForCausalLM keys (frozen weights only; bf16)."""
    sd = {}
    for i, layer in enumerate(stack.layers):
        for n, grp in PROJ.items():
            sd[f"model.layers.{i}.{grp}.{n}.weight"] = getattr(layer, grp)[n].weight.detach().clone()
        sd[f"model.layers.{i}.input_layernorm.weight"] = layer.input_layernorm_weight.detach().clone()
        sd[f"model.layers.{i}.post_attention_layernorm.weight"] = layer.post_attention_layernorm_weight.detach().clone()
    sd["model.norm.weight"] = stack.norm_weight.detach().clone()
    sd["lm_head.weight"] = stack.lm_head.weight.detach().clone()
    if embed is not None:
        sd["model.embed_tokens.weight"] = embed.detach().clone()
    return sd


def peft_adapter_state_dict(stack):
    """The trainable tensors under the keys of a peft adapter file (`get_peft_model_state_dict`: the adapter name is
    dropped) -- what the reference's checkpoints hold for the language model (leo_trainer.py:445-454 keeps the
    tensors with requires_grad)."""
    sd = {}
    for i, layer in enumerate(stack.layers):
        for n, grp in PROJ.items():
            mod = getattr(layer, grp)[n]
            base = f"{_PEFT_PREFIX}model.layers.{i}.{grp}.{n}"
            sd[f"{base}.lora_A.weight"] = mod.lora_A.weight.detach().clone()
            sd[f"{base}.lora_B.weight"] = mod.lora_B.weight.detach().clone()
    return sd


def reference_trainer_state_dict(stack):
    """The trainable language-model tensors under the keys of the REFERENCE TRAINER's `pytorch_model.bin`
    (/root/reference/trainer/leo_trainer.py:445-454 saves `named_parameters()` with requires_grad of the whole MSR3D module:
    `llm_model.` + peft's in-memory spelling, adapter name `default` included)."""
    sd = {}
    for k, v in peft_adapter_state_dict(stack).items():
        k = k.replace(".lora_A.weight", ".lora_A.default.weight").replace(".lora_B.weight", ".lora_B.default.weight")
        sd["llm_model." + k] = v
    return sd
