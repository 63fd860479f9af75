"""Golden vectors for the LoRA-Llama stack (msr3d_amd/llm/stack.py): transformers' LlamaForCausalLM (eager attention)
called the way the reference calls it -- `llm_model(inputs_embeds=..., attention_mask=...)`,
/root/reference/model/msr3d/msr3d.py:409-415 -- with peft's LoRA formula on q/k/v/o/gate/up/down_proj of every layer
(:103-112) and the reference's per-sequence mean cross-entropy (:426-441), evaluated on the CPU in float32 on
bf16-rounded weights and inputs.  Weights are regenerated on both sides from `seed` (tests/helpers.py:
llama_stack_weights).  Stored: inputs_embeds, the left-padding mask, the targets, the loss per sequence, its gradient
with respect to inputs_embeds and to every LoRA matrix.      python tests/golden/make_golden_llama_stack.py
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from tests.golden.make_golden_llama_layer import NAMES, LoRA  # noqa: E402
from tests.helpers import llama_stack_weights  # noqa: E402

CFG = dict(layers=2, hidden=512, heads=8, inter=1024, vocab=1024, r=16, alpha=16, eps=1e-6, theta=10000.0, B=2, T=128)


def main(seed=0):
    from transformers.models.llama.modeling_llama import LlamaConfig, LlamaForCausalLM
    c = CFG
    cfg = LlamaConfig(hidden_size=c["hidden"], num_attention_heads=c["heads"], num_key_value_heads=c["heads"],
                      intermediate_size=c["inter"], num_hidden_layers=c["layers"], vocab_size=c["vocab"],
                      rms_norm_eps=c["eps"], rope_theta=c["theta"], max_position_embeddings=c["T"],
                      attention_bias=False, mlp_bias=False, tie_word_embeddings=False)
    cfg._attn_implementation = "eager"
    model = LlamaForCausalLM(cfg).float().eval()
    w = llama_stack_weights(seed, c["layers"], c["hidden"], c["inter"], c["r"], c["vocab"])
    s = c["alpha"] / c["r"]
    with torch.no_grad():
        for i, layer in enumerate(model.model.layers):
            lw = w["layers"][i]
            for n in NAMES:
                parent = layer.self_attn if n in ("q_proj", "k_proj", "v_proj", "o_proj") else layer.mlp
                setattr(parent, n, LoRA(torch.from_numpy(lw[n]), torch.from_numpy(lw[n + ".A"]),
                                        torch.from_numpy(lw[n + ".B"]), s))
            layer.input_layernorm.weight.copy_(torch.from_numpy(lw["ln1"]))
            layer.post_attention_layernorm.weight.copy_(torch.from_numpy(lw["ln2"]))
        model.model.norm.weight.copy_(torch.from_numpy(w["norm"]))
        model.lm_head.weight.copy_(torch.from_numpy(w["head"]))
    for p in model.parameters():
        p.requires_grad_(False)
    for layer in model.model.layers:
        for m in list(layer.self_attn.children()) + list(layer.mlp.children()):
            if isinstance(m, LoRA):
                m.A.requires_grad_(True)
                m.Bm.requires_grad_(True)
    rng = np.random.default_rng(seed + 2000)
    bf = lambda a: torch.from_numpy(a.astype(np.float32)).to(torch.bfloat16).float()      # noqa: E731
    x = bf(rng.standard_normal((c["B"], c["T"], c["hidden"])) * 0.5).requires_grad_(True)
    keep = np.ones((c["B"], c["T"]), np.int64)
    keep[1, :37] = 0                                      # left padding, as the LLM batches are padded
    targets = rng.integers(0, c["vocab"], size=(c["B"], c["T"])).astype(np.int64)
    targets[:, :60] = -100                                # the prompt is not supervised
    targets[0, 100:] = -100
    out = model(inputs_embeds=x, attention_mask=torch.from_numpy(keep), return_dict=True)
    logits = out.logits.float()
    tg = torch.from_numpy(targets)
    shift_logits = logits[..., :-1, :].contiguous()
    shift_labels = tg[..., 1:].contiguous()
    num = (shift_labels >= 0).int().sum(1)
    loss = F.cross_entropy(shift_logits.view(-1, c["vocab"]), shift_labels.view(-1), reduction="none")
    loss = loss.view(c["B"], -1).sum(1) / num
    gl = torch.tensor([1.0, 0.5])
    (loss * gl).sum().backward()
    rec = {"cfg": np.array([c[k] for k in ("layers", "hidden", "heads", "inter", "vocab", "r", "alpha", "B", "T")], np.int64),
           "eps": np.float64(c["eps"]), "theta": np.float64(c["theta"]), "seed": np.int64(seed),
           "x": x.detach().numpy(), "keep": keep.astype(np.uint8), "targets": targets, "loss": loss.detach().numpy(),
           "grad_loss": gl.numpy(), "dx": x.grad.numpy(),
           "logits_first": logits[0, 60:64].detach().numpy()}
    for i, layer in enumerate(model.model.layers):
        for n in NAMES:
            parent = layer.self_attn if n in ("q_proj", "k_proj", "v_proj", "o_proj") else layer.mlp
            m = getattr(parent, n)
            rec[f"dA/{i}/{n}"] = m.A.grad.numpy()
            rec[f"dB/{i}/{n}"] = m.Bm.grad.numpy()
    path = os.path.join(HERE, f"llama_stack_seed{seed}.npz")
    np.savez_compressed(path, **rec)
    print(path, os.path.getsize(path), "bytes; loss", loss.tolist())


if __name__ == "__main__":
    main(0)
