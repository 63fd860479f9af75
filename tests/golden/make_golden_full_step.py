"""Golden vectors for the FULL MSR3D training forward + backward (msr3d_amd/model/msr3d_full.py): the gradient of the
language-model loss THROUGH the LoRA-Llama layers and the scene-token scatter into `llm_proj` and the prompter.

Built in the build container from the reference's own pieces (nothing of it travels):
  * `OSE3DSituation` -- the REFERENCE's class, imported from /root/reference (recipe of make_golden.py; `_ext` := CPU
    oracle), eval mode, Bs = 2, 60 objects (7 padded) x 1024 points, weights from fill_state_dict(seed);
  * `llm_proj = nn.Linear(256, E)` and the statements of `MSR3D.build_embeds` / `MSR3D.forward` that join the two halves,
    restated here line by line because `MSR3D` itself needs Vicuna / CLIP / ConvNeXt weights, peft and clip to construct
    (/root/reference/model/msr3d/msr3d.py:231-232 embedding lookup, :274-287 projection, cast, indexed write of embeddings and
    mask, :368-392 answer tokens appended and targets, :409-415 LLM call, :426-441 per-sequence mean cross-entropy);
  * the language model: transformers' LlamaForCausalLM (2 layers, hidden 512, eager attention) with peft's LoRA formula on
    all seven projections (as make_golden_llama_stack.py), float32 on bf16-rounded weights;
  * `loss.mean().backward()` as /root/reference/trainer/leo_trainer.py:184-189.

Stored: the seeds and shapes (inputs are regenerated on both sides by msr3d_amd.synth), loss (B,), scene_embeds and its
gradient, gradients of llm_proj and of the prompter (full tensors for the set make_golden_fullsize.py keeps, norm + sum of
the rest) and of every LoRA matrix.        python tests/golden/make_golden_full_step.py
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import make_golden as mg  # noqa: E402
from make_golden_fullsize import full_grad  # noqa: E402
from tests.golden.make_golden_llama_layer import NAMES, LoRA  # noqa: E402
from tests.helpers import _bf16_round, llama_stack_weights  # noqa: E402

CFG = dict(layers=2, hidden=512, heads=8, inter=1024, vocab=1024, r=16, alpha=16, eps=1e-6, theta=10000.0,
           B=2, O=60, P=1024, n_pad=7, T_in=100, T_out=28, scene_token=1000)


def build_llm(seed):
    from transformers.models.llama.modeling_llama import LlamaConfig, LlamaForCausalLM
    c = CFG
    cfg = LlamaConfig(hidden_size=c["hidden"], num_attention_heads=c["heads"], num_key_value_heads=c["heads"],
                      intermediate_size=c["inter"], num_hidden_layers=c["layers"], vocab_size=c["vocab"],
                      rms_norm_eps=c["eps"], rope_theta=c["theta"], max_position_embeddings=c["T_in"] + c["T_out"],
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
        model.model.embed_tokens.weight.copy_(torch.from_numpy(embed_table(seed)))
    for p in model.parameters():
        p.requires_grad_(False)
    for layer in model.model.layers:
        for m in list(layer.self_attn.children()) + list(layer.mlp.children()):
            if isinstance(m, LoRA):
                m.A.requires_grad_(True)
                m.Bm.requires_grad_(True)
    return model


def embed_table(seed):
    """(vocab, hidden) bf16-representable embedding table, shared with the test."""
    rng = np.random.default_rng(seed + 4242)
    return _bf16_round((rng.standard_normal((CFG["vocab"], CFG["hidden"])) * 0.5).astype(np.float32))


def main(seed=0):
    torch.manual_seed(0)
    torch.set_num_threads(8)
    c = CFG
    pu, ose = mg.import_reference()
    from msr3d_amd.synth import synth_batch, synth_text
    data_seed, text_seed = 6000 + seed, 6100 + seed
    B, O, E = c["B"], c["O"], c["hidden"]
    batch = synth_batch(data_seed, B, O=O, P=c["P"], n_valid=[O - c["n_pad"]] * B, device="cpu")
    text = synth_text(text_seed, B, L=O, T_in=c["T_in"], T_out=c["T_out"], vocab=c["vocab"],
                      scene_token=c["scene_token"])
    prompter = ose.OSE3DSituation(mg.ref_cfg("as_transform_for_objects")).eval()
    prompter.load_state_dict(mg.fill_state_dict(prompter.state_dict(), seed), strict=True)
    with torch.no_grad():                       # the zero-initialised constant would hide its path
        prompter.object_orientation_feat.copy_(torch.from_numpy(
            np.random.default_rng(seed + 55).standard_normal((1, 1, 256)).astype(np.float32) * 0.3))
    llm_proj = torch.nn.Linear(256, E)
    llm_proj.load_state_dict(mg.fill_state_dict(llm_proj.state_dict(), seed + 100))
    llm_model = build_llm(seed)

    # ---- MSR3D.build_embeds (msr3d.py:231-232, 274-287)
    input_ids, attention_mask = text["input_ids"], text["attention_mask"]
    inputs_embeds = llm_model.get_input_embeddings()(input_ids)
    scene_dict = prompter({k: v.clone() for k, v in batch.items()})
    scene_embeds = llm_proj(scene_dict["obj_tokens"])
    scene_embeds.retain_grad()
    scene_cast = scene_embeds.to(dtype=inputs_embeds.dtype)
    scene_embeds_index = torch.where(input_ids == c["scene_token"])
    scene_mask = scene_dict["obj_masks"]
    inputs_embeds = inputs_embeds.clone()
    inputs_embeds[scene_embeds_index] = scene_cast.reshape(-1, scene_cast.shape[-1])
    scene_mask = scene_mask.unsqueeze(-1)
    attention_mask = attention_mask.unsqueeze(-1)
    attention_mask = attention_mask.to(dtype=scene_mask.dtype)
    attention_mask[scene_embeds_index] = scene_mask.reshape(-1, scene_mask.shape[-1])
    attention_mask = attention_mask.squeeze(-1)
    # ---- MSR3D.forward (msr3d.py:378-392)
    out_ids, out_mask = text["output_ids"], text["output_mask"]
    text_output_embeds = llm_model.get_input_embeddings()(out_ids)
    inputs_embeds = torch.cat([inputs_embeds, text_output_embeds], dim=1)
    attention_mask = torch.cat([attention_mask, out_mask.to(attention_mask.dtype)], dim=1)
    targets = torch.zeros_like(attention_mask).long().fill_(-100)
    targets_idx = out_mask.bool()
    targets[:, -targets_idx.shape[1]:][targets_idx] = out_ids[targets_idx]
    targets[:, -targets_idx.shape[1]] = -100
    # ---- LLM + loss (msr3d.py:409-415, 426-441)
    outputs = llm_model(inputs_embeds=inputs_embeds, attention_mask=attention_mask.long(), return_dict=True)
    logits = outputs.logits.float()
    shift_logits = logits[..., :-1, :].contiguous()
    shift_labels = targets[..., 1:].contiguous()
    num_tokens_for_loss = (shift_labels >= 0).int().sum(1)
    loss = F.cross_entropy(shift_logits.view(-1, c["vocab"]), shift_labels.view(-1), reduction="none")
    loss = loss.view(B, -1).sum(1) / num_tokens_for_loss
    loss.mean().backward()                               # leo_trainer.py:184-189

    rec = {"cfg": np.array([c[k] for k in ("layers", "hidden", "heads", "inter", "vocab", "r", "alpha", "B", "O", "P",
                                           "n_pad", "T_in", "T_out", "scene_token")], np.int64),
           "eps": np.float64(c["eps"]), "theta": np.float64(c["theta"]), "seed": np.int64(seed),
           "data_seed": np.int64(data_seed), "text_seed": np.int64(text_seed),
           "orientation_feat": prompter.object_orientation_feat.detach().numpy().copy(),
           "loss": loss.detach().numpy(), "scene_embeds": scene_embeds.detach().numpy(),
           "d_scene_embeds": scene_embeds.grad.numpy(), "attention_mask": attention_mask.long().numpy(),
           "targets": targets.numpy()}
    names, norms, sums = [], [], []
    for n, p in list(prompter.named_parameters()) + [("llm_proj." + k, v) for k, v in llm_proj.named_parameters()]:
        if p.grad is None:
            continue
        gr = p.grad.detach()
        names.append(n)
        norms.append(gr.double().norm().item())
        sums.append(gr.double().sum().item())
        if full_grad(n):
            rec["grad/" + n] = gr.numpy().copy()
        elif n.startswith("spatial_encoder.1.linear") and gr.dim() == 2:
            rec["grad8/" + n] = gr[::8].numpy().copy()
    rec["grad_names"] = np.array(names)
    rec["grad_norms"] = np.array(norms)
    rec["grad_sums"] = np.array(sums)
    for i, layer in enumerate(llm_model.model.layers):
        for n in NAMES:
            parent = layer.self_attn if n in ("q_proj", "k_proj", "v_proj", "o_proj") else layer.mlp
            m = getattr(parent, n)
            rec[f"dA/{i}/{n}"] = m.A.grad.numpy()
            rec[f"dB/{i}/{n}"] = m.Bm.grad.numpy()
    path = os.path.join(HERE, f"full_step_seed{seed}.npz")
    np.savez_compressed(path, **rec)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB; loss", loss.tolist(),
          "|d scene|", float(scene_embeds.grad.norm()))


if __name__ == "__main__":
    main(0)
