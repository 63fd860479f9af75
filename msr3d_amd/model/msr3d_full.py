"""The full MSR3D training forward, from `obj_fts` to `loss (B,)`: the hot path joined to the language model
(/root/reference/model/msr3d/msr3d.py:274-287 prompter -> llm_proj -> indexed write into `inputs_embeds`,
:378-392 answer tokens appended and targets built, :409-415 LLM forward from `inputs_embeds` + `attention_mask`,
:426-441 per-sequence mean cross-entropy).

    scene dict ─► MSR3DHotPath (visual_prompter + llm_proj)                 ─► scene_embeds (B, L, E) fp32
    input_ids | output_ids ─► embed_tokens (frozen, bf16)                     ─► inputs_embeds (B, T, E) bf16
                 scatter_scene_embeds_train_ (placeholder rows, mask rows)    ─► inputs_embeds, attention_mask
                 LoRALlamaStack (n decoder layers + norm + head) + seq-CE     ─► loss (B,)

What is NOT here (out of SURVEY §8's scope): tokenizer and prompt assembly (the step takes
token ids), the 2D image branch (`image_encoder`, `llm_proj_img`: one placeholder token per sequence), CLIP fusion,
`generate`.  Parameter names: `visual_prompter.*` and `llm_proj.*` as in the reference's MSR3D; the language model
under `llm_model.*` (msr3d_amd/llm/checkpoint.py maps Hugging Face / peft keys onto it).

GPU only beyond the prompter: the language-model kernels have no CPU fallback."""
import torch
import torch.nn.functional as F

from .build import MODEL_REGISTRY
from .scene_embeds import SCENE_SP_TOKEN, MSR3DHotPath, scatter_scene_embeds_train_


def build_targets(T_in, output_ids, output_mask):
    """msr3d.py:384-392: -100 everywhere except the answer's real tokens; the answer's first token (bos) is a
    condition, not a target.  No host sync (the reference's boolean-mask assignment is one)."""
    B = output_ids.shape[0]
    tail = torch.where(output_mask.bool(), output_ids, torch.full_like(output_ids, -100))
    tail[:, 0] = -100
    head = torch.full((B, T_in), -100, dtype=output_ids.dtype, device=output_ids.device)
    return torch.cat([head, tail], 1)


@MODEL_REGISTRY.register()
class MSR3DFullStep(MSR3DHotPath):
    """cfg: `prompter`, `llm_hidden_size` as MSR3DHotPath, plus `llm`: {num_layers, hidden_size, num_heads,
    intermediate_size, vocab_size, lora: {rank, alpha}, rms_eps, rope_theta, base: 'bf16' | 'fp8'}, optional `scene_sp_token` and `device`
    (the language model's weights -- 26 GB in both orientations for Vicuna-7B -- are created there directly)."""

    def __init__(self, cfg):
        super().__init__(cfg)
        from ..llm import LoRALlamaStack
        llm = cfg.llm
        if int(llm.hidden_size) != int(cfg.llm_hidden_size):
            raise ValueError("llm.hidden_size must equal llm_hidden_size (llm_proj's output width)")
        lora = llm.get("lora", {}) if hasattr(llm, "get") else {}
        dev = cfg.get("device", None) if hasattr(cfg, "get") else None
        self.llm_model = LoRALlamaStack(int(llm.num_layers), int(llm.hidden_size), int(llm.num_heads),
                                        int(llm.intermediate_size), int(llm.vocab_size),
                                        r=int(lora.get("rank", 16)), lora_alpha=int(lora.get("alpha", 16)),
                                        rms_eps=float(llm.get("rms_eps", 1e-6)),
                                        rope_theta=float(llm.get("rope_theta", 10000.0)), device=dev,
                                        base=str(llm.get("base", "bf16")))
        self.register_buffer("embed_tokens", torch.zeros((int(llm.vocab_size), int(llm.hidden_size)),
                                                         dtype=torch.bfloat16, device=dev))
        self.scene_sp_token = int(cfg.get("scene_sp_token", SCENE_SP_TOKEN)) if hasattr(cfg, "get") else SCENE_SP_TOKEN
        if not 0 <= self.scene_sp_token < int(llm.vocab_size):
            raise ValueError("scene_sp_token outside the vocabulary")

    def get_opt_params(self):
        """Registration order = prompter, llm_proj, LoRA pairs layer 0..n-1: reversed by the gradient engine it is
        the order backward produces them (last decoder layer first, the prompter last)."""
        return [p for p in self.parameters() if p.requires_grad]

    def embed_inputs(self, data_dict, scene_embeds, scene_mask):
        """-> inputs_embeds (B, T, E) bf16 with the scene tokens written in, attention_mask (B, T) int64 with the
        object mask at the placeholder positions, targets (B, T)."""
        ids_in, am_in = data_dict["input_ids"], data_dict["attention_mask"]
        ids_out, am_out = data_dict["output_ids"], data_dict["output_mask"]
        ids = torch.cat([ids_in, ids_out], 1).contiguous()
        am = torch.cat([am_in, am_out], 1).to(torch.int64).contiguous()
        if ids.shape[1] % 64:
            raise ValueError("prompt + answer length must be a multiple of 64 tokens (left-pad the prompt)")
        emb = F.embedding(ids, self.embed_tokens)                         # (B, T, E) bf16, frozen table
        emb = scatter_scene_embeds_train_(emb, am, ids, scene_embeds, scene_mask, self.scene_sp_token)
        return emb, am, build_targets(ids_in.shape[1], ids_out, am_out)

    def forward(self, data_dict):
        """data_dict: the scene keys of MSR3DHotPath.forward + input_ids / attention_mask (B, T1) int64 (left-padded
        prompt with L scene placeholders per row) + output_ids / output_mask (B, T3) (right-padded answer + eos).
        -> data_dict with `loss` (B,) fp32 (and scene_embeds, obj_tokens, obj_masks from the hot path)."""
        if not data_dict["input_ids"].is_cuda:
            raise RuntimeError("MSR3DFullStep runs on the GPU only (the language-model kernels have no CPU fallback)")
        d = MSR3DHotPath.forward(self, data_dict)
        emb, am, targets = self.embed_inputs(d, d["scene_embeds"], d["obj_masks"])
        # (targets are -100 over the whole prompt by construction: the head and the loss run over the answer span only)
        d["loss"] = self.llm_model(emb, attention_mask=am, targets=targets, supervised_from=data_dict["input_ids"].shape[1])
        return d
