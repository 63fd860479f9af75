"""The two lines of `MSR3D.build_embeds` that belong to the hot path
(/root/reference/model/msr3d/msr3d.py:84-86 `llm_proj`, :277-287 projection, cast and
scatter of the scene tokens into the LLM's `inputs_embeds` / `attention_mask`).

`MSR3DHotPath` carries the reference's parameter names (`visual_prompter.*`,
`llm_proj.*`), so the trainable tensors of a reference checkpoint
(`pytorch_model.bin`, leo_trainer.py:445-454) load into it unchanged.
"""
import torch
import torch.nn as nn

from .. import hipops
from ..modules.utils import disabled_train
from .build import MODEL_REGISTRY, build_model

SCENE_SP_TOKEN = 31495   # vicuna id of the scene placeholder (msr3d.py:213)


def scatter_scene_embeds(inputs_embeds, attention_mask, input_ids, scene_embeds, scene_mask,
                         scene_sp_token=SCENE_SP_TOKEN):
    """Write scene_embeds (B,L,E) / scene_mask (B,L) at the positions where
    input_ids == scene_sp_token (row-major order, L placeholders per row), as
    msr3d.py:279-287 does with torch.where + indexed assignment -- but without the
    host sync: the k-th placeholder (global running count) takes the k-th scene token.
    Returns (inputs_embeds, attention_mask) as new tensors."""
    B, T = input_ids.shape
    hit = input_ids == scene_sp_token                                  # (B,T)
    flat_hit = hit.reshape(-1)
    rank = torch.cumsum(flat_hit.to(torch.int64), 0) - 1               # k-th placeholder -> k
    rank = rank.clamp_(min=0, max=scene_embeds.shape[0] * scene_embeds.shape[1] - 1)
    src = scene_embeds.reshape(-1, scene_embeds.shape[-1]).to(inputs_embeds.dtype)
    picked = src.index_select(0, rank).reshape(B, T, -1)
    out_embeds = torch.where(hit.unsqueeze(-1), picked, inputs_embeds)
    m = scene_mask.reshape(-1).index_select(0, rank).reshape(B, T)
    out_mask = torch.where(hit, m, attention_mask.to(m.dtype))
    return out_embeds, out_mask


_DTYPE_CODE = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}


class _ScatterSceneEmbeds(torch.autograd.Function):
    """The training-time form of the hand-off: in place like msr3d.py:279-287, with the gradient
    the reference's indexed assignment has -- d scene_embeds = the placeholder rows of
    d inputs_embeds, and those rows of the overwritten embeddings receive none."""

    @staticmethod
    def forward(ctx, inputs_embeds, scene_embeds, attention_mask, input_ids, scene_mask, token):
        cnt, ws = _scatter_launch(inputs_embeds, attention_mask, input_ids, scene_embeds.detach(),
                                  scene_mask, token)
        ctx.mark_dirty(inputs_embeds)
        ctx.save_for_backward(ws, cnt)
        ctx.shape = tuple(scene_embeds.shape)
        ctx.cnt = cnt
        return inputs_embeds

    @staticmethod
    def backward(ctx, g):
        ws, cnt = ctx.saved_tensors
        n = ws.numel()
        E = g.shape[-1]
        g2 = g.reshape(-1, E)
        BT = g2.shape[0]
        # map entries beyond the count (malformed batch) carry no gradient; no host sync: they are
        # routed to a scratch row past the end
        live = torch.arange(n, device=ws.device) < cnt.to(torch.int64)
        rows = torch.where(live, ws.long().clamp(0, BT - 1), torch.full_like(ws, BT, dtype=torch.int64))
        ext = torch.cat([g2, g2.new_zeros(1, E)], 0)
        d_scene = ext.index_select(0, rows).to(torch.float32).reshape(ctx.shape)
        ext.index_fill_(0, rows, 0)
        return ext[:BT].view_as(g), d_scene, None, None, None, None


def scatter_scene_embeds_train_(inputs_embeds, attention_mask, input_ids, scene_embeds, scene_mask,
                                scene_sp_token=SCENE_SP_TOKEN):
    """In-place HIP hand-off WITH autograd (use this one where msr3d.py:279-287 sits in the
    training forward): returns inputs_embeds; gradients reach scene_embeds (hence llm_proj and the
    prompter).  The placeholder count stays on the device: `placeholder_count(inputs_embeds)`."""
    out = _ScatterSceneEmbeds.apply(inputs_embeds, scene_embeds, attention_mask, input_ids, scene_mask,
                                    scene_sp_token)
    return out


def _scatter_launch(inputs_embeds, attention_mask, input_ids, scene_embeds, scene_mask, scene_sp_token):
    import ctypes

    from .. import _lib
    B, T = input_ids.shape
    E = scene_embeds.shape[-1]
    n = scene_embeds.shape[0] * scene_embeds.shape[1]
    if not (inputs_embeds.is_cuda and inputs_embeds.is_contiguous() and input_ids.dtype == torch.int64):
        raise RuntimeError("scatter_scene_embeds_: contiguous GPU tensors and int64 ids expected")
    if attention_mask is not None and (attention_mask.dtype != torch.int64 or not attention_mask.is_contiguous()):
        raise RuntimeError("attention_mask must be contiguous int64")
    src = scene_embeds.reshape(n, E).float().contiguous()
    msk = scene_mask.reshape(n).contiguous().view(torch.uint8) if scene_mask is not None else None
    ws = torch.empty(n, dtype=torch.int32, device=input_ids.device)
    cnt = torch.zeros(1, dtype=torch.int32, device=input_ids.device)
    p = lambda t: ctypes.c_void_p(t.data_ptr() if t is not None else 0)
    lib = _lib.load()
    with torch.cuda.device(input_ids.device):
        rc = lib.msr3d_scene_scatter(B, T, n, E, p(input_ids.contiguous()), int(scene_sp_token), p(src), p(msk),
                                     _DTYPE_CODE[inputs_embeds.dtype], p(inputs_embeds),
                                     p(attention_mask), p(ws), p(cnt),
                                     _lib.current_stream_ptr(input_ids.device))
    _lib.check(rc, "msr3d_scene_scatter")
    return cnt, ws


def scatter_scene_embeds_(inputs_embeds, attention_mask, input_ids, scene_embeds, scene_mask,
                          scene_sp_token=SCENE_SP_TOKEN, validate=False):
    """In-place HIP version of the same hand-off (msr3d_scene_scatter): two launches, no host
    sync.  inputs_embeds (B,T,E) f32/f16/bf16 contiguous, attention_mask (B,T) int64 (or None),
    input_ids (B,T) int64, scene_embeds (B,L,E) f32, scene_mask (B,L) bool.  Returns the device
    int holding the number of placeholders found (== B*L in a well-formed batch).

    INFERENCE / no-grad only: the kernel writes through raw pointers, autograd does not see it.
    Called with gradients enabled on a scene_embeds that requires grad it raises (silently cutting
    the trainable hot path off from the loss is the failure mode this guards against) -- training
    uses scatter_scene_embeds_train_ (same kernels, autograd node) or scatter_scene_embeds.
    validate=True syncs and raises if the placeholder count differs from B*L, as the reference's
    indexed assignment does on a shape mismatch (msr3d.py:285)."""
    if torch.is_grad_enabled() and (scene_embeds.requires_grad or inputs_embeds.requires_grad):
        raise RuntimeError("scatter_scene_embeds_ has no autograd: use scatter_scene_embeds_train_ "
                           "(or run under torch.no_grad() for inference)")
    cnt, _ = _scatter_launch(inputs_embeds, attention_mask, input_ids, scene_embeds, scene_mask,
                             scene_sp_token)
    if validate:
        n = scene_embeds.shape[0] * scene_embeds.shape[1]
        found = int(cnt.item())
        if found != n:
            raise RuntimeError(f"{found} scene placeholders in input_ids but {n} scene tokens")
    return cnt


def project_and_scatter_(inputs_embeds, attention_mask, input_ids, obj_tokens, llm_proj, scene_mask,
                         scene_sp_token=SCENE_SP_TOKEN):
    """`llm_proj` + cast + scatter in ONE kernel (msr3d_project_scatter_bf16):
    inputs_embeds[placeholder k] = cast(obj_tokens[k] @ W^T + b), attention_mask likewise; the fp32
    (B*L, E) projector output of msr3d.py:277 is never written.  bf16 MFMA with fp32 accumulation,
    for 16-bit `inputs_embeds` (the LLM's dtype); inference / generation path -- no autograd.
    obj_tokens (B,L,K) f32, llm_proj an nn.Linear(K, E).  Returns the device placeholder count."""
    import ctypes

    from .. import _lib
    B, T = input_ids.shape
    n, K = obj_tokens.shape[0] * obj_tokens.shape[1], obj_tokens.shape[-1]
    E = llm_proj.out_features
    if not (inputs_embeds.is_cuda and inputs_embeds.is_contiguous() and input_ids.dtype == torch.int64):
        raise RuntimeError("project_and_scatter_: contiguous GPU tensors and int64 ids expected")
    if attention_mask is not None and (attention_mask.dtype != torch.int64 or not attention_mask.is_contiguous()):
        raise RuntimeError("attention_mask must be contiguous int64")
    if E % 128 or K % 32 or llm_proj.weight.dtype != torch.float32:
        raise RuntimeError("project_and_scatter_: E % 128 == 0, K % 32 == 0 and fp32 llm_proj expected")
    tok = obj_tokens.detach().reshape(n, K).float().contiguous()
    w = llm_proj.weight.detach().contiguous()
    b = llm_proj.bias.detach().contiguous() if llm_proj.bias is not None else None
    msk = scene_mask.reshape(n).contiguous().view(torch.uint8) if scene_mask is not None else None
    ws = torch.empty(n, dtype=torch.int32, device=input_ids.device)
    cnt = torch.zeros(1, dtype=torch.int32, device=input_ids.device)
    p = lambda t: ctypes.c_void_p(t.data_ptr() if t is not None else 0)   # noqa: E731
    lib = _lib.load()
    with torch.cuda.device(input_ids.device):
        rc = lib.msr3d_project_scatter_bf16(B, T, n, E, K, p(input_ids.contiguous()), int(scene_sp_token),
                                            p(tok), p(w), p(b), p(msk), _DTYPE_CODE[inputs_embeds.dtype],
                                            p(inputs_embeds), p(attention_mask), p(ws), p(cnt),
                                            _lib.current_stream_ptr(input_ids.device))
    _lib.check(rc, "msr3d_project_scatter_bf16")
    return cnt


@MODEL_REGISTRY.register()
class MSR3DHotPath(nn.Module):
    """visual_prompter (OSE3DSituation) + llm_proj: the trainable, LLM-independent part of
    `MODEL_REGISTRY["MSR3D"]`.  `cfg.prompter` is the prompter block; `cfg.llm_hidden_size`
    replaces `llm_model.config.hidden_size` (4096 for Vicuna-7B, 5120 for 13B)."""

    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.visual_prompter = build_model(cfg.prompter)
        if cfg.prompter.model.vision.args.freeze:
            self.visual_prompter.obj_encoder.train = disabled_train.__get__(
                self.visual_prompter.obj_encoder)
        self.llm_proj = nn.Linear(cfg.prompter.model.hidden_size, cfg.llm_hidden_size)

    def get_opt_params(self):
        return [p for p in self.parameters() if p.requires_grad]

    def forward(self, scene_dict):
        """-> scene_dict with obj_tokens, obj_masks (from the prompter) and scene_embeds (B,L,E)."""
        sched = getattr(self, "_schedule", None)
        if sched is not None and "obj_tokens" not in scene_dict:
            # training on the flat-buffer engine: the whole trainable part as one fixed schedule of
            # fused launches (msr3d_amd/fused_model.py); anything it does not cover falls through
            if "obj_embeds" not in scene_dict and scene_dict.get("obj_fts") is not None \
                    and scene_dict["obj_fts"].is_cuda and "single_obj" not in scene_dict:
                # (an unfrozen encoder runs under autograd here; the schedule hands back d obj_embeds)
                scene_dict["obj_embeds"] = self.visual_prompter.encode_objects(
                    scene_dict["obj_fts"], scene_dict.get("obj_masks"))
            if sched.eligible(scene_dict):
                from .. import fused_model
                return fused_model.run(self, scene_dict)
        if "obj_tokens" not in scene_dict:
            scene_dict = self.visual_prompter(scene_dict)
        scene_dict["scene_embeds"] = hipops.module_linear(self.llm_proj, scene_dict["obj_tokens"])
        return scene_dict
