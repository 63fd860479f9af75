"""One LoRA-Llama decoder layer on the HIP kernels: what >99 % of a full MSR3D
training step spends its time in (/root/reference/model/msr3d/msr3d.py:103-112 LoRA on q/k/v/o/gate/up/down,
:409-415 the LLM forward under bf16 autocast).  Same computation graph as
transformers.models.llama.modeling_llama.LlamaDecoderLayer with eager attention:

    h = RMSNorm(x); q, k, v = proj(h); RoPE(q, k); P = softmax(q k^T / sqrt(d) + causal + key padding)
    x = x + o_proj(P v); h = RMSNorm(x); x = x + down_proj(silu(gate_proj(h)) * up_proj(h))

with every projection a LoRALinear (msr3d_bf16_gemm_lowrank: frozen bf16 weight + rank-r update riding as one
extra K step), the attention fused (msr3d_attn_fwd / _bwd, csrc/llm_attn.hip: the scores never leave the registers),
and the row-local pieces on csrc/llm_layer.hip.  bf16 storage, fp32 accumulation.
Forward + backward: dx, and dA / dB of the seven LoRA pairs (the base weights and norm weights are frozen).
GPU only."""
import ctypes
import math

import torch
import torch.nn as nn

from .. import _lib
from .lora import PAD_R, LoRALinear, group_inputs


def _p(t):
    return ctypes.c_void_p(t.data_ptr() if t is not None else 0)


def _st(dev):
    return _lib.current_stream_ptr(dev)


def _call(name, *args):
    rc = getattr(_lib.load(), name)(*args)
    _lib.check(rc, name)


class _RMSNormFn(torch.autograd.Function):
    """(x, delta, w) -> (s = x + delta, y = RMSNorm(s) w); delta may be None (then s is x)."""

    @staticmethod
    def forward(ctx, x, delta, w, eps):
        D = x.shape[-1]
        x2 = x.reshape(-1, D)
        x2 = x2 if x2.is_contiguous() else x2.contiguous()
        M, dev = x2.shape[0], x.device
        d2 = None
        if delta is not None:
            d2 = delta.reshape(-1, D)
            d2 = d2 if d2.is_contiguous() else d2.contiguous()
        s = torch.empty_like(x2) if delta is not None else x2
        y = torch.empty_like(x2)
        rstd = torch.empty(M, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _call("msr3d_rmsnorm_fwd", M, D, _p(x2), _p(d2), _p(w), ctypes.c_float(eps),
                  _p(s) if delta is not None else _p(None), _p(y), _p(rstd), _st(dev))
        ctx.save_for_backward(s, w, rstd)
        ctx.has_delta = delta is not None
        ctx.shape = x.shape
        return s.view(x.shape), y.view(x.shape)

    @staticmethod
    def backward(ctx, ds, dy):
        s, w, rstd = ctx.saved_tensors
        M, D = s.shape
        dev = s.device
        dy2 = dy.reshape(M, D).contiguous() if dy is not None else torch.zeros_like(s)
        ds2 = ds.reshape(M, D).contiguous() if ds is not None else None
        dx = torch.empty_like(s)
        with torch.cuda.device(dev):
            _call("msr3d_rmsnorm_bwd", M, D, _p(dy2), _p(s), _p(w), _p(rstd), _p(ds2), _p(dx), _st(dev))
        dx = dx.view(ctx.shape)
        return dx, (dx if ctx.has_delta else None), None, None


class _RopeFn(torch.autograd.Function):
    """Rotary embedding IN PLACE on a projection's output (nothing else reads that tensor: _LoRAFn saves its inputs,
    not its output), and in place on the incoming gradient in backward: no copies.  Takes the projection's OWN output
    tensor (tokens, H D) -- not a view of it: an in-place op on a view of a custom function's output is rebased by
    autograd (CopySlices), which costs two full copies per call in backward (round 4: 128 copies a step)."""

    @staticmethod
    def forward(ctx, x, cos, sin, B, T, H, D):          # x (B T, H D) or (B, T, H, D) bf16, contiguous
        if not x.is_contiguous() or x.numel() != B * T * H * D:
            raise RuntimeError("_RopeFn: a contiguous (B T, H D) tensor expected")
        with torch.cuda.device(x.device):
            _call("msr3d_rope_inplace", B, T, H, D, _p(x), _p(cos), _p(sin), 0, _st(x.device))
        ctx.mark_dirty(x)
        ctx.save_for_backward(cos, sin)
        ctx.dims = (B, T, H, D)
        return x

    @staticmethod
    def backward(ctx, g):
        cos, sin = ctx.saved_tensors
        B, T, H, D = ctx.dims
        g = g.contiguous()                  # (the attention backward hands over fresh contiguous tensors: no copy)
        with torch.cuda.device(g.device):
            _call("msr3d_rope_inplace", B, T, H, D, _p(g), _p(cos), _p(sin), 1, _st(g.device))
        return g, None, None, None, None, None, None


class _Rope2Fn(torch.autograd.Function):
    """_RopeFn on q AND k in one launch each way (msr3d_rope_inplace2)."""

    @staticmethod
    def forward(ctx, q, k, cos, sin, B, T, H, D):
        for x in (q, k):
            if not x.is_contiguous() or x.numel() != B * T * H * D:
                raise RuntimeError("_Rope2Fn: contiguous (B T, H D) tensors expected")
        with torch.cuda.device(q.device):
            _call("msr3d_rope_inplace2", B, T, H, D, _p(q), _p(k), _p(cos), _p(sin), 0, _st(q.device))
        ctx.mark_dirty(q, k)
        ctx.save_for_backward(cos, sin)
        ctx.dims = (B, T, H, D)
        return q, k

    @staticmethod
    def backward(ctx, gq, gk):
        cos, sin = ctx.saved_tensors
        B, T, H, D = ctx.dims
        gq, gk = gq.contiguous(), gk.contiguous()
        with torch.cuda.device(gq.device):
            _call("msr3d_rope_inplace2", B, T, H, D, _p(gq), _p(gk), _p(cos), _p(sin), 1, _st(gq.device))
        return gq, gk, None, None, None, None, None, None


class _AttentionFn(torch.autograd.Function):
    """q, k, v (B, T, H, D) bf16 (RoPE applied), keep (B, T) uint8 or None -> context (B, T, H D) bf16.
    Fused (csrc/llm_attn.hip): the scores stay in registers -- online softmax forward, the row's log-sum-exp kept; the
    backward recomputes the probabilities from it (one kernel for dq, one for dk / dv).  Round 4 ran seven batched
    GEMMs, two softmax launches and six transposes per layer through 85 MB of fp32 scores."""

    @staticmethod
    def forward(ctx, q, k, v, keep):
        B, T, H, D = q.shape
        if T % 64 or D not in (64, 128):
            raise ValueError("attention: sequence length a multiple of 64, head size 64 or 128")
        for t in (q, k, v):
            if not t.is_contiguous():
                raise RuntimeError("attention: contiguous (B, T, H, D) tensors expected")
        dev, HD = q.device, H * D
        scale = 1.0 / math.sqrt(D)
        out = torch.empty((B, T, HD), dtype=torch.bfloat16, device=dev)
        lse = torch.empty((B, H, T), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _call("msr3d_attn_fwd", B, T, H, D, _p(q), _p(k), _p(v), HD, _p(keep), ctypes.c_float(scale), _p(out), _p(lse),
                  _st(dev))
        ctx.save_for_backward(q, k, v, out, lse, keep)
        ctx.scale = scale
        return out

    @staticmethod
    def backward(ctx, do):
        q, k, v, out, lse, keep = ctx.saved_tensors
        B, T, H, D = q.shape
        dev, HD = q.device, H * D
        do = do.reshape(B, T, HD)
        do = do if do.is_contiguous() else do.contiguous()
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        delta = torch.empty_like(lse)
        with torch.cuda.device(dev):
            _call("msr3d_attn_bwd", B, T, H, D, _p(q), _p(k), _p(v), _p(out), _p(do), HD, _p(keep), ctypes.c_float(ctx.scale),
                  _p(lse), _p(delta), _p(dq), _p(dk), _p(dv), _st(dev))
        return dq, dk, dv, None


class _SwiGLUFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, gate, up):
        gate, up = gate.contiguous(), up.contiguous()
        out = torch.empty_like(gate)
        with torch.cuda.device(gate.device):
            _call("msr3d_swiglu_fwd", gate.numel(), _p(gate), _p(up), _p(out), _st(gate.device))
        ctx.save_for_backward(gate, up)
        return out

    @staticmethod
    def backward(ctx, dh):
        gate, up = ctx.saved_tensors
        dh = dh.contiguous()
        dg, du = torch.empty_like(gate), torch.empty_like(up)
        with torch.cuda.device(gate.device):
            _call("msr3d_swiglu_bwd", gate.numel(), _p(gate), _p(up), _p(dh), _p(dg), _p(du), _st(gate.device))
        return dg, du


def rope_tables(T, D, theta=10000.0, device=None):
    """cos / sin (T, D) fp32 for positions 0..T-1 (modeling_llama.LlamaRotaryEmbedding, default rope type)."""
    inv = 1.0 / (theta ** (torch.arange(0, D, 2, dtype=torch.float32, device=device) / D))
    fr = torch.arange(T, dtype=torch.float32, device=device)[:, None] * inv[None]
    emb = torch.cat([fr, fr], -1)
    return emb.cos().contiguous(), emb.sin().contiguous()


class LoRALlamaDecoderLayer(nn.Module):
    """State-dict keys: `self_attn.{q,k,v,o}_proj.{weight, lora_A.weight, lora_B.weight}`, `mlp.{gate,up,down}_proj.*`,
    and the two norm weights as buffers of the layer (`input_layernorm_weight`, `post_attention_layernorm_weight`).
    Hugging Face / peft spell these `input_layernorm.weight`, `q_proj.base_layer.weight`, `lora_A.default.weight`:
    msr3d_amd/llm/checkpoint.py maps both ways (load_hf_state_dict, hf_state_dict, peft_adapter_state_dict)."""

    def __init__(self, hidden_size=4096, num_heads=32, intermediate_size=11008, r=16, lora_alpha=16, rms_eps=1e-6,
                 rope_theta=10000.0, device=None, base="bf16"):
        super().__init__()
        if hidden_size % num_heads or (hidden_size // num_heads) % 64:
            raise ValueError("head size must be a multiple of 64")
        self.hidden_size, self.num_heads, self.head_dim = hidden_size, num_heads, hidden_size // num_heads
        self.eps, self.theta = rms_eps, rope_theta
        mk = lambda i, o: LoRALinear(i, o, r=r, lora_alpha=lora_alpha, device=device, base=base)     # noqa: E731
        self.self_attn = nn.ModuleDict(dict(q_proj=mk(hidden_size, hidden_size), k_proj=mk(hidden_size, hidden_size),
                                            v_proj=mk(hidden_size, hidden_size), o_proj=mk(hidden_size, hidden_size)))
        self.mlp = nn.ModuleDict(dict(gate_proj=mk(hidden_size, intermediate_size), up_proj=mk(hidden_size, intermediate_size),
                                      down_proj=mk(intermediate_size, hidden_size)))
        if device is not None and torch.device(device).type == "cuda":
            # one x A^T product per input: as many members as fit the 64 columns of the shared low-rank activation
            # (r = 16: q, k, v and gate, up; r = 32: q, k and gate, up -- v keeps its own product).  When ALL consumers of
            # the norm's output are members, they also share one d-input buffer in backward; with v outside the group
            # its gradient reaches autograd between two members, so every member then returns its own.
            fit = max(1, PAD_R // r)
            for names, table in ((("q_proj", "k_proj", "v_proj"), self.self_attn), (("gate_proj", "up_proj"), self.mlp)):
                mods = [table[n] for n in names][:fit]
                if len(mods) > 1:
                    group_inputs(mods, shared_grad=len(mods) == len(names))
        self.register_buffer("input_layernorm_weight", torch.ones(hidden_size, dtype=torch.bfloat16, device=device))
        self.register_buffer("post_attention_layernorm_weight", torch.ones(hidden_size, dtype=torch.bfloat16, device=device))
        self._rope = None

    def _tables(self, T, dev):
        if self._rope is None or self._rope[0] != (T, str(dev)):
            self._rope = ((T, str(dev)), rope_tables(T, self.head_dim, self.theta, dev))
        return self._rope[1]

    def forward(self, x, attention_mask=None, delta=None, defer_residual=False, tail_from=0):
        """x (B, T, hidden) bf16; attention_mask (B, T) (1 / True = real token) or None.
        delta: a residual term still to be added to x (the previous layer's MLP output): the input norm's launch
        adds it.  defer_residual: return (x_after_attention, mlp_output) instead of their sum, for the next layer's
        `delta` -- a stack of layers then has no stand-alone add kernels in forward or backward.
        tail_from = p > 0 (the LAST layer of a stack whose head only reads positions >= p): the attention still runs over
        all T tokens, but the second norm and the MLP -- token-local -- only over positions p .. T-1; the outputs are
        (B, T - p, hidden)."""
        if not x.is_cuda:
            raise RuntimeError("LoRALlamaDecoderLayer runs on the GPU only (no CPU fallback)")
        B, T, Hd = x.shape
        H, D = self.num_heads, self.head_dim
        x = x.to(torch.bfloat16)
        keep = None if attention_mask is None else attention_mask.to(torch.uint8).contiguous()
        cos, sin = self._tables(T, x.device)
        a = self.self_attn
        # (x0 = x + delta; with delta None it is x itself -- and the ONLY use of the layer input from here on, so its
        # gradient arrives in one piece through the norm's backward)
        x0, h = _RMSNormFn.apply(x, delta, self.input_layernorm_weight, self.eps)
        # RoPE in place on the projections' own (tokens, hidden) outputs, not on views of them
        q, k = _Rope2Fn.apply(a["q_proj"].forward2d(h), a["k_proj"].forward2d(h), cos, sin, B, T, H, D)
        q, k = q.view(B, T, H, D), k.view(B, T, H, D)
        v = a["v_proj"](h).view(B, T, H, D)
        ctx = _AttentionFn.apply(q, k, v, keep)
        o = a["o_proj"](ctx)
        if tail_from > 0:
            x0, o = x0[:, tail_from:].contiguous(), o[:, tail_from:].contiguous()
        x1, h2 = _RMSNormFn.apply(x0, o, self.post_attention_layernorm_weight, self.eps)
        m = self.mlp
        y = m["down_proj"](_SwiGLUFn.apply(m["gate_proj"](h2), m["up_proj"](h2)))
        if defer_residual:
            return x1, y
        return x1 + y
