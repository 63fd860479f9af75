"""One LoRA-Llama decoder layer on the HIP kernels: what >99 % of a full MSR3D
training step spends its time in (/root/reference/model/msr3d/msr3d.py:103-112 LoRA on q/k/v/o/gate/up/down,
:409-415 the LLM forward under bf16 autocast).  Same computation graph as
transformers.models.llama.modeling_llama.LlamaDecoderLayer with eager attention:

    h = RMSNorm(x); q, k, v = proj(h); RoPE(q, k); P = softmax(q k^T / sqrt(d) + causal + key padding)
    x = x + o_proj(P v); h = RMSNorm(x); x = x + down_proj(silu(gate_proj(h)) * up_proj(h))

with every projection a LoRALinear (msr3d_bf16_gemm_lowrank: frozen bf16 weight + rank-r update riding as one
extra K step), the per-(sequence, head) products of the attention on msr3d_bf16_gemm_batched (the scores of a
576-token sequence are 85 MB per layer on a 288 GB part: no tiling of the softmax needed; the attention is 2 %
of the layer's FLOPs), and the row-local pieces on csrc/llm_layer.hip.  bf16 storage, fp32 accumulation.
Forward + backward: dx, and dA / dB of the seven LoRA pairs (the base weights and norm weights are frozen).
GPU only."""
import ctypes
import math

import torch
import torch.nn as nn

from .. import _lib
from .lora import LoRALinear


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


def _bgemm(dev, B, H, M, N, K, P, ldp, po, pi, Q, ldq, qo, qi, C, ldc, co, ci, c_f32, scale):
    with torch.cuda.device(dev):
        _call("msr3d_bf16_gemm_batched", B, H, M, N, K, _p(P), ldp, po, pi, _p(Q), ldq, qo, qi, _p(C), ldc, co, ci,
              int(c_f32), ctypes.c_float(scale), _st(dev))


def _transpose(dev, B, H, rows, cols, src, lds, so, si, dst, ldd, do, di):
    with torch.cuda.device(dev):
        _call("msr3d_transpose_bf16", B, H, rows, cols, _p(src), lds, so, si, _p(dst), ldd, do, di, _st(dev))


class _AttentionFn(torch.autograd.Function):
    """q, k, v (B, T, H, D) bf16 (RoPE applied), keep (B, T) uint8 or None -> context (B, T, H D) bf16."""

    @staticmethod
    def forward(ctx, q, k, v, keep):
        B, T, H, D = q.shape
        if T % 64 or D % 64:
            raise ValueError("attention: sequence length and head size must be multiples of 64")
        dev, HD = q.device, H * D
        scale = 1.0 / math.sqrt(D)
        S = torch.empty((B, H, T, T), dtype=torch.float32, device=dev)
        # scores[b, h] = scale q_h k_h^T: rows of (B, T, H D), head h at column offset h D
        _bgemm(dev, B, H, T, T, D, q, HD, T * HD, D, k, HD, T * HD, D, S, T, H * T * T, T * T, True, scale)
        P = torch.empty((B, H, T, T), dtype=torch.bfloat16, device=dev)
        with torch.cuda.device(dev):
            _call("msr3d_causal_softmax_fwd", B, H, T, _p(S), _p(keep), _p(P), _st(dev))
        del S
        vt = torch.empty((B, H, D, T), dtype=torch.bfloat16, device=dev)       # v_h^T: the contraction index contiguous
        _transpose(dev, B, H, T, D, v, HD, T * HD, D, vt, T, H * D * T, D * T)
        out = torch.empty((B, T, HD), dtype=torch.bfloat16, device=dev)
        _bgemm(dev, B, H, T, D, T, P, T, H * T * T, T * T, vt, T, H * D * T, D * T, out, HD, T * HD, D, False, 1.0)
        ctx.save_for_backward(q, k, v, P)
        ctx.scale = scale
        return out

    @staticmethod
    def backward(ctx, do):
        q, k, v, P = ctx.saved_tensors
        B, T, H, D = q.shape
        dev, HD, scale = q.device, H * D, ctx.scale
        do = do.reshape(B, T, HD)
        do = do if do.is_contiguous() else do.contiguous()
        bf = dict(dtype=torch.bfloat16, device=dev)
        # dP = dO V^T (no transposes: both operands have D contiguous); dS = softmax backward
        dP = torch.empty((B, H, T, T), dtype=torch.float32, device=dev)
        _bgemm(dev, B, H, T, T, D, do, HD, T * HD, D, v, HD, T * HD, D, dP, T, H * T * T, T * T, True, 1.0)
        dS = torch.empty((B, H, T, T), **bf)
        with torch.cuda.device(dev):
            _call("msr3d_causal_softmax_bwd", B, H, T, _p(dP), _p(P), _p(dS), _st(dev))
        del dP
        # dV = P^T dO
        Pt = torch.empty((B, H, T, T), **bf)
        _transpose(dev, B, H, T, T, P, T, H * T * T, T * T, Pt, T, H * T * T, T * T)
        dot = torch.empty((B, H, D, T), **bf)
        _transpose(dev, B, H, T, D, do, HD, T * HD, D, dot, T, H * D * T, D * T)
        dv = torch.empty((B, T, H, D), **bf)
        _bgemm(dev, B, H, T, D, T, Pt, T, H * T * T, T * T, dot, T, H * D * T, D * T, dv, HD, T * HD, D, False, 1.0)
        del Pt, dot
        # dQ = scale dS K ; dK = scale dS^T Q
        kt = torch.empty((B, H, D, T), **bf)
        _transpose(dev, B, H, T, D, k, HD, T * HD, D, kt, T, H * D * T, D * T)
        dq = torch.empty((B, T, H, D), **bf)
        _bgemm(dev, B, H, T, D, T, dS, T, H * T * T, T * T, kt, T, H * D * T, D * T, dq, HD, T * HD, D, False, scale)
        dSt = torch.empty((B, H, T, T), **bf)
        _transpose(dev, B, H, T, T, dS, T, H * T * T, T * T, dSt, T, H * T * T, T * T)
        qt = kt
        _transpose(dev, B, H, T, D, q, HD, T * HD, D, qt, T, H * D * T, D * T)
        dk = torch.empty((B, T, H, D), **bf)
        _bgemm(dev, B, H, T, D, T, dSt, T, H * T * T, T * T, qt, T, H * D * T, D * T, dk, HD, T * HD, D, False, scale)
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
        self.register_buffer("input_layernorm_weight", torch.ones(hidden_size, dtype=torch.bfloat16, device=device))
        self.register_buffer("post_attention_layernorm_weight", torch.ones(hidden_size, dtype=torch.bfloat16, device=device))
        self._rope = None

    def _tables(self, T, dev):
        if self._rope is None or self._rope[0] != (T, str(dev)):
            self._rope = ((T, str(dev)), rope_tables(T, self.head_dim, self.theta, dev))
        return self._rope[1]

    def forward(self, x, attention_mask=None, delta=None, defer_residual=False):
        """x (B, T, hidden) bf16; attention_mask (B, T) (1 / True = real token) or None.
        delta: a residual term still to be added to x (the previous layer's MLP output): the input norm's launch
        adds it.  defer_residual: return (x_after_attention, mlp_output) instead of their sum, for the next layer's
        `delta` -- a stack of layers then has no stand-alone add kernels in forward or backward."""
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
        q = _RopeFn.apply(a["q_proj"].forward2d(h), cos, sin, B, T, H, D).view(B, T, H, D)
        k = _RopeFn.apply(a["k_proj"].forward2d(h), cos, sin, B, T, H, D).view(B, T, H, D)
        v = a["v_proj"](h).view(B, T, H, D)
        ctx = _AttentionFn.apply(q, k, v, keep)
        x1, h2 = _RMSNormFn.apply(x0, a["o_proj"](ctx), self.post_attention_layernorm_weight, self.eps)
        m = self.mlp
        y = m["down_proj"](_SwiGLUFn.apply(m["gate_proj"](h2), m["up_proj"](h2)))
        if defer_residual:
            return x1, y
        return x1 + y
