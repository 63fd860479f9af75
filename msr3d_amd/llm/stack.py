"""The language-model side of one MSR3D training step, assembled from the C-ABI pieces
(/root/reference/model/msr3d/msr3d.py:95-112 peft LoRA on q/k/v/o/gate/up/down, :409-415 the LLM forward from
`inputs_embeds` under bf16 autocast, :426-441 the per-sequence mean cross-entropy):

    inputs_embeds (B, T, H) -> n x LoRALlamaDecoderLayer -> RMSNorm -> lm_head (frozen, bf16) -> logits (B, T, V) bf16
                            -> seq_mean_cross_entropy(logits, targets)  (B,)

Everything frozen is bf16 and stays frozen (embeddings are not part of the step: the caller scatters the scene
tokens into `inputs_embeds`, msr3d.py:277-287); the only trainable tensors are the layers' lora_A / lora_B (fp32),
which is what a data-parallel step has to exchange: 14 small tensors per layer.  GPU only."""
import ctypes

import torch
import torch.nn as nn

from .. import _lib
from .decoder import LoRALlamaDecoderLayer, _RMSNormFn
from .lora import LoRALinear, refresh_shadows
from .losses import seq_mean_cross_entropy


def _p(t):
    return ctypes.c_void_p(t.data_ptr() if t is not None else 0)


def _gemm(M, N, K, P, Q, C, dev):
    """C (M, N) bf16 = P (M, K) Q (N, K)^T, bf16 operands."""
    with torch.cuda.device(dev):
        rc = _lib.load().msr3d_bf16_gemm_lowrank(M, N, K, 0, _p(P), K, _p(Q), K, None, 0, None, 0, _p(C), N, 0,
                                                 ctypes.c_float(1.0), _lib.current_stream_ptr(dev))
    _lib.check(rc, "msr3d_bf16_gemm_lowrank")


class _FrozenLinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mod):
        K, N = mod.in_features, mod.out_features
        x2 = x.reshape(-1, K)
        x2 = x2 if x2.is_contiguous() else x2.contiguous()
        y = torch.empty((x2.shape[0], N), dtype=torch.bfloat16, device=x.device)
        _gemm(x2.shape[0], N, K, x2, mod.weight, y, x.device)
        ctx.mod, ctx.shape = mod, x.shape
        return y.view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        mod = ctx.mod
        K, N = mod.in_features, mod.out_features
        dy2 = dy.reshape(-1, N).to(torch.bfloat16)
        dy2 = dy2 if dy2.is_contiguous() else dy2.contiguous()
        M = dy2.shape[0]
        S = next((c for c in (10, 8, 5, 4) if N % (64 * c) == 0), 0)
        if M <= 1024 and N >= 8192 and S:
            # few rows against a long reduction (the 32000-way head over the answer span: 260 x 4096 x 32000): one tile per
            # 128 x 128 of the output is 96 workgroups walking 500 K steps each (507 us); the reduction cut into S chunks
            # as a batch with fp32 partials puts 10 x as many workgroups on it, the partials are added in chunk order
            part = torch.empty((S, M, K), dtype=torch.float32, device=dy.device)
            with torch.cuda.device(dy.device):
                rc = _lib.load().msr3d_bf16_gemm_batched(S, 1, M, K, N // S, _p(dy2), N, N // S, 0, _p(mod.weight_t), N, N // S, 0,
                                                         _p(part), K, M * K, 0, 1, ctypes.c_float(1.0),
                                                         _lib.current_stream_ptr(dy.device))
            _lib.check(rc, "msr3d_bf16_gemm_batched")
            dx = part.sum(0).to(torch.bfloat16)
        else:
            dx = torch.empty((M, K), dtype=torch.bfloat16, device=dy.device)
            _gemm(M, K, N, dy2, mod.weight_t, dx, dy.device)
        return dx.view(ctx.shape), None


class FrozenLinear(nn.Module):
    """nn.Linear(in_features, out_features, bias=False) with a frozen bf16 weight, kept in both orientations
    (forward and dx read k-contiguous rows): the language-model head."""

    def __init__(self, in_features, out_features, device=None):
        super().__init__()
        if in_features % 64 or out_features % 64:
            raise ValueError("feature sizes must be multiples of 64")
        self.in_features, self.out_features = in_features, out_features
        self.register_buffer("weight", torch.empty((out_features, in_features), dtype=torch.bfloat16, device=device))
        self.register_buffer("weight_t", torch.empty((in_features, out_features), dtype=torch.bfloat16, device=device),
                             persistent=False)
        self._wt_version = None

    def _sync(self):
        key = (self.weight._version, self.weight.data_ptr())
        if self._wt_version != key:
            with torch.no_grad():
                self.weight_t.copy_(self.weight.t())
            self._wt_version = (self.weight._version, self.weight.data_ptr())

    def load_weight(self, w):
        self.weight.copy_(w.to(torch.bfloat16))
        self._sync()

    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError("FrozenLinear runs on the GPU only (no CPU fallback)")
        self._sync()
        return _FrozenLinearFn.apply(x.to(torch.bfloat16), self)


class LoRALlamaStack(nn.Module):
    """`layers` decoder layers + final RMSNorm + head.  Own state-dict keys: `layers.i.*` (see LoRALlamaDecoderLayer),
    `norm_weight`, `lm_head.weight`; Hugging Face LlamaForCausalLM / peft checkpoints load through
    msr3d_amd/llm/checkpoint.py::load_hf_state_dict (and save through hf_state_dict / peft_adapter_state_dict)."""

    def __init__(self, num_layers, hidden_size=4096, num_heads=32, intermediate_size=11008, vocab_size=32000, r=16,
                 lora_alpha=16, rms_eps=1e-6, rope_theta=10000.0, device=None, base="bf16"):
        """base = "fp8": the decoder layers' frozen projections on e4m3 operands (LoRALinear); the head stays bf16."""
        super().__init__()
        self.layers = nn.ModuleList([LoRALlamaDecoderLayer(hidden_size, num_heads, intermediate_size, r, lora_alpha,
                                                           rms_eps, rope_theta, device=device, base=base)
                                     for _ in range(num_layers)])
        self.register_buffer("norm_weight", torch.ones(hidden_size, dtype=torch.bfloat16, device=device))
        self.lm_head = FrozenLinear(hidden_size, vocab_size, device=device)
        self.eps = rms_eps
        self._lora_mods = None

    def lora_parameters(self):
        return [p for p in self.parameters() if p.requires_grad]

    def _pairs(self):
        if self._lora_mods is None:
            self._lora_mods = [m for m in self.modules() if isinstance(m, LoRALinear)]
        return self._lora_mods

    def logits(self, inputs_embeds, attention_mask=None, from_position=0):
        """from_position = p: only the logits of positions p .. T-1 (B, T - p, V) -- the attention of every layer still runs
        over all T tokens; the LAST layer's token-local half (second norm + MLP), the final norm and the head only over that
        tail (nothing else ever reads the other rows of those)."""
        x = inputs_embeds.to(torch.bfloat16)
        # the bf16 images of all 7 x layers LoRA pairs: ONE launch per optimiser step (224 pairs x 4 copies before)
        mods = self._pairs()
        capturing = x.is_cuda and torch.cuda.is_current_stream_capturing()
        refresh_shadows(mods, capturing)
        keep = None if attention_mask is None else attention_mask.to(torch.uint8).contiguous()     # once, not per layer
        for m in mods:
            m._fresh_in_capture = capturing
        try:
            delta = None
            last = len(self.layers) - 1
            for i, layer in enumerate(self.layers):
                # the residual sum x + mlp(x) is left to the NEXT layer's input norm (one fused launch, no add kernel);
                # the last layer's token-local half (second norm + MLP) only over the positions the head reads
                x, delta = layer(x, attention_mask=keep, delta=delta, defer_residual=True,
                                 tail_from=from_position if i == last else 0)
            if from_position > 0 and not self.layers:
                x = x[:, from_position:].contiguous()
            _, h = _RMSNormFn.apply(x, delta, self.norm_weight, self.eps)
        finally:
            for m in mods:
                m._fresh_in_capture = False
        return self.lm_head(h)

    def forward(self, inputs_embeds, attention_mask=None, targets=None, supervised_from=None):
        """-> logits (B, T, V) bf16, or with `targets` (B, T) int64 (negative = not supervised) the per-sequence mean
        cross-entropy (B,) of msr3d.py:426-441.
        supervised_from = p (with targets): the caller GUARANTEES targets[:, :p] < 0 (msr3d.py:384-392 builds them that
        way: -100 over the whole prompt, p = its length).  The loss reads logits[t] against targets[t + 1], so only the
        logits of positions p - 1 .. T - 2 can contribute: final norm, the 32000-way head and the cross-entropy then run
        over T - p + 1 positions instead of T (65 of 576 in the benchmarked shape) -- same loss, same gradients (the
        skipped rows' d logits are exactly zero), 0.9 ms of a step's head products not spent on rows nothing reads."""
        if targets is None:
            return self.logits(inputs_embeds, attention_mask)
        p0 = 0 if supervised_from is None else max(int(supervised_from) - 1, 0)
        lg = self.logits(inputs_embeds, attention_mask, from_position=p0)
        return seq_mean_cross_entropy(lg, targets[:, p0:] if p0 else targets)
