"""LoRA-augmented linear layer of the frozen bf16 language model
(/root/reference/model/msr3d/msr3d.py:103-112: peft LoraConfig(r=16, lora_alpha=16, dropout 0) on
q/k/v/o/gate/up/down_proj; the LLM runs under bf16 autocast, msr3d.py:409-415).

    y = x W^T + s (x A^T) B^T,   s = alpha / r;   W frozen, A (r, K) and B (N, r) trainable

`LoRALinear` keeps the frozen weight in bf16 in BOTH orientations (W for the forward, W^T for dx:
HBM capacity buys a transposing load path away) and fp32 masters of A / B like peft; forward and
backward are msr3d_bf16_gemm_lowrank (bf16 MFMA, fp32 accumulate) with the low-rank term as one
extra K step, the weight gradients msr3d_lora_grad.  GPU only (raises on CPU tensors)."""
import ctypes
import math

import torch
import torch.nn as nn

from .. import _lib

PAD_R = 64          # the low-rank pair rides as one extra 64-wide K step (zero-padded)


def _p(t):
    return ctypes.c_void_p(t.data_ptr() if t is not None else 0)


def _gemm(M, N, K, R, P, ldp, Q, ldq, P2, ldp2, Q2, ldq2, C, ldc, c_f32, scale, dev):
    with torch.cuda.device(dev):
        rc = _lib.load().msr3d_bf16_gemm_lowrank(M, N, K, R, _p(P), ldp, _p(Q), ldq, _p(P2), ldp2, _p(Q2), ldq2,
                                                 _p(C), ldc, int(c_f32), ctypes.c_float(scale),
                                                 _lib.current_stream_ptr(dev))
    _lib.check(rc, "msr3d_bf16_gemm_lowrank")


def _skinny(M, N, K, P, Q, C, ldc, scale, dev):
    """C[:, :N] = scale P Q^T, C[:, N:ldc] = 0 (N = r rows of Q)."""
    with torch.cuda.device(dev):
        rc = _lib.load().msr3d_bf16_gemm_skinny(M, N, K, _p(P), K, _p(Q), K, _p(C), ldc, ldc, ctypes.c_float(scale),
                                                _lib.current_stream_ptr(dev))
    _lib.check(rc, "msr3d_bf16_gemm_skinny")


def quant_rows_fp8(x2):
    """x2 (M, K) bf16 contiguous -> (q (M, K) uint8 = OCP e4m3 codes, scale (M,) f32): q = rne(x / scale),
    scale = max |row| / 448 (msr3d_quant_rows_fp8)."""
    M, K = x2.shape
    q = torch.empty((M, K), dtype=torch.uint8, device=x2.device)
    sc = torch.empty((M,), dtype=torch.float32, device=x2.device)
    with torch.cuda.device(x2.device):
        rc = _lib.load().msr3d_quant_rows_fp8(M, K, _p(x2), x2.stride(0), _p(q), K, _p(sc), _lib.current_stream_ptr(x2.device))
    _lib.check(rc, "msr3d_quant_rows_fp8")
    return q, sc


def _gemm_fp8(M, N, K, Pq, sp, Qq, sq, P2, Q2, C, dev):
    with torch.cuda.device(dev):
        rc = _lib.load().msr3d_fp8_gemm_lowrank(M, N, K, _p(Pq), K, _p(sp), _p(Qq), K, _p(sq), _p(P2), PAD_R, _p(Q2), PAD_R,
                                                _p(C), N, _lib.current_stream_ptr(dev))
    _lib.check(rc, "msr3d_fp8_gemm_lowrank")


def _quant_cached(x, x2):
    """The e4m3 image of an activation, shared by the projections that read the same tensor (q / k / v, gate / up):
    cached on the tensor OBJECT the layer handed in (dies with it)."""
    c = getattr(x, "_msr3d_fp8", None)
    if c is None or c[2] != x._version:
        q, sc = quant_rows_fp8(x2)
        c = (q, sc, x._version)
        try:
            x._msr3d_fp8 = c
        except AttributeError:
            pass
    return c[0], c[1]


_ws = {}


def _grad_workspace(dev, floats):
    """Partial sums of the weight-gradient row chunks (msr3d_lora_grad): one buffer per device and stream, grown
    to the largest layer seen (64 chunks x r x 11008 floats = 45 MB at r = 16)."""
    key = (dev, torch.cuda.current_stream(dev).cuda_stream)
    w = _ws.get(key)
    if w is None or w.numel() < floats:
        w = _ws[key] = torch.empty(floats, dtype=torch.float32, device=dev)
    return w


class _LoRAFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, lora_A, lora_B, mod):
        K, N, r, s = mod.in_features, mod.out_features, mod.r, mod.scaling
        x2 = x.reshape(-1, K)
        x2 = x2 if x2.is_contiguous() else x2.contiguous()
        M, dev = x2.shape[0], x.device
        # bf16 shadows of the trainable pair in the orientations the products read: built once per weight
        # version (i.e. once per optimiser step), not per call
        a_pad, b2, _, _ = mod._shadows(forward=True)
        # u = s x A^T  (M, r) bf16 in a zero-padded (M, 64): the r-row product has its own kernel
        u = torch.empty((M, PAD_R), dtype=torch.bfloat16, device=dev)
        _skinny(M, r, K, x2, a_pad, u, PAD_R, s, dev)
        y = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
        if mod.base == "fp8" and M >= 128:
            # frozen W as e4m3 with per-output-channel scales (quantised once), x per token row (per call, shared by the
            # projections reading the same tensor); the LoRA term rides in bf16 -- csrc/lora_fp8.hip
            xq, sx = _quant_cached(x, x2)
            _gemm_fp8(M, N, K, xq, sx, mod.weight_q, mod.weight_scale, u, b2, y, dev)
        else:
            _gemm(M, N, K, PAD_R, x2, K, mod.weight, K, u, PAD_R, b2, PAD_R, y, N, False, 1.0, dev)
        ctx.save_for_backward(x2, u, lora_A, lora_B)
        ctx.mod = mod
        ctx.shape = x.shape
        return y            # (M, N): the caller reshapes -- a view made in here could not be updated in place (RoPE)

    @staticmethod
    def backward(ctx, dy):
        x2, u, lora_A, lora_B = ctx.saved_tensors
        mod = ctx.mod
        K, N, r, s = mod.in_features, mod.out_features, mod.r, mod.scaling
        dev = dy.device
        dy2 = dy.reshape(-1, N).to(torch.bfloat16)
        dy2 = dy2 if dy2.is_contiguous() else dy2.contiguous()
        M = dy2.shape[0]
        _, _, bt_pad, at2 = mod._shadows()
        # v = s dy B  (M, r) bf16 in a zero-padded (M, 64)
        v = torch.empty((M, PAD_R), dtype=torch.bfloat16, device=dev)
        _skinny(M, r, N, dy2, bt_pad, v, PAD_R, s, dev)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty((M, K), dtype=torch.bfloat16, device=dev)
            if mod.base == "fp8" and mod.fp8_backward and M >= 128:
                dq, sdy = quant_rows_fp8(dy2)              # the upstream gradient, per token row
                _gemm_fp8(M, K, N, dq, sdy, mod.weight_t_q, mod.weight_t_scale, v, at2, dx, dev)
            else:
                _gemm(M, K, N, PAD_R, dy2, N, mod.weight_t, N, v, PAD_R, at2, PAD_R, dx, K, False, 1.0, dev)
            dx = dx.view(ctx.shape)
        lib = _lib.load()
        # On the flat-gradient engine (dp.py) the pair's .grad are views of the flat buffer: the kernels ADD into them
        # (accumulate = 1) and report readiness themselves -- no AccumulateGrad add launch per parameter (448 a step for
        # a 32-layer stack) and no temporaries.
        from .. import hipops
        pA, pB = mod.lora_A.weight, mod.lora_B.weight          # (the Parameter objects themselves: they carry the engine)
        direct = hipops._direct_targets(pA, pB) if (ctx.needs_input_grad[1] and ctx.needs_input_grad[2]) else None
        if direct is not None:
            dA, dB, acc = pA.grad, pB.grad, 1
        else:
            dA = torch.empty((r, K), dtype=torch.float32, device=dev)      # (written, not added to: accumulate = 0)
            dB = torch.empty((N, r), dtype=torch.float32, device=dev)
            acc = 0
        with torch.cuda.device(dev):
            st = _lib.current_stream_ptr(dev)
            # dA = (s dy B)^T x = v^T x ; dB = dy^T (s x A^T) = dy^T u   (s already inside u and v)
            ws = _grad_workspace(dev, 64 * r * max(K, N))
            rc = lib.msr3d_lora_grad(M, r, K, _p(v), PAD_R, _p(x2), K, _p(dA), 0, ctypes.c_float(1.0), acc, _p(ws), ws.numel(), st)
            _lib.check(rc, "msr3d_lora_grad")
            rc = lib.msr3d_lora_grad(M, r, N, _p(u), PAD_R, _p(dy2), N, _p(dB), 1, ctypes.c_float(1.0), acc, _p(ws), ws.numel(), st)
            _lib.check(rc, "msr3d_lora_grad")
        if direct is not None:
            hipops._direct_done(direct)
            return dx, None, None, None
        return dx, dA, dB, None


class LoRALinear(nn.Module):
    """nn.Linear(in_features, out_features, bias=False) with a frozen bf16 weight plus a rank-r
    update, peft's parameter names (`lora_A.weight (r, K)`, `lora_B.weight (N, r)`) and init
    (A: kaiming-uniform(a = sqrt 5), B: zeros)."""

    def __init__(self, in_features, out_features, r=16, lora_alpha=16, device=None, base="bf16", fp8_backward=True):
        """base = "fp8": the frozen weight is ALSO kept as OCP e4m3 with one scale per output channel, in both
        orientations (quantised whenever the bf16 weight is written), and forward / dx multiply on the MX fp8 matrix
        instruction with activations quantised per token row (csrc/lora_fp8.hip); the LoRA pair stays bf16 / fp32.
        fp8_backward = False keeps dx on the bf16 product."""
        super().__init__()
        if r not in (16, 32) or in_features % 64 or out_features % 64:
            raise ValueError("r in {16, 32}; feature sizes must be multiples of 64")
        if base not in ("bf16", "fp8"):
            raise ValueError("base must be 'bf16' or 'fp8'")
        if base == "fp8" and (in_features % 128 or out_features % 128):
            raise ValueError("fp8 base weights: feature sizes must be multiples of 128")
        self.base, self.fp8_backward = base, bool(fp8_backward)
        self.weight_q = self.weight_scale = self.weight_t_q = self.weight_t_scale = None
        self.in_features, self.out_features, self.r = in_features, out_features, r
        self.scaling = lora_alpha / r
        self.register_buffer("weight", torch.empty((out_features, in_features), dtype=torch.bfloat16, device=device))
        self.register_buffer("weight_t", torch.empty((in_features, out_features), dtype=torch.bfloat16, device=device),
                             persistent=False)
        self.lora_A = nn.Linear(in_features, r, bias=False, device=device)
        self.lora_B = nn.Linear(r, out_features, bias=False, device=device)
        nn.init.kaiming_uniform_(self.lora_A.weight, a=math.sqrt(5))
        nn.init.zeros_(self.lora_B.weight)
        self._wt_version = None       # `weight._version` the transposed copy was made from

    def _shadows(self, forward=False):
        """Zero-padded bf16 copies of A / B in the four orientations forward and backward read
        (a_pad (r, K), b2 (N, 64), bt_pad (r, N), at2 (K, 64)); rebuilt when A or B has been written."""
        A, Bw = self.lora_A.weight, self.lora_B.weight
        key = (A._version, Bw._version, A.data_ptr(), Bw.data_ptr())
        # Inside a graph capture the copies are ALWAYS rebuilt (into the same storage): a replayed optimiser step
        # changes A / B without running this host code again, so the rebuild has to be part of the graph.
        capturing = forward and A.is_cuda and torch.cuda.is_current_stream_capturing()     # (backward reuses forward's)
        if getattr(self, "_shadow_key", None) == "captured" and not forward and A.is_cuda and \
                torch.cuda.is_current_stream_capturing():
            return self._shadow                 # backward of the captured step: forward's copies
        if capturing or getattr(self, "_shadow_key", None) != key:
            r, K, N, dev = self.r, self.in_features, self.out_features, A.device
            sh = getattr(self, "_shadow", None)
            if sh is None or sh[0].device != dev:
                # allocated once: the padding columns r..63 of b2 / at2 are zero and stay zero
                sh = (torch.empty((r, K), dtype=torch.bfloat16, device=dev), torch.zeros((N, PAD_R), dtype=torch.bfloat16, device=dev),
                      torch.empty((r, N), dtype=torch.bfloat16, device=dev), torch.zeros((K, PAD_R), dtype=torch.bfloat16, device=dev))
                self._shadow = sh
            a_pad, b2, bt_pad, at2 = sh
            with torch.no_grad():               # four conversion launches per rebuild (it was nine ops with fresh buffers)
                a_pad.copy_(A)
                b2[:, :r].copy_(Bw)
                bt_pad.copy_(Bw.t())
                at2[:, :r].copy_(A.t())
            self._shadow_key = "captured" if capturing else key
        return self._shadow

    def invalidate_shadows(self):
        """Force the bf16 copies of A / B to be rebuilt at the next forward / backward (for writers that changed
        the parameters without going through autograd or FlatAdamW.mark_written)."""
        self._shadow_key = None

    def _sync_weight_t(self):
        """weight_t is a cache of weight^T (non-persistent): rebuilt whenever `weight` has been written --
        load_state_dict, .copy_(), a dtype / device move -- so dx = dy W never reads a stale transpose."""
        w = self.weight
        if self._wt_version != (w._version, w.data_ptr()) or self.weight_t.device != w.device:
            with torch.no_grad():
                if self.weight_t.shape != (w.shape[1], w.shape[0]) or self.weight_t.device != w.device:
                    self.weight_t = torch.empty((w.shape[1], w.shape[0]), dtype=w.dtype, device=w.device)
                self.weight_t.copy_(w.t())
                if self.base == "fp8" and w.is_cuda:
                    self.weight_q, self.weight_scale = quant_rows_fp8(w)
                    self.weight_t_q, self.weight_t_scale = quant_rows_fp8(self.weight_t)
            self._wt_version = (w._version, w.data_ptr())

    @torch.no_grad()
    def load_base_weight(self, w):
        """w (N, K): the frozen projection; stored in both orientations."""
        self.weight.copy_(w.to(torch.bfloat16))
        self._sync_weight_t()

    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError("LoRALinear runs on the GPU only (no CPU fallback)")
        self._sync_weight_t()
        y = _LoRAFn.apply(x.to(torch.bfloat16), self.lora_A.weight, self.lora_B.weight, self)
        return y.view(*x.shape[:-1], self.out_features)
