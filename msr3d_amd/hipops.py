"""Autograd-level ops of the trainable part, over the C ABI (include/msr3d_hip.h).

`linear` is nn.Linear's functional form on the HIP f32-MFMA GEMM (msr3d_gemm_f32): forward
y = x W^T + b [+ GELU], backward dx = dy W, dW = dy^T x, db = colsum(dy) -- three launches of
the same kernel reading x / W / dy in place.  GPU fp32 tensors take the HIP kernels; CPU
tensors (the host-logic unit tests) take torch's own ops.  There is no silent GPU fallback:
a GPU tensor either runs the HIP kernel or raises.
"""
import ctypes

import torch
import torch.nn.functional as F

from . import _lib


def _p(t):
    return ctypes.c_void_p(t.data_ptr() if t is not None else 0)


def _gemm(a_kc, b_kc, M, N, K, A, lda, B, ldb, C, ldc, bias=None, c_pre=None, flags=0, beta=0.0):
    lib = _lib.load()
    dev = C.device
    with torch.cuda.device(dev):
        rc = lib.msr3d_gemm_f32(int(a_kc), int(b_kc), M, N, K, _p(A), lda, _p(B), ldb, _p(C), ldc,
                                _p(bias), _p(c_pre), flags, ctypes.c_float(beta),
                                _lib.current_stream_ptr(dev))
    _lib.check(rc, "msr3d_gemm_f32")


def _colsum(X, M, N, out, accumulate=False):
    lib = _lib.load()
    with torch.cuda.device(X.device):
        rc = lib.msr3d_colsum_f32(M, N, _p(X), N, _p(out), int(accumulate),
                                  _lib.current_stream_ptr(X.device))
    _lib.check(rc, "msr3d_colsum_f32")


class _HipLinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, gelu):
        x2 = x.reshape(-1, x.shape[-1])
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        w = weight if weight.is_contiguous() else weight.contiguous()
        M, K = x2.shape
        N = w.shape[0]
        y = torch.empty((M, N), dtype=torch.float32, device=x.device)
        pre = torch.empty_like(y) if gelu else None
        _gemm(True, True, M, N, K, x2, K, w, K, y, N, bias=bias, c_pre=pre, flags=1 if gelu else 0)
        ctx.save_for_backward(x2, w, pre)
        ctx.has_bias = bias is not None
        ctx.x_shape = x.shape
        return y.reshape(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        x2, w, pre = ctx.saved_tensors
        M, K = x2.shape
        N = w.shape[0]
        dy2 = dy.reshape(M, N)
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        if pre is not None:
            lib = _lib.load()
            g = torch.empty_like(dy2)
            with torch.cuda.device(dy2.device):
                rc = lib.msr3d_gelu_bwd_f32(dy2.numel(), _p(dy2), _p(pre), _p(g),
                                            _lib.current_stream_ptr(dy2.device))
            _lib.check(rc, "msr3d_gelu_bwd_f32")
            dy2 = g
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty((M, K), dtype=torch.float32, device=dy.device)
            _gemm(True, False, M, K, N, dy2, N, w, K, dx, K)          # dx = dy @ W
            dx = dx.reshape(ctx.x_shape)
        if ctx.needs_input_grad[1]:
            dw = torch.empty((N, K), dtype=torch.float32, device=dy.device)
            _gemm(False, False, N, K, M, dy2, N, x2, K, dw, K)        # dW = dy^T @ x
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = torch.empty((N,), dtype=torch.float32, device=dy.device)
            _colsum(dy2, M, N, db)
        return dx, dw, db, None


def linear(x, weight, bias=None, gelu=False):
    """F.linear (optionally followed by exact GELU) on the HIP GEMM for GPU fp32 tensors."""
    if x.is_cuda:
        if x.dtype != torch.float32 or weight.dtype != torch.float32:
            raise RuntimeError("msr3d_amd.hipops.linear: fp32 tensors expected on the GPU path")
        if gelu and weight.shape[0] % 4 != 0:
            raise RuntimeError("fused GELU needs an output width that is a multiple of 4")
        return _HipLinear.apply(x, weight, bias, gelu)
    y = F.linear(x, weight, bias)
    return F.gelu(y) if gelu else y


def module_linear(mod, x, gelu=False):
    """Apply an nn.Linear module through `linear` (its parameters stay where they are)."""
    return linear(x, mod.weight, mod.bias, gelu=gelu)
