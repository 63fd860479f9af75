"""`loss (B,)` of /root/reference/model/msr3d/msr3d.py:426-441 -- the mean cross-entropy of each
sequence over its supervised tokens -- as one autograd node over msr3d_seq_ce_fwd / _bwd."""
import ctypes

import torch
import torch.nn.functional as F

from .. import _lib

_DTYPE = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


class _SeqCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, targets):
        B, T, V = logits.shape
        x = logits if logits.is_contiguous() else logits.contiguous()
        tg = targets.to(device=x.device, dtype=torch.int64).contiguous()
        dev = x.device
        lse = torch.empty((B, T - 1), dtype=torch.float32, device=dev)
        tok = torch.empty((B, T - 1), dtype=torch.float32, device=dev)
        loss = torch.empty((B,), dtype=torch.float32, device=dev)
        count = torch.empty((B,), dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            rc = _lib.load().msr3d_seq_ce_fwd(B, T, V, _p(x), _DTYPE[x.dtype], _p(tg), _p(lse), _p(tok), _p(loss),
                                              _p(count), _lib.current_stream_ptr(dev))
        _lib.check(rc, "msr3d_seq_ce_fwd")
        ctx.save_for_backward(x, tg, lse, count)
        return loss

    @staticmethod
    def backward(ctx, g):
        x, tg, lse, count = ctx.saved_tensors
        B, T, V = x.shape
        dx = torch.empty_like(x)
        gg = g.to(torch.float32).contiguous()
        with torch.cuda.device(x.device):
            rc = _lib.load().msr3d_seq_ce_bwd(B, T, V, _p(x), _DTYPE[x.dtype], _p(tg), _p(lse), _p(count), _p(gg),
                                              _p(dx), _lib.current_stream_ptr(x.device))
        _lib.check(rc, "msr3d_seq_ce_bwd")
        return dx, None


def seq_mean_cross_entropy(logits, targets):
    """logits (B, T, V) f32 / f16 / bf16, targets (B, T) int64 with negative = not supervised
    -> loss (B,) f32.  GPU tensors run the fused HIP kernels (no fp32 copy of the logits, no shifted
    copies); CPU tensors take the reference's formulation verbatim."""
    if logits.is_cuda and logits.dtype in _DTYPE and logits.shape[1] >= 2 and \
            (logits.shape[2] * logits.element_size()) % 16 == 0:
        return _SeqCE.apply(logits, targets)
    lg = logits.float()
    shift_logits = lg[..., :-1, :].contiguous()
    shift_labels = targets[..., 1:].contiguous().to(lg.device)
    n = (shift_labels >= 0).int().sum(1)
    B = lg.shape[0]
    loss = F.cross_entropy(shift_logits.view(-1, lg.shape[-1]), shift_labels.view(-1), reduction="none")
    return loss.view(B, -1).sum(1) / n
