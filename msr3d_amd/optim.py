"""Flat-buffer AdamW (+ global-norm clipping) for the hot path, over msr3d_adamw_flat.

Parameters are re-pointed at views of ONE flat fp32 buffer laid out exactly like the flat
gradient buffer of `FlatGradAllReduce` (msr3d_amd/dp.py), so an optimiser step is: squared
norm -> AdamW -> tick, three launches, graph-capturable (step counter and norm live on the
device).  Same update rule as torch.optim.AdamW with the reference's settings; `state_dict`
/`load_state_dict` round-trip the moments per parameter name order.
"""
import ctypes

import torch

from . import _lib

SCHEDULES = {"constant": 0, "warmup_cosine_instructblip": 1}


class FlatAdamW:
    def __init__(self, dp, lr=3e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.05,
                 max_grad_norm=5.0, schedule="constant", warmup_steps=400, total_steps=1):
        self.dp = dp
        self.lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        self.max_grad_norm = max_grad_norm
        self.schedule = SCHEDULES[schedule]
        self.warmup_steps, self.total_steps = warmup_steps, total_steps
        flat_g = dp.flat
        if not flat_g.is_cuda:
            raise RuntimeError("FlatAdamW runs on the GPU only (no CPU fallback)")
        if flat_g.numel() % 4:
            raise RuntimeError("flat buffer length must be a multiple of 4")
        self.flat_p = torch.empty_like(flat_g)
        # same element order as the gradient buffer: reversed(params)
        off = 0
        with torch.no_grad():
            for p in dp.order:
                k = p.numel()
                view = self.flat_p[off:off + k].view_as(p)
                view.copy_(p.data)
                p.data = view
                off += k
        self.exp_avg = torch.zeros_like(flat_g)
        self.exp_avg_sq = torch.zeros_like(flat_g)
        self.sumsq = torch.zeros(1024, dtype=torch.float32, device=flat_g.device)   # block partials
        self.step_ctr = torch.zeros(1, dtype=torch.int32, device=flat_g.device)
        self.fused_clip = True

    def step(self, zero_grad=False):
        lib = _lib.load()
        dev = self.flat_p.device
        p = lambda t: ctypes.c_void_p(t.data_ptr())
        f = ctypes.c_float
        with torch.cuda.device(dev):
            rc = lib.msr3d_adamw_flat(self.flat_p.numel(), p(self.flat_p), p(self.dp.flat),
                                      p(self.exp_avg), p(self.exp_avg_sq), p(self.sumsq),
                                      p(self.step_ctr), f(self.lr), f(self.betas[0]), f(self.betas[1]),
                                      f(self.eps), f(self.wd), f(self.max_grad_norm or 0.0),
                                      self.schedule, self.warmup_steps, self.total_steps,
                                      int(zero_grad), _lib.current_stream_ptr(dev))
        _lib.check(rc, "msr3d_adamw_flat")

    def state_dict(self):
        return {"exp_avg": self.exp_avg.clone(), "exp_avg_sq": self.exp_avg_sq.clone(),
                "step": int(self.step_ctr.item())}

    def load_state_dict(self, sd):
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        self.step_ctr.fill_(int(sd["step"]))
