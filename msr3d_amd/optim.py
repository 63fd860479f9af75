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
                 max_grad_norm=5.0, schedule="constant", warmup_steps=400, total_steps=1,
                 sched_steps_per_update=1):
        """schedule 'warmup_cosine_instructblip' follows optim/scheduler.py:17-20 in units of
        SCHEDULER steps.  In the reference accelerate's AcceleratedScheduler advances the LambdaLR
        `num_processes` times per optimiser step (split_batches False), so on N GPUs the 400-step
        warm-up lasts 400/N optimiser steps: pass sched_steps_per_update=N (and the reference's
        total = epochs * len(loader) * N scheduler steps) to reproduce its learning-rate curve."""
        if schedule not in SCHEDULES:
            raise ValueError("schedule must be one of %s" % sorted(SCHEDULES))
        if SCHEDULES[schedule] == 1 and not total_steps > warmup_steps >= 1:
            raise ValueError("warmup_cosine_instructblip needs total_steps > warmup_steps >= 1 "
                             f"(got total_steps={total_steps}, warmup_steps={warmup_steps})")
        if sched_steps_per_update < 1:
            raise ValueError("sched_steps_per_update must be >= 1")
        self.sched_mult = int(sched_steps_per_update)
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
        self.flat_p.zero_()      # (alignment gaps between parameters stay zero)
        with torch.no_grad():
            for p in dp.order:
                k, off = p.numel(), dp.offset[id(p)]
                view = self.flat_p[off:off + k].view_as(p)
                view.copy_(p.data)
                p.data = view
        self.exp_avg = torch.zeros_like(flat_g)
        self.exp_avg_sq = torch.zeros_like(flat_g)
        self.sumsq = torch.zeros(1024, dtype=torch.float32, device=flat_g.device)   # block partials
        self.step_ctr = torch.zeros(1, dtype=torch.int32, device=flat_g.device)
        self.fused_clip = True
        # one byte per float4 of the flat buffers: 0 = the parameter it belongs to receives no gradient and
        # is left alone (set_unused); None until then = everything is updated
        self.active = None
        self.sync_replicas()

    def sync_replicas(self, src=0):
        """Every rank takes rank `src`'s parameters, moments and step counter (what torch DDP's wrap
        does for the parameters; call it again after a rank-0-only checkpoint load)."""
        import torch.distributed as dist
        if not (dist.is_initialized() and self.dp.world > 1):
            return
        for t in (self.flat_p, self.exp_avg, self.exp_avg_sq, self.step_ctr):
            dist.broadcast(t, src=src, group=self.dp.group)
        self.mark_written()

    def mark_written(self):
        """The parameters were written through `flat_p` behind autograd's back (the fused kernel, a broadcast, a
        roll-back copy): bump their version counters so that anything keyed on `param._version` -- LoRALinear's bf16
        shadows of A / B -- rebuilds.  Host side only, no launch.  Call it after ANY direct write to `flat_p`."""
        inc = getattr(torch.autograd.graph, "increment_version", None)
        if inc is None:
            raise RuntimeError("torch.autograd.graph.increment_version is missing: caches keyed on parameter versions "
                               "(LoRALinear shadows) could go stale silently -- need torch >= 2.3")
        for q in self.dp.order:
            inc(q)

    def set_unused(self, params):
        """Parameters that receive no gradient in this configuration (`anchor_feat`, `loc_layers` with
        situation_type 'as_transform_for_objects', a classification head nobody reads): torch.optim.AdamW
        skips a parameter whose .grad is None -- no moment update, NO weight decay -- and the reference
        trains exactly so (DDP with find_unused_parameters=True, trainer/leo_trainer.py:50-52).  Here
        every parameter has a (zero) gradient view, so the set is stated: these are left untouched."""
        n4 = self.flat_p.numel() // 4
        act = torch.ones(n4, dtype=torch.uint8, device=self.flat_p.device)
        for p in params:
            off, k = self.dp.offset[id(p)], p.numel()         # (offsets are multiples of 4 elements)
            act[off // 4:(off + k + 3) // 4] = 0
        self.active = act if int((act == 0).sum()) else None
        self.unused = [id(p) for p in params]

    def step(self, zero_grad=False):
        lib = _lib.load()
        dev = self.flat_p.device
        p = lambda t: ctypes.c_void_p(t.data_ptr())
        f = ctypes.c_float
        with torch.cuda.device(dev):
            # scale_in_optimizer: the gradient engine left the all-reduced SUM in the buffer (dp.py)
            gscale = 1.0 / self.dp.world if getattr(self.dp, "scale_in_optimizer", False) else 1.0
            rc = lib.msr3d_adamw_flat_scaled(
                self.flat_p.numel(), p(self.flat_p), p(self.dp.flat), p(self.exp_avg), p(self.exp_avg_sq),
                p(self.sumsq), p(self.step_ctr), f(self.lr), f(self.betas[0]), f(self.betas[1]), f(self.eps),
                f(self.wd), f(self.max_grad_norm or 0.0), self.schedule | (self.sched_mult << 8),
                self.warmup_steps, self.total_steps, int(zero_grad),
                p(self.active) if self.active is not None else None, f(gscale), _lib.current_stream_ptr(dev))
        _lib.check(rc, "msr3d_adamw_flat_scaled")
        self.mark_written()       # (the kernel wrote the parameters behind autograd's back)

    def state_dict(self, names=None):
        """Per-parameter moments keyed by position in `dp.order` (or by `names[i]`, the parameter
        names in that order): independent of the flat layout, so it survives a change of pack groups
        or of the trainable set."""
        keys = list(names) if names is not None else [str(i) for i in range(len(self.dp.order))]
        if len(keys) != len(self.dp.order):
            raise ValueError("names must have one entry per parameter of dp.order")
        state = {}
        for k, p in zip(keys, self.dp.order):
            n, off = p.numel(), self.dp.offset[id(p)]
            state[k] = {"exp_avg": self.exp_avg[off:off + n].view_as(p).clone(),
                        "exp_avg_sq": self.exp_avg_sq[off:off + n].view_as(p).clone()}
        return {"state": state, "step": int(self.step_ctr.item())}

    def load_state_dict(self, sd, names=None):
        keys = list(names) if names is not None else [str(i) for i in range(len(self.dp.order))]
        for k, p in zip(keys, self.dp.order):
            n, off = p.numel(), self.dp.offset[id(p)]
            if k in sd["state"]:
                if tuple(sd["state"][k]["exp_avg"].shape) != tuple(p.shape):
                    raise ValueError(f"optimizer state of {k}: shape mismatch")
                self.exp_avg[off:off + n].copy_(sd["state"][k]["exp_avg"].reshape(-1))
                self.exp_avg_sq[off:off + n].copy_(sd["state"][k]["exp_avg_sq"].reshape(-1))
        self.step_ctr.fill_(int(sd["step"]))

    def load_params_flat(self, flat):
        """Overwrite every parameter from a flat fp32 tensor laid out like `flat_p` (checkpoint restore, roll-back)."""
        with torch.no_grad():
            self.flat_p.copy_(flat)
        self.mark_written()
