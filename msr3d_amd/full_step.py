"""One optimiser step of the FULL MSR3D model on one rank of a data-parallel job -- what
`LeoTrainer.train_step` + accelerate's DDP do around `MSR3D.forward`
(/root/reference/trainer/leo_trainer.py:180-195: loss.mean() -> backward -> clip_grad_norm_(5.0) -> AdamW;
:50-52,135 the DDP wrap whose all-reduce this engine replaces):

    [no_grad]  frozen PointNet++ encoder                      obj_fts -> obj_embeds
    forward    prompter schedule -> llm_proj -> embedding + scatter -> n LoRA-Llama layers -> norm -> head -> seq-CE
    backward   seq-CE -> head -> layers n-1..0 -> scatter -> llm_proj / prompter schedule
               every gradient lands in ONE flat fp32 buffer (prompter + llm_proj + all LoRA pairs; 181 MB for
               Vicuna-7B r = 16) whose buckets are contiguous slices in the order backward produces them; with
               world > 1 a bucket's all-reduce is issued on the communication stream from the hook of its last
               gradient, so the LoRA buckets of the upper layers travel over xGMI while the lower layers' backward
               still runs, and only the prompter's 21 MB are exposed at the end
    update     fused global-norm clip + AdamW over the flat buffers (1 / world folded in; gradients cleared as consumed)

The reference runs 4 sequences per GPU and accumulates 5 micro-batches because of the language model's activation
memory; with 288 GB a rank takes the whole window (or more) in one pass -- `loss (B,)`'s mean over 20 sequences IS the
mean of five micro-batch means -- so there is no accumulation loop here: pass the window as the batch.
"""
import torch

from . import hipops
from .dp import FlatGradAllReduce
from .optim import FlatAdamW


class FullTrainStep:
    def __init__(self, model, lr=3e-5, betas=(0.9, 0.999), weight_decay=0.05, max_grad_norm=5.0,
                 bucket_bytes=8 << 20, overlap=True, process_group=None, zero_in_optimizer=True, use_graph=False, **opt_kw):
        """model: MSR3DFullStep on its GPU.  bucket_bytes: 8 MiB ~ one and a half decoder layers' LoRA pairs --
        large enough that a ring over point-to-point xGMI links is bandwidth- rather than latency-bound per call,
        small enough that ~20 of them leave during backward."""
        self.model = model
        params = model.get_opt_params()
        self.dp = FlatGradAllReduce(params, bucket_bytes=bucket_bytes, overlap=overlap, process_group=process_group,
                                    pack_groups=hipops.collect_pack_groups(model))
        self.opt = FlatAdamW(self.dp, lr=lr, betas=betas, weight_decay=weight_decay, max_grad_norm=max_grad_norm,
                             **opt_kw)
        if self.dp.distributed:
            self.dp.scale_in_optimizer = True          # the exchange leaves the SUM; AdamW reads it times 1 / world
        hipops.attach_packed_views(model, self.dp, self.opt)       # (also attaches the fused prompter schedule)
        # zero_in_optimizer: AdamW clears each gradient as it consumes it (no 181 MB fill per step); the gradients are
        # then NOT readable after a step -- tests that inspect them pass False
        self.zero_in_optimizer = bool(zero_in_optimizer)
        self._probed = False
        self.unused_parameters = []
        self.loss = None
        # use_graph (one rank only): after ONE eager step on a batch of the same shapes, the whole step -- seed bump, forward,
        # backward, clip + AdamW: ~1,500 launches -- is captured into a HIP graph and replayed from static copies of the
        # batch's tensors; a kernel boundary inside a replayed graph costs the GPU a fraction of what a stream-ordered launch
        # does.  With world > 1 the exchange is issued from the backward hooks as eager RCCL calls: no graph then.
        self.use_graph = bool(use_graph) and not self.dp.distributed
        self.graph = None
        self.static = None
        self._eager_steps = 0
        self._cap_stream = None

    # ------------------------------------------------------------------
    def _forward_backward(self, batch):
        d = dict(batch)
        out = self.model(d)
        loss = out["loss"].mean()                    # leo_trainer.py:184 `loss.mean()`
        loss.backward()
        return loss.detach()

    def _probe_unused(self, batch):
        """Once: which parameters never receive a gradient in this configuration (`anchor_feat`, `loc_layers` under
        'as_transform_for_objects' -- why the reference wraps with find_unused_parameters=True); FlatAdamW leaves
        them alone exactly as torch's AdamW leaves a parameter whose .grad is None."""
        self._probed = True
        snap = [t.clone() for t in (self.opt.flat_p, self.opt.exp_avg, self.opt.exp_avg_sq, self.opt.step_ctr)]
        seed = hipops.seed_word(self.opt.flat_p.device).clone()
        comm, self.dp.hold = self.dp.hold, True       # the probe exchanges nothing
        try:
            self.dp.zero_grad()
            unused = self.dp.probe_unused(lambda: self._forward_backward(batch))
        finally:
            self.dp.hold = comm
        self.dp.zero_grad()
        with torch.no_grad():
            for t, v in zip((self.opt.flat_p, self.opt.exp_avg, self.opt.exp_avg_sq, self.opt.step_ctr), snap):
                t.copy_(v)
            hipops.seed_word(self.opt.flat_p.device).copy_(seed)
        self.opt.mark_written()
        self.opt.set_unused(unused)
        self.unused_parameters = unused

    def _step(self, batch):
        hipops.bump_seed(self.opt.flat_p.device)      # fresh dropout masks per step
        sched = getattr(self.model, "_schedule", None)
        if sched is not None:
            sched.bump_seed = False
        if not self.zero_in_optimizer:
            self.dp.zero_grad()
        self.dp.begin_micro(last=True)
        loss = self._forward_backward(batch)
        self.dp.finish()                              # flush what the hooks did not send; wait for the exchange
        self.opt.step(zero_grad=self.zero_in_optimizer)      # clip + AdamW (+ gradients cleared as they are consumed)
        self.dp.reset_marks()
        return loss

    def _capture(self, batch):
        self.static = {k: v.clone() for k, v in batch.items() if torch.is_tensor(v)}
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        # (on the stream the eager step ran on: work queues, workspaces and LDS attributes exist per stream / per kernel
        # from that step -- creating them is not something a capture can hold)
        with torch.cuda.graph(self.graph, stream=self._cap_stream, capture_error_mode="thread_local"):
            self.loss = self._step(self.static)

    def __call__(self, batch):
        """batch: scene keys (obj_fts, obj_masks, obj_locs, anchor_locs, anchor_orientation) + input_ids,
        attention_mask, output_ids, output_mask, all on the model's GPU.  -> mean loss (device scalar)."""
        if not self._probed:
            self._probe_unused(batch)
        if self.use_graph:
            if self.graph is None and self._eager_steps >= 1:
                self._capture(batch)
            if self.graph is not None:
                keys = [k for k in self.static if k in batch]
                if any(batch[k].shape != self.static[k].shape for k in keys):
                    raise RuntimeError("the captured step was built for other batch shapes")
                torch._foreach_copy_([self.static[k] for k in keys], [batch[k] for k in keys])
                self.graph.replay()
                # the replayed optimiser launch rewrote the parameters without running its host code: bump their versions
                # so that version-keyed caches (the LoRA pairs' bf16 images, weight packs) are rebuilt by the next EAGER
                # forward instead of serving the images of a step ago
                self.opt.mark_written()
                return self.loss
            # the eager step(s) before the capture run on the capture's own stream
            if self._cap_stream is None:
                self._cap_stream = torch.cuda.Stream(device=self.opt.flat_p.device)
            cur = torch.cuda.current_stream(self.opt.flat_p.device)
            self._cap_stream.wait_stream(cur)
            with torch.cuda.stream(self._cap_stream):
                self.loss = self._step(batch)
            cur.wait_stream(self._cap_stream)
            self._eager_steps += 1
            return self.loss
        self.loss = self._step(batch)
        return self.loss
