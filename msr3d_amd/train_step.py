"""The hot-path training step as the GPU sees it:

    [eager]  frozen PointNet++ encoder: FPS, ball query, three fused set-abstraction launches and
             the fc GEMM  (obj_fts -> (B,O,768)), then the inputs are copied into static buffers
    [graph]  zero grads -> prompter forward from the encoder features -> llm_proj -> loss
             -> backward -> clip -> AdamW                      (one GPU: ~95 kernels, one submission)
    world > 1 or gradient accumulation: the graph holds forward/backward of one micro-batch; zeroing,
             the RCCL all-reduce of the flat gradient buffer and the 3-launch optimiser are issued
             eagerly around it (no host-driven collective inside a graph)

Issued eagerly the trainable part is host-bound; captured once into a HIP graph it replays as one
submission and the host runs ahead of the GPU (tools/gap_probe.py).  The encoder stays eager:
it is a handful of launches, and keeping it outside the graph lets bench.py time its kernels with
HIP events inside the timed region.

The encoder is frozen, so the features of batch k+1 depend neither on the gradients nor on the
update of step k.  `step(batch, next_batch)` uses that in two ways:
  * one GPU: the encoder of `next_batch` runs on a side HIP stream while the graph of `batch`
    replays (optional; +6 % end to end -- the trainable part is a chain of small latency-bound
    kernels that leaves most of the chip idle; CU-masked and priority streams measured worse);
  * data-parallel: it is issued on the compute stream between the start of the all-reduce and the
    optimiser, so the exchange over xGMI is hidden behind 1.2 ms of independent work.
Results are identical to the sequential order: every step still encodes and trains one batch.

`freeze: False` (unfrozen backbone): the encoder is part of the differentiated step -- the point
clouds are a static input, its forward + backward ride in the same graph, and none of the above
overlap applies (nothing of the next batch can run before this step's update).
"""
import torch

from . import hipops


_SIDE_STREAMS = {}


def _side_stream(device):
    """ONE encoder side stream per device for the whole process.  HIP streams share a handful of hardware queues
    (GPU_MAX_HW_QUEUES, four by default), handed out as streams are created: a step object that made a stream of its own
    could land on the queue the compute stream uses, and its prefetched encoder then ran IN LINE with the step instead
    of beside it (measured: the second and later step objects of a process 7-12 % slower than the first, same kernels,
    same kernel times: bench.py's extra.* variants).  The first stream made is the one every step object uses."""
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=device)
    return _SIDE_STREAMS[key]


class HotPathTrainStep:
    def __init__(self, model, optimizer, dp, loss_fn, example_batch, use_graph=True, accum_steps=1,
                 zero_in_optimizer=False, micro_batches=1):
        """model: MSR3DHotPath; dp: FlatGradAllReduce over its trainable params;
        loss_fn(scene_dict) -> scalar, or (scalar, tensor, d scalar / d tensor) when the caller
        already holds the upstream gradient; example_batch fixes the (static) shapes.
        accum_steps: gradient accumulation as the reference trains (`gradient_accumulation_steps: 5`,
        configs/msr3d.yaml:33; accelerate scales each micro-batch loss by 1/accum_steps and
        synchronises / steps on the last one, trainer/leo_trainer.py:180-195): every call is one
        micro-batch; gradients add up in the flat buffer, the exchange, clip and AdamW run on every
        accum_steps-th call.
        micro_batches = m > 1: the accumulation window as ONE pass -- `example_batch` (and every batch
        given to a call) holds the m micro-batches of a window back to back along dim 0; encoder and
        trainable part run once over all of them, loss_fn is called per micro-batch on its slice of the
        outputs, each loss scaled by 1/m as accelerate scales it.  Same gradients as m accumulated calls
        (no operation of the path couples scenes; dropout masks are drawn per window instead of per
        micro-batch), one optimiser step per call; what stays per-micro-batch is whatever consumes
        `scene_embeds` downstream (the language model, whose memory is why the reference accumulates)."""
        self.micro_batches = int(micro_batches)
        if self.micro_batches < 1 or example_batch["obj_fts"].shape[0] % self.micro_batches:
            raise ValueError("micro_batches must divide the window's scene count")
        if self.micro_batches > 1 and int(accum_steps) != 1:
            raise ValueError("micro_batches (a window per call) and accum_steps (a micro-batch per call) exclude each other")
        self.model, self.opt, self.dp, self.loss_fn = model, optimizer, dp, loss_fn
        self.prompter = model.visual_prompter
        # freeze: False -- the encoder is part of the differentiated (and captured) step: its pass cannot
        # be hoisted out of autograd, prefetched or run ahead, and the point clouds become a static input
        enc_params = [p for p in self.prompter.obj_encoder.parameters() if p.requires_grad]
        self.unfrozen = bool(enc_params)
        if self.unfrozen and not all(any(p is q for q in dp.order) for p in enc_params):
            raise ValueError("unfrozen object encoder: its parameters must be owned by the gradient engine (dp)")
        self.use_graph = use_graph and example_batch["obj_fts"].is_cuda
        self.static = {k: torch.empty_like(v) for k, v in example_batch.items()
                       if k != "obj_fts" or self.unfrozen}
        B, O = example_batch["obj_fts"].shape[:2]
        self.static["obj_embeds"] = torch.empty(
            (B, O, self.prompter.obj_linear_projection.in_features),
            dtype=torch.float32, device=example_batch["obj_fts"].device)
        self.loss = None
        self.graph = None
        self.split = False
        self.accum_steps = int(accum_steps)
        if self.accum_steps < 1:
            raise ValueError("accum_steps must be >= 1")
        self._micro = 0
        self._zero_in_graph = False
        # zero_in_optimizer: the fused AdamW kernel clears each gradient as it consumes it (its
        # zero_grad flag) instead of a separate 21 MB fill at the start of the next step; without
        # accumulation only, and the gradients are then NOT readable after a step
        fused = bool(getattr(optimizer, "fused_clip", False))
        self._opt_zeroes = bool(zero_in_optimizer) and self.accum_steps == 1 and fused
        # world > 1 with the fused optimiser: the exchange leaves the SUM, the optimiser reads it times
        # 1 / world (no separate averaging pass over the buffer)
        if fused and dp.distributed and hasattr(dp, "scale_in_optimizer"):
            dp.scale_in_optimizer = True
        # The exchange captured with the rest -- the whole data-parallel step is ONE graph (zero -> forward ->
        # backward -> all-reduce on the side stream -> clip + AdamW), no next_batch needed.
        #   MSR3D_DP_GRAPH_COMM=1   always (no check)
        #   MSR3D_DP_GRAPH_COMM=0   never: eager RCCL calls between the captured forward / backward and the optimiser
        #   unset                   world > 1 on the GPU: captured, after capture()'s start-up self-check (two replays
        #                           against two eager steps from the same state, every rank must agree); a failed
        #                           check falls back to the eager exchange, loudly (graph_comm_check says why)
        import os
        env = os.environ.get("MSR3D_DP_GRAPH_COMM")
        self._graph_comm = env == "1" and self.accum_steps == 1
        self._graph_comm_auto = (env is None or env == "auto") and bool(dp.distributed) and self.accum_steps == 1 \
            and self.use_graph and self._backend_captures(dp)
        self.graph_comm_check = None
        self._sched_direct = False
        self._probed = False
        self.unused_parameters = []
        # encoder prefetch (software pipelining over steps)
        self._enc_stream = _side_stream(self.static["obj_embeds"].device) if self.static["obj_embeds"].is_cuda else None
        self._pref = {"key": None, "feats": torch.empty_like(self.static["obj_embeds"]), "event": None}
        self._win = {"feats": None, "index": {}, "B": 0}        # encode_window(): features of a whole accumulation window

    @staticmethod
    def _backend_captures(dp):
        """RCCL calls can be stream-captured; gloo's (host-side) cannot."""
        try:
            import torch.distributed as dist
            return dist.get_backend(getattr(dp, "group", None)) == "nccl"
        except Exception:      # noqa: BLE001
            return False

    # ---- the trainable part, on static buffers -------------------------------------
    def _fwd_bwd(self, zero=True):
        sched = getattr(self.model, "_schedule", None)
        if sched is not None and sched.enabled and sched.eligible(self.static):
            sched.bump_seed = True      # fresh dropout masks per replay, from the schedule's first launch
        elif self.static["obj_embeds"].is_cuda:
            if sched is not None:
                sched.bump_seed = False
            hipops.bump_seed(self.static["obj_embeds"].device)   # fresh dropout masks per replay
        if zero:
            self.dp.zero_grad()
        inp = dict(self.static)
        if self.unfrozen:
            inp.pop("obj_embeds")        # computed under autograd from the static point clouds
        out = self.model(inp)
        if self.micro_batches > 1:
            return self._window_loss(out)
        res = self.loss_fn(out)
        scale = 1.0 / self.accum_steps
        if isinstance(res, tuple):
            # (loss value, tensor, upstream gradient): how the path is driven in the real model --
            # the gradient of `scene_embeds` arrives from the language model's backward
            loss, y, gy = res
            torch.autograd.backward([y], [gy if self.accum_steps == 1 else gy * scale])
        else:
            loss = res
            (loss if self.accum_steps == 1 else loss * scale).backward()
        return loss.detach()

    def _window_loss(self, out):
        """loss_fn per micro-batch on its slice of the window's outputs, 1/m each; ONE backward."""
        m = self.micro_batches
        n = out["scene_embeds"].shape[0]
        per = n // m
        parts = []
        for i in range(m):
            parts.append(self.loss_fn({k: (v[i * per:(i + 1) * per] if torch.is_tensor(v) and v.dim() > 0 and v.shape[0] == n
                                           else v) for k, v in out.items()}))
        if isinstance(parts[0], tuple):
            # (loss, slice of an output, its upstream gradient): the slices are views of one tensor
            y0 = parts[0][1]
            base = [v for v in out.values() if torch.is_tensor(v) and v.requires_grad and v.shape[0] == n
                    and v.data_ptr() == y0.data_ptr() and v.shape[1:] == y0.shape[1:]]
            if not base:
                raise ValueError("window step: loss_fn must return (a slice of) one of the model's outputs")
            base = base[0]
            gy = torch.cat([g for _, _, g in parts], 0)
            torch.autograd.backward([base], [gy * (1.0 / m)])
            return torch.stack([l.detach() for l, _, _ in parts]).mean()
        total = parts[0]
        for l in parts[1:]:
            total = total + l
        total = total * (1.0 / m)
        total.backward()
        return total.detach()

    def _update(self, between=None):
        """Exchange + optimiser.  `between` (world > 1) is enqueued on the compute stream while the
        all-reduce runs on the communication stream: the NEXT batch's frozen encoder -- 1.2 ms of
        work that needs neither the gradients nor the updated weights -- hides the 21 MB exchange."""
        self.dp.start()
        if between is not None:
            between()
        self.dp.wait()
        if self._opt_zeroes:
            self.opt.step(zero_grad=True)         # ... and the gradients are cleared as they are consumed
            self.dp.reset_marks()
        elif getattr(self.opt, "fused_clip", False):
            self.opt.step()                       # clip + AdamW in the flat-buffer kernels
        else:
            self.dp.clip_grad_norm_(5.0)
            self.opt.step()

    def _train_part(self):
        loss = self._fwd_bwd(zero=not self._opt_zeroes)
        if self._opt_zeroes:
            self.dp.start()
            self.dp.wait()
            self.opt.step(zero_grad=True)
            self.dp.reset_marks()
        else:
            self._update()
        return loss

    def _micro_step(self, run, between=None):
        """Split schedule: zero the gradients before the first micro-batch, `run` forward/backward
        (eagerly or by graph replay), exchange + optimiser after the last one."""
        if self._micro == 0 and not self._zero_in_graph and not self._opt_zeroes:
            self.dp.zero_grad()
        # only the last micro-batch may exchange (overlap mode launches buckets from the hooks)
        self.dp.begin_micro(last=self._micro + 1 == self.accum_steps)
        loss = run()
        self._micro += 1
        if self._micro == self.accum_steps:
            self._micro = 0
            self._update(between)
        elif between is not None:
            between()
        return loss

    def encode_ahead(self, batch):
        """Run the frozen encoder for `batch` NOW on the compute stream; the step that later
        receives this batch finds its features ready (same hand-over as prefetch())."""
        if self.unfrozen or self._win["index"].get(id(batch["obj_fts"])):   # (already encoded with its window)
            return
        with torch.no_grad():
            self.prompter.encode_objects(batch["obj_fts"], batch.get("obj_masks"), out=self._pref["feats"])
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        self._pref["key"], self._pref["event"] = id(batch["obj_fts"]), ev

    def encode_window(self, batches):
        """Gradient accumulation with a frozen encoder: the features of ALL micro-batches of an optimiser
        step depend on nothing the step updates, so they are encoded in ONE pass (accum x B x O objects per
        launch: the persistent set-abstraction kernels are sized for a full chip and a 4-scene micro-batch
        fills a quarter of it) and each micro-step picks up its slice.  Same values as encoding each
        micro-batch on its own (every object's feature is independent of what else is in the launch: tested
        bit for bit).  Call it before the window's first micro-step; batches whose features are not in the
        window are encoded on demand as before."""
        if self.unfrozen or not batches:
            return
        fts = [b["obj_fts"] for b in batches]
        B = fts[0].shape[0]
        n = len(fts)
        # micro-batches that already lie back to back IN ONE STORAGE (slices of a window-major loader's buffer) are
        # encoded in place; separately allocated tensors that merely happen to be neighbours in the caching
        # allocator's block are not one storage and are concatenated
        step = fts[0].numel() * fts[0].element_size()
        st0 = fts[0].untyped_storage()
        room = st0.nbytes() - (fts[0].data_ptr() - st0.data_ptr())
        if room >= n * step and all(
                f.is_contiguous() and f.shape == fts[0].shape and f.data_ptr() == fts[0].data_ptr() + i * step
                and f.untyped_storage().data_ptr() == st0.data_ptr() for i, f in enumerate(fts)):
            allf = torch.as_strided(fts[0], (n * B,) + tuple(fts[0].shape[1:]), fts[0].stride())
        else:
            allf = torch.cat(fts, 0)
        masks = None
        if all("obj_masks" in b for b in batches):
            masks = torch.cat([b["obj_masks"] for b in batches], 0)
        w = self._win
        shape = (n * B,) + tuple(self.static["obj_embeds"].shape[1:])
        if w.get("feats") is None or tuple(w["feats"].shape) != shape:
            w["feats"] = torch.empty(shape, dtype=torch.float32, device=allf.device)
        with torch.no_grad():
            self.prompter.encode_objects(allf, masks, out=w["feats"])
        w["index"] = {}
        for i, f in enumerate(fts):           # (the same batch object may sit in a window twice)
            w["index"].setdefault(id(f), []).append(i)
        w["B"] = B

    def window_ready(self, batches):
        """True when encode_window() has been run for exactly these micro-batches and none was consumed."""
        need = {}
        for b in batches:
            need[id(b["obj_fts"])] = need.get(id(b["obj_fts"]), 0) + 1
        return bool(need) and all(len(self._win["index"].get(k, ())) == n for k, n in need.items())

    def prefetch(self, batch):
        """Start the frozen encoder for `batch` on the side stream (returns immediately)."""
        if self._enc_stream is None or self.unfrozen or self._win["index"].get(id(batch["obj_fts"])):
            return
        main = torch.cuda.current_stream()
        self._enc_stream.wait_stream(main)          # inputs exist; previous prefetch consumed
        # (the encoder's fc GEMM runs concurrently with the main stream's: its own split-K workspace)
        with torch.cuda.stream(self._enc_stream), torch.no_grad(), hipops.gemm_lane(1):
            self.prompter.encode_objects(batch["obj_fts"], batch.get("obj_masks"), out=self._pref["feats"])
            ev = torch.cuda.Event()
            ev.record(self._enc_stream)
        self._pref["key"], self._pref["event"] = id(batch["obj_fts"]), ev

    def _load(self, batch):
        sched = getattr(self.model, "_schedule", None)
        with torch.no_grad():
            if self.unfrozen:
                self.static["obj_fts"].copy_(batch["obj_fts"])
            elif self._win["index"].get(id(batch["obj_fts"])):  # encoded with its accumulation window
                i, Bm = self._win["index"][id(batch["obj_fts"])].pop(0), self._win["B"]
                self.static["obj_embeds"].copy_(self._win["feats"][i * Bm:(i + 1) * Bm])
            elif self._pref["key"] == id(batch["obj_fts"]):     # features were prefetched
                torch.cuda.current_stream().wait_event(self._pref["event"])
                self.static["obj_embeds"].copy_(self._pref["feats"])
                self._pref["key"] = None
            else:
                # the encoder's last GEMM writes the static buffer itself
                self.prompter.encode_objects(batch["obj_fts"], batch.get("obj_masks"),
                                             out=self.static["obj_embeds"])
            direct = sched is not None and sched.enabled and self.static["obj_embeds"].is_cuda and \
                sched.eligible(dict(batch, obj_embeds=self.static["obj_embeds"]), ignore_grad_mode=True)
            if direct:
                # the schedule's one-launch prologue reads the batch's small tensors where they are
                # and writes every derived static buffer (key mask, pairwise / Fourier features,
                # obj_locs and obj_masks copies): no input copies at all
                anchors = None
                if all(k in self.static and k in batch and self.static[k].dtype == torch.float32 and batch[k].dtype == torch.float32 and batch[k].is_contiguous()
                       for k in ("anchor_locs", "anchor_orientation")):
                    anchors = (self.static["anchor_locs"], self.static["anchor_orientation"])
                sched.stage(dict(batch, obj_embeds=self.static["obj_embeds"]), anchor_out=anchors)
                self.static["obj_masks"] = sched.valid
                self.static["obj_locs"] = sched.arena["loc6"].view(sched.valid.shape[0], sched.valid.shape[1], 6)
                self.static["_staged"] = True
                # everything else the batch carries (anchor pose, and whatever a loss_fn reads from the scene
                # dict: ids, targets, labels) still goes into the static buffers the captured step sees
                rest = [k for k in self.static if k not in ("obj_embeds", "obj_fts", "obj_masks", "obj_locs", "_staged")
                        and not (anchors and k in ("anchor_locs", "anchor_orientation"))]     # (the prologue copied those)
                missing = [k for k in rest if k not in batch]
                if missing:
                    raise KeyError(f"batch lacks {missing}, which the example batch of this step had")
                if rest:
                    torch._foreach_copy_([self.static[k] for k in rest], [batch[k] for k in rest])
            else:
                self.static.pop("_staged", None)
                keys = [k for k in self.static if k not in ("obj_embeds", "_staged", "obj_fts")]
                dst = [self.static[k] for k in keys]
                src = [batch[k] for k in keys]
                if all(d.is_cuda and s.is_cuda and d.dtype == s.dtype and d.shape == s.shape for d, s in zip(dst, src)):
                    torch._foreach_copy_(dst, src)
                else:
                    for d, s in zip(dst, src):
                        d.copy_(s)
            if direct != self._sched_direct and self.graph is not None:
                raise RuntimeError("the captured step was built for the other input-staging mode")
            self._sched_direct = direct

    def _snapshot(self):
        """Everything the warm-up steps move: weights, optimiser moments and step counter (LR
        schedule / bias-correction phase), the device dropout seed, the accumulation phase."""
        opt = self.opt
        snap = {"micro": self._micro}
        if self.unfrozen:      # BatchNorm running statistics and batch counters move in training mode
            snap["buffers"] = [b.clone() for b in self.prompter.obj_encoder.buffers()]
        if hasattr(opt, "flat_p"):
            snap["flat"] = [t.clone() for t in (opt.flat_p, opt.exp_avg, opt.exp_avg_sq, opt.step_ctr)]
        else:
            snap["params"] = [p.detach().clone() for p in self.dp.order]
            snap["opt"] = {id(p): {k: (v.clone() if torch.is_tensor(v) else v) for k, v in st.items()}
                           for p, st in opt.state.items()}
        dev = self.static["obj_embeds"].device
        if dev.type == "cuda":
            snap["seed"] = hipops.seed_word(dev).clone()
        return snap

    def _restore(self, snap):
        opt = self.opt
        with torch.no_grad():
            if "flat" in snap:
                for t, v in zip((opt.flat_p, opt.exp_avg, opt.exp_avg_sq, opt.step_ctr), snap["flat"]):
                    t.copy_(v)
                if hasattr(opt, "mark_written"):
                    opt.mark_written()         # (flat_p was written directly: version-keyed caches must rebuild)
            else:
                for p, v in zip(self.dp.order, snap["params"]):
                    p.copy_(v)
                # in place: the state tensors the warm-up created stay allocated (a capturable
                # optimiser must not initialise its state inside the capture); state that did not
                # exist before the warm-up goes back to its initial zeros
                for p, st in opt.state.items():
                    before = snap["opt"].get(id(p), {})
                    for k, v in st.items():
                        if torch.is_tensor(v):
                            v.copy_(before[k]) if k in before else v.zero_()
                        elif k in before:
                            st[k] = before[k]
            if "seed" in snap:
                hipops.seed_word(self.static["obj_embeds"].device).copy_(snap["seed"])
            if "buffers" in snap:
                for b, v in zip(self.prompter.obj_encoder.buffers(), snap["buffers"]):
                    b.copy_(v)
        self._micro = snap["micro"]
        self.dp.zero_grad()

    def _probe_unused(self, batch):
        """Once per step object: which parameters receive no gradient in this configuration (one eager
        forward + backward under the gradient engine's probe) -> FlatAdamW leaves them untouched, as
        torch.optim.AdamW leaves a parameter whose .grad is None.  Training state the probe moves (dropout
        seed, BatchNorm buffers of an unfrozen backbone) is put back."""
        self._probed = True
        if not hasattr(self.opt, "set_unused") or not hasattr(self.dp, "probe_unused"):
            return
        snap = self._snapshot()

        def run():
            self._load(batch)
            self._fwd_bwd(zero=True)
        unused = self.dp.probe_unused(run)
        self._restore(snap)              # (also clears the gradients the probe left)
        self.opt.set_unused(unused)
        self.unused_parameters = unused

    def capture(self, batch, warmup=3):
        """Warm up on a side stream (allocator, autotune, lazy inits), then capture.  The warm-up
        runs REAL steps (with world > 1: real all-reduces, the same on every rank); their effect on
        the weights, the optimiser state, the step counter and the dropout seed is rolled back
        before the capture, so training starts from exactly the state the caller prepared."""
        if not self.use_graph:
            return
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            if not self._probed:
                self._probe_unused(batch)
        snap = self._snapshot()
        with torch.cuda.stream(s):
            for _ in range(warmup * self.accum_steps):
                self._load(batch)
                if self.accum_steps > 1:
                    self._micro_step(lambda: self._fwd_bwd(zero=False))
                else:
                    self._train_part()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        self._restore(snap)
        torch.cuda.synchronize()
        if self._graph_comm_auto:
            self._capture_checked(batch)
        else:
            self._capture_graph(batch)

    def _state_vector(self):
        flat = getattr(self.opt, "flat_p", None)
        if flat is not None:
            return flat.detach().clone()
        return torch.cat([p.detach().reshape(-1).float() for p in self.dp.order])

    def _all_ranks_agree(self, ok):
        if not self.dp.distributed:
            return ok
        import torch.distributed as dist
        t = torch.tensor([1 if ok else 0], device=self.static["obj_embeds"].device, dtype=torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=getattr(self.dp, "group", None))
        return bool(t.item())

    def _sync(self):
        if self.static["obj_embeds"].is_cuda:
            torch.cuda.synchronize()

    def _capture_checked(self, batch):
        """world > 1, MSR3D_DP_GRAPH_COMM unset: take the one-graph step (exchange captured) if it survives a
        self-check, the eager exchange otherwise.  The check, from one saved state: after two replays of the captured
        step (a) every rank holds the SAME weights -- a collective that did not run on replay, or delivered different
        sums, shows here, bit for bit -- and (b) they lie where two eager steps (real collectives) put them, up to what
        a step can move: AdamW's first updates are lr * g / (|g| + eps), so a parameter whose gradient is ~0 follows the
        run-to-run rounding of its gradient by up to 2 lr a step, eager against eager as well; the bound is twice the
        distance the eager steps travelled, not a rounding tolerance.

        Every rank must issue the SAME sequence of collectives whatever happens to it locally, so the check runs in
        phases and the ranks agree (one MIN all-reduce) at the end of each: a capture is host-side work only -- it records
        the exchange, it does not run it -- so a rank whose capture raised reaches the first agreement while no peer is
        inside a replayed collective, and all of them fall back together.  The replays and the eager reference steps
        follow only once every rank holds a graph."""
        import sys
        state = {"why": None}
        snap = self._snapshot()
        self._graph_comm = True

        def phase(fn):
            """Run `fn` here unless this rank has already failed; never raise; -> every rank is still fine."""
            if state["why"] is None:
                try:
                    fn()
                except Exception as e:      # noqa: BLE001 -- a capture that fails must not take the run with it
                    state["why"] = f"{type(e).__name__}: {e}"
            return self._all_ranks_agree(state["why"] is None)

        res = {}

        def p_capture():
            res["start"] = self._state_vector()
            self._capture_graph(batch)

        def p_replay():
            for _ in range(2):
                self.graph.replay()
            self._sync()
            res["got"] = self._state_vector()

        def p_spread():
            _, res["spread"] = self.dp.replica_checksum(res["got"])

        def p_eager():
            self._restore(snap)
            saved_defer = self.dp.defer_comm
            try:
                for _ in range(2):
                    self._load(batch)
                    self._train_part()
                self._sync()
            finally:
                self.dp.defer_comm = saved_defer
            res["want"] = self._state_vector()

        ok = phase(p_capture) and phase(p_replay) and phase(p_spread) and phase(p_eager)
        if ok:
            got, want, start, spread = res["got"], res["want"], res["start"], res["spread"]
            diff = float((got - want).abs().max())
            moved = float((want - start).abs().max())
            scale = float(want.abs().max())
            if spread != 0.0:
                state["why"] = f"replicas differ after two captured steps: checksum spread {spread:.3e}"
            elif not bool(torch.isfinite(got).all()) or not (diff <= 2.0 * moved + 1e-6 * max(scale, 1e-30)):
                state["why"] = (f"two captured steps end {diff:.3e} from two eager ones, which moved the weights by "
                                f"{moved:.3e} (scale {scale:.3e})")
            self.graph_comm_check = {"captured": True, "max_abs_diff": diff, "eager_moved": moved, "scale": scale,
                                     "replica_checksum_spread": spread}
            ok = self._all_ranks_agree(state["why"] is None)
        if not ok:
            why = state["why"] or "another rank's check failed"
            print(f"[msr3d] captured gradient exchange NOT taken ({why}); falling back to the eager exchange",
                  file=sys.stderr, flush=True)
            self.graph_comm_check = {"captured": False, "why": why}
            self._graph_comm = False
            self.graph = None
            self._sync()
            self._restore(snap)
            self._sync()
            self._capture_graph(batch)
            return
        self._restore(snap)
        self._sync()
        self._load(batch)

    def _replay_whole(self):
        """Replay a graph that holds the optimiser: the captured AdamW kernel writes the parameters behind autograd's
        back and no host code of opt.step() runs on a replay, so the version counters that the caches are keyed on
        (pointnet2/fused.get_plan's state key, LoRA shadows, FrozenLinear packs) are bumped here, as full_step.py does."""
        self.graph.replay()
        if hasattr(self.opt, "mark_written"):
            self.opt.mark_written()

    def _capture_graph(self, batch):
        self.graph = torch.cuda.CUDAGraph()
        self._load(batch)
        # world > 1: the gradient exchange stays OUTSIDE the graph (RCCL calls are issued eagerly
        # between the captured forward/backward and the 3-launch optimiser) -- a handful of host
        # launches per step, and no dependence on collective capture support.
        # Gradient accumulation likewise: the graph holds one micro-batch's forward/backward.
        self.split = (self.dp.distributed and not self._graph_comm) or self.accum_steps > 1
        if self.split:
            self.dp.defer_comm = True
        elif self.dp.distributed and self.unfrozen:
            # one graph, exchange inside: a bucket is sent (on the communication stream, a fork of the
            # capture) as soon as backward has produced it -- the trainable part's gradients, complete
            # first, travel while the backbone's backward still runs.  No next batch needed.
            self.dp.defer_comm = False
        # thread_local: other threads (RCCL's watchdog polling its events, loader threads) may keep
        # calling the HIP runtime while this thread captures
        # one micro-batch per optimiser step: the gradient zero-fill rides in the graph as well
        self._zero_in_graph = self.split and self.accum_steps == 1 and not self._opt_zeroes
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
            self.loss = self._fwd_bwd(zero=self._zero_in_graph) if self.split else self._train_part()

    def __call__(self, batch, next_batch=None):
        """One training step on `batch`.  If `next_batch` is given its frozen-encoder pass is run
        early: on one GPU on a side stream, overlapping this step's trainable part; data-parallel
        (world > 1) on the compute stream right after backward, where it hides the gradient
        all-reduce.  With gradient accumulation `next_batch` may be the LIST of the next window's
        micro-batches, passed with the window's last micro-batch: data-parallel, encode_window() of that
        list is what runs beside the exchange (one GPU: ignored, the caller's encode_window() does it)."""
        next_window = None
        if isinstance(next_batch, (list, tuple)):
            next_window, next_batch = list(next_batch), None
        if not self._probed and self.static["obj_embeds"].is_cuda:
            self._probe_unused(batch)
        self._load(batch)
        if self.unfrozen:
            next_batch = None           # nothing of the next batch can run before this step's update
        # (a step whose graph holds the exchange as well is scheduled like the one-GPU step)
        whole = self.graph is not None and not self.split
        hide_comm = (next_batch is not None or next_window is not None) and self.dp.distributed and \
            self.static["obj_embeds"].is_cuda and not whole and not self.unfrozen
        if next_batch is not None and not hide_comm:
            self.prefetch(next_batch)
        between = None
        if hide_comm and next_window is not None:
            def between():
                if not any(self._win["index"].values()):      # (never over features still to be consumed)
                    self.encode_window(next_window)
        elif hide_comm:
            between = lambda: self.encode_ahead(next_batch)   # noqa: E731
        if self.graph is not None:
            if self.split:
                return self._micro_step(lambda: (self.graph.replay(), self.loss)[1], between)
            self._replay_whole()
            return self.loss
        if self.accum_steps > 1 or hide_comm:
            self.loss = self._micro_step(lambda: self._fwd_bwd(zero=False), between)
        else:
            self.loss = self._train_part()
        return self.loss
