"""Data-parallel gradient exchange for the hot path: one process per GPU, RCCL over xGMI.

Replaces what the reference gets implicitly from accelerate -> torch DDP
(`DistributedDataParallelKwargs(find_unused_parameters=True)`,
/root/reference/trainer/leo_trainer.py:50-52,135) with an explicit
engine shaped for this model and for MI355X's point-to-point xGMI:

  * ONE flat fp32 gradient buffer; every trainable parameter's `.grad` is a view into it,
    laid out in reverse registration order (~ the order backward produces them), so a bucket
    is a contiguous slice and needs no copy in or out;
  * the exchange runs on a SIDE stream that waits on the producing stream; by default as ONE
    all-reduce of the whole buffer right after backward (`start()` / `wait()`, or `finish()`):
    the step's forward/backward is replayed from a HIP graph, which must not contain host-driven
    collectives, and the caller (train_step.HotPathTrainStep) enqueues the NEXT batch's frozen
    encoder between `start()` and `wait()`, so the 21 MB exchange is hidden behind 1 ms of
    independent work.  `overlap=True` launches each bucket from post-accumulate-grad hooks as soon
    as its last gradient exists, so communication overlaps the rest of backward (eager mode);
  * parameters that receive no gradient in a step (`anchor_feat`, `loc_layers` in
    'as_transform_for_objects' mode -- the reason the reference needs
    find_unused_parameters) simply leave zeros in the buffer: `finish()` flushes the
    buckets whose hooks never completed, no graph traversal, no extra sync;
  * BN is frozen/eval on this path, so there is no buffer broadcast in forward.

The hot-path gradient is 5.28 M params = 21.1 MB: with 7 x ~153 GB/s links per GPU a
direct reduce-scatter/all-gather moves 2*S/8 per link (~35 us).  A ring over point-to-point xGMI
links is latency-bound per call at this size, hence one collective in the default mode; in overlap
mode the bucket size is chosen large (8 MiB -> 3 buckets) for the same reason.
"""
import os

import threading

import torch
import torch.distributed as dist


class FlatGradAllReduce:
    def __init__(self, params, bucket_bytes=8 << 20, process_group=None, overlap=False,
                 pack_groups=None, sync_params=True):
        """pack_groups: lists of parameters that must sit back to back (in the given order) in the
        flat buffer, e.g. the q/k/v/cond projection weights of one attention block, so that a
        consumer can treat them as ONE tensor (hipops.linear_packed) -- see collect_pack_groups.
        sync_params: broadcast rank 0's parameter values at construction, as torch DDP's wrap does
        (the reference gets it from accelerate.prepare, trainer/leo_trainer.py:135): replicas start
        identical whatever each rank seeded or loaded."""
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        dev = self.params[0].device
        if any(p.device != dev or p.dtype != torch.float32 for p in self.params):
            raise ValueError("all trainable parameters must be fp32 on one device")
        self.device = dev
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # `distributed`: the exchange path is taken.  MSR3D_DP_FORCE_EXCHANGE=1 takes it with a
        # single rank too (an all-reduce over a one-rank communicator): the way to run the real
        # RCCL + side-stream + graph-replay schedule on a box with one GPU
        # (tests/test_bench_ranks_gpu.py)
        self.distributed = self.world > 1 or (
            dist.is_initialized() and os.environ.get("MSR3D_DP_FORCE_EXCHANGE") == "1")
        self.on_gpu = dev.type == "cuda"
        if self.distributed and self.on_gpu and self.world > 1:
            # The all-reduce runs BESIDE the next batch's frozen encoder (train_step.py), whose persistent
            # kernels size their grids to the chip: leave RCCL's workgroups their CUs, or the blocks that find
            # no CU free start a second round and double the kernel's time.  One CU per channel; the channel
            # count is RCCL's (NCCL_MAX_NCHANNELS, which bench.py caps) or MSR3D_RESERVE_CUS.
            n = int(os.environ.get("MSR3D_RESERVE_CUS", os.environ.get("NCCL_MAX_NCHANNELS", "16")))
            try:
                from . import _lib
                _lib.set_reserved_cus(max(0, min(n, 128)))
                self.reserved_cus = max(0, min(n, 128))
            except (OSError, RuntimeError):
                self.reserved_cus = 0

        group_of = {}
        for g in (pack_groups or []):
            g = [p for p in g if p.requires_grad]
            for p in g:
                group_of[id(p)] = g
        order, seen = [], set()
        for p in reversed(self.params):                     # ~ backward order
            if id(p) in seen:
                continue
            for q in group_of.get(id(p), [p]):              # a group is emitted whole, in its order
                if id(q) not in seen:
                    seen.add(id(q))
                    order.append(q)
        self.order = order
        self.offset = {}
        # Every parameter (every pack group) starts on a 16-byte boundary: the fused kernels read
        # LayerNorm / bias vectors of the flat buffers as float4.  The gaps (a 607-way head, 3-wide
        # biases of an unfrozen backbone ...) hold zeros in the parameter, gradient and moment buffers.
        starts, off = {}, 0
        i = 0
        while i < len(order):
            g = group_of.get(id(order[i]), [order[i]])
            off = (off + 3) // 4 * 4
            for q in g:
                starts[id(q)] = off
                off += q.numel()
            i += len(g)
        total = off
        # float4 kernels (fused optimiser) see whole vectors; a multiple of 32 also splits evenly over up
        # to 8 ranks for the reduce-scatter / all-gather exchange
        total_padded = (total + 31) // 32 * 32
        self.flat = torch.zeros(total_padded, dtype=torch.float32, device=dev)
        self.numel = total
        self.buckets = []                                   # (start, end) element ranges
        self._bucket_of = {}
        off = b_start = 0
        per_bucket = max(1, bucket_bytes // 4)
        for p in order:
            n = p.numel()
            off = starts[id(p)]
            p.grad = self.flat[off:off + n].view_as(p)
            self.offset[id(p)] = off
            # lets the HIP linear backward accumulate dW/db straight into these views (no
            # temporary, no AccumulateGrad add) and report readiness itself: hipops._HipLinear
            p._msr3d_dp = self
            self._bucket_of[id(p)] = len(self.buckets)
            off += n
            if off - b_start >= per_bucket:
                self.buckets.append((b_start, off))
                b_start = off
        if off > b_start:
            self.buckets.append((b_start, off))
        counts = [0] * len(self.buckets)
        for p in order:
            counts[self._bucket_of[id(p)]] += 1
        self._bucket_size = counts
        # readiness is counted per DISTINCT parameter: a module applied twice in one forward
        # (loc_embedding_encoder in 'as_embedding') reports its gradient twice
        self._ready = [set() for _ in self.buckets]
        self._launched = [False] * len(self.buckets)
        # hold: gradient accumulation -- micro-batches before the last one must not exchange
        # (begin_micro); the buckets then launch from the LAST micro-batch's hooks, or from start()
        self.hold = False
        self._probe = None
        # MSR3D_DP_EXCHANGE: "allreduce" (default: one all-reduce per flush) or "rs_ag" (reduce-scatter
        # + all-gather of the same buffer: the two halves of a ring all-reduce as separate collectives,
        # so that the 8-GPU run can A/B what RCCL does with each over the 7 xGMI links)
        self.exchange_mode = os.environ.get("MSR3D_DP_EXCHANGE", "allreduce")
        if self.exchange_mode not in ("allreduce", "rs_ag"):
            raise ValueError("MSR3D_DP_EXCHANGE must be 'allreduce' or 'rs_ag'")
        # timing (bench.py): HIP events on the communication stream around every exchange, and on the
        # compute stream around the wait for it (= the part of the exchange that was NOT hidden)
        # scale_in_optimizer: leave the SUM in the buffer; the consumer (FlatAdamW, msr3d_adamw_flat_scaled)
        # reads every gradient times 1 / world -- same bits as the separate pass, one 21 MB round trip less.
        # Set by HotPathTrainStep when the optimiser is the fused one; the gradients in the buffer are then
        # sums, not means, after an exchange.
        self.scale_in_optimizer = False
        self.timing = False
        self.comm_events, self.wait_events = [], []
        self.comm_stream = torch.cuda.Stream(device=dev) if self.on_gpu else None
        # defer_comm (default): collectives are issued by finish(), after backward has been
        # enqueued -- required when backward is replayed from a HIP graph, and the safe choice
        # with producers that write the flat buffer directly (hipops).  overlap=True launches
        # each bucket from the gradient hooks as soon as it is complete instead.
        self.defer_comm = not overlap
        if self.distributed:
            for p in self.params:
                p.register_post_accumulate_grad_hook(self._on_grad)
        if sync_params:
            self.broadcast_params()

    # ------------------------------------------------------------------ hooks
    def packed_range(self, group):
        """(start, length) of a pack group inside the flat buffer (asserts it is contiguous)."""
        start = self.offset[id(group[0])]
        pos = start
        for p in group:
            if self.offset[id(p)] != pos:
                raise RuntimeError("parameters of a pack group are not contiguous in the flat buffer")
            pos += p.numel()
        return start, pos - start

    def mark_ready(self, p):
        """A producer wrote p's gradient into the flat buffer directly (bypassing autograd's
        AccumulateGrad, hence its hook): same bookkeeping as the hook."""
        if self._probe is not None:
            self._probe.add(id(p))
        if self.distributed:
            self._on_grad(p)

    def probe_unused(self, run):
        """Run `run()` (one forward + backward) and return the parameters that received NO gradient --
        neither through autograd (post-accumulate hooks) nor from a producer that writes the flat buffer
        directly (mark_ready).  What DDP's find_unused_parameters establishes per step in the reference;
        here once per configuration, for FlatAdamW.set_unused."""
        seen = set()
        self._probe = seen
        handles = [p.register_post_accumulate_grad_hook(lambda q: seen.add(id(q))) for p in self.order]
        try:
            run()
        finally:
            for h in handles:
                h.remove()
            self._probe = None
        return [p for p in self.order if id(p) not in seen]

    def _on_grad(self, p):
        b = self._bucket_of[id(p)]
        self._ready[b].add(id(p))
        if len(self._ready[b]) == self._bucket_size[b] and not self.defer_comm and not self.hold:
            self._launch(b)

    def begin_micro(self, last):
        """Gradient accumulation: call before every micro-batch's backward.  Gradients add up in the
        flat buffer; only the LAST micro-batch may exchange (overlap mode would otherwise send a
        bucket after the first micro-batch and never again: replicas diverge silently)."""
        self._ready = [set() for _ in self.buckets]
        self.hold = not last

    # ------------------------------------------------------------------ replica consistency
    def broadcast_params(self, src=0):
        """Every rank takes rank `src`'s parameter values (one flat broadcast)."""
        if not (dist.is_initialized() and self.world > 1):
            return
        with torch.no_grad():
            flat = torch.cat([p.detach().reshape(-1) for p in self.order])
            dist.broadcast(flat, src=src, group=self.group)
            off = 0
            for p in self.order:
                n = p.numel()
                p.copy_(flat[off:off + n].view_as(p))
                off += n

    def replica_checksum(self, tensor=None):
        """(sum, sum of squares) in float64 of the parameters (or of `tensor`), and the largest
        difference of that pair across ranks -- 0.0 when the replicas hold identical bits' worth of
        values.  Costs one tiny all-reduce pair; used by bench.py and the tests."""
        with torch.no_grad():
            t = tensor if tensor is not None else torch.cat([p.detach().reshape(-1) for p in self.order])
            t = t.double()
            mine = torch.stack([t.sum(), (t * t).sum()])
        if not (dist.is_initialized() and self.world > 1):
            return mine.tolist(), 0.0
        hi, lo = mine.clone(), mine.clone()
        dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=self.group)
        dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=self.group)
        return mine.tolist(), float((hi - lo).abs().max())

    def _reduce(self, view):
        # SUM then scale: works on every backend (gloo has no AVG), one tiny launch
        n = view.numel()
        if self.exchange_mode == "rs_ag" and n % self.world == 0:
            shard = n // self.world
            rank = dist.get_rank(self.group)
            mine = view[rank * shard:(rank + 1) * shard]
            try:
                dist.reduce_scatter_tensor(mine, view, op=dist.ReduceOp.SUM, group=self.group)
            except RuntimeError:                      # a backend without reduce-scatter (gloo)
                self.exchange_mode = "allreduce"
                return self._reduce(view)
            if not self.scale_in_optimizer:
                mine.mul_(1.0 / self.world)
            dist.all_gather_into_tensor(view, mine.clone(), group=self.group)
            return
        dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group)
        if not self.scale_in_optimizer:
            view.mul_(1.0 / self.world)

    def _exchange(self, view):
        if self.on_gpu:
            self.comm_stream.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(self.comm_stream):
                if self.timing and not torch.cuda.is_current_stream_capturing():
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(self.comm_stream)
                    self._reduce(view)
                    e1.record(self.comm_stream)
                    self.comm_events.append((e0, e1))
                else:
                    self._reduce(view)
        else:
            self._reduce(view)

    def _launch(self, b):
        if self._launched[b] or not self.distributed:
            return
        if (self.on_gpu and threading.current_thread() is not threading.main_thread()
                and torch.cuda.is_current_stream_capturing()):
            # A bucket completed inside a backward hook, on an autograd worker thread, while the step is being CAPTURED
            # (MSR3D_DP_GRAPH_COMM=1 with an unfrozen backbone).  torch's ProcessGroupNCCL decides from the calling
            # thread's current stream whether a collective belongs to a capture (and then keeps it away from its
            # watchdog); issued from a worker thread, one run in four ended with the watchdog polling an event "last
            # recorded in a capturing stream" (hipErrorCapturedEvent) and taking the process down.  Such buckets are
            # left to start(), which runs on the thread that owns the capture, in bucket order.
            return
        self._launched[b] = True
        s, e = self.buckets[b]
        self._exchange(self.flat[s:e])

    # ------------------------------------------------------------------ step API
    def reset_marks(self):
        """The bookkeeping half of zero_grad(), for a consumer that cleared the buffer itself (the fused
        optimiser's zero_grad flag)."""
        self._ready = [set() for _ in self.buckets]
        self._launched = [False] * len(self.buckets)

    def zero_grad(self):
        """One memset for every gradient; `.grad` views stay attached."""
        self.flat.zero_()
        self._ready = [set() for _ in self.buckets]
        self._launched = [False] * len(self.buckets)

    def finish(self):
        """Call after backward: flush buckets that never completed (unused params keep their
        zeros) and make the compute stream wait for the exchange."""
        self.start()
        self.wait()

    def wait(self):
        """The compute stream waits for the exchange started by start()."""
        if self.distributed and self.on_gpu:
            cur = torch.cuda.current_stream(self.device)
            if self.timing and not torch.cuda.is_current_stream_capturing():
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(cur)
                cur.wait_stream(self.comm_stream)
                e1.record(cur)
                self.wait_events.append((e0, e1))
            else:
                cur.wait_stream(self.comm_stream)

    def start(self):
        """Launch the exchange on the communication stream and return: work that does not touch
        the gradients (the next batch's frozen encoder) can be enqueued on the compute stream
        before wait()."""
        if self.distributed:
            if self.defer_comm:
                # nothing is in flight and nothing is left to overlap with: ONE collective over the
                # whole buffer (21 MB) instead of one per bucket -- fewer launches, and a ring over
                # point-to-point xGMI links is latency-bound per call at this size
                whole = self.exchange_mode == "rs_ag" and self.flat.numel() % max(self.world, 1) == 0
                self._exchange(self.flat if whole else self.flat[:self.numel])
                self._launched = [True] * len(self.buckets)
            else:
                for b in range(len(self.buckets)):
                    self._launch(b)

    def grad_norm(self):
        return torch.linalg.vector_norm(self.flat)

    def clip_grad_norm_(self, max_norm):
        """Global-norm clipping on the flat buffer (accelerator.clip_grad_norm_,
        leo_trainer.py:192-193): no host sync."""
        norm = self.grad_norm()
        scale = torch.clamp(max_norm / (norm + 1e-6), max=1.0)
        self.flat.mul_(scale)
        return norm
