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


def _gemm_acc(M, N, K, R, P, ldp, Q, ldq, P2, ldp2, Q2, ldq2, C, ldc, scale, dev):
    """C (bf16) += the product (msr3d_bf16_gemm_lowrank_acc); -> False where the shape is outside that entry's domain."""
    with torch.cuda.device(dev):
        rc = _lib.load().msr3d_bf16_gemm_lowrank_acc(M, N, K, R, _p(P), ldp, _p(Q), ldq, _p(P2), ldp2, _p(Q2), ldq2,
                                                     _p(C), ldc, ctypes.c_float(scale), _lib.current_stream_ptr(dev))
    if rc == -22:
        return False
    _lib.check(rc, "msr3d_bf16_gemm_lowrank_acc")
    return True


def _skinny(M, N, K, P, Q, C, ldc, scale, dev):
    """C[:, :N] = scale P Q^T, C[:, N:ldc] = 0 (N = r rows of Q)."""
    with torch.cuda.device(dev):
        rc = _lib.load().msr3d_bf16_gemm_skinny(M, N, K, _p(P), K, _p(Q), K, _p(C), ldc, ldc, ctypes.c_float(scale),
                                                _lib.current_stream_ptr(dev))
    _lib.check(rc, "msr3d_bf16_gemm_skinny")


def _skinny_quant(M, N, K, P, Q, C, ldc, scale, dev):
    """_skinny + the e4m3 image of P with its row scales in the same launch (msr3d_bf16_gemm_skinny_quant)."""
    q = torch.empty((M, K), dtype=torch.uint8, device=dev)
    sc = torch.empty((M,), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = _lib.load().msr3d_bf16_gemm_skinny_quant(M, N, K, _p(P), K, _p(Q), K, _p(C), ldc, ldc, ctypes.c_float(scale),
                                                      _p(q), K, _p(sc), _lib.current_stream_ptr(dev))
    _lib.check(rc, "msr3d_bf16_gemm_skinny_quant")
    return q, sc


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


def _gemm_fp8(M, N, K, Pq, sp, Qq, sq, P2, Q2, C, dev, accumulate=False):
    fn = "msr3d_fp8_gemm_lowrank_acc" if accumulate else "msr3d_fp8_gemm_lowrank"
    with torch.cuda.device(dev):
        rc = getattr(_lib.load(), fn)(M, N, K, _p(Pq), K, _p(sp), _p(Qq), K, _p(sq), _p(P2), PAD_R, _p(Q2), PAD_R,
                                      _p(C), N, _lib.current_stream_ptr(dev))
    _lib.check(rc, fn)


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


_tables = {}


def group_inputs(mods, shared_grad=False):
    """Declare that `mods` (LoRALinear, same in_features / r / scaling / device) always read the SAME input tensor (q / k / v,
    gate / up): their A matrices' bf16 images are stacked into one (n r, K) operand, so ONE r-row product s x [A_1; ..; A_n]^T
    serves all of them (n r <= 64 columns of the shared (M, 64) low-rank activation; member i's B image sits in columns
    i r .. of ITS zero-padded (N, 64) operand, so its GEMM picks out its own slice).  Reads the input once instead of n times.
    shared_grad: the caller GUARANTEES that the members are the only consumers of that input tensor (the decoder layer's
    norm outputs): in backward the first member to run hands autograd its d input and the others ADD theirs into that
    buffer and return nothing.  Without the guarantee (default) every member returns its own d input -- a gradient from
    another consumer arriving between two members would make autograd replace the buffer and the later adds would be lost."""
    m0 = mods[0]
    r, K, dev = m0.r, m0.in_features, m0.lora_A.weight.device
    if len(mods) * r > PAD_R or any(m.r != r or m.in_features != K or m.scaling != m0.scaling or
                                    m.lora_A.weight.device != dev for m in mods):
        raise ValueError("group_inputs: members must share in_features, r, scaling and device, n r <= 64")
    a_cat = torch.empty((len(mods) * r, K), dtype=torch.bfloat16, device=dev)
    grp = {"a_cat": a_cat, "mods": list(mods), "shared_grad": bool(shared_grad)}
    for i, m in enumerate(mods):
        N = m.out_features
        m._shadow = (a_cat[i * r:(i + 1) * r], torch.zeros((N, PAD_R), dtype=torch.bfloat16, device=dev),
                     torch.empty((r, N), dtype=torch.bfloat16, device=dev), torch.zeros((K, PAD_R), dtype=torch.bfloat16, device=dev))
        m._shadow_key = None
        m._group, m._u_col = grp, i * r


def refresh_shadows(mods, capturing=False):
    """bf16 images of the LoRA pairs of `mods` (LoRALinear) in the four orientations the products read, ONE launch for
    all of them (msr3d_lora_shadows) -- once per weight version, i.e. once per optimiser step; inside a graph capture
    always (a replayed optimiser step changes A / B without running this host code again)."""
    stale = [m for m in mods if capturing or m._shadow_key != m._pair_key()]
    if not stale:
        return
    dev = stale[0].lora_A.weight.device
    jobs = []
    for m in stale:
        A, Bw = m.lora_A.weight, m.lora_B.weight
        r, K, N = m.r, m.in_features, m.out_features
        sh = m._shadow
        if sh is None or sh[0].device != A.device:
            m._group, m._u_col = None, 0          # (a module moved to another device leaves its input group)
            # allocated once: the padding columns r..63 of b2 / at2 are zero and stay zero
            sh = m._shadow = (torch.empty((r, K), dtype=torch.bfloat16, device=dev),
                              torch.zeros((N, PAD_R), dtype=torch.bfloat16, device=dev),
                              torch.empty((r, N), dtype=torch.bfloat16, device=dev),
                              torch.zeros((K, PAD_R), dtype=torch.bfloat16, device=dev))
        jobs.append((A.data_ptr(), Bw.data_ptr(), sh[0].data_ptr(), sh[1].data_ptr() + 2 * m._u_col, sh[2].data_ptr(),
                     sh[3].data_ptr(), r, K, N, 0))
    key = tuple(jobs)
    tab = _tables.get(key)
    if tab is None:
        if capturing:
            raise RuntimeError("LoRA shadow table not built before the graph capture (run a warm-up step first)")
        arr = (_lib.LoraShadowJob * len(jobs))(*[_lib.LoraShadowJob(*j) for j in jobs])
        tab = _tables[key] = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
        # NEVER evicted: a captured graph holds the table's ADDRESS as a kernel argument (a cleared cache would hand the
        # block back to the allocator and a replay would read job pointers out of whatever lives there next).  A table is
        # a few hundred bytes per distinct stale set; it is also pinned on every module it serves.
        for m in stale:
            m._shadow_tables = getattr(m, "_shadow_tables", [])
            m._shadow_tables.append(tab)
    with torch.cuda.device(dev):
        rc = _lib.load().msr3d_lora_shadows(len(jobs), _p(tab), _lib.current_stream_ptr(dev))
    _lib.check(rc, "msr3d_lora_shadows")
    for m in stale:
        m._shadow_key = "captured" if capturing else m._pair_key()


class _LoRAFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, lora_A, lora_B, mod):
        K, N, r, s = mod.in_features, mod.out_features, mod.r, mod.scaling
        x2 = x.reshape(-1, K)
        x2 = x2 if x2.is_contiguous() else x2.contiguous()
        M, dev = x2.shape[0], x.device
        # bf16 shadows of the trainable pair in the orientations the products read: built once per weight
        # version (i.e. once per optimiser step), not per call
        if mod._group is not None and not mod._fresh_in_capture:
            # the shared product reads EVERY member's A image: all of them current, not only this module's
            refresh_shadows(mod._group["mods"], x.is_cuda and torch.cuda.is_current_stream_capturing())
        a_pad, b2, _, _ = mod._shadows(forward=True)
        # u = s x A^T  (M, r) bf16 in a zero-padded (M, 64): the r-row product has its own kernel; members of an input
        # group (q / k / v, gate / up) share ONE product over their stacked A's, cached on the input tensor they share
        grp, col = mod._group, mod._u_col
        fp8 = mod.base == "fp8" and M >= 128
        qc = getattr(x, "_msr3d_fp8", None) if fp8 else None
        if qc is not None and qc[2] != x._version:
            qc = None
        uc = getattr(x, "_msr3d_u", None) if grp is not None else None
        if uc is not None and (uc[1] != x._version or uc[2] is not grp):
            uc = None
        if uc is None:
            u = torch.empty((M, PAD_R), dtype=torch.bfloat16, device=dev)
            a_op = grp["a_cat"] if grp is not None else a_pad
            if fp8 and qc is None:
                # the e4m3 image of x (per token row; shared by the projections reading the same tensor) comes out of the
                # same pass over x as the r-row product
                xq, sx = _skinny_quant(M, a_op.shape[0], K, x2, a_op, u, PAD_R, s, dev)
                qc = (xq, sx, x._version)
                try:
                    x._msr3d_fp8 = qc
                except AttributeError:
                    pass
            else:
                _skinny(M, a_op.shape[0], K, x2, a_op, u, PAD_R, s, dev)
            if grp is not None:
                # [3]: the members' shared input-gradient buffer of the backward pass in progress, [4]: members that have
                # run in it, [5]: members that took this record in forward
                uc = [u, x._version, grp, None, 0, 0]
                try:
                    x._msr3d_u = uc
                except AttributeError:
                    uc = None
        else:
            u = uc[0]
        y = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
        if fp8:
            # frozen W as e4m3 with per-output-channel scales (quantised once); the LoRA term rides in bf16 -- csrc/lora_fp8.hip
            xq, sx = (qc[0], qc[1]) if qc is not None else _quant_cached(x, x2)
            _gemm_fp8(M, N, K, xq, sx, mod.weight_q, mod.weight_scale, u, b2, y, dev)
        else:
            _gemm(M, N, K, PAD_R, x2, K, mod.weight, K, u, PAD_R, b2, PAD_R, y, N, False, 1.0, dev)
        ctx.save_for_backward(x2, u, lora_A, lora_B)
        ctx.mod = mod
        ctx.u_col = col if grp is not None else 0
        ctx.rec = uc if (grp is not None and grp.get("shared_grad")) else None
        if ctx.rec is not None:
            ctx.rec[5] += 1
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
        fp8_dx = ctx.needs_input_grad[0] and mod.base == "fp8" and mod.fp8_backward and M >= 128
        if fp8_dx:
            dq, sdy = _skinny_quant(M, r, N, dy2, bt_pad, v, PAD_R, s, dev)     # + the upstream gradient as e4m3, per token row
        else:
            _skinny(M, r, N, dy2, bt_pad, v, PAD_R, s, dev)
        dx = None
        if ctx.needs_input_grad[0]:
            # A member of this input group may already have produced its share of d input in THIS backward: the product is
            # then ADDED into that buffer -- autograd holds it as the input's gradient, and the input's producer runs only
            # after every member -- and this member hands back nothing: no third tensor, no add launch.
            rec = ctx.rec
            added = False
            if rec is not None and rec[3] is not None:
                if fp8_dx:
                    _gemm_fp8(M, K, N, dq, sdy, mod.weight_t_q, mod.weight_t_scale, v, at2, rec[3], dev, accumulate=True)
                    added = True
                else:
                    added = _gemm_acc(M, K, N, PAD_R, dy2, N, mod.weight_t, N, v, PAD_R, at2, PAD_R, rec[3], K, 1.0, dev)
            if not added:
                dx = torch.empty((M, K), dtype=torch.bfloat16, device=dev)
                if fp8_dx:
                    _gemm_fp8(M, K, N, dq, sdy, mod.weight_t_q, mod.weight_t_scale, v, at2, dx, dev)
                else:
                    _gemm(M, K, N, PAD_R, dy2, N, mod.weight_t, N, v, PAD_R, at2, PAD_R, dx, K, False, 1.0, dev)
                if rec is not None and rec[3] is None:
                    rec[3] = dx
                dx = dx.view(ctx.shape)
            if rec is not None:
                rec[4] += 1
                if rec[4] >= rec[5]:           # the last member of THIS backward pass: a second pass over the same graph
                    rec[3], rec[4] = None, 0   # (retain_graph) starts with a buffer of its own
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
            # dA = (s dy B)^T x = v^T x ; dB = dy^T (s x A^T) = dy^T u   (s already inside u and v): both in ONE launch,
            # a workgroup per 64 output columns over all tokens -- no partial sums, no workspace
            jobs = (_lib.LoraGradJob * 2)(
                _lib.LoraGradJob(K, v.data_ptr(), PAD_R, x2.data_ptr(), K, dA.data_ptr(), 0),
                _lib.LoraGradJob(N, u.data_ptr() + 2 * ctx.u_col, PAD_R, dy2.data_ptr(), N, dB.data_ptr(), 1))
            rc = lib.msr3d_lora_grad_pair(M, r, 2, jobs, ctypes.c_float(1.0), acc, st)
            _lib.check(rc, "msr3d_lora_grad_pair")
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
        self._shadow, self._shadow_key = None, None
        self._fresh_in_capture = False    # set by LoRALlamaStack around a captured forward (its one refresh launch)
        self._group, self._u_col = None, 0    # group_inputs(): modules reading the same tensor share one r-row product

    def _shadows(self, forward=False):
        """Zero-padded bf16 copies of A / B in the four orientations forward and backward read
        (a_pad (r, K), b2 (N, 64), bt_pad (r, N), at2 (K, 64)); rebuilt when A or B has been written."""
        A, Bw = self.lora_A.weight, self.lora_B.weight
        key = self._pair_key()
        # Inside a graph capture the copies are ALWAYS rebuilt (into the same storage): a replayed optimiser step
        # changes A / B without running this host code again, so the rebuild has to be part of the graph.
        capturing = forward and A.is_cuda and torch.cuda.is_current_stream_capturing()     # (backward reuses forward's)
        if getattr(self, "_shadow_key", None) == "captured" and not forward and A.is_cuda and \
                torch.cuda.is_current_stream_capturing():
            return self._shadow                 # backward of the captured step: forward's copies
        if capturing and self._fresh_in_capture:
            return self._shadow                 # the enclosing stack rebuilt every pair at the top of this capture
        if capturing or getattr(self, "_shadow_key", None) != key:
            refresh_shadows([self], capturing)        # (a stack refreshes all of its pairs in one launch before this)
        return self._shadow

    def _pair_key(self):
        A, Bw = self.lora_A.weight, self.lora_B.weight
        return (A._version, Bw._version, A.data_ptr(), Bw.data_ptr())

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

    def forward2d(self, x):
        """-> y (tokens, out_features): the autograd function's own output, NOT a view of it -- what a caller that goes
        on IN PLACE (RoPE) must take: an in-place op on a view of a custom function's output makes autograd rebase the
        view (CopySlices: two full copies per op in backward)."""
        if not x.is_cuda:
            raise RuntimeError("LoRALinear runs on the GPU only (no CPU fallback)")
        self._sync_weight_t()
        return _LoRAFn.apply(x.to(torch.bfloat16), self.lora_A.weight, self.lora_B.weight, self)

    def forward(self, x):
        return self.forward2d(x).view(*x.shape[:-1], self.out_features)
