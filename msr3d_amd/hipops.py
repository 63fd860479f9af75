"""Autograd-level ops of the trainable part, over the C ABI (include/msr3d_hip.h).

`linear` is nn.Linear's functional form on the HIP f32-MFMA GEMM (msr3d_gemm_f32): forward
y = x W^T + b [+ GELU] [+ dropout], backward dx = dy W, dW = dy^T x, db = colsum(dy) -- the same
kernel body reading x / W / dy in place; dx and dW+db share one launch when the gradients live in
the flat buffer of the data-parallel engine.  GPU fp32 tensors take the HIP kernels; CPU
tensors (the host-logic unit tests) take torch's own ops.  There is no silent GPU fallback:
a GPU tensor either runs the HIP kernel or raises.
"""
import ctypes

import torch
import torch.nn.functional as F

from . import _lib


import os as _os
_NO_DIRECT = bool(int(_os.environ.get("MSR3D_NO_DIRECT_GRAD", "0")))   # debugging aid


def _p(t):
    return ctypes.c_void_p(t.data_ptr() if t is not None else 0)


# ---- split-K workspace (include/msr3d_hip.h, msr3d_gemm_f32) ---------------------------------
# MSR3D_GEMM_DETERMINISTIC=1 (or set_deterministic(True)) selects the ordered split-K meeting point:
# bit-reproducible GEMMs at ~5 % step time; default is the atomic one.
# One persistent buffer per (device, lane).  Launches that share a lane must be ordered on one
# stream; work issued concurrently on a second stream (the encoder prefetch of train_step.py)
# selects another lane with `gemm_lane(1)`.  Allocated (and zeroed, once) at first use -- before
# any HIP-graph capture, since the eager warm-up steps run the same GEMMs.
GEMM_WS_BYTES = 16 << 20
_ws = {}
_lane = [0]


class gemm_lane:
    def __init__(self, lane):
        self.lane = lane

    def __enter__(self):
        self.prev = _lane[0]
        _lane[0] = self.lane

    def __exit__(self, *a):
        _lane[0] = self.prev


def _workspace(dev):
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), _lane[0])
    ws = _ws.get(key)
    if ws is None:
        ws = _ws[key] = torch.zeros(GEMM_WS_BYTES // 4, dtype=torch.int32, device=dev)
    return ws


_deterministic = [_os.environ.get("MSR3D_GEMM_DETERMINISTIC", "0") == "1"
                  or _os.environ.get("MSR3D_DETERMINISTIC", "0") == "1"]


def set_deterministic(on):
    """Bit-reproducible reductions from now on: ordered split-K in every GEMM, LayerNorm parameter
    gradients and bias column sums summed in a fixed order (instead of meeting by float atomics).
    Same seed + same data => identical weights, run after run; ~5 % slower.  Returns the previous
    setting."""
    was, _deterministic[0] = _deterministic[0], bool(on)
    return was


LN_BWD_ROWS = 16     # MSR3D_LN_BWD_ROWS


def ln_partials(M, D, n_arrays, device):
    """Workspace for the ordered LayerNorm-gradient reduction (None in the default atomic mode)."""
    if not _deterministic[0]:
        return None
    return torch.empty(((M + LN_BWD_ROWS - 1) // LN_BWD_ROWS) * n_arrays * D, dtype=torch.float32,
                       device=device)


def _ws_args(dev, force=False):
    if not (_deterministic[0] or force):
        return ctypes.c_void_p(0), ctypes.c_size_t(0)
    ws = _workspace(dev)
    return ctypes.c_void_p(ws.data_ptr()), ctypes.c_size_t(GEMM_WS_BYTES)


def _gemm(a_kc, b_kc, M, N, K, A, lda, B, ldb, C, ldc, bias=None, c_pre=None, flags=0, beta=0.0,
          p_drop=0.0, salt=0, ordered=False):
    """p_drop > 0: epilogue dropout keyed by (seed word of the device, salt).  ordered: K-splits meet in
    a fixed order through the workspace (bit-reproducible, every output row independent of the
    others) whatever the global deterministic switch says."""
    lib = _lib.load()
    dev = C.device
    wp, wb = _ws_args(dev, ordered)
    seed = seed_word(dev) if p_drop > 0 else None
    with torch.cuda.device(dev):
        rc = lib.msr3d_gemm_f32(int(a_kc), int(b_kc), M, N, K, _p(A), lda, _p(B), ldb, _p(C), ldc,
                                _p(bias), _p(c_pre), flags, ctypes.c_float(beta),
                                ctypes.c_float(p_drop), _p(seed), salt, wp, wb,
                                _lib.current_stream_ptr(dev))
    _lib.check(rc, "msr3d_gemm_f32")


# Tall token GEMMs (an unfrozen backbone's SharedMLP rows: >= 8192 rows) go to
# msr3d_rows_gemm_split: bf16 pipe at fp32 accuracy, HBM-bound.  MSR3D_ROWS_GEMM=f32 keeps them on the fp32 pipe.
_ROWS_SPLIT = _os.environ.get("MSR3D_ROWS_GEMM", "split") != "f32"


def _rows_split_ok(M, N, K, A, C):
    # K <= 256: the weight stays in LDS for the whole launch.  Wider reductions re-fill LDS per 128-wide super-slab
    # and row block: right for the last level's short, wide layers (15 k rows), not for half a million rows.
    return (_ROWS_SPLIT and M >= 8192 and K % 4 == 0 and N % 4 == 0 and N <= 1024
            and (K <= 256 or (K <= 1024 and M <= 65536))
            and A.data_ptr() % 16 == 0 and C.data_ptr() % 16 == 0)


ROWS_GEMM_BLOCK = 256      # MSR3D_ROWS_GEMM_BLOCK


def _rows_gemm(M, N, K, A, lda, B, ldb, b_trans, C, ldc, stats=None, a_bn=None):
    """C (M, N) = A (M, K) op(B)^T, op(B) = B (N, K) or, b_trans, B (K, N)^T.  stats: (ceil(M / 256), 2, N) floats that
    receive C's per-block column sums and sums of squares.  a_bn (4, K) = [gamma | beta | mean | rstd]: the product
    is taken of relu(batch_norm(A)), formed on load."""
    with torch.cuda.device(C.device):
        rc = _lib.load().msr3d_rows_gemm_split(M, N, K, _p(A), lda, _p(B), ldb, int(b_trans), _p(C), ldc, _p(stats),
                                               _p(a_bn), _lib.current_stream_ptr(C.device))
    _lib.check(rc, "msr3d_rows_gemm_split")


_wgrad_ws = {}


def _wgrad_rows_ok(M, N, K):
    # (any width: the chunk count adapts -- 256 / tiles chunks of n_out x k_in floats never exceed 256 x 128 x 128)
    return _ROWS_SPLIT and M >= 8192


def _wgrad_rows(M, N, K, dy, x, dw, accumulate=False, x_bn=None):
    """dw (N, K) (+)= dy (M, N)^T x (M, K) over tall row counts (msr3d_wgrad_rows_split).  The chunk workspace (33 MB:
    256 / tiles chunks of n_out x k_in floats never exceed it) is one buffer per (device, stream) -- two tall
    weight-gradient launches on different streams must not share partial slabs (as lora._grad_workspace and `dot`'s
    scratch).  A capture stream gets its own from the graph's pool, held here for the replays."""
    dev = dy.device
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
    ws = _wgrad_ws.get(key)
    if ws is None:
        ws = _wgrad_ws[key] = torch.empty(256 * 256 * 128, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = _lib.load().msr3d_wgrad_rows_split(M, N, K, _p(dy), dy.stride(0), _p(x), x.stride(0), _p(dw), dw.stride(0),
                                                int(accumulate), _p(ws), ws.numel(), _p(x_bn),
                                                _lib.current_stream_ptr(dev))
    _lib.check(rc, "msr3d_wgrad_rows_split")


def _linear_bwd(M, N, K, dy, x, w, dx, dx_beta, dw, db):
    """dx (M,K) = dx_beta*dx + dy (M,N) @ w (N,K);  dw += dy^T x;  db += colsum(dy): one launch."""
    lib = _lib.load()
    wp, wb = _ws_args(dy.device)
    with torch.cuda.device(dy.device):
        rc = lib.msr3d_linear_bwd_f32(M, N, K, _p(dy), _p(x), _p(w), _p(dx), ctypes.c_float(dx_beta),
                                      _p(dw), _p(db), wp, wb, _lib.current_stream_ptr(dy.device))
    _lib.check(rc, "msr3d_linear_bwd_f32")


def _gelu_bwd(dy2, pre, p_drop=0.0, salt=0):
    lib = _lib.load()
    g = torch.empty_like(dy2)
    seed = seed_word(dy2.device) if p_drop > 0 else None
    with torch.cuda.device(dy2.device):
        rc = lib.msr3d_gelu_bwd_f32(dy2.numel(), _p(dy2), _p(pre), _p(g), ctypes.c_float(p_drop),
                                    _p(seed), salt, _lib.current_stream_ptr(dy2.device))
    _lib.check(rc, "msr3d_gelu_bwd_f32")
    return g


def _colsum(X, M, N, out, accumulate=False):
    lib = _lib.load()
    with torch.cuda.device(X.device):
        rc = lib.msr3d_colsum_f32(M, N, _p(X), N, _p(out), int(accumulate) | (2 if _deterministic[0] else 0),
                                  _lib.current_stream_ptr(X.device))
    _lib.check(rc, "msr3d_colsum_f32")


class _HipLinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, gelu, stats_out=None):
        x2 = x.reshape(-1, x.shape[-1])
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        w = weight if weight.is_contiguous() else weight.contiguous()
        M, K = x2.shape
        N = w.shape[0]
        y = torch.empty((M, N), dtype=torch.float32, device=x.device)
        pre = torch.empty_like(y) if gelu else None
        if bias is None and not gelu and _rows_split_ok(M, N, K, x2, y):
            stats = None
            if stats_out is not None:      # the BatchNorm that follows takes its first-stage statistics from here
                stats = torch.empty((-(-M // ROWS_GEMM_BLOCK), 2, N), dtype=torch.float32, device=x.device)
                stats_out.append(stats)
            _rows_gemm(M, N, K, x2, K, w, K, False, y, N, stats)
        else:
            _gemm(True, True, M, N, K, x2, K, w, K, y, N, bias=bias, c_pre=pre, flags=1 if gelu else 0)
        ctx.save_for_backward(x2, w, pre)
        ctx.has_bias = bias is not None
        ctx.x_shape = x.shape
        # leaf parameters whose .grad lives in the data-parallel engine's flat buffer
        dpw = getattr(weight, "_msr3d_dp", None)
        ok = dpw is not None and weight.is_leaf and weight.grad is not None and weight.is_contiguous()
        if ok and bias is not None:
            ok = getattr(bias, "_msr3d_dp", None) is dpw and bias.is_leaf and bias.grad is not None
        ctx.direct = (dpw, weight, bias) if (ok and not _NO_DIRECT) else None
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
            dy2 = _gelu_bwd(dy2, pre)
        dx = dw = db = None
        direct = ctx.direct is not None and ctx.needs_input_grad[1]
        if direct and ctx.needs_input_grad[0]:
            # dx, and dW (+ db) straight into the flat gradient buffer's views: ONE launch
            dpw, wparam, bparam = ctx.direct
            dx = torch.empty((M, K), dtype=torch.float32, device=dy.device)
            _linear_bwd(M, N, K, dy2, x2, w, dx, 0.0, wparam.grad,
                        bparam.grad if bparam is not None else None)
            dpw.mark_ready(wparam)
            if bparam is not None:
                dpw.mark_ready(bparam)
            return dx.reshape(ctx.x_shape), None, None, None, None
        if ctx.needs_input_grad[0]:
            dx = torch.empty((M, K), dtype=torch.float32, device=dy.device)
            if _rows_split_ok(M, K, N, dy2, dx):
                _rows_gemm(M, K, N, dy2, N, w, K, True, dx, K)            # dx = dy @ W: op(B)[k][n] = W[n][k]
            elif M >= 8192 and K >= 64 and 16 <= N <= 256 and N % 16 == 0:
                # tall and skinny (the unfrozen backbone's SharedMLP layers, up to 983 k rows): against a
                # transposed copy of the small weight this is a forward-shaped product with a short
                # reduction, which the A-resident kernel takes (one strip of dy in LDS, no per-slab barrier)
                wt = w.t().contiguous()
                _gemm(True, True, M, K, N, dy2, N, wt, N, dx, K)
            else:
                _gemm(True, False, M, K, N, dy2, N, w, K, dx, K)          # dx = dy @ W
            dx = dx.reshape(ctx.x_shape)
        want_db = ctx.has_bias and ctx.needs_input_grad[2]
        if direct:
            # accumulate dW (+ db) straight into the flat gradient buffer's views
            dpw, wparam, bparam = ctx.direct
            lib = _lib.load()
            wp, wb = _ws_args(dy.device)
            with torch.cuda.device(dy.device):
                rc = lib.msr3d_linear_wgrad_acc_f32(M, N, K, _p(dy2), _p(x2), _p(wparam.grad),
                                                    _p(bparam.grad if bparam is not None else None),
                                                    wp, wb, _lib.current_stream_ptr(dy.device))
            _lib.check(rc, "msr3d_linear_wgrad_acc_f32")
            dpw.mark_ready(wparam)
            if bparam is not None:
                dpw.mark_ready(bparam)
            return dx, None, None, None, None
        if ctx.needs_input_grad[1] and want_db:
            # dW = dy^T @ x and db = colsum(dy) from ONE launch into one buffer
            buf = torch.empty((N * K + N,), dtype=torch.float32, device=dy.device)
            lib = _lib.load()
            wp, wb = _ws_args(dy.device)
            with torch.cuda.device(dy.device):
                rc = lib.msr3d_linear_wgrad_f32(M, N, K, _p(dy2), _p(x2), _p(buf), wp, wb,
                                                _lib.current_stream_ptr(dy.device))
            _lib.check(rc, "msr3d_linear_wgrad_f32")
            dw, db = buf[:N * K].view(N, K), buf[N * K:]
        else:
            if ctx.needs_input_grad[1]:
                dw = torch.empty((N, K), dtype=torch.float32, device=dy.device)
                if _wgrad_rows_ok(M, N, K):
                    _wgrad_rows(M, N, K, dy2, x2, dw)
                else:
                    _gemm(False, False, N, K, M, dy2, N, x2, K, dw, K)    # dW = dy^T @ x
            if want_db:
                db = torch.empty((N,), dtype=torch.float32, device=dy.device)
                _colsum(dy2, M, N, db)
        return dx, dw, db, None, None


def linear(x, weight, bias=None, gelu=False, stats_out=None):
    """F.linear (optionally followed by exact GELU) on the HIP GEMM for GPU fp32 tensors.  stats_out: a list; when
    the product runs on the tall-rows kernel, the per-block column statistics of y are appended to it."""
    if x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32:
        # (the path is fp32; other dtypes -- e.g. the float64 references of the tests -- are not
        # part of it and use torch's own GEMM)
        if gelu and weight.shape[0] % 4 != 0:
            raise RuntimeError("fused GELU needs an output width that is a multiple of 4")
        return _HipLinear.apply(x, weight, bias, gelu, stats_out)
    y = F.linear(x, weight, bias)
    return F.gelu(y) if gelu else y


def pack_split_weight(w):
    """(N, K) fp32 weight -> the PACK operand of the split-MFMA kernels (include/msr3d_hip.h, msr3d_split_pack's layout,
    transposed = 0): three exact bf16 terms, [K / 32 slabs][N / 16 tiles][3 planes][64 lanes][8] -- int16 tensor.
    For weights that do not change between steps (a frozen `fc`); trainable ones go through msr3d_split_pack."""
    n, k = w.shape
    if n % 16 or k % 32:
        raise ValueError("pack_split_weight: N % 16 == 0 and K % 32 == 0")
    wf = w.detach().float()
    w0 = wf.to(torch.bfloat16)
    r1 = wf - w0.float()
    w1 = r1.to(torch.bfloat16)
    w2 = (r1 - w1.float()).to(torch.bfloat16)
    planes = torch.stack([w0, w1, w2])                                              # (3, n, k)
    frag = planes.view(3, n // 16, 16, k // 32, 4, 8).permute(3, 1, 0, 4, 2, 5)      # (s, t, p, g, i, j)
    return frag.contiguous().reshape(-1).view(torch.int16)


def rows_linear_split_ok(m, n, k):
    return n % 32 == 0 and k % 128 == 0


def rows_linear_split(x, pack, n_out, bias=None, out=None):
    """y (M, n_out) = x (M, K) . W^T + bias with W as pack_split_weight / msr3d_split_pack left it (msr3d_rows_linear_split:
    fp32 accuracy on the bf16 matrix pipe, no K split -- row-independent, bit-reproducible).  Forward only."""
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1):
        raise ValueError("rows_linear_split: a 2-D fp32 GPU tensor with unit column stride")
    M, K = x.shape
    y = out if out is not None else torch.empty((M, n_out), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        rc = _lib.load().msr3d_rows_linear_split(M, n_out, K, _p(x), x.stride(0), _p(pack), pack.numel() * 2, _p(bias),
                                                 _p(y), y.stride(0), _lib.current_stream_ptr(x.device))
    _lib.check(rc, "msr3d_rows_linear_split")
    return y


class _HipLinearPacked(torch.autograd.Function):
    """y = x Wp^T + bp where Wp / bp are VIEWS of the flat parameter buffer spanning several
    nn.Linear modules laid out back to back (q | k | v | cond projections), and their gradient
    views in the flat gradient buffer.  One forward GEMM, one dx GEMM, one dW+db launch
    accumulating straight into the flat gradients; no cat, no split, no AccumulateGrad adds.
    `members` (the real Parameters) are passed only so that autograd schedules this node."""

    @staticmethod
    def forward(ctx, x, packed, *members):
        wv, bv, gwv, gbv, dp = packed
        x2 = x.reshape(-1, x.shape[-1])
        x2 = x2 if x2.is_contiguous() else x2.contiguous()
        M, K = x2.shape
        N = wv.shape[0]
        y = torch.empty((M, N), dtype=torch.float32, device=x.device)
        _gemm(True, True, M, N, K, x2, K, wv, K, y, N, bias=bv)
        ctx.save_for_backward(x2)
        ctx.packed = packed
        ctx.members = members
        ctx.x_shape = x.shape
        return y.reshape(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        (x2,) = ctx.saved_tensors
        wv, bv, gwv, gbv, dp = ctx.packed
        M, K = x2.shape
        N = wv.shape[0]
        dy2 = dy.reshape(M, N)
        dy2 = dy2 if dy2.is_contiguous() else dy2.contiguous()
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty((M, K), dtype=torch.float32, device=dy.device)
            _linear_bwd(M, N, K, dy2, x2, wv, dx, 0.0, gwv, gbv)
            dx = dx.reshape(ctx.x_shape)
        else:
            lib = _lib.load()
            wp, wb = _ws_args(dy.device)
            with torch.cuda.device(dy.device):
                rc = lib.msr3d_linear_wgrad_acc_f32(M, N, K, _p(dy2), _p(x2), _p(gwv), _p(gbv), wp, wb,
                                                    _lib.current_stream_ptr(dy.device))
            _lib.check(rc, "msr3d_linear_wgrad_acc_f32")
        for p in ctx.members:
            dp.mark_ready(p)
        return (dx, None) + (None,) * len(ctx.members)


def linear_packed(x, packed, members):
    return _HipLinearPacked.apply(x, packed, *members)


def collect_pack_groups(model):
    """Pack groups requested by the model's modules (see FlatGradAllReduce pack_groups)."""
    groups = []
    for m in model.modules():
        if hasattr(m, "pack_groups"):
            groups.extend(m.pack_groups())
    return groups


def attach_packed_views(model, dp, opt):
    """After the data-parallel engine and the flat optimiser own the storage: hand every
    module that asked for packing its (weight, bias, grad-weight, grad-bias) views."""
    flat_p = getattr(opt, "flat_p", None)
    if flat_p is None or _NO_DIRECT:
        return 0
    n = 0
    for m in model.modules():
        if hasattr(m, "pack_groups") and hasattr(m, "set_packed"):
            wg, bg = m.pack_groups()
            (ws, wl), (bs, bl) = dp.packed_range(wg), dp.packed_range(bg)
            K = wg[0].shape[1]
            m.set_packed((flat_p[ws:ws + wl].view(-1, K), flat_p[bs:bs + bl],
                          dp.flat[ws:ws + wl].view(-1, K), dp.flat[bs:bs + bl], dp), wg + bg)
            n += 1
    if n and hasattr(model, "visual_prompter") and hasattr(model, "llm_proj"):
        from . import fused_model
        fused_model.attach(model, dp)         # MSR3DHotPath: the whole trainable part as one schedule
    return n


def module_linear(mod, x, gelu=False):
    """Apply an nn.Linear module through `linear` (its parameters stay where they are)."""
    return linear(x, mod.weight, mod.bias, gelu=gelu)


# ---------------------------------------------------------------------------------------
# SharedMLP in TRAINING mode (unfrozen backbone): conv1x1 -> BatchNorm
# (batch statistics) -> ReLU per layer, then the max over the neighbourhood.  The convolutions are
# the token GEMMs above on a token-major copy of the grouped tensor; the normalisation is
# csrc/bn_train.hip (ordered two-stage statistics: bit-reproducible).
# ---------------------------------------------------------------------------------------
BN_CHUNK_ROWS = 512        # MSR3D_BN_CHUNK_ROWS


def _direct_targets(*params):
    """(engine, params) when every one of `params` is a leaf owned by ONE data-parallel engine with its gradient view in
    place: their producers then add into those views themselves and report through mark_ready -- no AccumulateGrad
    launch per parameter."""
    if _NO_DIRECT:
        return None
    dp = getattr(params[0], "_msr3d_dp", None)
    if dp is None:
        return None
    for q in params:
        if getattr(q, "_msr3d_dp", None) is not dp or not q.is_leaf or q.grad is None or not q.grad.is_contiguous():
            return None
    return dp, params


def _direct_ptrs(direct):
    """The two accumulate pointers of the BatchNorm backward entries (gamma.grad, beta.grad) or NULLs."""
    if direct is None:
        return ctypes.c_void_p(0), ctypes.c_void_p(0)
    return _p(direct[1][0].grad), _p(direct[1][1].grad)


def _direct_done(direct, n=None):
    dp, params = direct
    for q in params[:n]:
        dp.mark_ready(q)


class _BNReLUTrain(torch.autograd.Function):
    """y = relu(batch_norm(x)) over the rows of x (R, C), training mode; updates the module's
    running statistics like nn.BatchNorm2d.forward does."""

    @staticmethod
    def forward(ctx, x, gamma, beta, bn, partials=None):
        R, C = x.shape
        momentum, rm, rv = _bn_train_args(bn, x)
        x = x if x.is_contiguous() else x.contiguous()
        g, b = gamma.contiguous(), beta.contiguous()
        y = torch.empty_like(x)
        mean = torch.empty(C, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        ws, chunks = _bn_partials(partials, R, C, x.device)
        with torch.cuda.device(x.device):
            rc = _lib.load().msr3d_bn_relu_train_fwd(
                R, C, _p(x), _p(g), _p(b), float(bn.eps), momentum, _p(rm), _p(rv), _p(y), _p(mean),
                _p(rstd), _p(ws), chunks, _lib.current_stream_ptr(x.device))
        _lib.check(rc, "msr3d_bn_relu_train_fwd")
        ctx.save_for_backward(x, g, b, mean, rstd)
        ctx.direct = _direct_targets(gamma, beta)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, g, b, mean, rstd = ctx.saved_tensors
        R, C = x.shape
        dy = dy if dy.is_contiguous() else dy.contiguous()
        dx = torch.empty_like(x)
        dg = torch.empty(C, dtype=torch.float32, device=x.device)
        db = torch.empty_like(dg)
        ws = torch.empty(2 * C * max(1, -(-R // BN_CHUNK_ROWS)), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            rc = _lib.load().msr3d_bn_relu_train_bwd(
                R, C, _p(x), _p(dy), _p(g), _p(b), _p(mean), _p(rstd), _p(dx), _p(dg), _p(db), _p(ws),
                *_direct_ptrs(ctx.direct), _lib.current_stream_ptr(x.device))
        _lib.check(rc, "msr3d_bn_relu_train_bwd")
        if ctx.direct is not None:
            _direct_done(ctx.direct)
            return dx, None, None, None, None
        return dx, dg, db, None, None


def _bn_partials(partials, R, C, device):
    """(workspace, partial_chunks) for the forward statistics: the producer's per-block partials (the tall-rows GEMM's
    col_stats) if it left any, else scratch for the kernel's own first stage."""
    if partials is not None:
        assert partials.shape == (-(-R // ROWS_GEMM_BLOCK), 2, C) and partials.is_contiguous()
        return partials, partials.shape[0]
    return torch.empty(2 * C * max(1, -(-R // BN_CHUNK_ROWS)), dtype=torch.float32, device=device), 0


def _bn_train_args(bn, x):
    """(momentum, running_mean, running_var) for the kernels; bumps num_batches_tracked like
    nn.BatchNorm2d.forward."""
    if x.shape[0] <= 1:          # as torch.nn.functional.batch_norm in training mode
        raise ValueError(f"Expected more than 1 value per channel when training, got input size {x.shape}")
    if not (bn.track_running_stats and bn.running_mean is not None):
        return 0.0, None, None
    with torch.no_grad():
        bn.num_batches_tracked += 1
    # momentum None = cumulative average; the factor then depends on a device counter the host would
    # have to read -- nn.BatchNorm2d's default (0.1) is what the backbone uses
    if bn.momentum is None:
        raise NotImplementedError("cumulative-average BatchNorm is not on this path")
    return float(bn.momentum), bn.running_mean, bn.running_var


class _BNReLUMaxPoolTrain(torch.autograd.Function):
    """pooled (R / ns, C) = max over each group of ns consecutive rows of relu(batch_norm(x)): the last
    SharedMLP layer fused with the neighbourhood max-pool (first maximum wins, as F.max_pool2d);
    neither the (R, C) activation nor its gradient is materialised."""

    @staticmethod
    def forward(ctx, x, gamma, beta, bn, ns, partials=None):
        R, C = x.shape
        momentum, rm, rv = _bn_train_args(bn, x)
        x = x if x.is_contiguous() else x.contiguous()
        g, b = gamma.contiguous(), beta.contiguous()
        pooled = torch.empty((R // ns, C), dtype=torch.float32, device=x.device)
        arg = torch.empty((R // ns, C), dtype=torch.int32, device=x.device)
        xsel = torch.empty_like(pooled)
        mean = torch.empty(C, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        ws, chunks = _bn_partials(partials, R, C, x.device)
        with torch.cuda.device(x.device):
            rc = _lib.load().msr3d_bn_relu_maxpool_train_fwd(
                R, C, ns, _p(x), _p(g), _p(b), float(bn.eps), momentum, _p(rm), _p(rv), _p(pooled), _p(arg),
                _p(xsel), _p(mean), _p(rstd), _p(ws), chunks, _lib.current_stream_ptr(x.device))
        _lib.check(rc, "msr3d_bn_relu_maxpool_train_fwd")
        ctx.save_for_backward(x, g, mean, rstd, pooled, arg, xsel)
        ctx.direct = _direct_targets(gamma, beta)
        ctx.ns = ns
        return pooled

    @staticmethod
    def backward(ctx, dpooled):
        x, g, mean, rstd, pooled, arg, xsel = ctx.saved_tensors
        R, C = x.shape
        dpooled = dpooled if dpooled.is_contiguous() else dpooled.contiguous()
        dx = torch.empty_like(x)
        dg = torch.empty(C, dtype=torch.float32, device=x.device)
        db = torch.empty_like(dg)
        ws = torch.empty(2 * C * max(1, -(-R // BN_CHUNK_ROWS)), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            rc = _lib.load().msr3d_bn_relu_maxpool_train_bwd(
                R, C, ctx.ns, _p(x), _p(dpooled), _p(pooled), _p(arg), _p(xsel), _p(g), _p(mean), _p(rstd), _p(dx),
                _p(dg), _p(db), _p(ws), *_direct_ptrs(ctx.direct), _lib.current_stream_ptr(x.device))
        _lib.check(rc, "msr3d_bn_relu_maxpool_train_bwd")
        if ctx.direct is not None:
            _direct_done(ctx.direct)
            return dx, None, None, None, None, None
        return dx, dg, db, None, None, None


class _BNReLULinear(torch.autograd.Function):
    """z_next (R, N) = relu(batch_norm(z)) W^T for a tall z whose first-stage statistics the producing GEMM left in
    `partials`: the statistics are finalised (running statistics updated), and the normalisation + ReLU is applied to
    the operand on its way into the matrix pipe -- forward here, and again in the backward's weight gradient -- so the
    (R, C) activation is neither written nor re-read.  Backward: dy = dz_next W, dW = dz_next^T relu(bn(z)), then the
    BatchNorm + ReLU backward of msr3d_bn_relu_train_bwd on (z, dy)."""

    @staticmethod
    def forward(ctx, z, gamma, beta, bn, partials, weight, stats_out):
        R, C = z.shape
        N = weight.shape[0]
        momentum, rm, rv = _bn_train_args(bn, z)
        w = (weight if weight.is_contiguous() else weight.contiguous()).view(N, C)    # (the conv parameter: (N, C, 1, 1))
        ctx.direct_bn = _direct_targets(gamma, beta)
        ctx.direct_w = _direct_targets(weight)
        pro = torch.empty((4, C), dtype=torch.float32, device=z.device)     # [gamma | beta | mean | rstd]
        g, b = gamma.contiguous(), beta.contiguous()
        with torch.cuda.device(z.device):
            rc = _lib.load().msr3d_bn_train_stats(R, C, _p(partials), partials.shape[0], float(bn.eps), momentum, _p(rm),
                                                  _p(rv), _p(g), _p(b), _p(pro), _lib.current_stream_ptr(z.device))
        _lib.check(rc, "msr3d_bn_train_stats")
        y = torch.empty((R, N), dtype=torch.float32, device=z.device)
        stats = torch.empty((-(-R // ROWS_GEMM_BLOCK), 2, N), dtype=torch.float32, device=z.device)
        stats_out.append(stats)
        _rows_gemm(R, N, C, z, C, w, C, False, y, N, stats, pro)
        ctx.save_for_backward(z, pro, w)
        ctx.w_shape = tuple(weight.shape)
        return y

    @staticmethod
    def backward(ctx, dzn):
        z, pro, w = ctx.saved_tensors
        R, C = z.shape
        N = w.shape[0]
        dzn = dzn if dzn.is_contiguous() else dzn.contiguous()
        dw = None
        if ctx.needs_input_grad[5]:
            if ctx.direct_w is not None:         # added straight into the parameter's view of the flat gradients
                _wgrad_rows(R, N, C, dzn, z, ctx.direct_w[1][0].grad.view(N, C), accumulate=True, x_bn=pro)
                _direct_done(ctx.direct_w)
            else:
                dw = torch.empty((N, C), dtype=torch.float32, device=z.device).view(ctx.w_shape)
                _wgrad_rows(R, N, C, dzn, z, dw.view(N, C), x_bn=pro)
        dy = torch.empty((R, C), dtype=torch.float32, device=z.device)
        _rows_gemm(R, C, N, dzn, N, w, C, True, dy, C)
        dz = torch.empty_like(z)
        dg = torch.empty(C, dtype=torch.float32, device=z.device)
        db = torch.empty_like(dg)
        ws = torch.empty(2 * C * max(1, -(-R // BN_CHUNK_ROWS)), dtype=torch.float32, device=z.device)
        with torch.cuda.device(z.device):
            rc = _lib.load().msr3d_bn_relu_train_bwd(R, C, _p(z), _p(dy), _p(pro[0]), _p(pro[1]), _p(pro[2]), _p(pro[3]),
                                                     _p(dz), _p(dg), _p(db), _p(ws), *_direct_ptrs(ctx.direct_bn),
                                                     _lib.current_stream_ptr(z.device))
        _lib.check(rc, "msr3d_bn_relu_train_bwd")
        if ctx.direct_bn is not None:
            _direct_done(ctx.direct_bn)
            dg = db = None
        return dz, dg, db, None, None, dw, None


class _RowsLinear(torch.autograd.Function):
    """z (R, N) = t (R, KP) W^T for the FIRST layer of a SharedMLP on tall grouped rows: t is zero-padded from the
    weight's K to KP columns (whole 16-wide slabs) and the kernels read the (N, K) weight as it lies -- no padded copy,
    its gradient added straight into the parameter's view of the flat gradients when there is one."""

    @staticmethod
    def forward(ctx, t, weight, stats_out):
        R, KP = t.shape
        N = weight.shape[0]
        w = (weight if weight.is_contiguous() else weight.contiguous()).view(N, -1)
        K = w.shape[1]
        z = torch.empty((R, N), dtype=torch.float32, device=t.device)
        stats = torch.empty((-(-R // ROWS_GEMM_BLOCK), 2, N), dtype=torch.float32, device=t.device)
        stats_out.append(stats)
        _rows_gemm(R, N, KP, t, KP, w, K, False, z, N, stats)
        ctx.save_for_backward(t, w)
        ctx.direct_w = _direct_targets(weight)
        ctx.w_shape = tuple(weight.shape)
        return z

    @staticmethod
    def backward(ctx, dz):
        t, w = ctx.saved_tensors
        R, KP = t.shape
        N, K = w.shape
        dz = dz if dz.is_contiguous() else dz.contiguous()
        dw = None
        if ctx.needs_input_grad[1]:
            if ctx.direct_w is not None:
                _wgrad_rows(R, N, K, dz, t, ctx.direct_w[1][0].grad.view(N, K), accumulate=True)
                _direct_done(ctx.direct_w)
            else:
                dw = torch.empty((N, K), dtype=torch.float32, device=t.device)
                _wgrad_rows(R, N, K, dz, t, dw)
                dw = dw.view(ctx.w_shape)
        dt = None
        if ctx.needs_input_grad[0]:
            dt = torch.empty((R, KP), dtype=torch.float32, device=t.device)
            _rows_gemm(R, KP, N, dz, N, w, K, True, dt, KP)           # (columns K .. KP: zero)
        return dt, dw, None


def _rows_linear_ok(t, w):
    R, KP = t.shape
    N, K = w.shape
    return (_ROWS_SPLIT and _FUSE_BN and R >= 8192 and KP % 4 == 0 and N % 4 == 0 and K <= KP <= 256 and N <= 256
            and t.is_contiguous() and t.data_ptr() % 16 == 0)


def _bn_linear_fused_ok(z, part, w):
    """The normalisation of z can ride on the next product's operand load: statistics already there, both products of
    the pair on the tall-rows kernels (whole weight in LDS), no padding between the layers."""
    R, C = z.shape
    N, K = w.shape
    wide_ok = R <= 65536 and C <= 1024 and N <= 1024          # (the last level: LDS re-filled per super-slab)
    return (part is not None and K == C and _ROWS_SPLIT and _FUSE_BN and R >= 8192 and C % 4 == 0 and N % 4 == 0
            and ((C <= 256 and N <= 256) or wide_ok) and z.data_ptr() % 16 == 0)


_FUSE_BN = _os.environ.get("MSR3D_BN_FUSE", "1") != "0"


def _mlp_train_ok(mlp):
    """Module in training mode, every layer a bias-free 1x1 convolution followed by an affine,
    statistics-tracking BatchNorm2d and a ReLU, channel counts the kernels take."""
    if not (mlp.training and getattr(mlp, "use_hip_train", True)):
        return False
    pairs = mlp.conv_bn_pairs()
    for layer, (conv, bn) in zip(mlp, pairs):
        if conv is None or bn is None or conv.bias is not None or not bn.affine or bn.momentum is None:
            return False
        if conv.kernel_size != (1, 1) or conv.stride != (1, 1) or conv.padding != (0, 0) or conv.groups != 1:
            return False
        if conv.out_channels % 4 or conv.out_channels > 1024 or not isinstance(
                list(layer.children())[-1], torch.nn.ReLU):
            return False
        if list(layer.children())[0] is not conv:          # pre-activation order is not on this path
            return False
    return len(pairs) > 0


def shared_mlp_train_supported(mlp, x):
    """GPU fp32 (B, C, npoint, nsample) input and a SharedMLP `_mlp_train_ok` accepts."""
    return bool(x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and _mlp_train_ok(mlp))


def _mlp_rows(mlp, t, pool_ns):
    """The SharedMLP on token-major rows t (R, K >= C_in, zero-padded), then the max over every
    pool_ns consecutive rows -> (R / pool_ns, C_out); the last layer's normalisation and the pooling
    are one kernel.  On tall rows (>= 8192) every product runs on the tall-rows kernels, leaves the next
    BatchNorm's first-stage statistics behind, and applies the PREVIOUS layer's normalisation + ReLU to its operand
    on load (_BNReLULinear): between two layers only the pre-normalisation z exists in memory."""
    pairs = mlp.conv_bn_pairs()
    z = part = pending = None        # pending: the BatchNorm whose normalisation the NEXT product applies on load
    for j, (conv, bn) in enumerate(pairs):
        w = conv.weight.view(conv.out_channels, conv.in_channels)
        got = []
        if pending is not None and _bn_linear_fused_ok(z, part, w):
            z = _BNReLULinear.apply(z, pending.weight, pending.bias, pending, part, conv.weight, got)
        else:
            if pending is not None:
                t = _BNReLUTrain.apply(z, pending.weight, pending.bias, pending, part)
            if _rows_linear_ok(t, w):
                z = _RowsLinear.apply(t, conv.weight, got)
            else:
                if t.shape[1] != w.shape[1]:
                    w = F.pad(w, (0, t.shape[1] - w.shape[1]))      # zero columns against the operand's padding
                z = linear(t, w, stats_out=got)
        part = got[0] if got else None
        pending = bn
    return _BNReLUMaxPoolTrain.apply(z, pending.weight, pending.bias, pending, pool_ns, part)


def shared_mlp_train(mlp, x):
    """x (B, C, npoint, nsample) -> max over nsample of the SharedMLP's output, (B, C_out, npoint).
    Same values as `F.max_pool2d(mlp(x), [1, nsample])` in training mode (batch statistics, running
    statistics updated), computed on a token-major copy: rows = (b, point, sample)."""
    B, C, NP, NS = x.shape
    t = x.permute(0, 2, 3, 1).reshape(B * NP * NS, C)
    if C % 4:
        t = F.pad(t, (0, (-C) % 4))          # 16-byte rows for the GEMM's vector loads
    pooled = _mlp_rows(mlp, t, NS).view(B, NP, -1)
    return pooled.permute(0, 2, 1).contiguous()


# ---- SharedMLP on the GPU outside the two fast paths (frozen fused kernels; training-mode rows above) --------------
# nn.Conv2d / nn.BatchNorm2d on a GPU tensor are MIOpen calls (convolution find + run-time kernel compilation through
# comgr on a fresh box, BatchNorm forward AND backward kernels built the same way): the one piece of GPU code a
# composite set-abstraction or feature-propagation module would run that is not this build's.  The 1x1 convolutions
# are products over token-major rows, so they run on this build's own GEMM (`linear`), the normalisation is spelled
# with elementwise / reduction operators (autograd differentiates them, batch statistics included), and the
# activation module is applied as it is.  Same values as the module tree (pytorch_utils.py:11-66 of the reference).
def _conv_is_pointwise(conv):
    return (conv.kernel_size == (1, 1) and conv.stride == (1, 1) and conv.padding == (0, 0) and conv.dilation == (1, 1)
            and conv.groups == 1 and conv.padding_mode == "zeros")


# element-by-element activations: the same values on token-major rows as on the (B, C, H, W) tensor
_ROWWISE_ACTIVATIONS = (torch.nn.ReLU, torch.nn.LeakyReLU, torch.nn.GELU, torch.nn.SiLU, torch.nn.Identity,
                        torch.nn.Tanh, torch.nn.Sigmoid, torch.nn.ELU, torch.nn.ReLU6)


def shared_mlp_rows_supported(mlp, x):
    """GPU fp32 (B, C, H, W) input and a stack of [1x1 conv | BatchNorm2d | activation] layers."""
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4):
        return False
    for layer in mlp:
        for m in layer.children():
            if isinstance(m, torch.nn.Conv2d):
                if not _conv_is_pointwise(m) or m.weight.dtype != torch.float32:
                    return False
            elif isinstance(m, torch.nn.Sequential):
                if not (len(m) == 1 and isinstance(m[0], torch.nn.BatchNorm2d)):
                    return False
            elif not isinstance(m, _ROWWISE_ACTIVATIONS):
                # applied to (B*H*W, C) rows here: anything that looks at spatial dimensions or at dim=-1 of the
                # (B, C, H, W) tensor (Softmax, Dropout2d, instance norms ...) would compute something else
                return False
    return True


def batch_norm_rows(bn, z):
    """nn.BatchNorm2d.forward on rows z (R, C): batch statistics (biased variance for the normalisation, unbiased
    into the running estimate, momentum or cumulative average) in training mode or without running buffers,
    running statistics otherwise."""
    use_batch = bn.training or bn.running_mean is None
    if use_batch:
        mean = z.mean(dim=0)
        var = z.var(dim=0, unbiased=False)
        if bn.training and bn.track_running_stats and bn.running_mean is not None:
            with torch.no_grad():
                bn.num_batches_tracked += 1
                f = (1.0 / float(bn.num_batches_tracked)) if bn.momentum is None else bn.momentum
                R = z.shape[0]
                bn.running_mean.mul_(1 - f).add_(mean.detach(), alpha=f)
                bn.running_var.mul_(1 - f).add_(var.detach() * (R / max(R - 1, 1)), alpha=f)
    else:
        mean, var = bn.running_mean, bn.running_var
    y = (z - mean) * torch.rsqrt(var + bn.eps)
    if bn.affine:
        y = y * bn.weight + bn.bias
    return y


def shared_mlp_rows(mlp, x):
    """`mlp(x)` for x (B, C, H, W) -> (B, C_out, H, W): the layers over token-major rows (b, h, w)."""
    B, C, H, W = x.shape
    t = x.permute(0, 2, 3, 1).reshape(B * H * W, C)
    for layer in mlp:
        for m in layer.children():
            if isinstance(m, torch.nn.Conv2d):
                t = linear(t, m.weight.view(m.out_channels, m.in_channels), m.bias)
            elif isinstance(m, torch.nn.Sequential):
                t = batch_norm_rows(m[0], t)
            else:
                t = m(t)
    return t.view(B, H, W, -1).permute(0, 3, 1, 2).contiguous()


class _GroupRows(torch.autograd.Function):
    """QueryAndGroup's output as token-major rows (msr3d_group_rows), gradient to the features by
    the deterministic ordered scatter (msr3d_group_rows_grad)."""

    @staticmethod
    def forward(ctx, xyz, new_xyz, feats, idx, KP):
        b, n, _ = xyz.shape
        m, ns = idx.shape[1], idx.shape[2]
        C = 0 if feats is None else feats.shape[1]
        xyz, new_xyz, idx = xyz.contiguous(), new_xyz.contiguous(), idx.contiguous()
        f = None if feats is None else feats.contiguous()
        rows = torch.empty((b * m * ns, KP), dtype=torch.float32, device=xyz.device)
        with torch.cuda.device(xyz.device):
            rc = _lib.load().msr3d_group_rows(b, n, m, ns, C, KP, _p(xyz), _p(new_xyz), _p(f), _p(idx),
                                              _p(rows), _lib.current_stream_ptr(xyz.device))
        _lib.check(rc, "msr3d_group_rows")
        ctx.save_for_backward(idx)
        ctx.dims = (b, n, m, ns, C, KP)
        return rows

    @staticmethod
    def backward(ctx, d_rows):
        (idx,) = ctx.saved_tensors
        b, n, m, ns, C, KP = ctx.dims
        if C == 0 or not ctx.needs_input_grad[2]:
            return None, None, None, None, None
        d_rows = d_rows if d_rows.is_contiguous() else d_rows.contiguous()
        d_feats = torch.empty((b, C, n), dtype=torch.float32, device=d_rows.device)
        with torch.cuda.device(d_rows.device):
            rc = _lib.load().msr3d_group_rows_grad(b, n, m, ns, C, KP, _p(d_rows), _p(idx), _p(d_feats),
                                                   _lib.current_stream_ptr(d_rows.device))
        _lib.check(rc, "msr3d_group_rows_grad")
        return None, None, d_feats, None, None


def group_rows_supported(xyz, new_xyz, feats, nsample):
    """fp32 GPU tensors, coordinates that need no gradient (the level's inputs in MSR3D), and an
    inverted index of one cloud that fits the backward kernel's LDS."""
    if not (xyz.is_cuda and xyz.dtype == torch.float32 and new_xyz is not None and not xyz.requires_grad
            and not new_xyz.requires_grad):
        return False
    if feats is not None and (feats.dtype != torch.float32 or feats.dim() != 3):
        return False
    return 2 * xyz.shape[1] + 1 + new_xyz.shape[1] * nsample <= 36 * 1024


def sa_level_train(mlp, xyz, new_xyz, feats, idx):
    """One set-abstraction scale in training mode: grouped rows -> SharedMLP -> max over the
    neighbourhood, (B, C_out, npoint); idx (B, npoint, nsample) from ball_query."""
    B, NP, NS = idx.shape
    C = 0 if feats is None else feats.shape[1]
    rows = _GroupRows.apply(xyz, new_xyz, feats, idx, (3 + C + 15) // 16 * 16)   # whole 16-wide K slabs
    pooled = _mlp_rows(mlp, rows, NS).view(B, NP, -1)
    return pooled.permute(0, 2, 1).contiguous()


# ---------------------------------------------------------------------------------------
# fused spatial attention core ('cond' fusion)
# ---------------------------------------------------------------------------------------
# Operand precision of the attention kernels' matrix products (include/msr3d_hip.h, MSR3D_MMA_*).
# "f32" is the reference's arithmetic and the default; opt-in, labelled: "bf16" (forward + backward), "fp8" (OCP e4m3,
# forward only: inference) and "fp8_bf16" (round 6: the TRAINING form of BASELINE.json configs[4]'s "fp8 MFMA
# object-attention" -- QK^T / PV of the forward on v_mfma_f32_16x16x32_fp8_fp8, the backward's four products on bf16
# operands; softmax, the spatial term and every accumulation stay fp32).  MSR3D_ATTN_MMA=... or set_attention_mma(...).
ATTN_MMA = {"f32": 0, "bf16": 1, "fp8": 2, "fp8_bf16": 2}
_attn_mma = [_os.environ.get("MSR3D_ATTN_MMA", "f32")]
if _attn_mma[0] not in ATTN_MMA:
    raise ValueError("MSR3D_ATTN_MMA must be one of %s" % sorted(ATTN_MMA))


def set_attention_mma(name):
    """Select the operand precision of QK^T / PV (and the backward products): 'f32' | 'bf16' | 'fp8' | 'fp8_bf16'.
    Returns the previous setting."""
    if name not in ATTN_MMA:
        raise ValueError("attention mma must be one of %s" % sorted(ATTN_MMA))
    prev, _attn_mma[0] = _attn_mma[0], name
    return prev


def attention_mma(backward=False, training=False):
    """The MSR3D_MMA_* code to pass to the kernels.  training: a forward whose backward will follow (the fused
    schedules): 'fp8' has none."""
    mode = _attn_mma[0]
    if mode == "fp8_bf16":
        return ATTN_MMA["bf16"] if backward else ATTN_MMA["fp8"]
    if (backward or training) and mode == "fp8":
        raise RuntimeError("fp8 attention is forward-only (inference); train with 'f32', 'bf16' or 'fp8_bf16'")
    return ATTN_MMA[mode]


class _SpatialAttnCond(torch.autograd.Function):
    """Operates on the PACKED projection output qkvc (B*L, 3D + H*6) = [q | k | v | cond]: the
    kernels read the four column blocks in place (leading dimension = packed width) and the
    backward writes the four gradients into one packed buffer -- no split / cat copies."""

    @staticmethod
    def forward(ctx, qkvc, pairwise_locs, pad_mask, B, L, D, n_head):
        dh = D // n_head
        W = qkvc.shape[-1]
        x = qkvc.reshape(B * L, W)
        if not x.is_contiguous():
            x = x.contiguous()
        pl = pairwise_locs.contiguous()
        pad = pad_mask.contiguous().view(torch.uint8)
        out = torch.empty((B * L, D), dtype=torch.float32, device=x.device)
        probs = torch.empty((B, n_head, L, L), dtype=torch.float32, device=x.device)
        lib = _lib.load()
        base, fs = x.data_ptr(), 4
        with torch.cuda.device(x.device):
            rc = lib.msr3d_spatial_attn_fwd(
                B, L, n_head, dh, pl.shape[-1], ctypes.c_void_p(base), ctypes.c_void_p(base + D * fs),
                ctypes.c_void_p(base + 2 * D * fs), W, ctypes.c_void_p(base + 3 * D * fs), W, _p(pl),
                _p(pad), _p(out), _p(probs), attention_mma(), _lib.current_stream_ptr(x.device))
        _lib.check(rc, "msr3d_spatial_attn_fwd")
        ctx.save_for_backward(x, pl, pad, probs)
        ctx.dims = (B, L, D, n_head, dh, W)
        ctx.mark_non_differentiable(probs)
        ctx.set_materialize_grads(False)          # no zero tensor for the probabilities' gradient
        return out.view(B, L, D), probs

    @staticmethod
    def backward(ctx, dout, _dprobs):
        x, pl, pad, probs = ctx.saved_tensors
        B, L, D, H, dh, W = ctx.dims
        if dout is None:
            return (None,) * 7
        do = dout.reshape(B * L, D)
        if not do.is_contiguous():
            do = do.contiguous()
        g = torch.empty_like(x)
        lib = _lib.load()
        base, gb, fs = x.data_ptr(), g.data_ptr(), 4
        vp = ctypes.c_void_p
        with torch.cuda.device(do.device):
            rc = lib.msr3d_spatial_attn_bwd(
                B, L, H, dh, pl.shape[-1], vp(base), vp(base + D * fs), vp(base + 2 * D * fs), W,
                vp(base + 3 * D * fs), W, _p(pl), _p(pad), _p(probs), _p(do), vp(gb), vp(gb + D * fs),
                vp(gb + 2 * D * fs), W, vp(gb + 3 * D * fs), W, attention_mma(True),
                _lib.current_stream_ptr(do.device))
        _lib.check(rc, "msr3d_spatial_attn_bwd")
        return g.view(B, L, W), None, None, None, None, None, None


def spatial_attn_cond_supported(q, n_head, spatial_dim, spatial_n_head):
    B, L, D = q.shape
    return (q.is_cuda and q.dtype == torch.float32 and L <= 128 and D // n_head == 32
            and spatial_dim == 5 and spatial_n_head == n_head)


def spatial_attn_cond(qkvc, pairwise_locs, key_padding_mask, n_head, d_model):
    """qkvc (B,L,3D+H*6) = packed [q|k|v|cond] projections -> ctx (B,L,D), probs (B,H,L,L)."""
    B, L = qkvc.shape[:2]
    if key_padding_mask is None:
        key_padding_mask = torch.zeros((B, L), dtype=torch.bool, device=qkvc.device)
    return _SpatialAttnCond.apply(qkvc, pairwise_locs, key_padding_mask, B, L, d_model, n_head)


# ---------------------------------------------------------------------------------------
# y = LayerNorm(dropout(a) + r): fused row kernels (csrc/rowops.hip)
# ---------------------------------------------------------------------------------------
_seed_words = {}
_salt_counter = [0]


def seed_word(device):
    """Device-resident 64-bit dropout seed (one per device), initialised from torch's seed."""
    key = (device.type, device.index)
    if key not in _seed_words:
        _seed_words[key] = torch.tensor([torch.initial_seed() & 0x7FFFFFFFFFFFFFFF],
                                        dtype=torch.int64, device=device)
    return _seed_words[key]


def bump_seed(device):
    """Advance the seed word on the device (call once per step; capturable)."""
    w = seed_word(device)
    lib = _lib.load()
    with torch.cuda.device(device):
        rc = lib.msr3d_bump_seed(_p(w), _lib.current_stream_ptr(device))
    _lib.check(rc, "msr3d_bump_seed")


_dot_scratch = {}
_dot_pool = {}


def dot(a, b):
    """sum(a * b) of two fp32 GPU tensors in one bit-reproducible launch (msr3d_dot_f32) -> 0-d tensor."""
    if not (a.is_cuda and a.dtype == torch.float32 and b.dtype == torch.float32 and a.numel() == b.numel()):
        raise ValueError("dot: two fp32 GPU tensors of the same size")
    a, b = a.contiguous(), b.contiguous()
    dev = a.device
    key = (dev, torch.cuda.current_stream(dev).cuda_stream)     # concurrent dots on two streams must not share
    ws = _dot_scratch.get(key)
    if ws is None:
        # a row of a pool zeroed ahead of time: a torch.zeros here would put a fill kernel into a graph that
        # is being captured on a fresh stream (the kernel itself leaves its counter at zero)
        pool = _dot_pool.setdefault(dev, [])
        if not pool:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("hipops.dot: first use on this device inside a graph capture; call it once eagerly")
            pool.extend(torch.zeros(8, 1024 + 1, dtype=torch.float32, device=dev).unbind(0))
        ws = _dot_scratch[key] = pool.pop()
    out = torch.empty((), dtype=torch.float32, device=dev)
    n = a.numel()
    if n % 4 or a.data_ptr() % 16 or b.data_ptr() % 16:
        return torch.dot(a.reshape(-1), b.reshape(-1))
    lib = _lib.load()
    with torch.cuda.device(dev):
        rc = lib.msr3d_dot_f32(n, _p(a), _p(b), _p(ws), _p(out), _lib.current_stream_ptr(dev))
    _lib.check(rc, "msr3d_dot_f32")
    return out


def _next_salt():
    _salt_counter[0] = (_salt_counter[0] + 1) & 0x7FFFFFFF
    return _salt_counter[0]


class _DropoutAddLN(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, r, gamma, beta, eps, p_drop):
        shape = a.shape
        D = shape[-1]
        a2 = a.reshape(-1, D)
        a2 = a2 if a2.is_contiguous() else a2.contiguous()
        r2 = None
        if r is not None:
            r2 = r.reshape(-1, D)
            r2 = r2 if r2.is_contiguous() else r2.contiguous()
        M = a2.shape[0]
        y = torch.empty_like(a2)
        need_bwd = any(ctx.needs_input_grad[:4])
        s = torch.empty_like(a2) if need_bwd else None
        stats = torch.empty((M, 2), dtype=torch.float32, device=a.device) if need_bwd else None
        salt = _next_salt() if p_drop > 0 else 0
        seed = seed_word(a.device) if p_drop > 0 else None
        lib = _lib.load()
        with torch.cuda.device(a.device):
            rc = lib.msr3d_dropout_add_ln_fwd(M, D, _p(a2), _p(r2), _p(gamma), _p(beta),
                                              ctypes.c_float(eps), ctypes.c_float(p_drop), _p(seed),
                                              salt, _p(y), _p(s), _p(stats),
                                              _lib.current_stream_ptr(a.device))
        _lib.check(rc, "msr3d_dropout_add_ln_fwd")
        ctx.save_for_backward(s, stats, gamma)
        ctx.cfg = (M, D, p_drop, salt, r is not None, shape)
        dpg = getattr(gamma, "_msr3d_dp", None)
        ok = (dpg is not None and getattr(beta, "_msr3d_dp", None) is dpg and gamma.is_leaf
              and beta.is_leaf and gamma.grad is not None and beta.grad is not None)
        ctx.direct = (dpg, gamma, beta) if (ok and not _NO_DIRECT) else None
        return y.view(shape)

    @staticmethod
    def backward(ctx, dy):
        s, stats, gamma = ctx.saved_tensors
        M, D, p_drop, salt, has_r, shape = ctx.cfg
        dy2 = dy.reshape(M, D)
        dy2 = dy2 if dy2.is_contiguous() else dy2.contiguous()
        want_a, want_r = ctx.needs_input_grad[0], has_r and ctx.needs_input_grad[1]
        # without dropout d(a) == d(r): write once
        same = p_drop == 0 and want_a and want_r
        da = torch.empty_like(dy2) if want_a else None
        dr = torch.empty_like(dy2) if (want_r and not same) else None
        if ctx.direct is not None:
            dpg, gparam, bparam = ctx.direct
            dg, db = gparam.grad, bparam.grad
        else:
            dg = torch.zeros(D, dtype=torch.float32, device=dy.device)
            db = torch.zeros(D, dtype=torch.float32, device=dy.device)
        seed = seed_word(dy.device) if p_drop > 0 else None
        lib = _lib.load()
        with torch.cuda.device(dy.device):
            rc = lib.msr3d_dropout_add_ln_bwd(M, D, _p(dy2), _p(s), _p(stats), _p(gamma),
                                              ctypes.c_float(p_drop), _p(seed), salt, _p(da), _p(dr), 0,
                                              _p(dg), _p(db), _p(ln_partials(M, D, 2, dy.device)),
                                              _lib.current_stream_ptr(dy.device))
        _lib.check(rc, "msr3d_dropout_add_ln_bwd")
        ga = da.view(shape) if da is not None else None
        gr = (ga if same else (dr.view(shape) if dr is not None else None))
        if ctx.direct is not None:
            dpg.mark_ready(gparam)
            dpg.mark_ready(bparam)
            return ga, gr, None, None, None, None
        return ga, gr, dg, db, None, None


def dropout_add_layernorm(a, r, ln, p_drop=0.0, training=False):
    """ln(dropout(a) + r) for an nn.LayerNorm `ln` over the last dim; r may be None."""
    p = float(p_drop) if training else 0.0
    D = a.shape[-1]
    if (a.is_cuda and a.dtype == torch.float32 and D in (256, 512, 768, 1024)
            and ln.elementwise_affine and ln.bias is not None and tuple(ln.normalized_shape) == (D,)):
        return _DropoutAddLN.apply(a, r, ln.weight, ln.bias, ln.eps, p)
    x = F.dropout(a, p, True) if p > 0 else a
    if r is not None:
        x = x + r
    return ln(x)


# ---------------------------------------------------------------------------------------
# constant embeddings broadcast onto every token (csrc/prologue.hip)
# ---------------------------------------------------------------------------------------
class _AddTokenConstants(torch.autograd.Function):
    """out = x + table[row] + extra: every object token receives the same type-embedding row and
    the same learnt orientation vector (model/ose3d_situation.py:327-365).  Autograd's version
    costs two broadcast adds forward and eight launches backward (two reductions with their
    memsets, a zero table for the embedding row, copies, adds); here: one launch forward, and the
    column sum of the upstream gradient goes straight onto the two parameters' gradients."""

    @staticmethod
    def forward(ctx, x, table, row, extra):
        shape = x.shape
        D = shape[-1]
        x2 = x.reshape(-1, D)
        x2 = x2 if x2.is_contiguous() else x2.contiguous()
        v1 = table[row]
        v2 = extra.reshape(-1) if extra is not None else None
        out = torch.empty_like(x2)
        lib = _lib.load()
        with torch.cuda.device(x.device):
            rc = lib.msr3d_add_row_vectors(x2.shape[0], D, _p(x2), _p(v1), _p(v2), _p(out),
                                           _lib.current_stream_ptr(x.device))
        _lib.check(rc, "msr3d_add_row_vectors")
        ctx.row, ctx.shape = row, shape
        ctx.params = (table, extra)
        dpt = getattr(table, "_msr3d_dp", None)
        ok = dpt is not None and table.is_leaf and table.grad is not None
        if ok and extra is not None:
            ok = getattr(extra, "_msr3d_dp", None) is dpt and extra.is_leaf and extra.grad is not None
        ctx.direct = dpt if (ok and not _NO_DIRECT) else None
        return out.view(shape)

    @staticmethod
    def backward(ctx, dy):
        table, extra = ctx.params
        D = ctx.shape[-1]
        d2 = dy.reshape(-1, D)
        d2 = d2 if d2.is_contiguous() else d2.contiguous()
        M = d2.shape[0]
        if ctx.direct is not None:
            _colsum(d2, M, D, table.grad[ctx.row], accumulate=True)
            ctx.direct.mark_ready(table)
            if extra is not None:
                _colsum(d2, M, D, extra.grad.view(-1), accumulate=True)
                ctx.direct.mark_ready(extra)
            return dy, None, None, None
        col = torch.empty((D,), dtype=torch.float32, device=dy.device)
        _colsum(d2, M, D, col)
        gt = torch.zeros_like(table)
        gt[ctx.row] = col
        return dy, gt, None, (col.view_as(extra).clone() if extra is not None else None)


def add_token_constants(x, table, row, extra=None):
    """x (..., D) + table[row] + extra (any shape with D elements; optional)."""
    if x.is_cuda and x.dtype == torch.float32 and table.dtype == torch.float32 and x.shape[-1] % 4 == 0:
        return _AddTokenConstants.apply(x, table, row, extra)
    out = x + table[row]
    return out + extra.reshape(-1) if extra is not None else out


# ---------------------------------------------------------------------------------------
# data-only front of the encoder (csrc/prologue.hip)
# ---------------------------------------------------------------------------------------
def pairwise_locs_center5(obj_loc, eps=1e-10):
    """obj_loc (B,L,>=3) f32 contiguous (centres first) -> (B,L,L,5); GPU fp32 only."""
    B, L, W = obj_loc.shape
    x = obj_loc if obj_loc.is_contiguous() else obj_loc.contiguous()
    out = torch.empty((B, L, L, 5), dtype=torch.float32, device=x.device)
    lib = _lib.load()
    with torch.cuda.device(x.device):
        rc = lib.msr3d_pairwise_locs(B, L, _p(x), W, ctypes.c_float(eps), _p(out),
                                     _lib.current_stream_ptr(x.device))
    _lib.check(rc, "msr3d_pairwise_locs")
    return out


_freq_cache = {}


def agent_fourier(obj_loc, anchor_loc=None, anchor_ori=None, num_bands=10, max_freq=15):
    """Fourier features of the object centres, optionally first moved into the agent frame.
    obj_loc (B,L,>=3); anchor_loc (B,3), anchor_ori (B,4 xyzw) or None -> (B,L,3+6*num_bands)."""
    B, L, W = obj_loc.shape
    x = obj_loc if obj_loc.is_contiguous() else obj_loc.contiguous()
    key = (x.device, num_bands, max_freq)
    if key not in _freq_cache:
        _freq_cache[key] = torch.linspace(1.0, max_freq, steps=num_bands, device=x.device)
    freqs = _freq_cache[key]
    transform = anchor_loc is not None
    al = anchor_loc.contiguous() if transform else None
    ao = anchor_ori.contiguous() if transform else None
    out = torch.empty((B, L, 3 + 6 * num_bands), dtype=torch.float32, device=x.device)
    lib = _lib.load()
    with torch.cuda.device(x.device):
        rc = lib.msr3d_agent_fourier(B, L, _p(x), W, _p(al), _p(ao), _p(freqs), num_bands,
                                     int(transform), _p(out), _lib.current_stream_ptr(x.device))
    _lib.check(rc, "msr3d_agent_fourier")
    return out
