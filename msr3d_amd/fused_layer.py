"""One autograd node for a whole TransformerSpatialEncoderLayer ('cond' fusion, self-attention,
GELU FFN; /root/reference/modules/layers/transformers.py:200-252,314-329) in training.

Built from the same HIP kernels as the modular path (hipops), but scheduled by hand so that the
glue autograd would insert disappears from the captured graph:

  * the three split-K outputs of the forward (packed q|k|v|cond projection, attention `fc`,
    `linear2`) live in ONE zero-filled buffer: one fill instead of three memsets;
  * residual gradients meet in place: `linear1`'s and the packed projection's dx GEMMs add onto
    the buffer the LayerNorm backward already wrote (beta = 1: no zero-fill, no add kernel), and
    the second LayerNorm backward accumulates its residual gradient (`dr_accumulate`);
  * the FFN dropout rides in `linear1`'s GEMM epilogue (after GELU) and is regenerated inside the
    GELU backward kernel: no mask tensor, no dropout / masked-scale kernels;
  * every dW/db goes straight into the flat gradient buffer of the data-parallel engine.

  * each linear's two backward products (dx, and dW/db) are one launch.

Per layer this is 10 launches forward and 9 backward instead of 13 and 22.  Numerics are those of
the modular path (same kernels, same order of floating-point operations inside them); the dropout
masks differ only in which call site draws them.
"""
import ctypes

import torch
import torch.nn.functional as F

from . import _lib, hipops
from .hipops import _gemm, _gelu_bwd, _linear_bwd, _next_salt, _p, seed_word


def _dal_fwd(a, r, ln, p, salt):
    M, D = a.shape
    y = torch.empty_like(a)
    s = torch.empty_like(a)
    stats = torch.empty((M, 2), dtype=torch.float32, device=a.device)
    seed = seed_word(a.device) if p > 0 else None
    lib = _lib.load()
    with torch.cuda.device(a.device):
        rc = lib.msr3d_dropout_add_ln_fwd(M, D, _p(a), _p(r), _p(ln.weight), _p(ln.bias),
                                          ctypes.c_float(ln.eps), ctypes.c_float(p), _p(seed), salt,
                                          _p(y), _p(s), _p(stats), _lib.current_stream_ptr(a.device))
    _lib.check(rc, "msr3d_dropout_add_ln_fwd")
    return y, s, stats


def _dal_bwd(dy, s, stats, ln, p, salt, da, dr, accumulate):
    M, D = dy.shape
    seed = seed_word(dy.device) if p > 0 else None
    lib = _lib.load()
    with torch.cuda.device(dy.device):
        rc = lib.msr3d_dropout_add_ln_bwd(M, D, _p(dy), _p(s), _p(stats), _p(ln.weight),
                                          ctypes.c_float(p), _p(seed), salt, _p(da), _p(dr),
                                          int(accumulate), _p(ln.weight.grad), _p(ln.bias.grad),
                                          _p(hipops.ln_partials(M, D, 2, dy.device)),
                                          _lib.current_stream_ptr(dy.device))
    _lib.check(rc, "msr3d_dropout_add_ln_bwd")


def _dal2_fwd(a, r, ln1, p1, salt1, ln2, p2, salt2):
    """t = ln2(drop2(ln1(drop1(a) + r)) + r) in one launch (msr3d_dropout_add_ln2_fwd)."""
    M, D = a.shape
    dev = a.device
    y, s1, s2 = torch.empty_like(a), torch.empty_like(a), torch.empty_like(a)
    st1 = torch.empty((M, 2), dtype=torch.float32, device=dev)
    st2 = torch.empty((M, 2), dtype=torch.float32, device=dev)
    seed = seed_word(dev) if (p1 > 0 or p2 > 0) else None
    lib = _lib.load()
    with torch.cuda.device(dev):
        rc = lib.msr3d_dropout_add_ln2_fwd(M, D, _p(a), _p(r), _p(ln1.weight), _p(ln1.bias),
                                           ctypes.c_float(ln1.eps), ctypes.c_float(p1), salt1,
                                           _p(ln2.weight), _p(ln2.bias), ctypes.c_float(ln2.eps),
                                           ctypes.c_float(p2), salt2, _p(seed), _p(y), _p(s1), _p(st1),
                                           _p(s2), _p(st2), _lib.current_stream_ptr(dev))
    _lib.check(rc, "msr3d_dropout_add_ln2_fwd")
    return y, s1, st1, s2, st2


def _dal2_bwd(dy, s1, st1, ln1, p1, salt1, s2, st2, ln2, p2, salt2, da, dr):
    M, D = dy.shape
    dev = dy.device
    seed = seed_word(dev) if (p1 > 0 or p2 > 0) else None
    lib = _lib.load()
    with torch.cuda.device(dev):
        rc = lib.msr3d_dropout_add_ln2_bwd(M, D, _p(dy), _p(s1), _p(st1), _p(ln1.weight),
                                           ctypes.c_float(p1), salt1, _p(s2), _p(st2), _p(ln2.weight),
                                           ctypes.c_float(p2), salt2, _p(seed), _p(da), _p(dr),
                                           _p(ln1.weight.grad), _p(ln1.bias.grad), _p(ln2.weight.grad),
                                           _p(ln2.bias.grad), _p(hipops.ln_partials(M, D, 4, dev)),
                                           _lib.current_stream_ptr(dev))
    _lib.check(rc, "msr3d_dropout_add_ln2_bwd")


def _direct(dp, *params):
    return all(p is not None and getattr(p, "_msr3d_dp", None) is dp and p.is_leaf and p.grad is not None
               and p.is_contiguous() for p in params)


def eligible(layer, x, pairwise_locs):
    """The hand-scheduled node needs: GPU fp32 tokens, the fused attention core, packed projection
    views and every parameter's gradient living in the flat buffer (HotPathTrainStep's setup)."""
    sa = layer.self_attn
    packed = getattr(sa, "_packed", None)
    if (packed is None or not torch.is_grad_enabled() or not x.is_cuda or x.dtype != torch.float32
            or not getattr(layer, "use_fused_layer", True) or not getattr(sa, "use_fused_core", True)):
        return False
    if sa.spatial_attn_fusion != "cond" or layer.activation is not F.gelu or layer.prenorm:
        return False
    if not hipops.spatial_attn_cond_supported(x, sa.n_head, sa.spatial_dim, sa.spatial_n_head):
        return False
    D = x.shape[-1]
    if D not in (256, 512, 768, 1024) or layer.linear1.out_features % 4:
        return False
    dp = packed[4]
    return _direct(dp, sa.fc.weight, sa.fc.bias, sa.layer_norm.weight, sa.layer_norm.bias,
                   layer.norm1.weight, layer.norm1.bias, layer.norm2.weight, layer.norm2.bias,
                   layer.linear1.weight, layer.linear1.bias, layer.linear2.weight, layer.linear2.bias)


def layer_params(layer):
    sa = layer.self_attn
    return list(sa._packed_members) + [
        sa.fc.weight, sa.fc.bias, sa.layer_norm.weight, sa.layer_norm.bias, layer.norm1.weight,
        layer.norm1.bias, layer.norm2.weight, layer.norm2.bias, layer.linear1.weight,
        layer.linear1.bias, layer.linear2.weight, layer.linear2.bias]


class _SpatialLayerFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, pairwise_locs, pad_mask, layer, *params):
        sa = layer.self_attn
        wv, bv, gwv, gbv, dp = sa._packed
        B, L, D = x.shape
        M, W, H, FF = B * L, wv.shape[0], sa.n_head, layer.linear1.out_features
        dev = x.device
        train = layer.training
        p_attn = float(sa.dropout.p) if train else 0.0
        p1 = float(layer.dropout1.p) if train else 0.0
        p2 = float(layer.dropout2.p) if train else 0.0
        p_ffn = float(layer.dropout.p) if train else 0.0
        salts = [_next_salt() for _ in range(4)]
        x2 = x.reshape(M, D)
        x2 = x2 if x2.is_contiguous() else x2.contiguous()
        pl = pairwise_locs.contiguous()
        pad = pad_mask.contiguous().view(torch.uint8)

        # the three split-K meeting points of the forward and the one of the backward (fc's dx),
        # zeroed by one fill
        zero = torch.zeros(M * (W + 3 * D), dtype=torch.float32, device=dev)
        qkvc = zero[:M * W].view(M, W)
        fc_out = zero[M * W:M * (W + D)].view(M, D)
        ffn_out = zero[M * (W + D):M * (W + 2 * D)].view(M, D)
        d_attn = zero[M * (W + 2 * D):].view(M, D)

        _gemm(True, True, M, W, D, x2, D, wv, D, qkvc, W, bias=bv, beta=1.0)
        attn = torch.empty((M, D), dtype=torch.float32, device=dev)
        probs = torch.empty((B, H, L, L), dtype=torch.float32, device=dev)
        lib = _lib.load()
        base, fs, vp = qkvc.data_ptr(), 4, ctypes.c_void_p
        with torch.cuda.device(dev):
            rc = lib.msr3d_spatial_attn_fwd(
                B, L, H, D // H, pl.shape[-1], vp(base), vp(base + D * fs), vp(base + 2 * D * fs), W,
                vp(base + 3 * D * fs), W, _p(pl), _p(pad), _p(attn), _p(probs),
                hipops.attention_mma(False, training=True), _lib.current_stream_ptr(dev))
        _lib.check(rc, "msr3d_spatial_attn_fwd")
        _gemm(True, True, M, D, D, attn, D, sa.fc.weight, D, fc_out, D, bias=sa.fc.bias, beta=1.0)
        if D in (256, 512):     # transformers.py:250-251 then :324-325, one launch
            t, s1, st1, s2, st2 = _dal2_fwd(fc_out, x2, sa.layer_norm, p_attn, salts[0], layer.norm1, p1,
                                            salts[1])
        else:
            a, s1, st1 = _dal_fwd(fc_out, x2, sa.layer_norm, p_attn, salts[0])
            t, s2, st2 = _dal_fwd(a, x2, layer.norm1, p1, salts[1])
        h = torch.empty((M, FF), dtype=torch.float32, device=dev)
        pre = torch.empty((M, FF), dtype=torch.float32, device=dev)
        _gemm(True, True, M, FF, D, t, D, layer.linear1.weight, D, h, FF, bias=layer.linear1.bias,
              c_pre=pre, flags=1, p_drop=p_ffn, salt=salts[3])                  # gelu, dropout: :326
        _gemm(True, True, M, D, FF, h, FF, layer.linear2.weight, FF, ffn_out, D,
              bias=layer.linear2.bias, beta=1.0)
        out, s3, st3 = _dal_fwd(ffn_out, t, layer.norm2, p2, salts[2])          # :327-328

        ctx.save_for_backward(x2, qkvc, pl, pad, probs, attn, s1, st1, s2, st2, t, pre, h, s3, st3,
                              d_attn)
        ctx.set_materialize_grads(False)          # no zero tensor for the probabilities' gradient
        ctx.layer = layer
        ctx.cfg = (B, L, D, M, W, H, FF, p_attn, p1, p2, p_ffn, salts)
        ctx.mark_non_differentiable(probs)
        return out.view(B, L, D), probs

    @staticmethod
    def backward(ctx, d_out, _d_probs):
        (x2, qkvc, pl, pad, probs, attn, s1, st1, s2, st2, t, pre, h, s3, st3,
         d_attn) = ctx.saved_tensors
        layer = ctx.layer
        sa = layer.self_attn
        wv, bv, gwv, gbv, dp = sa._packed
        B, L, D, M, W, H, FF, p_attn, p1, p2, p_ffn, salts = ctx.cfg
        n_params = len(layer_params(layer))
        if d_out is None:                          # tokens unused downstream
            return (None,) * (4 + n_params)
        dev = d_out.device
        g = d_out.reshape(M, D)
        g = g if g.is_contiguous() else g.contiguous()
        new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)   # noqa: E731

        d_t, d_ffn = new(M, D), new(M, D)
        _dal_bwd(g, s3, st3, layer.norm2, p2, salts[2], d_ffn, d_t, False)
        # each linear's dx and dW/db are one launch (msr3d_linear_bwd_f32)
        d_h = new(M, FF)
        _linear_bwd(M, D, FF, d_ffn, h, layer.linear2.weight, d_h, 0.0,
                    layer.linear2.weight.grad, layer.linear2.bias.grad)
        d_pre = _gelu_bwd(d_h, pre, p_ffn, salts[3])
        _linear_bwd(M, FF, D, d_pre, t, layer.linear1.weight, d_t, 1.0,                       # joins d_t
                    layer.linear1.weight.grad, layer.linear1.bias.grad)

        d_x, d_fc = new(M, D), new(M, D)
        if D in (256, 512):
            _dal2_bwd(d_t, s1, st1, sa.layer_norm, p_attn, salts[0], s2, st2, layer.norm1, p1, salts[1],
                      d_fc, d_x)
        else:
            d_a = new(M, D)
            _dal_bwd(d_t, s2, st2, layer.norm1, p1, salts[1], d_a, d_x, False)
            _dal_bwd(d_a, s1, st1, sa.layer_norm, p_attn, salts[0], d_fc, d_x, True)         # joins d_x
        _linear_bwd(M, D, D, d_fc, attn, sa.fc.weight, d_attn, 1.0,                           # onto the zeros
                    sa.fc.weight.grad, sa.fc.bias.grad)
        d_qkvc = new(M, W)
        lib = _lib.load()
        base, gb, fs, vp = qkvc.data_ptr(), d_qkvc.data_ptr(), 4, ctypes.c_void_p
        with torch.cuda.device(dev):
            rc = lib.msr3d_spatial_attn_bwd(
                B, L, H, D // H, pl.shape[-1], vp(base), vp(base + D * fs), vp(base + 2 * D * fs), W,
                vp(base + 3 * D * fs), W, _p(pl), _p(pad), _p(probs), _p(d_attn), vp(gb),
                vp(gb + D * fs), vp(gb + 2 * D * fs), W, vp(gb + 3 * D * fs), W,
                hipops.attention_mma(True), _lib.current_stream_ptr(dev))
        _lib.check(rc, "msr3d_spatial_attn_bwd")
        _linear_bwd(M, W, D, d_qkvc, x2, wv, d_x, 1.0, gwv, gbv)                              # joins d_x

        for p in layer_params(layer):
            dp.mark_ready(p)
        return (d_x.view(B, L, D), None, None, None) + (None,) * n_params


def spatial_layer(layer, x, pairwise_locs, key_padding_mask):
    """-> (tokens (B,L,D), attention probabilities (H,B,L,L)) like the layer's forward."""
    if key_padding_mask is None:
        key_padding_mask = torch.zeros(x.shape[:2], dtype=torch.bool, device=x.device)
    out, probs = _SpatialLayerFn.apply(x, pairwise_locs, key_padding_mask, layer, *layer_params(layer))
    return out, probs.permute(1, 0, 2, 3)
