"""Micro-benchmark of the spatial-attention kernels in the three operand precisions
(f32 / bf16 / fp8-forward) at the bench shape (16 scenes x 60 tokens) and the stress shape
(16 x 120).  Times are per launch, back-to-back launches between two HIP events.
    python tools/bench_attn.py [--iters 200]"""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from msr3d_amd import _lib, hipops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=200)
args = ap.parse_args()
D, H = 256, 8


def timed(fn, iters):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def raw_bwd(qkvc, pl, mask, probs, dctx, g, mma):
    """msr3d_spatial_attn_bwd on the packed buffers, without the autograd node around it."""
    B, L, W = qkvc.shape
    vp, fs = ctypes.c_void_p, 4
    base, gb = qkvc.data_ptr(), g.data_ptr()
    rc = _lib.load().msr3d_spatial_attn_bwd(
        B, L, H, D // H, 5, vp(base), vp(base + D * fs), vp(base + 2 * D * fs), W, vp(base + 3 * D * fs), W,
        vp(pl.data_ptr()), vp(mask.view(torch.uint8).data_ptr()), vp(probs.data_ptr()), vp(dctx.data_ptr()),
        vp(gb), vp(gb + D * fs), vp(gb + 2 * D * fs), W, vp(gb + 3 * D * fs), W, mma,
        _lib.current_stream_ptr(qkvc.device))
    _lib.check(rc, "msr3d_spatial_attn_bwd")


for B, L in ((16, 60), (16, 120)):
    torch.manual_seed(0)
    qkvc = torch.randn(B, L, 3 * D + H * 6, device="cuda")
    pl = torch.randn(B, L, L, 5, device="cuda")
    mask = torch.zeros(B, L, dtype=torch.bool, device="cuda")
    dctx = torch.randn(B, L, D, device="cuda")
    g = torch.empty_like(qkvc)
    base = None
    for mma in ("f32", "bf16", "fp8"):
        hipops.set_attention_mma(mma)
        with torch.no_grad():
            t_f = timed(lambda: hipops.spatial_attn_cond(qkvc, pl, mask, H, D), args.iters)
            ctx, _ = hipops.spatial_attn_cond(qkvc, pl, mask, H, D)
        base = ctx if base is None else base
        err = float((ctx - base).norm() / base.norm())
        line = f"B={B} L={L:3d} {mma:4s} fwd {t_f:6.1f} us  rel-L2 vs f32 {err:.2e}"
        if mma != "fp8":
            _, probs = hipops.spatial_attn_cond(qkvc, pl, mask, H, D)
            t_b = timed(lambda: raw_bwd(qkvc, pl, mask, probs, dctx, g, hipops.ATTN_MMA[mma]), args.iters)
            line += f"   bwd {t_b:6.1f} us"
        print(line)
    hipops.set_attention_mma("f32")
