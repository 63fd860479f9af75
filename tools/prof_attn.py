"""Time the fused attention kernels (msr3d_attn_fwd / _bwd) at the language model's shape.
python tools/prof_attn.py [B T H D]"""
import ctypes
import math
import sys

import torch

sys.path.insert(0, ".")
from msr3d_amd import _lib  # noqa: E402

B, T, H, D = (int(v) for v in sys.argv[1:5]) if len(sys.argv) >= 5 else (4, 576, 32, 128)
lib = _lib.load()
dev = torch.device("cuda:0")
st = _lib.current_stream_ptr(dev)
mk = lambda: torch.randn(B, T, H, D, device=dev).to(torch.bfloat16)      # noqa: E731
q, k, v, do = mk(), mk(), mk(), mk()
keep = torch.ones(B, T, dtype=torch.uint8, device=dev)
keep[:, :17] = 0
out = torch.empty(B, T, H * D, dtype=torch.bfloat16, device=dev)
lse = torch.empty(B, H, T, device=dev)
delta = torch.empty_like(lse)
dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
p = lambda t: ctypes.c_void_p(t.data_ptr())      # noqa: E731
sc = ctypes.c_float(1.0 / math.sqrt(D))


def fwd():
    lib.msr3d_attn_fwd(B, T, H, D, p(q), p(k), p(v), H * D, p(keep), sc, p(out), p(lse), st)


def bwd():
    lib.msr3d_attn_bwd(B, T, H, D, p(q), p(k), p(v), p(out), p(do), H * D, p(keep), sc, p(lse), p(delta), p(dq), p(dk), p(dv), st)


flops = 4.0 * B * H * T * T * D / 2            # two products, causal half
for name, fn, mult in (("forward", fwd, 1.0), ("backward (dq + dk/dv)", bwd, 3.5)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"B={B} T={T} H={H} D={D}  {name:24s} {us:8.1f} us  {flops * mult / us / 1e6:7.0f} TFLOP/s", flush=True)
