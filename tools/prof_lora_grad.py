"""Time msr3d_lora_grad (two products + two reductions) against msr3d_lora_grad_pair (one launch) for the LoRA pairs of
a Vicuna-7B layer.  python tools/prof_lora_grad.py"""
import ctypes
import sys

import torch

sys.path.insert(0, ".")
from msr3d_amd import _lib  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda:0")
st = _lib.current_stream_ptr(dev)
r = 16
for M, K, N in [(2304, 4096, 4096), (2304, 4096, 11008), (2304, 11008, 4096), (11520, 4096, 4096)]:
    bf = lambda *s: torch.randn(*s, device=dev).to(torch.bfloat16)      # noqa: E731
    v, u, x, dy = bf(M, 64), bf(M, 64), bf(M, K), bf(M, N)
    dA, dB = torch.zeros(r, K, device=dev), torch.zeros(N, r, device=dev)
    ws_old = torch.zeros(64 * r * max(K, N), device=dev)
    jobs = (_lib.LoraGradJob * 2)(_lib.LoraGradJob(K, v.data_ptr(), 64, x.data_ptr(), K, dA.data_ptr(), 0),
                                  _lib.LoraGradJob(N, u.data_ptr(), 64, dy.data_ptr(), N, dB.data_ptr(), 1))
    p = lambda t: ctypes.c_void_p(t.data_ptr())      # noqa: E731

    def old():
        lib.msr3d_lora_grad(M, r, K, p(v), 64, p(x), K, p(dA), 0, ctypes.c_float(1.0), 1, p(ws_old), ws_old.numel(), st)
        lib.msr3d_lora_grad(M, r, N, p(u), 64, p(dy), N, p(dB), 1, ctypes.c_float(1.0), 1, p(ws_old), ws_old.numel(), st)

    def new():
        lib.msr3d_lora_grad_pair(M, r, 2, jobs, ctypes.c_float(1.0), 1, st)

    for name, fn in (("old 4 launches", old), ("pair 1 launch", new)):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            fn()
        e1.record()
        torch.cuda.synchronize()
        print(f"M={M} K={K} N={N}  {name:16s} {e0.elapsed_time(e1) / 50 * 1e3:8.1f} us", flush=True)
