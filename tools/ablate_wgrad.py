"""Where the pipe weight-gradient tile's time goes: the launch (whole tiles, bench step's problem set) built without one
of its parts at a time (tools/ablate_wgrad.sh).  python tools/ablate_wgrad.py"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from msr3d_amd import _lib  # noqa: E402
from msr3d_amd.scene_blocks import WgradTable  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
M, D, FF, W, E, KE = 960, 256, 2048, 816, 4096, 768
dev = torch.device("cuda")
t = WgradTable(dev)
t.mixed = False
keep = []


def add(n_out, k_in):
    dy, x = torch.randn(M, n_out, device=dev), torch.randn(M, k_in, device=dev)
    dW, db = torch.zeros(n_out, k_in, device=dev), torch.zeros(n_out, device=dev)
    keep.extend([dy, x, dW, db])
    t.add(dy.data_ptr(), n_out, n_out, x.data_ptr(), k_in, k_in, M, dW.data_ptr(), k_in, db.data_ptr())


which = sys.argv[1] if len(sys.argv) > 1 else "all"
if which in ("all", "llm"):
    add(E, D)
if which in ("all", "layers"):
    for _ in range(3):
        add(D, FF); add(FF, D); add(W, D); add(D, D)
if which == "all":
    add(D, 64); add(D, 3); add(D, KE)
st = _lib.current_stream_ptr(dev)
t.launch(st)
torch.cuda.synchronize()
names = ["full", "no MFMAs", "no split arithmetic", "no global loads", "no fragment reads", "no stash stores", "full, no sched groups"]
for n, name in enumerate(names):
    path = os.path.join(HERE, "_prof", f"libwgp_{n}.so")
    if not os.path.exists(path):
        continue
    lib = ctypes.CDLL(path)
    lib.msr3d_wgrad_split.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    lib.msr3d_wgrad_form.argtypes = [ctypes.c_int]
    lib.msr3d_wgrad_form(1)
    ts = []
    for rep in range(25):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = lib.msr3d_wgrad_split(len(t.probs), t._table.data_ptr(), t._pfx.data_ptr(), t.prefix[-1], st)
        e1.record()
        torch.cuda.synchronize()
        assert rc == 0
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    print(f"{which:7s} {t.prefix[-1]:4d} workgroups  {name:22s} median {ts[len(ts) // 2]:7.1f} us  min {ts[0]:7.1f} us")
