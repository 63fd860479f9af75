"""Phase stamps of msr3d_wgrad_split on the bench step's problem set (synthetic operands)."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from msr3d_amd import _lib  # noqa: E402
from msr3d_amd.scene_blocks import WgradTable  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(HERE, "_prof", os.environ.get("WG_LIB", "libwg_stamped.so")))
lib.msr3d_wgrad_split.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
lib.msr3d_prof_wgrad_stamps.argtypes = [ctypes.c_void_p]
M, D, FF, W, E, KE = 960, 256, 2048, 816, 4096, 768
dev = torch.device("cuda")
t = WgradTable(dev)
keep = []


PAD = int(os.environ.get("WG_PAD", "0"))


def add(n_out, k_in):
    dyw = torch.randn(M, n_out + PAD, device=dev); xw = torch.randn(M, k_in + PAD, device=dev)
    dy, x = dyw[:, :n_out], xw[:, :k_in]
    dW = torch.zeros(n_out, k_in, device=dev); db = torch.zeros(n_out, device=dev)
    keep.extend([dyw, xw, dW, db])
    t.add(dy.data_ptr(), dyw.stride(0), n_out, x.data_ptr(), xw.stride(0), k_in, M, dW.data_ptr(), k_in, db.data_ptr())


which = sys.argv[1] if len(sys.argv) > 1 else "all"
if which in ("all", "llm"):
    add(E, D)
if which in ("all", "layers"):
    for i in range(3):
        add(D, FF); add(FF, D); add(W, D); add(D, D)
if which == "all":
    add(D, 63); add(D, KE)
st = _lib.current_stream_ptr(dev)
t.launch(st)      # uploads the table (through the product library), warm
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for rep in range(3):
    e0.record()
    rc = lib.msr3d_wgrad_split(len(t.probs), t._table.data_ptr(), t._pfx.data_ptr(), t.prefix[-1], st)
    e1.record()
    torch.cuda.synchronize()
    assert rc == 0
print("problems", len(t.probs), "workgroups", t.prefix[-1], "ms", e0.elapsed_time(e1))
buf = np.zeros(8192 * 8, np.uint64)
assert lib.msr3d_prof_wgrad_stamps(buf.ctypes.data) == 0
s = buf.reshape(8192, 8).astype(np.int64)
NM = int(os.environ.get('WG_NMULT', '8'))
role = np.arange(8192) % 16 >= NM
ok = (s[:, 0] != 0) & (s[:, 6] != 0)
for name, sel in (("multiplier", ok & ~role), ("loader", ok & role)):
    q = s[sel]
    life = q[:, 6] - q[:, 0]
    print(f"{name:10s} waves {len(q):5d} life p50 {int(np.median(life)):7d} max {int(life.max()):7d} | barrier wait p50 {int(np.median(q[:, 1])):7d} max {int(q[:, 1].max()):7d}")
