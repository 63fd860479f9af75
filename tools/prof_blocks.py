"""Phase stamps of the scene-block kernels (tools/prof/scene_block_stamped.hip): one eager training step at
the bench shape; after every block launch the per-wave s_memtime marks are read back and summarised."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from msr3d_amd import _lib, fused_model, hipops, scene_blocks  # noqa: E402
from msr3d_amd._lib import SceneBlock  # noqa: E402

NST, MAXW = 16, 4096
HERE = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(HERE, "_prof", "libsb_stamped.so"))
lib.msr3d_scene_block.argtypes = [ctypes.POINTER(SceneBlock), ctypes.c_void_p]
lib.msr3d_scene_block.restype = ctypes.c_int
lib.msr3d_prof_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
KIND = {v: k for k, v in _lib.BLK.items()}
rows = []


def launch_block(stream, **kw):
    s = SceneBlock()
    for k, v in kw.items():
        if isinstance(v, torch.Tensor):
            v = v.data_ptr()
        elif isinstance(v, ctypes.c_void_p):
            v = v.value
        setattr(s, k, v if v is not None else 0)
    torch.cuda.synchronize()
    rc = lib.msr3d_scene_block(ctypes.byref(s), stream)
    assert rc == 0, rc
    torch.cuda.synchronize()
    buf = np.zeros(MAXW * NST, np.uint64)
    assert lib.msr3d_prof_stamps(buf.ctypes.data, 1) == 0
    st = buf.reshape(MAXW, NST).astype(np.int64)
    st = st[st[:, 0] != 0]
    t0 = st[:, 0].min()
    marks = [i for i in range(NST) if (st[:, i] != 0).all()]
    line = f"{KIND[s.kind]:14s} waves {len(st):5d} span {st[:, marks].max() - t0:7d} | start spread {int(st[:, 0].max() - t0):6d} |"
    prev = 0
    for i in marks[1:]:
        d = st[:, i] - st[:, prev]
        line += f" [{prev}->{i}] {int(np.median(d)):6d}/{int(d.max()):6d}"
        prev = i
    rows.append(line)


def main():
    from tests.test_fused_model_gpu import _setup
    B, O, E = 16, 60, 4096
    model, dp, batch = _setup(0.1, B=B, O=O, E=E)
    scene_blocks.launch_block = launch_block
    fused_model.set_mode("blocks")
    for it in range(2):
        rows.clear()
        dp.zero_grad()
        out = model(dict(batch))
        y = out["scene_embeds"]
        (y * y).mean().backward()
        dp.finish()
        torch.cuda.synchronize()
    print("# median / max over waves of the time between marks, s_memtime ticks")
    for r in rows:
        print(r)


if __name__ == "__main__":
    main()
