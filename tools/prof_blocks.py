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
    marks = [i for i in range(NST) if (st[:, i] != 0).all()]
    # a workgroup's waves are consecutive rows; its span = last mark of its slowest wave - first mark of its earliest
    # (one XCD, one counter: spans are comparable inside a workgroup only)
    name = KIND[s.kind]
    slices = 8 if name.startswith("attn") else int(s.ff) // 128 if name.startswith("ffn") else \
        int(s.N) // 256 if name == "linear" else int(s.lda0) // 256
    if name == "attn_fwd" and os.environ.get("MSR3D_ATTN_FWD_SPLIT", "1") != "0":
        slices = 16                      # two workgroups per (scene, head)
    nblk = int(s.B) * slices
    wpb = max(len(st) // max(nblk, 1), 1)
    blk = st[:nblk * wpb].reshape(nblk, wpb, NST)
    span = blk[:, :, marks[-1]].max(axis=1) - blk[:, :, 0].min(axis=1)
    # start skew inside an XCD (workgroups of one XCD = one counter): the spread of the workgroups' first marks
    x = np.arange(nblk) % 8
    first = blk[:, :, 0].min(axis=1)
    skew = max(int(first[x == k].max() - first[x == k].min()) for k in range(8) if (x == k).any())
    whole = max(int(blk[x == k][:, :, marks[-1]].max() - first[x == k].min()) for k in range(8) if (x == k).any())
    line = f"{KIND[s.kind]:14s} waves {len(st):5d} ({wpb}/wg) wg span {int(np.median(span)):6d}/{int(span.max()):6d} xcd skew {skew:6d} xcd whole {whole:6d} |"
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
