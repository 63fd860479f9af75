"""msr3d_wgrad_split on the bench step's problem set (synthetic operands), the launch forms side by side:
whole tiles / mixed (tiles of the partial round as two half-reductions) / every tile halved.
    python tools/bench_wgrad.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from msr3d_amd import _lib  # noqa: E402
from msr3d_amd.scene_blocks import WgradTable  # noqa: E402

M, D, FF, W, E, KE = 960, 256, 2048, 816, 4096, 768
dev = torch.device("cuda")


def build(mixed, halves=False):
    os.environ["MSR3D_WGRAD_MIXED"] = "1" if mixed else "0"
    os.environ["MSR3D_WGRAD_HALVES"] = "1" if halves else "0"
    t = WgradTable(dev)
    keep = []

    def add(n_out, k_in):
        dy, x = torch.randn(M, n_out, device=dev), torch.randn(M, k_in, device=dev)
        dW, db = torch.zeros(n_out, k_in, device=dev), torch.zeros(n_out, device=dev)
        keep.extend([dy, x, dW, db])
        t.add(dy.data_ptr(), n_out, n_out, x.data_ptr(), k_in, k_in, M, dW.data_ptr(), k_in, db.data_ptr())
    add(E, D)
    for _ in range(3):
        add(D, FF); add(FF, D); add(W, D); add(D, D)
    add(D, 64); add(D, 3); add(D, KE)
    return t, keep


st = _lib.current_stream_ptr(dev)
for form in (0, 1):
  _lib.load().msr3d_wgrad_form(form)
  print("tile kernel:", "pipe (8 waves, round 6)" if form else "loader + multiplier waves (rounds 4-5)")
  for name, kw in (("whole tiles", dict(mixed=False)), ("mixed", dict(mixed=True)), ("all halved", dict(mixed=False, halves=True)),
                   ("stream", dict(mixed=True, stream=True))):
      if kw.get("stream") and not form:
          continue
      want_stream = kw.pop("stream", False)
      t, keep = build(**kw)
      t.stream = want_stream
      if "WHOLE" in os.environ and kw.get("mixed"):
          t._whole_tiles()
          t._whole = int(os.environ["WHOLE"])
      for _ in range(5):
          t.launch(st)
      torch.cuda.synchronize()
      ts = []
      for _ in range(30):
          e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
          e0.record(); t.launch(st); e1.record()
          torch.cuda.synchronize()
          ts.append(e0.elapsed_time(e1) * 1e3)
      ts.sort()
      extra = f" whole {t._whole_tiles()} of {t.prefix[-1]} workgroups, xcd load {t.xcd_load}" if kw.get("mixed") else ""
      if t.stream and t._stream_plan:
          extra = f" {t._stream_plan[3]} workgroups, {t._stream_plan[2]} pieces, {t._stream_plan[4]} cut tiles, load {t._stream_plan[6]}..{t._stream_plan[5]} slab pairs"
      print(f"{name:12s} median {ts[len(ts) // 2]:7.1f} us  min {ts[0]:7.1f} us{extra}")
