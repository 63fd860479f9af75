"""Micro-benchmark of the frozen encoder alone (4 fused launches + fc), for profiling.
    python tools/bench_sa.py [--batch 16] [--iters 20]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--iters", type=int, default=20)
args = ap.parse_args()

from msr3d_amd.modules.layers.pointnet import PointNetPP  # noqa: E402
from msr3d_amd.synth import synth_batch  # noqa: E402

torch.manual_seed(0)
net = PointNetPP(sa_n_points=[32, 16, None], sa_n_samples=[32, 32, None], sa_radii=[0.2, 0.4, None],
                 sa_mlps=[[3, 64, 64, 128], [128, 128, 128, 256], [256, 256, 512, 768]]).cuda().eval()
pts = synth_batch(10000, args.batch, device="cuda")["obj_fts"].reshape(-1, 1024, 6).contiguous()
with torch.no_grad():
    for _ in range(3):
        net(pts)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.iters):
        net(pts)
    e1.record()
    torch.cuda.synchronize()
print(f"encoder forward: {e0.elapsed_time(e1) / args.iters * 1e3:.1f} us for {pts.shape[0]} objects")

# per-kernel timing with events through the loader's timing sink
from msr3d_amd import _lib  # noqa: E402
sink = {k: [] for k in ("msr3d_sa_fps2", "msr3d_sa_level1", "msr3d_sa_level2", "msr3d_sa_level3")}
_lib.set_timing_sink(sink)
with torch.no_grad():
    for _ in range(args.iters):
        net(pts)
torch.cuda.synchronize()
_lib.set_timing_sink(None)
for k, v in sink.items():
    ts = sorted(a.elapsed_time(b) * 1e3 for a, b in v)
    print(f"  {k:18s} median {ts[len(ts) // 2]:8.1f} us   min {ts[0]:8.1f} us")
