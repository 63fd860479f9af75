"""A/B timing and phase stamps of msr3d_sa_level_split (level 2): every tools/_prof/abl/*.so -- builds of
csrc/sa_split.hip with experimental edits -- is loaded beside the shipping library and timed on the bench's
shapes; tools/_prof/abl/stamp/stamp.so, built from tools/prof/sa_split_stamped.hip, records s_memtime stamps of one steady-state
tile per block (the phase table of DESIGN.md 4.1b).
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -shared -fPIC -I include tools/prof/sa_split_stamped.hip \
          msr3d_amd/csrc/sa_split.hip -o tools/_prof/abl/stamp/stamp.so
    python tools/ab_split.py [--batch 16]"""
import argparse
import ctypes
import glob
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--iters", type=int, default=30)
args = ap.parse_args()

from msr3d_amd import _lib  # noqa: E402
from msr3d_amd.modules.layers.pointnet import PointNetPP  # noqa: E402
from msr3d_amd.pointnet2 import fused  # noqa: E402
from msr3d_amd.synth import synth_batch  # noqa: E402

torch.manual_seed(0)
net = PointNetPP(sa_n_points=[32, 16, None], sa_n_samples=[32, 32, None], sa_radii=[0.2, 0.4, None],
                 sa_mlps=[[3, 64, 64, 128], [128, 128, 128, 256], [256, 256, 512, 768]]).cuda().eval()
pts = synth_batch(0, args.batch, device="cuda")["obj_fts"].reshape(-1, 1024, 6).contiguous()
b = pts.shape[0]
with torch.no_grad():
    net(pts)
plan = fused.get_plan(net)
S = plan["split2"]
new1 = torch.rand(b, 32, 3, device="cuda")
feat1 = torch.randn(b, 32, 128, device="cuda")
new2 = new1[:, :16].contiguous()
out = torch.empty(b, 16, 256, device="cuda")
p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
here = os.path.dirname(os.path.abspath(__file__))
libs = [("shipping", _lib.LIB_PATH)] + [(os.path.basename(f), f) for f in sorted(glob.glob(os.path.join(here, "_prof/abl/*.so")))]
for name, path in libs:
    lib = ctypes.CDLL(path)
    fn = lib.msr3d_sa_level_split
    fn.argtypes = [ctypes.c_int] * 4 + [ctypes.c_float] + [ctypes.c_void_p] * 13
    fn.restype = ctypes.c_int
    def call():
        return fn(2, b, 32, 16, ctypes.c_float(0.4), p(new1), p(feat1), p(new2), p(S[0][0]), p(S[0][1]), p(S[1][0]),
                  p(S[1][1]), p(S[2][0]), p(S[2][1]), p(out), p(None), p(None), st)
    for _ in range(3):
        assert call() == 0
    torch.cuda.synchronize()
    ts = []
    for _ in range(args.iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); call(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    print(f"{name:24s} median {ts[len(ts)//2]:8.1f} us  min {ts[0]:8.1f} us")

# phase stamps (s_memtime): tools/prof/sa_split_stamped.hip built into tools/_prof/abl/stamp/stamp.so
sp = os.path.join(here, "_prof/abl/stamp/stamp.so")
if os.path.exists(sp):
    lib = ctypes.CDLL(sp)
    fn = lib.msr3d_sa_level_split
    fn.argtypes = [ctypes.c_int] * 4 + [ctypes.c_float] + [ctypes.c_void_p] * 13
    fn.restype = ctypes.c_int
    import numpy as np
    lib.msr3d_prof_sa2_stamps.argtypes = [ctypes.c_void_p]
    for _ in range(3):
        fn(2, b, 32, 16, ctypes.c_float(0.4), p(new1), p(feat1), p(new2), p(S[0][0]), p(S[0][1]), p(S[1][0]),
           p(S[1][1]), p(S[2][0]), p(S[2][1]), p(out), p(None), p(None), st)
    torch.cuda.synchronize()
    hbuf = np.zeros(512 * 4 * 16, np.uint64)
    assert lib.msr3d_prof_sa2_stamps(hbuf.ctypes.data) == 0
    t = torch.from_numpy(hbuf.astype(np.int64)).view(-1, 4, 16)[:, :, :11].double()
    d = (t[:, :, 1:] - t[:, :, :-1])           # ticks of 10 ns
    names = ["L1 gemm", "L1 epi+geo", "L2 gemm+query", "L2 epi", "L3 gemm", "L3 epi", "barrier E", "gather", "barrier F", "loop top"]
    med = d.reshape(-1, 10).median(dim=0).values
    mean = d.reshape(-1, 10).mean(dim=0)
    tot = (t[:, :, 10] - t[:, :, 0]).reshape(-1)
    print("phase medians / means (cycles):")
    for k, nme in enumerate(names):
        print(f"  {nme:14s} {med[k]:8.0f} {mean[k]:8.0f}")
    print(f"  block total    {tot.median():8.0f} {tot.mean():8.0f}   blocks {t.shape[0]}")
