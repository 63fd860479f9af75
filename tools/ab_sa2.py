"""A/B timing of msr3d_sa_level across alternative builds of sa_fused.hip
(tools/_prof/lib*.so), interleaved in one process.
    python tools/ab_sa2.py <level> "<glob>"      (default glob: libsa*.so; build variants with e.g.
    hipcc ... -DMSR3D_SA2_CPB=4 -shared -o tools/_prof/libsa_cpb4.so msr3d_amd/csrc/sa_fused.hip)"""
import ctypes
import glob
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from msr3d_amd.modules.layers.pointnet import PointNetPP  # noqa: E402
from msr3d_amd.pointnet2 import fused  # noqa: E402
from msr3d_amd.synth import synth_batch  # noqa: E402

level = int(sys.argv[1]) if len(sys.argv) > 1 else 2
pat = sys.argv[2] if len(sys.argv) > 2 else "libsa*.so"
torch.manual_seed(0)
net = PointNetPP(sa_n_points=[32, 16, None], sa_n_samples=[32, 32, None], sa_radii=[0.2, 0.4, None],
                 sa_mlps=[[3, 64, 64, 128], [128, 128, 128, 256], [256, 256, 512, 768]]).cuda().eval()
pts = synth_batch(0, 16, device="cuda")["obj_fts"].reshape(-1, 1024, 6).contiguous()
with torch.no_grad():
    _, d = fused.forward(net, pts, return_internals=True)
plan = fused.get_plan(net)
b = pts.shape[0]
p = lambda t: ctypes.c_void_p(t.data_ptr() if t is not None else 0)
libs = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "_prof", pat)))
libs = [l for l in libs if "prof_cpb" not in l]
handles = {os.path.basename(l): ctypes.CDLL(l) for l in libs}
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def call(lib, out):
    L = plan["levels"][level - 1]
    if level == 1:
        return lib.msr3d_sa_level(1, b, 1024, 32, ctypes.c_float(0.2), p(pts), p(None), p(d["new_xyz1"]),
                                  plan["dims"][0], p(L[0]), p(L[1]), p(L[2]), p(out), p(d["ball1"]) if "ball1" in d else p(None), p(None), st)
    if level == 2:
        return lib.msr3d_sa_level(2, b, 32, 16, ctypes.c_float(0.4), p(d["new_xyz1"]), p(d["feat1"]),
                                  p(d["new_xyz2"]), plan["dims"][1], p(L[0]), p(L[1]), p(L[2]), p(out),
                                  p(None), p(None), st)
    return lib.msr3d_sa_level(3, b, 16, 1, ctypes.c_float(0.0), p(d["new_xyz2"]), p(d["feat2"]), p(None),
                              plan["dims"][2], p(L[0]), p(L[1]), p(L[2]), p(out), p(None), p(None), st)


ref = {1: d["feat1"], 2: d["feat2"], 3: d["pooled"]}[level]
res = {k: [] for k in handles}
for rnd in range(12):
    for k, lib in handles.items():
        out = torch.empty_like(ref)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = call(lib, out)
        e1.record()
        torch.cuda.synchronize()
        assert rc == 0
        if rnd == 0:
            assert torch.allclose(out, ref, rtol=1e-4, atol=1e-5), k
        else:
            res[k].append(e0.elapsed_time(e1) * 1e3)
for k, v in res.items():
    v = sorted(v)
    print(f"{k:24s} median {v[len(v) // 2]:8.1f} us   min {v[0]:8.1f} us")
