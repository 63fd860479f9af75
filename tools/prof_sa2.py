"""Phase timing of sa2_kernel from an MSR3D_PROF build (clock64 stamps written to the debug
buffer).  Build, from msr3d_amd/csrc:
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I ../../include -ffp-contract=off \\
        -DMSR3D_PROF -shared -o ../../tools/_prof/libprof_cpb2.so sa_fused.hip
"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from msr3d_amd.modules.layers.pointnet import PointNetPP  # noqa: E402
from msr3d_amd.pointnet2 import fused  # noqa: E402
from msr3d_amd.synth import synth_batch  # noqa: E402

torch.manual_seed(0)
net = PointNetPP(sa_n_points=[32, 16, None], sa_n_samples=[32, 32, None], sa_radii=[0.2, 0.4, None],
                 sa_mlps=[[3, 64, 64, 128], [128, 128, 128, 256], [256, 256, 512, 768]]).cuda().eval()
pts = synth_batch(0, 16, device="cuda")["obj_fts"].reshape(-1, 1024, 6).contiguous()
with torch.no_grad():
    _, d = fused.forward(net, pts, return_internals=True)
plan = fused.get_plan(net)
b = pts.shape[0]
names = ["stage+ballq", "gather", "L1 mfma", "L1 epi+bar", "L2 mfma", "L2 epi+bar", "L3 mfma", "L3 epi"]
for cpb in (2, 4):
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_prof", f"libprof_cpb{cpb}.so")
    if not os.path.exists(path):
        continue
    lib = ctypes.CDLL(path)
    p = lambda t: ctypes.c_void_p(t.data_ptr() if t is not None else 0)
    out = torch.empty((b, 16, 256), device="cuda")
    dbg = torch.zeros((b, 16, 32), dtype=torch.int32, device="cuda")
    L = plan["levels"][1]
    for it in range(3):
        rc = lib.msr3d_sa_level(2, b, 32, 16, ctypes.c_float(0.4), p(d["new_xyz1"]), p(d["feat1"]),
                                p(d["new_xyz2"]), plan["dims"][1], p(L[0]), p(L[1]), p(L[2]), p(out),
                                p(dbg), p(None), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0, rc
    torch.cuda.synchronize()
    nwg = b * (16 // cpb)
    t = dbg.reshape(-1)[:nwg * 8].reshape(nwg, 8).double()
    mean = t.mean(0).tolist()
    tot = sum(mean)
    print(f"CPB={cpb}: {nwg} workgroups, mean cycles per phase (total {tot:.0f}):")
    for n, v in zip(names, mean):
        print(f"   {n:14s} {v:9.0f}  {100 * v / tot:5.1f}%")
    assert torch.allclose(out, d["feat2"], rtol=1e-4, atol=1e-5)
