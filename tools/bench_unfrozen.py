"""Unfrozen-backbone step of the object encoder (forward + backward, BN in training mode) on the
SharedMLP path of this build (token GEMMs + csrc/bn_train.hip) against torch's conv / batch_norm
modules over the same HIP index ops.   python tools/bench_unfrozen.py [--scenes 4]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from msr3d_amd.modules.vision.pcd_pointnet_encoder import PcdObjEncoder  # noqa: E402
from msr3d_amd.synth import synth_batch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--scenes", type=int, default=4)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--only", choices=["own", "torch"], default=None)
args = ap.parse_args()
kw = dict(sa_n_points=[32, 16, None], sa_n_samples=[32, 32, None], sa_radii=[0.2, 0.4, None],
          sa_mlps=[[3, 64, 64, 128], [128, 128, 128, 256], [256, 256, 512, 768]], dropout=0.0, freeze=False)
torch.manual_seed(0)
enc = PcdObjEncoder(None, **kw).cuda().train()
fts = synth_batch(3, args.scenes, O=60, P=1024, device="cuda")["obj_fts"]
w = torch.randn(args.scenes, 60, 768, device="cuda")


def step():
    enc.zero_grad(set_to_none=True)
    out, _ = enc(fts)
    (out * w).sum().backward()


for name, on in (("own kernels (token GEMM + bn_train.hip)", True), ("torch conv2d / batch_norm modules", False)):
    if args.only and (args.only == "own") != on:
        continue
    for m in enc.modules():
        if hasattr(m, "conv_bn_pairs"):
            m.use_hip_train = on
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.iters):
        step()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / args.iters
    print(f"{name:45s} {ms:8.2f} ms / step  ({args.scenes} scenes x 60 objects x 1024 points, "
          f"{args.scenes * 1e3 / ms:7.1f} scenes/s, peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB)")
