"""PointNet++ object encoder (mirror of /root/reference/modules/layers/pointnet.py:6-63):
three set-abstraction levels + `fc`.  State-dict keys `encoder.{i}.mlps.0.layer{j}.*`, `fc.*`."""
import torch.nn as nn

from ...pointnet2 import fused
from ...pointnet2.pointnet2_modules import PointnetSAModule


def break_up_pc(pc):
    """(..., N, 3+C) -> xyz (..., N, 3) contiguous, features (..., C, N) contiguous or None."""
    xyz = pc[..., 0:3].contiguous()
    features = pc[..., 3:].transpose(1, 2).contiguous() if pc.size(-1) > 3 else None
    return xyz, features


class PointNetPP(nn.Module):
    def __init__(self, sa_n_points: list, sa_n_samples: list, sa_radii: list, sa_mlps: list,
                 bn=True, use_xyz=True):
        super().__init__()
        n_sa = len(sa_n_points)
        if not (n_sa == len(sa_n_samples) == len(sa_radii) == len(sa_mlps)):
            raise ValueError("Lens of given hyper-params are not compatible")
        self.encoder = nn.ModuleList(
            PointnetSAModule(npoint=sa_n_points[i], nsample=sa_n_samples[i], radius=sa_radii[i],
                             mlp=sa_mlps[i], bn=bn, use_xyz=use_xyz)
            for i in range(n_sa))
        out_n_points = sa_n_points[-1] if sa_n_points[-1] is not None else 1
        self.fc = nn.Linear(out_n_points * sa_mlps[-1][-1], sa_mlps[-1][-1])

    def forward(self, features):
        """(b, P, 3+C) -> (b, D).  Frozen/eval backbone on the GPU: the fused set-abstraction
        kernels (pointnet2/fused.py).  Otherwise the composite, autograd-capable path."""
        if fused.can_fuse(self, features):
            return fused.forward(self, features)
        xyz, features = break_up_pc(features)
        for sa in self.encoder:
            xyz, features = sa(xyz, features)
        return self.fc(features.reshape(features.size(0), -1))
