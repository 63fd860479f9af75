"""PointNet++ object encoder behind the reference's `PointNetPP` interface
(/root/reference/modules/layers/pointnet.py:6-63): one set-abstraction module per entry of the four
hyper-parameter lists, then `fc`.  State-dict keys `encoder.{i}.mlps.0.layer{j}.*`, `fc.*`.

On the GPU with a frozen, eval-mode backbone (every shipped config) forward() is the fused HIP
path of pointnet2/fused.py; anything else takes the composite, autograd-capable path below."""
import torch.nn as nn

from ...pointnet2 import fused
from ...pointnet2.pointnet2_modules import PointnetSAModule


def break_up_pc(pc):
    """(..., N, 3+C) -> (xyz (..., N, 3), features (..., C, N) or None), both contiguous --
    the split the SA modules expect (coordinates point-major, features channel-major)."""
    coords, extra = pc[..., :3], pc[..., 3:]
    feats = extra.transpose(1, 2).contiguous() if extra.size(-1) else None
    return coords.contiguous(), feats


class PointNetPP(nn.Module):
    def __init__(self, sa_n_points: list, sa_n_samples: list, sa_radii: list, sa_mlps: list,
                 bn=True, use_xyz=True):
        super().__init__()
        specs = (sa_n_points, sa_n_samples, sa_radii, sa_mlps)
        if len({len(s) for s in specs}) != 1:
            raise ValueError("Lens of given hyper-params are not compatible")
        levels = [PointnetSAModule(npoint=n, nsample=k, radius=r, mlp=m, bn=bn, use_xyz=use_xyz)
                  for n, k, r, m in zip(*specs)]
        self.encoder = nn.ModuleList(levels)
        width = sa_mlps[-1][-1]
        pooled_points = sa_n_points[-1] or 1           # None = group-all: one pooled point
        self.fc = nn.Linear(pooled_points * width, width)

    def forward(self, features, valid=None, out=None):
        """(b, P, 3+C) -> (b, D).  valid (b,) bool: padding slots to skip (fused path only; see
        fused.forward) -- ignored by the composite path, which encodes whatever the slots hold.
        out (b, D): optional destination (fused path writes it directly)."""
        if fused.can_fuse(self, features):
            return fused.forward(self, features, valid=valid, out=out)
        if out is not None:
            return out.copy_(self.forward(features, valid=valid))
        xyz, feats = break_up_pc(features)
        for level in self.encoder:
            xyz, feats = level(xyz, feats)
        return self.fc(feats.flatten(1))
