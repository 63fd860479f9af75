"""Set-abstraction modules (the part of
/root/reference/modules/third_party/pointnet2/pointnet2_modules.py:26-161 MSR3D uses).

`forward` is the composite path: HIP index ops + torch conv/BN/ReLU/max, usable
with autograd and BN in train mode.  The frozen, eval-mode encoder does not come
through here level by level: PointNetPP routes it to the fused SA kernels.
"""
from typing import List

import torch
import torch.nn as nn

from .. import hipops
from . import pointnet2_utils
from . import pytorch_utils as pt_utils


class _PointnetSAModuleBase(nn.Module):
    def __init__(self):
        super().__init__()
        self.npoint = None
        self.groupers = None
        self.mlps = None

    def sample_centres(self, xyz):
        """(B,N,3) -> (B,npoint,3) by FPS + gather, or None for a group-all level."""
        if self.npoint is None:
            return None
        idx = pointnet2_utils.furthest_point_sample(xyz, self.npoint)
        flipped = xyz.transpose(1, 2).contiguous()
        return pointnet2_utils.gather_operation(flipped, idx).transpose(1, 2).contiguous()

    def forward(self, xyz, features=None):
        new_xyz = self.sample_centres(xyz)
        pooled = []
        for grouper, mlp in zip(self.groupers, self.mlps):
            if (isinstance(grouper, pointnet2_utils.QueryAndGroup) and grouper.use_xyz
                    and not grouper.normalize_xyz and not grouper.ret_grouped_xyz and hipops._mlp_train_ok(mlp)
                    and hipops.group_rows_supported(xyz, new_xyz, features, grouper.nsample)):
                # unfrozen backbone on the GPU: neighbourhood rows written token-major, straight into
                # the layout the SharedMLP's token GEMMs read
                idx = pointnet2_utils.ball_query(grouper.radius, grouper.nsample, xyz, new_xyz)
                pooled.append(hipops.sa_level_train(mlp, xyz, new_xyz, features, idx))
                continue
            grouped = grouper(xyz, new_xyz, features)     # (B, C_in, npoint, nsample)
            if hipops.shared_mlp_train_supported(mlp, grouped):
                # unfrozen backbone on the GPU: the convolutions as token GEMMs, BatchNorm (batch
                # statistics) + ReLU in csrc/bn_train.hip
                pooled.append(hipops.shared_mlp_train(mlp, grouped))
                continue
            x = mlp(grouped)                              # (B, C_out, npoint, nsample)
            pooled.append(torch.amax(x, dim=3))           # max over the neighbourhood
        return new_xyz, torch.cat(pooled, dim=1)


class PointnetSAModuleMSG(_PointnetSAModuleBase):
    def __init__(self, *, npoint: int, radii: List[float], nsamples: List[int],
                 mlps: List[List[int]], bn: bool = True, use_xyz: bool = True,
                 sample_uniformly: bool = False):
        super().__init__()
        if not (len(radii) == len(nsamples) == len(mlps)):
            raise AssertionError("radii / nsamples / mlps must have one entry per scale")
        self.npoint = npoint
        self.groupers = nn.ModuleList()
        self.mlps = nn.ModuleList()
        for radius, nsample, spec in zip(radii, nsamples, mlps):
            if npoint is not None:
                self.groupers.append(pointnet2_utils.QueryAndGroup(
                    radius, nsample, use_xyz=use_xyz, sample_uniformly=sample_uniformly))
            else:
                self.groupers.append(pointnet2_utils.GroupAll(use_xyz))
            spec = list(spec)
            if use_xyz:
                # the reference adds the 3 in place (pointnet2_modules.py:120-122), which corrupts the
                # caller's list -- PcdObjEncoder's mutable default included, so a second construction
                # with defaults fails there; a private copy gives the same module without that
                spec[0] += 3
            self.mlps.append(pt_utils.SharedMLP(spec, bn=bn))


class PointnetSAModule(PointnetSAModuleMSG):
    def __init__(self, *, mlp: List[int], npoint: int = None, radius: float = None,
                 nsample: int = None, bn: bool = True, use_xyz: bool = True):
        super().__init__(mlps=[mlp], npoint=npoint, radii=[radius], nsamples=[nsample], bn=bn,
                         use_xyz=use_xyz)


class PointnetFPModule(nn.Module):
    """Feature propagation: carry `known_feats` (B, C2, m) from the `known` points (B, m, 3) to the
    `unknown` points (B, n, 3) by inverse-distance weighting over the three nearest known points,
    stack with the unknown points' own features and run the SharedMLP
    (/root/reference/modules/third_party/pointnet2/pointnet2_modules.py:331-393).  The only consumer
    of three_nn / three_interpolate; not instantiated by any MSR3D config, kept so
    `pointnet2_modules` offers the reference's decoder-side module next to the SA modules."""

    def __init__(self, *, mlp: List[int], bn: bool = True):
        super().__init__()
        self.mlp = pt_utils.SharedMLP(mlp, bn=bn)

    @staticmethod
    def propagate(unknown, known, known_feats):
        """(B, C2, n): three-NN interpolation, or a broadcast when there are no known positions."""
        if known is None:
            return known_feats.expand(known_feats.size(0), known_feats.size(1), unknown.size(1))
        dist, idx = pointnet2_utils.three_nn(unknown, known)       # Euclidean (the wrapper takes the root)
        inv = 1.0 / (dist + 1e-8)
        weight = inv / inv.sum(dim=2, keepdim=True)
        return pointnet2_utils.three_interpolate(known_feats, idx, weight)

    def forward(self, unknown, known, unknow_feats, known_feats):
        feats = self.propagate(unknown, known, known_feats)
        if unknow_feats is not None:
            feats = torch.cat([feats, unknow_feats], dim=1)          # (B, C2 + C1, n)
        return self.mlp(feats.unsqueeze(-1)).squeeze(-1)
