"""Autograd wrappers and groupers over the HIP ops.

Same public names and call signatures as
/root/reference/modules/third_party/pointnet2/pointnet2_utils.py
(furthest_point_sample :48-77, gather_operation :80-114, three_nn :117-146,
three_interpolate :149-203, grouping_operation :206-254, ball_query :257-288,
QueryAndGroup :291-373, GroupAll :376-419).  `_ext` is a module attribute so the
reference's monkey-patch point (`pointnet2_utils._ext = ...`) keeps working.
"""
import torch
import torch.nn as nn
from torch.autograd import Function

from . import _ext


class FurthestPointSampling(Function):
    @staticmethod
    def forward(ctx, xyz, npoint):
        idx = _ext.furthest_point_sampling(xyz, npoint)
        ctx.mark_non_differentiable(idx)
        return idx

    @staticmethod
    def backward(ctx, grad=None):
        return None, None


furthest_point_sample = FurthestPointSampling.apply


class GatherOperation(Function):
    @staticmethod
    def forward(ctx, features, idx):
        ctx.n = features.size(2)
        ctx.save_for_backward(idx)
        return _ext.gather_points(features, idx)

    @staticmethod
    def backward(ctx, grad_out):
        (idx,) = ctx.saved_tensors
        return _ext.gather_points_grad(grad_out.contiguous(), idx, ctx.n), None


gather_operation = GatherOperation.apply


class ThreeNN(Function):
    @staticmethod
    def forward(ctx, unknown, known):
        dist2, idx = _ext.three_nn(unknown, known)
        dist = torch.sqrt(dist2)
        ctx.mark_non_differentiable(dist, idx)
        return dist, idx

    @staticmethod
    def backward(ctx, a=None, b=None):
        return None, None


three_nn = ThreeNN.apply


class ThreeInterpolate(Function):
    @staticmethod
    def forward(ctx, features, idx, weight):
        ctx.m = features.size(2)
        ctx.save_for_backward(idx, weight)
        return _ext.three_interpolate(features, idx, weight)

    @staticmethod
    def backward(ctx, grad_out):
        idx, weight = ctx.saved_tensors
        return _ext.three_interpolate_grad(grad_out.contiguous(), idx, weight, ctx.m), None, None


three_interpolate = ThreeInterpolate.apply


class GroupingOperation(Function):
    @staticmethod
    def forward(ctx, features, idx):
        ctx.n = features.size(2)
        ctx.save_for_backward(idx)
        return _ext.group_points(features, idx)

    @staticmethod
    def backward(ctx, grad_out):
        (idx,) = ctx.saved_tensors
        return _ext.group_points_grad(grad_out.contiguous(), idx, ctx.n), None


grouping_operation = GroupingOperation.apply


class BallQuery(Function):
    @staticmethod
    def forward(ctx, radius, nsample, xyz, new_xyz):
        idx = _ext.ball_query(new_xyz, xyz, radius, nsample)   # centres first at the _ext level
        ctx.mark_non_differentiable(idx)
        return idx

    @staticmethod
    def backward(ctx, grad=None):
        return None, None, None, None


ball_query = BallQuery.apply


class QueryAndGroup(nn.Module):
    """ball query -> gather neighbourhood -> recentre xyz -> [xyz(3), features(C)] stack."""

    def __init__(self, radius, nsample, use_xyz=True, ret_grouped_xyz=False, normalize_xyz=False,
                 sample_uniformly=False, ret_unique_cnt=False):
        super().__init__()
        if sample_uniformly or ret_unique_cnt:
            # votenet-only extension (pointnet2_utils.py:331-340); no MSR3D config enables it
            raise NotImplementedError("sample_uniformly is outside the MSR3D hot path")
        self.radius, self.nsample, self.use_xyz = radius, nsample, use_xyz
        self.ret_grouped_xyz = ret_grouped_xyz
        self.normalize_xyz = normalize_xyz

    def forward(self, xyz, new_xyz, features=None):
        idx = ball_query(self.radius, self.nsample, xyz, new_xyz)
        grouped_xyz = grouping_operation(xyz.transpose(1, 2).contiguous(), idx)
        grouped_xyz = grouped_xyz - new_xyz.transpose(1, 2).unsqueeze(-1)
        if self.normalize_xyz:
            grouped_xyz = grouped_xyz / self.radius
        if features is None:
            if not self.use_xyz:
                raise AssertionError("Cannot have not features and not use xyz as a feature!")
            new_features = grouped_xyz
        else:
            grouped = grouping_operation(features, idx)
            new_features = torch.cat([grouped_xyz, grouped], dim=1) if self.use_xyz else grouped
        return (new_features, grouped_xyz) if self.ret_grouped_xyz else new_features


class GroupAll(nn.Module):
    """One group holding every point: (B, 3+C, 1, N)."""

    def __init__(self, use_xyz=True, ret_grouped_xyz=False):
        super().__init__()
        self.use_xyz = use_xyz

    def forward(self, xyz, new_xyz, features=None):
        grouped_xyz = xyz.transpose(1, 2).unsqueeze(2)
        if features is None:
            return grouped_xyz
        grouped = features.unsqueeze(2)
        return torch.cat([grouped_xyz, grouped], dim=1) if self.use_xyz else grouped
