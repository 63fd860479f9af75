"""Helpers of the situated encoder (mirror of the used part of
/root/reference/modules/utils.py): module helpers :12-55, agent-frame transform
:60-82, pairwise spatial features :88-137."""
import contextlib
import copy

import torch
import torch.nn as nn
import torch.nn.functional as F


def disabled_train(self, mode=True):
    """Bound over `.train` of a frozen sub-module so mode switches do not reach it."""
    return self


def get_activation_fn(activation_type):
    if activation_type not in ("relu", "gelu", "glu"):
        raise RuntimeError(f"activation function currently support relu/gelu, not {activation_type}")
    return getattr(F, activation_type)


def get_mlp_head(input_size, hidden_size, output_size, dropout=0):
    return nn.Sequential(
        nn.Linear(input_size, hidden_size),
        nn.ReLU(),
        nn.LayerNorm(hidden_size, eps=1e-12),
        nn.Dropout(dropout),
        nn.Linear(hidden_size, output_size),
    )


def layer_repeat(module, N):
    """N layers: N-1 deep copies followed by the original instance (order matters for
    which instance init hooks touch -- utils.py:40-41)."""
    return nn.ModuleList([copy.deepcopy(module) for _ in range(N - 1)] + [module])


def maybe_autocast(model, dtype="bf16", enabled=True):
    if model.device == torch.device("cpu"):
        return contextlib.nullcontext()
    # the reference's fp16/fp32 branches are no-op comparisons (utils.py:49-54): the dtype that
    # reaches autocast is bf16 only when asked for, otherwise the string is passed through.
    torch_dtype = torch.bfloat16 if dtype == "bf16" else None
    return torch.autocast("cuda", dtype=torch_dtype, enabled=enabled)


def quaternion_to_matrix(quaternions):
    """(B,4) xyzw quaternion -> (B,3,3) rotation of the INVERSE orientation (x,y,z negated),
    laid out for right-multiplication `p @ R` (utils.py:60-75)."""
    x, y, z = -quaternions[:, 0], -quaternions[:, 1], -quaternions[:, 2]
    w = quaternions[:, 3]
    xx, yy, zz = x * x, y * y, z * z
    xy, xz, xw = x * y, x * z, x * w
    yz, yw, zw = y * z, y * w, z * w
    rows = [
        torch.stack([1 - 2 * (yy + zz), 2 * (xy + zw), 2 * (xz - yw)], dim=-1),
        torch.stack([2 * (xy - zw), 1 - 2 * (xx + zz), 2 * (yz + xw)], dim=-1),
        torch.stack([2 * (xz + yw), 2 * (yz - xw), 1 - 2 * (xx + yy)], dim=-1),
    ]
    return torch.stack(rows, dim=-2)


def transform_to_agent_coor(obj_centers, anchor_loc, anchor_ori):
    """Object centres in the agent's frame: (p - anchor) @ R(q) (utils.py:77-82)."""
    rel = obj_centers - anchor_loc.unsqueeze(1)
    return torch.matmul(rel, quaternion_to_matrix(anchor_ori))


def calc_pairwise_locs(obj_centers, obj_whls, eps=1e-10, pairwise_rel_type="center",
                       spatial_dist_norm=True, spatial_dim=5):
    """(B,L,3) centres -> (B,L,L,spatial_dim) pairwise features
    [d/max d, dz/d, d_xy/d, dy/d_xy, dx/d_xy] with d = sqrt(sum^2 + eps); the max runs over
    ALL pairs of the sample, padded objects included (utils.py:88-137)."""
    if pairwise_rel_type == "mlp":
        locs = torch.cat([obj_centers, obj_whls], 2)
        L = locs.size(1)
        return torch.cat([locs.unsqueeze(2).expand(-1, -1, L, -1),
                          locs.unsqueeze(1).expand(-1, L, -1, -1)], dim=3)

    diff = obj_centers.unsqueeze(2) - obj_centers.unsqueeze(1)          # [b, l, t, 3] = c_l - c_t
    dist = torch.sqrt(torch.sum(diff ** 2, 3) + eps)
    if spatial_dist_norm:
        max_d = dist.flatten(1).max(dim=1)[0]
        norm_dist = dist / max_d[:, None, None]
    else:
        norm_dist = dist
    if spatial_dim == 1:
        return norm_dist.unsqueeze(3)

    dist_2d = torch.sqrt(torch.sum(diff[..., :2] ** 2, 3) + eps)
    if pairwise_rel_type == "center":
        feats = [norm_dist, diff[..., 2] / dist, dist_2d / dist, diff[..., 1] / dist_2d,
                 diff[..., 0] / dist_2d]
    elif pairwise_rel_type == "vertical_bottom":
        bottom = obj_centers.clone()
        bottom[:, :, 2] -= obj_whls[:, :, 2]
        bdiff = bottom.unsqueeze(2) - bottom.unsqueeze(1)
        bdist = torch.sqrt(torch.sum(bdiff ** 2, 3) + eps)
        bdist_2d = torch.sqrt(torch.sum(bdiff[..., :2] ** 2, 3) + eps)
        feats = [norm_dist, bdiff[..., 2] / bdist, bdist_2d / bdist, diff[..., 1] / dist_2d,
                 diff[..., 0] / dist_2d]
    else:
        raise NotImplementedError(f"pairwise_rel_type {pairwise_rel_type}")
    out = torch.stack(feats, dim=3)
    return out[..., 1:] if spatial_dim == 4 else out
